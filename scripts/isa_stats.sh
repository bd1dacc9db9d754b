#!/bin/bash
# Per-kernel register / LDS / scratch figures of one .hip file as the product build compiles it (device code only, no GPU needed).
#   usage: scripts/isa_stats.sh render.hip [filter regex]
cd "$(dirname "$0")/../splatam_amd/csrc"
f=$1; pat=${2:-.}
out=/tmp/isa_$(basename $f .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize --cuda-device-only -S $f -o $out 2>/dev/null
python3 - "$out" "$pat" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not pat.search(name):
        continue
    body = m.group(2)
    g = lambda k: (re.search(r"\." + k + r"\s+(\S+)", body) or [None, "?"])[1]
    short = re.sub(r"\(.*", "", name)[:110]
    print(f"{short:110s} vgpr {g('amdhsa_next_free_vgpr'):>4} sgpr {g('amdhsa_next_free_sgpr'):>4} lds {g('amdhsa_group_segment_fixed_size'):>6} scratch {g('amdhsa_private_segment_fixed_size'):>5}")
# instruction counts per kernel body are in the text too: lines between the kernel's label and its s_endpgm
PY
