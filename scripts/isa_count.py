#!/usr/bin/env python3
"""Static instruction mix per kernel of an assembly file written by scripts/isa_stats.sh (/tmp/isa_<file>.s).
usage: scripts/isa_count.py /tmp/isa_fused.s [name regex]"""
import re, subprocess, sys
from collections import Counter
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not pat.search(name):
        continue
    ins = [l.split()[0] for l in m.group(2).split("\n") if l.startswith("\t") and l.split() and not l.split()[0].startswith((".", ";"))]
    c = Counter()
    for i in ins:
        c["valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += 1
    print(re.sub(r"\(.*", "", name).replace("splat::(anonymous namespace)::", "")[:90], dict(c))
    print("    ", Counter(ins).most_common(16))
