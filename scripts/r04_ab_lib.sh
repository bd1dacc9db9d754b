#!/bin/bash
# same-call A/B of two BUILDS of the library over the bench line: splatam_amd/lib (working tree) vs splatam_amd/lib_ab
#   (make -C splatam_amd/csrc OUTDIR=../lib_ab EXTRA=...).   usage (on the GPU box): scripts/r04_ab_lib.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py $BENCH_ARGS --no-cpu-baseline --no-slam-loop --sustain-s 2 > gpurun_out/abl_$tag.log 2> gpurun_out/abl_$tag.err
  python - "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/abl_{sys.argv[1]}.log") if l.startswith("{")][-1])
print(sys.argv[1], "value", d["value"], "sustained", d["sustained"].get("iters_per_s"), "tracking", d["tracking_iters_per_s"], "mapping", d["mapping_iters_per_s"],
      "K6", d["roofline"]["other"]["render_forward_ms"], "K7", d["roofline"]["kernel_ms"])
PY
}
for r in 1 2; do
  run lib_$r A=1
  run lib_ab_$r SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so
done
