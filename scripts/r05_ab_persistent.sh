#!/bin/bash
# same-call A/B of the persistent composites (SplatState.tile_queue): [tests first], then the bench line at B and B-loop for
#   P1  splatam_amd/lib, SPLAT_PERSISTENT=1    (tile loop compiled in, queues on)
#   P0  splatam_amd/lib, SPLAT_PERSISTENT=0    (tile loop compiled in, one-shot launches)
#   AB  splatam_amd/lib_ab                     (make -C splatam_amd/csrc OUTDIR=../lib_ab EXTRA=-DSPLAT_TILE_LOOP=0: the loop compiled out)
# usage (on the GPU box): scripts/r05_ab_persistent.sh [rounds] ["pytest args"]
rounds=${1:-1}; tests=$2
export TMPDIR=/tmp
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
if [ -n "$tests" ]; then
  timeout 1500 python -m pytest $tests -m gpu -q -x -p no:cacheprovider > gpurun_out/abp_tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/abp_tests.log
fi
run() {
  tag=$1; wl=$2; shift; shift
  env "$@" timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-slam-loop --sustain-s 2 > gpurun_out/abp_${tag}_$wl.log 2> gpurun_out/abp_${tag}_$wl.err
  python - "$tag" "$wl" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/abp_{sys.argv[1]}_{sys.argv[2]}.log") if l.startswith("{")][-1])
    o = d["roofline"]["other"]
    k = o.get("kernels", {})
    print(sys.argv[2], sys.argv[1], "value", d["value"], "sustained", d["sustained"].get("iters_per_s"), "tracking", d["tracking_iters_per_s"], "mapping",
          d["mapping_iters_per_s"], "K6", o["render_forward_ms"], "K7", d["roofline"]["kernel_ms"], "K7 30-in-a-row", o["render_backward_30_in_a_row_ms"],
          "fused track us", (k.get("render_track_fused_kernel") or {}).get("avg_us"))
except Exception as e:
    print(sys.argv[2], sys.argv[1], "FAILED", e)
PY
}
for r in $(seq $rounds); do
  for wl in B B-loop; do
    run P1_$r $wl SPLAT_PERSISTENT=1
    run P0_$r $wl SPLAT_PERSISTENT=0
    [ -f splatam_amd/lib_ab/libsplat_hip.so ] && run AB_$r $wl SPLAT_PERSISTENT=0 SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so
  done
done
