#!/bin/bash
# same-call A/B of process-level switches over the bench line (no profiler): usage scripts/r04_ab_env.sh "VAR=0" ["VAR2=0" ...]
export TMPDIR=/tmp
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py $BENCH_ARGS --no-cpu-baseline --no-slam-loop --sustain-s 2 > gpurun_out/ab_$tag.log 2> gpurun_out/ab_$tag.err
  python - "$tag" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/ab_{sys.argv[1]}.log") if l.startswith("{")][-1])
print(sys.argv[1], "value", d["value"], "sustained", d["sustained"].get("iters_per_s"), "tracking", d["tracking_iters_per_s"], "mapping", d["mapping_iters_per_s"],
      "K6", d["roofline"]["other"]["render_forward_ms"], "K7", d["roofline"]["kernel_ms"])
PY
}
run base A=1
i=0
for kv in "$@"; do i=$((i+1)); run "v$i" $kv; done
run base2 A=1
