#!/bin/bash
# same-call A/B of process-level switches with the kernel trace on: per-kernel averages of the bench loop under every setting.
#   usage: scripts/r04_ab_prof.sh "<kernel name regex>" "VAR=1" ["VAR2=1" ...]      (base / each setting / base again)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
pat=$1; shift
run() {
  tag=$1; shift
  rm -rf /tmp/abp_$tag
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/abp_$tag -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --no-roofline --sustain-s 1 > $GRAFT_REPO_ROOT/gpurun_out/abp_$tag.log 2> $GRAFT_REPO_ROOT/gpurun_out/abp_$tag.err )
  f=$(find /tmp/abp_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/abp_${tag}_kernel_stats.csv
  python - "$tag" "$f" "$pat" <<'PY'
import csv, json, re, sys
tag, f, pat = sys.argv[1:4]
d = json.loads([l for l in open(f"gpurun_out/abp_{tag}.log") if l.startswith("{")][-1])
print(tag, "value", d["value"], "tracking", d["tracking_iters_per_s"], "mapping", d["mapping_iters_per_s"])
for r in csv.DictReader(open(f)):
    if re.search(pat, r["Name"]):
        print("    %-90s calls %6s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
run base A=1
i=0
for kv in "$@"; do i=$((i+1)); run "v$i" $kv; done
run base2 A=1
