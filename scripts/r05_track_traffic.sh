#!/bin/bash
# Where the fused tracking composite's memory traffic comes from: FETCH_SIZE / WRITE_SIZE of the product kernel and of
# its measurement builds, two rocprofv3 --pmc runs (one counter set each, --kernel-trace only).   usage: scripts/r05_track_traffic.sh <tag> [workload]
tag=${1:-r05}; wl=${2:-B}
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_track_traffic_$wl.txt
: > $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tt_$ctr
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/tt_$ctr -o p -- python $GRAFT_REPO_ROOT/scripts/track_traffic_driver.py $wl 5 > /tmp/tt_$ctr.log 2>&1)
  f=$(find /tmp/tt_$ctr -name "*counter_collection.csv" | head -1)
  echo "## $ctr (KiB per launch, mean over launches)" >> $out
  [ -z "$f" ] && { tail -5 /tmp/tt_$ctr.log >> $out; continue; }
  python - "$f" >> $out <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    m = re.search(r'(render_track_fused_kernel<[^>]*>|fused_backward_kernel<[^>]*>|fused_preprocess_kernel<[^>]*>)', k)
    if m:
        acc[m.group(1)].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f"{k}: {sum(v) / len(v):.1f} (n={len(v)})")
PY
done
cat $out
