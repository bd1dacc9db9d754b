#!/bin/bash
# rocprofv3 kernel trace (+ stats) of one command; the kernel_stats CSV lands in gpurun_out/<tag>_kernel_stats.csv and its head is printed.
#   usage (GPU box): scripts/r06_trace.sh <tag> <command ...>
tag=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.log 2>&1)
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { echo "no kernel_stats for $tag"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.log; exit 1; }
cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, os, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(os.environ.get("TOP", "16"))]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {int(r['Calls']):6d}  {float(r['Percentage']):6.2f} %  {r['Name'][:110]}")
PY
