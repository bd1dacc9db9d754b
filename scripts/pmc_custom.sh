#!/bin/bash
# ad-hoc counter passes over scripts/fused_driver.py: usage scripts/pmc_custom.sh <tag> <workload> "<kernel regex>" "<counters of pass 1>" ["<counters of pass 2>" ...]
tag=$1; wl=$2; pat=$3; shift 3
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcc_$tag.txt
: > $out
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmcc_$i
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmcc_$i -o p -- python $GRAFT_REPO_ROOT/scripts/fused_driver.py $wl 3 > /tmp/pmcc_$i.log 2>&1)
  f=$(find /tmp/pmcc_$i -name "*counter_collection.csv" | head -1)
  echo "## pass $i: $ctrs" >> $out
  if [ -z "$f" ]; then echo "no counter file; log tail:" >> $out; tail -5 /tmp/pmcc_$i.log >> $out; continue; fi
  python - "$f" "$pat" >> $out <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    if not re.search(sys.argv[2], k):
        continue
    m = re.search(r'([A-Za-z_0-9]+_kernel[0-9]*(?:_w[0-9])?(?:<[^>]*>)?)', k)
    acc[m.group(1) if m else k[:56]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())), f"(n={len(next(iter(d.values())))})")
PY
done
cat $out
