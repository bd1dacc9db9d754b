#!/bin/bash
# Round 6, VERDICT r5 item 4: what the fused tracking composite's accumulator atomics (62 MB of 64-byte line requests at B) COST -- the
# product kernel beside its measurement builds (splat_debug_option(4, bits): 16 forward + loss only, 2 + staging, 4 everything but the
# atomics): launch duration (kernel trace), and per launch SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_VALU
# and WRITE_SIZE (separate rocprofv3 --pmc passes, --kernel-trace only).   usage: scripts/r06_track_atomics.sh <tag> [workload]
tag=${1:-r06}; wl=${2:-B}
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_track_atomics_$wl.txt
: > $out
rm -rf /tmp/ta_trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ta_trace -o p -- python $GRAFT_REPO_ROOT/scripts/track_traffic_driver.py $wl 40 > /tmp/ta_trace.log 2>&1)
f=$(find /tmp/ta_trace -name "*kernel_stats.csv" | head -1)
echo "## launch duration (us, rocprofv3 --kernel-trace --stats, 40 launches per build)" >> $out
[ -n "$f" ] && python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "render_track_fused_kernel" in r["Name"]:
        print(f"{r['Name'].split('(')[0].replace('void splat::', '')}: {float(r['AverageNs']) / 1e3:.1f} (n={r['Calls']})")
PY
pass() {
  name=$1; shift
  rm -rf /tmp/ta_$name
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/ta_$name -o p -- python $GRAFT_REPO_ROOT/scripts/track_traffic_driver.py $wl 5 > /tmp/ta_$name.log 2>&1)
  f=$(find /tmp/ta_$name -name "*counter_collection.csv" | head -1)
  echo "## pass $name: $* (per launch, mean)" >> $out
  [ -z "$f" ] && { tail -5 /tmp/ta_$name.log >> $out; return; }
  python - "$f" >> $out <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(render_track_fused_kernel<[^>]*>)', r.get('Kernel_Name', ''))
    if m:
        acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k, ' '.join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(d.items())), f"(n={len(next(iter(d.values())))})")
PY
}
pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
pass write WRITE_SIZE
pass fetch FETCH_SIZE
cat $out
