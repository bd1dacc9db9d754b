#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$GRAFT_REPO_ROOT/splatam_amd/lib_ab/libsplat_hip.so; else unset SPLAT_HIP_LIB; fi
  timeout 300 python scripts/slam_loop_profile.py B 4 > gpurun_out/r03_slamloop_$v.log 2>&1
  rm -rf /tmp/sl_$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sl_$v -o k -- python $GRAFT_REPO_ROOT/scripts/slam_loop_profile.py B 4 > /tmp/sl_$v.log 2>&1)
  f=$(find /tmp/sl_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r03_slamloop_${v}_kernel_stats.csv
  python - "$f" >> gpurun_out/r03_slamloop_$v.log <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'splat' in r['Name']]
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} tot_ms {float(r['TotalDurationNs'])/1e6:8.1f}")
PY
done
cat gpurun_out/r03_slamloop_A.log; echo =====; cat gpurun_out/r03_slamloop_B.log
