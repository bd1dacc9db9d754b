#!/bin/bash
bash scripts/r03_ab_quick.sh r03_q2
unset SPLAT_HIP_LIB
export TMPDIR=/tmp
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so; fi
  for wl in E-clustered E-clustered-5M; do
    rm -rf /tmp/prof_q2$v$wl
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q2$v$wl -o b -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /tmp/q2_$v$wl.log 2>&1)
    f=$(find /tmp/prof_q2$v$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_q2_${wl}_${v}_kernel_stats.csv
    echo "== $v $wl: $(tail -1 /tmp/q2_$v$wl.log | cut -c1-130)"
  done
done
