#!/bin/bash
# same-call A/B of the working tree (lib) against HEAD (lib_ab): full GPU suite first, then the bench line + kernel stats of both libraries
tag=${1:-r03_c12}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so; fi
  timeout 120 python scripts/track_loop_spread.py 4 > gpurun_out/${tag}_spread_$v.log 2>&1
  timeout 600 python bench.py --no-cpu-baseline --no-slam-loop --sustain-s 1 > gpurun_out/${tag}_bench_$v.log 2>&1
  rm -rf /tmp/prof_$tag$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag$v -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /dev/null 2>&1)
  f=$(find /tmp/prof_$tag$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${v}_kernel_stats.csv
done
tail -5 gpurun_out/${tag}_tests.log
for v in A B; do echo == $v; cat gpurun_out/${tag}_spread_$v.log | tail -4; tail -1 gpurun_out/${tag}_bench_$v.log | cut -c1-300; done
