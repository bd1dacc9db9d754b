#!/bin/bash
# round 3: the profile set of one state of the code (tag $1): bench line, rocprofv3 kernel stats of the same command, PMC passes, B-loop, stress workloads
tag=${1:-r03_v2}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_$tag.log 2>&1
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --sustain-s 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1)
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_bench_kernel_stats.csv
bash scripts/pmc.sh $tag B fused > /dev/null 2>&1
timeout 900 python bench.py --workload B-loop --no-cpu-baseline --no-slam-loop --sustain-s 2 > gpurun_out/bench_${tag}_Bloop.log 2>&1
rm -rf /tmp/prof_${tag}L
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}L -o b -- python $GRAFT_REPO_ROOT/bench.py --workload B-loop --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --sustain-s 1 > /dev/null 2>&1)
f=$(find /tmp/prof_${tag}L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_Bloop_kernel_stats.csv
for wl in E-clustered E-clustered-5M; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --sustain-s 1 > gpurun_out/bench_${tag}_$wl.log 2>&1
  rm -rf /tmp/prof_${tag}_$wl
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$wl -o b -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --sustain-s 0.5 > /dev/null 2>&1)
  f=$(find /tmp/prof_${tag}_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_kernel_stats.csv
done
for f in gpurun_out/bench_$tag*.log; do echo == $f; tail -1 $f | cut -c1-400; done
