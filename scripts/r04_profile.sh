#!/bin/bash
# round 4: the profile set of one state of the code (tag $1): [tests named in $2], K7 account (B, B-loop), PMC passes (B, B-loop), kernel stats of the
# bench command (B, B-loop), the bench line.   usage (on the GPU box): scripts/r04_profile.sh <tag> ["<pytest args>"] [quick|mid]
# (quick: no counter passes, no stress workloads; mid: counter passes at B only -- a pass takes ~2 minutes)
tag=${1:-r04_v1}; tests=$2; quick=$3
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
if [ -n "$tests" ]; then
  timeout 1200 python -m pytest $tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
  rc=$?
  echo "pytest rc $rc" >> gpurun_out/${tag}_tests.log
  tail -6 gpurun_out/${tag}_tests.log
  [ $rc -ne 0 ] && { echo "tests failed: no profile of a broken tree"; exit 1; }
fi
for wl in B B-loop; do
  timeout 300 python scripts/k7_account.py $wl > gpurun_out/${tag}_k7_account_$wl.md 2> gpurun_out/${tag}_k7_account_$wl.err || tail -3 gpurun_out/${tag}_k7_account_$wl.err
done
cat gpurun_out/${tag}_k7_account_B.md
if [ "$quick" != "quick" ]; then
  bash scripts/pmc.sh ${tag} B fused > /dev/null 2>&1
  [ -z "$quick" ] && bash scripts/pmc.sh ${tag}_Bloop B-loop fused > /dev/null 2>&1
fi
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --sustain-s 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1)
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_bench_kernel_stats.csv
rm -rf /tmp/prof_${tag}L
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}L -o b -- python $GRAFT_REPO_ROOT/bench.py --workload B-loop --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --sustain-s 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_${tag}_Bloop_prof.log 2>&1)
f=$(find /tmp/prof_${tag}L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_Bloop_kernel_stats.csv
timeout 900 python bench.py > gpurun_out/bench_$tag.log 2> gpurun_out/bench_$tag.err
echo "bench rc $?"
timeout 900 python bench.py --workload B-loop --no-slam-loop > gpurun_out/bench_${tag}_Bloop.log 2> gpurun_out/bench_${tag}_Bloop.err
echo "bench B-loop rc $?"
tail -1 gpurun_out/bench_$tag.log | cut -c1-300
head -12 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-160
if [ -z "$quick" ]; then
  for wl in E-clustered E-clustered-5M; do
    timeout 600 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --sustain-s 1 > gpurun_out/bench_${tag}_$wl.log 2>&1
    tail -1 gpurun_out/bench_${tag}_$wl.log | cut -c1-160
  done
  timeout 600 python bench.py --gpus 2 --steps 20 --warmup 10 --no-roofline --sustain-s 1 > gpurun_out/bench_${tag}_mix2_gloo.log 2>&1
  tail -1 gpurun_out/bench_${tag}_mix2_gloo.log | cut -c1-200
fi
