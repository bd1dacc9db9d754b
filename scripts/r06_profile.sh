#!/bin/bash
# round 6: ONE profile set of one state of the code, from one gpurun call: the bench line + kernel trace + counter
# passes at B and B-loop, un-profiled bench lines at A, D, E (uniform) and E-clustered, the K7 account.  Everything lands in gpurun_out/
# under <tag>; scripts/r06_collect.py <tag> (run in the repository afterwards) copies the set into profiles/, converts the counters and
# writes <tag>_meta.json (the git head of the call) and r06_workloads.md.
#   usage (on the GPU box): scripts/r05_profile.sh <tag> <git head of the tree> [quick]        (quick: counter passes at B only)
tag=${1:-r06_v1}; head=${2:-unknown}; quick=$3
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
echo "$head" > gpurun_out/${tag}_head.txt
# --- un-profiled bench lines (the default command first: what the driver runs)
timeout 900 python bench.py > gpurun_out/bench_$tag.log 2> gpurun_out/bench_$tag.err; echo "bench B rc $?"
tail -1 gpurun_out/bench_$tag.log | cut -c1-260
timeout 900 python bench.py --workload B-loop --no-slam-loop > gpurun_out/bench_${tag}_Bloop.log 2> gpurun_out/bench_${tag}_Bloop.err; echo "bench B-loop rc $?"
for wl in A D E E-clustered; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 10 --no-slam-loop --sustain-s 1 > gpurun_out/bench_${tag}_$wl.log 2> gpurun_out/bench_${tag}_$wl.err
  echo "bench $wl rc $?"; tail -1 gpurun_out/bench_${tag}_$wl.log | cut -c1-160
done
# --- kernel traces of the bench command
for wl in B B-loop; do
  sfx=$([ $wl = B ] && echo bench || echo Bloop)
  rm -rf /tmp/prof_${tag}_$sfx
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$sfx -o b -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --sustain-s 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$sfx.log 2>&1)
  f=$(find /tmp/prof_${tag}_$sfx -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${sfx}_kernel_stats.csv
done
head -12 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-160
# --- the drop-in rasterizer (BASELINE metric (i)): forward + backward per capacity policy, host vs GPU time, kernel trace of the auto policy
timeout 300 python scripts/r06_dropin.py B 2 > gpurun_out/${tag}_dropin_B.json 2> gpurun_out/${tag}_dropin_B.err; tail -1 gpurun_out/${tag}_dropin_B.json | cut -c1-300
timeout 300 python scripts/r06_dropin_host.py B 300 > gpurun_out/${tag}_dropin_host_B.json 2>/dev/null; cat gpurun_out/${tag}_dropin_host_B.json
rm -rf /tmp/prof_${tag}_dropin
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_dropin -o d -- python $GRAFT_REPO_ROOT/scripts/r06_dropin_host.py B 300 > /dev/null 2>&1)
f=$(find /tmp/prof_${tag}_dropin -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_dropin_kernel_stats.csv
# --- K7 account + XCD balance
for wl in B B-loop; do
  timeout 300 python scripts/k7_account.py $wl > gpurun_out/${tag}_k7_account_$wl.md 2> gpurun_out/${tag}_k7_account_$wl.err || tail -3 gpurun_out/${tag}_k7_account_$wl.err
done
# --- counter passes (separate rocprofv3 --pmc runs with --kernel-trace only)
bash scripts/pmc.sh ${tag} B fused > /dev/null 2>&1
[ "$quick" != "quick" ] && bash scripts/pmc.sh ${tag}_Bloop B-loop fused > /dev/null 2>&1
ls gpurun_out | grep $tag | head -40
