#!/bin/bash
# stress path (BASELINE config E, clustered): working tree (lib) against HEAD (lib_ab) in one call -- sort / list tests first, then
# bench.py + kernel stats at E-clustered (1 M) and E-clustered-5M for both libraries
tag=${1:-r03_s1}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_dist_pipeline.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
timeout 120 python scripts/time_radix_sort.py 2>&1 | tail -12
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so; fi
  for wl in E-clustered E-clustered-5M; do
    timeout 600 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 1 > gpurun_out/${tag}_bench_${wl}_$v.log 2>&1
    rm -rf /tmp/prof_$tag$v$wl
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag$v$wl -o b -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /dev/null 2>&1)
    f=$(find /tmp/prof_$tag$v$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_${v}_kernel_stats.csv
    echo "== $v $wl: $(tail -1 gpurun_out/${tag}_bench_${wl}_$v.log | cut -c1-200)"
  done
done
