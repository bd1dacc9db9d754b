#!/bin/bash
# single SSIM kernel (F4 + F5 fused) vs HEAD (lib_ab): tests first, then the bench line + kernel stats of both libraries in one call
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_plugin.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/r03_c10_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_c10_tests.log
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-slam-loop --sustain-s 1 > gpurun_out/r03_c10_bench_$v.log 2>&1
  rm -rf /tmp/prof_c10$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c10$v -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /dev/null 2>&1)
  f=$(find /tmp/prof_c10$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_c10_${v}_kernel_stats.csv
done
tail -5 gpurun_out/r03_c10_tests.log
for v in A B; do echo == $v; tail -1 gpurun_out/r03_c10_bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'iters_per_s' in k or k in ('value','sustained')})"; head -12 gpurun_out/r03_c10_${v}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150; done
