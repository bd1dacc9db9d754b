#!/bin/bash
# same-call A/B (lib vs lib_ab): loss / fused tests, then kernel stats + bench line of both libraries
tag=${1:-r03_q1}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
tail -3 gpurun_out/${tag}_tests.log
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-slam-loop --sustain-s 1 > gpurun_out/${tag}_bench_$v.log 2>&1
  rm -rf /tmp/prof_$tag$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag$v -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /dev/null 2>&1)
  f=$(find /tmp/prof_$tag$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${v}_kernel_stats.csv
  echo "== $v $(tail -1 gpurun_out/${tag}_bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['tracking_iters_per_s'], d['mapping_iters_per_s'])")"
done
