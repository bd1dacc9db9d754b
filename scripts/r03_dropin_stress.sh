#!/bin/bash
# drop-in path on very long exact lists (dense K1 count + scatter): list tests, then kernel stats of bench.py at the clustered stress workloads
tag=${1:-r03_d1}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
tail -3 gpurun_out/${tag}_tests.log
for wl in E-clustered E-clustered-5M; do
  rm -rf /tmp/prof_$tag$wl
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag$wl -o b -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --sustain-s 0.5 > /tmp/$tag$wl.log 2>&1)
  f=$(find /tmp/prof_$tag$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_kernel_stats.csv
  echo "== $wl: $(tail -1 /tmp/$tag$wl.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['dropin_iters_per_s'])")"
done
