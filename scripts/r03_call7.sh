#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_c7_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_c7_tests.log
timeout 900 python bench.py > gpurun_out/r03_c7_bench.log 2>&1
timeout 900 python bench.py --workload B-loop --steps 30 --no-cpu-baseline --no-slam-loop --sustain-s 2 > gpurun_out/r03_c7_bench_loop.log 2>&1
tail -4 gpurun_out/r03_c7_tests.log; tail -1 gpurun_out/r03_c7_bench.log | cut -c1-1500; tail -3 gpurun_out/r03_c7_bench_loop.log | cut -c1-1200
