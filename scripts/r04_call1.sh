#!/bin/bash
# round 4, call 1: GPU suite, then the bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/c1_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/c1_tests.log
tail -25 gpurun_out/c1_tests.log
s=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c1_bench.log 2> gpurun_out/c1_bench.err
echo "bench rc $? in $(( $(date +%s) - s )) s"
tail -3 gpurun_out/c1_bench.err
tail -1 gpurun_out/c1_bench.log | cut -c1-600
