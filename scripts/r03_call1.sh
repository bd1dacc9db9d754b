#!/bin/bash
# round 3, GPU call 1: VALU issue microbenchmark, the whole GPU suite with the printed parity reports, baseline bench
mkdir -p gpurun_out
export TMPDIR=/tmp
(python -c "import torch; print(torch.cuda.get_device_name(0))"; nproc) > gpurun_out/r03_device.log 2>&1
timeout 300 scripts/micro/valu_issue_bench 200 > gpurun_out/r03_valu_issue.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s --timeout 900 -p no:cacheprovider > gpurun_out/r03_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r03_bench0.log 2>&1
tail -3 gpurun_out/r03_gpu_tests.log; head -30 gpurun_out/r03_valu_issue.txt; tail -2 gpurun_out/r03_bench0.log | cut -c1-600
