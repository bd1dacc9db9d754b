#!/bin/bash
# what the driver runs at round end: the GPU suite, smoke(), the default bench line (timed)
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/check_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/check_tests.log
tail -3 gpurun_out/check_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
s=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check_bench.log 2>&1
echo "bench rc $? in $(( $(date +%s) - s )) s"
tail -1 gpurun_out/check_bench.log | cut -c1-330
