#!/bin/bash
# round 3: parity suite on the current kernels, then same-call A/B against the round-2 kernels (lib_ab), optional microbenchmark
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r03_c2_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_c2_tests.log
bash scripts/ab_lib.sh B 3 > gpurun_out/r03_c2_ab.log 2>&1
[ -n "$1" ] && timeout 200 scripts/micro/valu_issue_bench 200 > gpurun_out/r03_valu_issue3.txt 2>&1
tail -4 gpurun_out/r03_c2_tests.log; cat gpurun_out/r03_c2_ab.log
