#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_c6_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_c6_tests.log
bash scripts/ab_lib.sh B 2 > gpurun_out/r03_c6_ab.log 2>&1
bash scripts/pmc_ab.sh lists2 B > /dev/null 2>&1
tail -4 gpurun_out/r03_c6_tests.log; cat gpurun_out/r03_c6_ab.log; for v in A B; do echo == $v; grep -A5 "kernel stats" gpurun_out/pmcab_lists2_$v.txt; grep "pass 1" -A6 gpurun_out/pmcab_lists2_$v.txt | grep "<6, 8" | cut -c1-260; done
