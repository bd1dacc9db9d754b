#!/bin/bash
# same-call A/B of process-level knobs over scripts/time_k67.py (kernel times + iteration rates at workload B)
for r in 1 2; do
echo "base:         $(timeout 200 python scripts/time_k67.py B 2>&1 | tail -1)"
echo "dev kernarg:  $(HIP_FORCE_DEV_KERNARG=1 timeout 200 python scripts/time_k67.py B 2>&1 | tail -1)"
done
