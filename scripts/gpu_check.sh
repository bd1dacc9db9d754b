#!/bin/bash
# One gpurun call: GPU self-tests, parity tests, smoke, bench; logs land in gpurun_out/.
# usage: scripts/gpu_check.sh [quick|full|prof]
mode=${1:-quick}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.log 2>&1
nproc >> gpurun_out/device.log
for f in test_gpu_primitives test_gpu_parity; do
  timeout 600 python -m pytest tests/$f.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/$f.log
done
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
timeout 300 python scripts/kernel_driver.py --time --reps 20 > gpurun_out/stages.log 2>&1
if [ "$mode" != "quick" ]; then
  timeout 900 python bench.py --steps 30 --warmup 10 > gpurun_out/bench.log 2>&1
fi
if [ "$mode" = "prof" ]; then
  rm -rf /tmp/prof && mkdir -p /tmp/prof gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o st -- python $GRAFT_REPO_ROOT/scripts/kernel_driver.py --reps 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_stages.log 2>&1)
  find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
  ls -la /tmp/prof/* > gpurun_out/prof_files.log 2>&1
fi
for f in gpurun_out/*.log; do echo "== $f"; tail -n 5 $f; done
