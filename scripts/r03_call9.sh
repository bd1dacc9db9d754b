#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s > gpurun_out/r03_c9_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_c9_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-slam-loop --sustain-s 1 > gpurun_out/r03_c9_bench.log 2>&1
SPLAT_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 10 --sustain-s 1 > gpurun_out/r03_c9_bench2.log 2>&1
grep -E "passed|failed|rc " gpurun_out/r03_c9_tests.log | tail -3; grep "tracking statements" gpurun_out/r03_c9_tests.log; tail -1 gpurun_out/r03_c9_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'iters_per_s' in k or k in ('value','sustained')})"; tail -1 gpurun_out/r03_c9_bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'iters_per_s' in k or 'allreduce' in k or k in ('value','sustained')})"
