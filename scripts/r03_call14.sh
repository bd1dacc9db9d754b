#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03_c14
timeout 900 python -m pytest tests/test_gpu_dist_pipeline.py tests/test_gpu_primitives.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
wl=E-clustered-5M
timeout 600 python bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 1 > gpurun_out/${tag}_bench_${wl}.log 2>&1
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /dev/null 2>&1)
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_kernel_stats.csv
echo "== $wl: $(tail -1 gpurun_out/${tag}_bench_${wl}.log | cut -c1-200)"
timeout 900 python bench.py --workload B-loop --no-cpu-baseline --no-slam-loop > gpurun_out/${tag}_bench_Bloop.log 2>&1
echo "== B-loop: $(tail -1 gpurun_out/${tag}_bench_Bloop.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['sustained'])")"
SPLAT_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 10 --sustain-s 1 --no-cpu-baseline --no-slam-loop > gpurun_out/${tag}_bench2.log 2>&1
echo "== 2 ranks (gloo, one GPU): $(tail -1 gpurun_out/${tag}_bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'iters_per_s' in k or 'allreduce' in k or k in ('value','collectives')})")"
