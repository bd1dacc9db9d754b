#!/bin/bash
# full GPU suite, then the profile set of the state (tag $1), then the F4 tile-order variant (lib_ab) kernel stats
tag=${1:-r03_v3}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
bash scripts/r03_profile.sh $tag
export SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so
rm -rf /tmp/prof_${tag}_ab
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_ab -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 0.5 > /dev/null 2>&1)
f=$(find /tmp/prof_${tag}_ab -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_f4plain_kernel_stats.csv
grep -E "ssim_forward|map_loss_backward" gpurun_out/${tag}_bench_kernel_stats.csv gpurun_out/${tag}_f4plain_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
