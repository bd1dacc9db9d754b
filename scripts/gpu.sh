#!/bin/bash
# Rebuilds the HIP library + both oracle builds (a stale .so travels to the GPU box otherwise), then runs ONE gpurun call.
#   usage: scripts/gpu.sh <timeout seconds> '<command run on the GPU box>'
set -e
cd "$(dirname "$0")/.."
make -C splatam_amd/csrc -j8 -s
make -C oracle -s
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
