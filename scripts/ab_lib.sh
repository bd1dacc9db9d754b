#!/bin/bash
# A/B of two builds of the HIP library inside ONE gpurun call (timings of different calls land on different boxes and differ by a
# few per cent).  Build the variant into splatam_amd/lib_ab first:   make -C splatam_amd/csrc OUTDIR=../lib_ab
# usage (on the GPU box): scripts/ab_lib.sh [workload] [rounds]
wl=${1:-B}; rounds=${2:-2}
for r in $(seq $rounds); do
  echo "A (lib):    $(timeout 200 python scripts/time_k67.py $wl 2>&1 | tail -1)"
  echo "B (lib_ab): $(SPLAT_HIP_LIB=$PWD/splatam_amd/lib_ab/libsplat_hip.so timeout 200 python scripts/time_k67.py $wl 2>&1 | tail -1)"
done
