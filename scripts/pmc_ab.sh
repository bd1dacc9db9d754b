#!/bin/bash
# Same-call counter A/B of two builds of the library (lib vs lib_ab) over scripts/fused_driver.py: rocprofv3 --pmc passes (kernel-trace
# only) + one --kernel-trace --stats run each; per-kernel means land in gpurun_out/pmcab_<tag>_{A,B}.txt.   usage: scripts/pmc_ab.sh <tag> [workload]
tag=${1:-x}; wl=${2:-B}
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() {
python - "$1" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get('Kernel_Name', '')
    if 'splat' not in k:
        continue
    m = re.search(r'([A-Za-z_0-9]+_kernel[0-9]*(?:_w[0-9])?(?:<[^>]*>)?)', k)
    acc[m.group(1) if m else k[:56]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    if 'render' not in k: continue
    print(k, ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())), f"(n={len(next(iter(d.values())))})")
PY
}
for v in A B; do
  if [ $v = B ]; then export SPLAT_HIP_LIB=$GRAFT_REPO_ROOT/splatam_amd/lib_ab/libsplat_hip.so; else unset SPLAT_HIP_LIB; fi
  out=$GRAFT_REPO_ROOT/gpurun_out/pmcab_${tag}_$v.txt
  : > $out
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
             "SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1)); rm -rf /tmp/pm_$v$i
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm_$v$i -o p -- python $GRAFT_REPO_ROOT/scripts/fused_driver.py $wl 3 > /tmp/pm_$v$i.log 2>&1)
    f=$(find /tmp/pm_$v$i -name "*counter_collection.csv" | head -1)
    echo "## pass $i: $set" >> $out
    if [ -z "$f" ]; then echo "no counter file; log tail:" >> $out; tail -3 /tmp/pm_$v$i.log >> $out; else summ "$f" >> $out; fi
  done
  rm -rf /tmp/ks_$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python $GRAFT_REPO_ROOT/scripts/fused_driver.py $wl 30 > /tmp/ks_$v.log 2>&1)
  f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1)
  echo "## kernel stats (30 iterations)" >> $out
  [ -n "$f" ] && python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'splat' in r['Name']:
        print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
