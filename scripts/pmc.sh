#!/bin/bash
# Counter passes (rocprofv3 --pmc, kernel-trace only) over scripts/kernel_driver.py (or scripts/fused_driver.py); per-kernel
# means land in gpurun_out/pmc_<tag>.txt.   usage: scripts/pmc.sh <tag> [workload] [kernel|fused]
tag=${1:-r1}; wl=${2:-B}; drv=${3:-kernel}
if [ "$drv" = "fused" ]; then driver="scripts/fused_driver.py $wl 3"; else driver="scripts/kernel_driver.py --workload $wl --reps 3"; fi
mkdir -p gpurun_out
bash "$(dirname "$0")/gpu_probe.sh" || exit 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
: > $out
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/$driver > /tmp/pmc_$name.log 2>&1)
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "## pass $name: $*" >> $out
  if [ -z "$f" ]; then echo "no counter file; log tail:" >> $out; tail -5 /tmp/pmc_$name.log >> $out; return; fi
  python - "$f" >> $out <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get('Kernel_Name', '')
    if 'splat' not in k:
        continue
    m = re.search(r'([A-Za-z_0-9]+_kernel[0-9]*(?:_w[0-9])?(?:<[^>]*>)?)', k)
    acc[m.group(1) if m else k[:56]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())), f"(n={len(next(iter(d.values())))})")
PY
}
pass inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
pass trans SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
cat $out
