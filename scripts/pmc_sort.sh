#!/bin/bash
# counters of ONE LDS sort workgroup (splat_selftest through scripts/time_radix_sort.py): where does a run sort's time go?
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python scripts/time_radix_sort.py 2>&1 | head -6
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1)); rm -rf /tmp/pms_$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pms_$i -o p -- python $GRAFT_REPO_ROOT/scripts/time_radix_sort.py > /tmp/pms_$i.log 2>&1)
  f=$(find /tmp/pms_$i -name "*counter_collection.csv" | head -1)
  echo "## pass $i"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    if 'selftest' not in k: continue
    acc[k[k.find('selftest'):][:60] + ' grid' + r.get('Grid_Size', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k, ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())), f"(n={len(next(iter(d.values())))})")
PY
done
