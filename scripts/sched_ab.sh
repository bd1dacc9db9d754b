#!/bin/bash
# same-call A/B of process-level knobs: the Adam step of the map inside F6 (SPLAT_FUSED_ADAM_MAX_ROWS) at large maps
export TMPDIR=/tmp
for wl in B-loop E-clustered-5M; do
for rows in 500000 100000000; do
  st=10; [ $wl = B-loop ] && st=50
  echo "$wl max_rows=$rows: $(SPLAT_FUSED_ADAM_MAX_ROWS=$rows timeout 300 python bench.py --workload $wl --steps $st --warmup 5 --no-cpu-baseline --no-slam-loop --no-roofline --engine fused --sustain-s 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['tracking_iters_per_s'], d['mapping_iters_per_s'])")"
done; done
