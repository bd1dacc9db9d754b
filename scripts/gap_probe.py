#!/usr/bin/env python3
"""Inter-kernel gaps of the fused iteration from a rocprofv3 kernel trace (run under `rocprofv3 --kernel-trace --output-format csv`):
mode 'run' executes 150 tracking + 150 mapping iterations at B; mode 'read <trace.csv>' prints, per kernel, its duration and the idle
time between the end of its predecessor and its start.  Developer tool."""
import csv
import os
import sys
import collections

if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene("B", dev, 3)
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    eng.begin_tracking(1)
    for _ in range(2):
        eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
        torch.cuda.synchronize()
        assert not eng.check_overflow()
    for _ in range(150):
        eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
    torch.cuda.synchronize()
    for _ in range(150):
        eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING)
    torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 3:]                      # steady state only
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][-60:]
        a = acc[name]
        a[0] += 1
        a[1] += e - s
        if prev_end is not None:
            a[2] += max(0, s - prev_end)
        prev_end = e
    for name, (n, dur, gap) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:62s} n={n:5d}  dur {dur / n / 1e3:8.1f} us   gap before {gap / n / 1e3:6.2f} us")
