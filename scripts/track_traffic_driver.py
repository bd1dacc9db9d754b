#!/usr/bin/env python3
"""Target of the counter passes behind profiles/r05_track_fused_traffic.md: fused tracking iterations at a workload with the product
kernel and with its measurement builds (splat_debug_option(4, bits): 16 forward + loss only, 2 backward pass stages but visits nothing,
4 everything but the accumulator atomics) -- the builds are distinct template instances, so ONE rocprofv3 run separates them by name.
   usage: scripts/track_traffic_driver.py [workload] [iterations per build]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import _capi, slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene(wl, dev, 2)
eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
eng.begin_tracking(1)
for _ in range(3):
    eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
    torch.cuda.synchronize()
    assert not eng.check_overflow()
assert eng.tile_stride > 0
snapshot = {k: v.detach().clone() for k, v in eng.params.items()}
L = _capi.lib()
for bits in (0, 16, 2, 4):
    L.splat_debug_option(4, bits)
    for _ in range(reps):
        eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
    torch.cuda.synchronize()
    L.splat_debug_option(4, 0)
    with torch.no_grad():                    # (a measurement build leaves no usable gradient: put the pose back)
        for k, v in snapshot.items():
            eng.params[k].copy_(v)
    eng.begin_tracking(1)
    eng.buf['accum'].zero_()
    eng.buf['sums'].zero_()
print("track traffic driver ok")
