#!/usr/bin/env python3
"""A/B of group binning (FusedEngine.group_bins: one record per (Gaussian, 2 x 2-tile group) through an LDS histogram instead of one
returning global atomic per (Gaussian, tile) instance): fused iteration rates at a workload.  Developer tool (run through gpurun)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene(wl, dev, 3)
ref = None
for groups in (False, True, False, True):
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    eng.group_bins = groups
    eng.begin_tracking(1)
    for _ in range(2):
        eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
        torch.cuda.synchronize()
        assert not eng.check_overflow()
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize()
    g = eng.grads['means3D'].clone()
    if ref is None:
        ref = g
    else:
        print(f"  groups {groups} vs first: max |d means3D grad| {float((g - ref).abs().max()):.3e} of {float(ref.abs().max()):.3e}")

    def rate(fn, n=100):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)
    tr = rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING))
    mp = rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING))
    assert not eng.check_overflow(grow=False)
    print(f"group binning {groups}: stride {eng.tile_stride} hint {eng.max_list_hint}  tracking {tr:.0f} it/s  mapping {mp:.0f} it/s  loss {eng.loss():.6f}", flush=True)
