#!/usr/bin/env python3
"""A/B of the tracking form of the backward composite (splat_debug_option(2, 3) = with the opacity sum, one list entry per
loop trip; 2 = without it, two entries per trip): fused tracking iterations per second at workload B.  Developer tool."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from splatam_amd import _capi, slam
from splatam_amd.fused import FusedEngine
dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene("B", dev, 3)
L = _capi.lib()
for mode in (3, 2, 3, 2):
    L.splat_debug_option(2, mode)
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    eng.begin_tracking(1)
    eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
    torch.cuda.synchronize(); assert not eng.check_overflow()
    for _ in range(10): eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print("mode", mode, "(3 = with opacity sum, one entry per trip; 2 = without, pairs)", round(dt * 1e6, 1), "us/iter", round(1 / dt), "it/s", "d_cam", [round(float(x), 6) for x in eng.buf['d_cam'][:8]], flush=True)
