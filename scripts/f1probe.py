#!/usr/bin/env python3
"""Timing probe for the per-Gaussian kernel F1 (developer tool): fused tracking iterations per second."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene("B", dev, 3)
eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]["cam"])
eng.begin_tracking(1)
eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
torch.cuda.synchronize()
eng.check_overflow()
for _ in range(10):
    eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
for _ in range(n):
    eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"F1 mode {os.environ.get('SPLAT_F1_DBG', '0')}: {dt * 1e6:.1f} us per tracking iteration", flush=True)
