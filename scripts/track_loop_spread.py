#!/usr/bin/env python3
"""Run-to-run spread of tests/test_gpu_fused.py::test_tracking_loop_matches_reference_loop's per-iteration loss difference (the pose
optimisation amplifies summation-order noise ~10x per iteration).  Developer tool (gpurun)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from test_gpu_fused import _scene  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    params, variables, frame, cam = _scene(12000, 256, 192, aniso=False, seed=11)
    cfg = slam.REPLICA_TRACKING
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    opt = slam.initialize_optimizer(ref, cfg['lrs'], tracking=True)
    state = slam.TrackingState(ref, 1)
    eng = FusedEngine(params, cam)
    eng.begin_tracking(1)
    out = []
    for it in range(6):
        loss, _ = slam.tracking_iteration(ref, frame, dict(variables), 1, opt, state, cfg)
        eng.tracking_iteration(frame, cfg)
        out.append(abs(eng.loss() - float(loss.detach())) / abs(float(loss.detach())))
    print("rel loss diff per iteration:", " ".join(f"{v:.2e}" for v in out), flush=True)
