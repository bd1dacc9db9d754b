"""Round 6: bench.py's cpu_baseline leg THROUGH THE REFERENCE'S OWN MODULES at workload B, run where the reference is (this container:
8 host cores, no GPU).  The frame is rendered by the C oracle (no GPU here); then bench.cpu_baseline_reference times the reference's
get_loss / backward / Adam around the oracle, and bench.cpu_baseline's mirror leg rides along.  usage: python scripts/r06_cpu_baseline_here.py [workload]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import c_ref  # noqa: E402
from splatam_amd import slam  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
N, W, H, fx, fy, cx, cy = bench.WORKLOADS[wl]
params, variables = slam.synthetic_params(N, W, H, fx, fy, cx, cy, num_frames=3, seed=0, device="cpu")
w2c = torch.eye(4)
cam = slam.setup_camera(W, H, [[fx, 0, cx], [0, fy, cy], [0, 0, 1]], w2c.numpy(), device="cpu")
saved = slam.Renderer
slam.Renderer = c_ref.CRasterizer
try:
    im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.5, trans_m=0.01)
finally:
    slam.Renderer = saved
frames = {1: {'cam': cam, 'im': im, 'depth': depth, 'id': 1, 'w2c': w2c}}
out = bench.cpu_baseline(wl, params, frames, budget_s=120.0)
print(json.dumps(out), flush=True)
