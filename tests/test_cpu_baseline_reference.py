"""bench.py's cpu_baseline leg through the REFERENCE's own modules (scripts/cpu_baseline_reference.py): where /root/reference is present
(this container) the leg imports utils/slam_helpers.py, utils/slam_external.py and runs get_loss / initialize_optimizer of
scripts/splatam.py around the C oracle; elsewhere (the GPU box) it reports None and bench.py falls back to the mirror.  CPU only."""
import os

import pytest
import torch

import bench
from oracle import c_ref
from splatam_amd import slam


def _scene_a():
    N, W, H, fx, fy, cx, cy = 3000, 160, 112, 150.0, 150.0, 79.5, 55.5
    params, variables = slam.synthetic_params(N, W, H, fx, fy, cx, cy, num_frames=3, seed=0, device="cpu")
    w2c = torch.eye(4)
    cam = slam.setup_camera(W, H, [[fx, 0, cx], [0, fy, cy], [0, 0, 1]], w2c.numpy(), device="cpu")
    saved = slam.Renderer
    slam.Renderer = c_ref.CRasterizer
    try:
        im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.4, trans_m=0.01)
        cfg = slam.REPLICA_TRACKING
        p = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
        frame = {'cam': cam, 'im': im, 'depth': depth, 'id': 1, 'w2c': w2c}
        loss, _, _ = slam.get_loss(p, frame, {'max_2D_radius': torch.zeros(N)}, 1, cfg['loss_weights'], cfg['use_sil_for_loss'],
                                   cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'], tracking=True)
    finally:
        slam.Renderer = saved
    return params, {1: frame}, float(loss.detach())


def test_reference_leg_is_absent_without_the_reference(monkeypatch):
    monkeypatch.setattr(bench, "REFERENCE_DIR", "/nonexistent/reference")
    params, frames, _ = _scene_a()
    assert bench.cpu_baseline_reference("A", params, frames) is None


@pytest.mark.skipif(not os.path.isdir(os.path.join(bench.REFERENCE_DIR, "utils")), reason="the reference is not on this machine")
def test_reference_leg_runs_the_references_own_get_loss():
    """One timed tracking iteration through the reference's modules; its first loss is the mirror's loss of the same inputs (the glue
    the fused path is held to is the reference's, not only its mirror)."""
    params, frames, mirror_loss = _scene_a()
    out = bench.cpu_baseline_reference("A", params, frames, budget_s=5.0, max_iters=1)
    assert out is not None and out["kind"] == "reference glue + oracle" and out["value"] > 0
    assert "utils/slam_helpers.py" in out["sample"] and str(os.cpu_count()) in out["sample"]
    assert abs(out["first_loss"] - mirror_loss) <= 1e-5 * abs(mirror_loss), (out["first_loss"], mirror_loss)
