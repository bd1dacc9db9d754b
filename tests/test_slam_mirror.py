"""The host-side mirror (splatam_amd/slam.py) against golden vectors produced by
the REFERENCE's own Python (tests/golden/make_golden.py, run where /root/reference
exists).  CPU only; the rasterizer inside get_loss is the oracle here, exactly as
it was when the fixture was generated."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_ref as R
from splatam_amd import slam

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slam_reference.npz"))
PARAM_KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales', 'cam_unnorm_rots', 'cam_trans')


def _params(name, grad=False):
    return {k: torch.tensor(GOLD[f"{name}/param/{k}"]).requires_grad_(grad) for k in PARAM_KEYS}


@pytest.mark.parametrize("name", ["iso", "aniso"])
def test_render_variable_assembly(name):
    P = _params(name)
    tg = slam.transform_to_frame(P, 1, gaussians_grad=True, camera_grad=True)
    np.testing.assert_allclose(tg['means3D'].numpy(), GOLD[f"{name}/tg/means3D"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tg['unnorm_rotations'].numpy(), GOLD[f"{name}/tg/unnorm_rotations"], rtol=1e-5, atol=1e-6)
    rv = slam.transformed_params2rendervar(P, tg)
    for k in ('rotations', 'opacities', 'scales', 'colors_precomp'):
        np.testing.assert_allclose(rv[k].detach().numpy(), GOLD[f"{name}/rv/{k}"], rtol=1e-5, atol=1e-6)
    assert rv['means2D'].shape == P['means3D'].shape and not rv['means2D'].is_leaf
    dv = slam.transformed_params2depthplussilhouette(P, torch.eye(4), tg)
    np.testing.assert_allclose(dv['colors_precomp'].numpy(), GOLD[f"{name}/dv/colors_precomp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(slam.build_rotation(P['cam_unnorm_rots'][..., 1]).numpy(), GOLD[f"{name}/build_rotation"],
                               rtol=1e-6, atol=1e-7)


def test_detach_wiring():
    P = _params("iso", grad=True)
    tg = slam.transform_to_frame(P, 1, gaussians_grad=False, camera_grad=True)
    tg['means3D'].sum().backward()
    assert P['means3D'].grad is None and P['cam_trans'].grad is not None and P['cam_unnorm_rots'].grad is not None
    P = _params("iso", grad=True)
    tg = slam.transform_to_frame(P, 1, gaussians_grad=True, camera_grad=False)
    tg['means3D'].sum().backward()
    assert P['means3D'].grad is not None and P['cam_trans'].grad is None


def test_ssim_and_camera():
    gt = torch.tensor(GOLD["iso/gt_im"])
    assert abs(slam.calc_ssim(gt, gt * 0.9 + 0.02).item() - float(GOLD["iso/ssim"])) < 1e-6
    n, W, H, f, cx, cy = GOLD["iso/meta"]
    W, H = int(W), int(H)
    k = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]])
    cam = slam.setup_camera(W, H, k, np.eye(4), device="cpu")
    ref = R.make_camera(W, H, f, f, cx, cy)
    assert cam.image_height == H and cam.image_width == W and len(cam) == 11
    assert abs(cam.tanfovx - ref.tanfovx) < 1e-9 and abs(cam.tanfovy - ref.tanfovy) < 1e-9
    assert torch.allclose(cam.viewmatrix, ref.viewmatrix) and torch.allclose(cam.projmatrix, ref.projmatrix)
    assert not cam.viewmatrix.is_contiguous() or cam.viewmatrix.shape == (1, 4, 4)


@pytest.mark.parametrize("name", ["iso", "aniso"])
@pytest.mark.parametrize("mode", ["tracking", "mapping"])
def test_get_loss_matches_reference_code(monkeypatch, name, mode):
    monkeypatch.setattr(slam, "Renderer", R.OracleRasterizer)
    n, W, H, f, cx, cy = GOLD[f"{name}/meta"]
    n, W, H = int(n), int(W), int(H)
    cam = R.make_camera(W, H, f, f, cx, cy)
    P = {k: torch.nn.Parameter(torch.tensor(GOLD[f"{name}/param/{k}"])) for k in PARAM_KEYS}
    variables = {'max_2D_radius': torch.zeros(n), 'means2D_gradient_accum': torch.zeros(n), 'denom': torch.zeros(n),
                 'timestep': torch.zeros(n)}
    curr = {'cam': cam, 'im': torch.tensor(GOLD[f"{name}/gt_im"]), 'depth': torch.tensor(GOLD[f"{name}/gt_depth"]),
            'id': 1, 'w2c': torch.eye(4)}
    cfg = slam.REPLICA_TRACKING if mode == "tracking" else slam.REPLICA_MAPPING
    loss, variables, wl = slam.get_loss(P, curr, variables, 1, cfg['loss_weights'], cfg['use_sil_for_loss'],
                                        cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'],
                                        tracking=mode == "tracking", mapping=mode == "mapping")
    loss.backward()
    want = GOLD[f"{name}/{mode}/loss"]
    got = np.array([loss.item(), wl['im'].item(), wl['depth'].item()])
    np.testing.assert_allclose(got, want, rtol=2e-5)
    for k in PARAM_KEYS:
        ref = GOLD[f"{name}/{mode}/grad/{k}"]
        g = torch.zeros_like(P[k]) if P[k].grad is None else P[k].grad
        scale = np.abs(ref).max() + 1e-20
        assert np.abs(g.numpy() - ref).max() <= 2e-4 * scale + 1e-12, (k, np.abs(g.numpy() - ref).max(), scale)
    np.testing.assert_array_equal(variables['max_2D_radius'].numpy(), GOLD[f"{name}/{mode}/max_2D_radius"])
    ref = GOLD[f"{name}/{mode}/means2D_grad"]
    assert np.abs(variables['means2D'].grad.numpy() - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-20)


@pytest.mark.parametrize("mode", ["tracking", "mapping"])
def test_mirror_get_loss_makes_the_reference_call_sequence(mode, monkeypatch):
    """splatam_amd.slam.get_loss against the recording of the REFERENCE's get_loss at the rasterizer boundary
    (tests/golden/caller_reference.npz): same kwargs into both Renderer calls, same loss, same gradient planes back."""
    from oracle import c_ref
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caller_reference.npz"))
    n, W, H = (int(x) for x in gold["meta"][:3])
    f, cx, cy = (float(x) for x in gold["meta"][3:6])
    calls = []

    class Rec:
        def __init__(self, raster_settings):
            self.inner = c_ref.CRasterizer(raster_settings)

        def __call__(self, **kw):
            out = self.inner(**kw)
            out[0].retain_grad()
            calls.append((kw, out))
            return out
    monkeypatch.setattr(slam, "Renderer", Rec)
    params = {k[len("param/"):]: torch.nn.Parameter(torch.tensor(gold[k])) for k in gold.files if k.startswith("param/")}
    cam = slam.setup_camera(W, H, [[f, 0, cx], [0, f, cy], [0, 0, 1]], np.eye(4, dtype=np.float32), device="cpu")
    for fld in ("viewmatrix", "projmatrix"):
        assert np.array_equal(getattr(cam, fld).numpy(), gold[f"cam/{fld}"])
    variables = {'max_2D_radius': torch.zeros(n), 'means2D_gradient_accum': torch.zeros(n), 'denom': torch.zeros(n), 'timestep': torch.zeros(n)}
    curr = {'cam': cam, 'im': torch.tensor(gold["gt_im"]), 'depth': torch.tensor(gold["gt_depth"]), 'id': 1, 'w2c': torch.eye(4)}
    tracking = mode == "tracking"
    loss, variables, _ = slam.get_loss(params, curr, variables, 1, dict(im=0.5, depth=1.0), tracking, 0.99 if tracking else 0.5, True, False,
                                       tracking=tracking, mapping=not tracking)
    loss.backward()
    assert len(calls) == 2
    for ci, (kw, out) in enumerate(calls):
        assert set(kw) == {'means3D', 'colors_precomp', 'rotations', 'opacities', 'scales', 'means2D'}
        for k, v in kw.items():
            ref = gold[f"call{ci}/in/{k}"] if f"call{ci}/in/{k}" in gold.files else gold[f"call0/in/{k}"]
            np.testing.assert_allclose(v.detach().numpy(), ref, rtol=2e-6, atol=1e-7, err_msg=f"call {ci} {k}")
        g = out[0].grad if out[0].grad is not None else torch.zeros_like(out[0])
        ref = gold[f"{mode}/call{ci}/grad_out/color"]
        # the planes agree except where a loss term sits on a kink (|gt - render| ~ 0, a mask edge): a handful of pixels
        bad = np.abs(g.numpy() - ref) > 1e-6 * max(1.0, np.abs(ref).max())
        assert bad.mean() < 2e-4, (ci, bad.sum())
    assert abs(float(loss) - float(gold[f"{mode}/loss"])) <= 1e-5 * abs(float(gold[f"{mode}/loss"]))
    assert variables['means2D'] is calls[0][0]['means2D'] and variables['means2D'].grad is not None
    assert np.array_equal(variables['seen'].numpy(), gold[f"{mode}/seen"])
    assert np.array_equal(variables['max_2D_radius'].numpy(), gold[f"{mode}/max_2D_radius"])
