"""Parity at EVERY configuration BASELINE.json names, at full size, against the C oracle (oracle/raster_ref.c):

  A  10 000 Gaussians, 320x240                      B  300 000, 1200x680 (/root/reference/configs/data/replica.yaml:3-8)
  D  150 000, 640x480, fx 517.3 / fy 516.5, cx 318.6 / cy 255.3 (/root/reference/configs/data/TUM/freiburg1_desk.yaml:3-8)
  E  1 000 000, 1752x1168 (/root/reference/datasets/gradslam_datasets/scannetpp.py:28-29) + the CLUSTERED variant of
     SURVEY.md 8(d): every Gaussian inside 5 % of the image, per-tile lists far beyond the 4 096 keys one workgroup sorts
     in LDS (the "lists spilling to HBM" stress of BASELINE config 5)

through (1) the drop-in surface (GaussianRasterizer: forward + all six gradients) and (2) the fused iteration
(FusedEngine: six rendered planes, loss, every gradient) -- the latter compared DIRECTLY with the oracle's two renders and
two backward passes through the reference-shaped get_loss (splatam_amd.slam.get_loss running on CPU tensors with the C
oracle as its Renderer), not with the HIP drop-in path.

Tolerances: 1e-4 colour / depth; lists / radii exact; gradients: the north star's 1e-3 of the tensor's maximum AND, per
element, no further from the float64 evaluation of the oracle than 1.3x the float32 oracle itself is (tests/util.py:
assert_grad_calibrated -- a flat per-element 1e-3 is not attainable by ANY float32 evaluation of this backward pass: the
float32 and float64 builds of the oracle differ by more than that on 0.2-0.7 % of the elements)."""
import numpy as np
import pytest
import torch

from oracle import c_ref
from oracle import raster_ref as R
from tests.util import (assert_close_outliers, assert_grad_calibrated, assert_grad_outliers_explained, assert_outliers_explained,
                        flip_pixels, oracle_flip_bounds)

pytestmark = pytest.mark.gpu

CONFIGS = {
    'A': dict(n=10_000, W=320, H=240, fx=300.0, fy=300.0, cx=159.5, cy=119.5),
    'B': dict(n=300_000, W=1200, H=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5),
    'D': dict(n=150_000, W=640, H=480, fx=517.3, fy=516.5, cx=318.6, cy=255.3),
    'E': dict(n=1_000_000, W=1752, H=1168, fx=1200.0, fy=1200.0, cx=875.5, cy=583.5),
}
CLUSTER = (0.40, 0.40, 0.40 + 0.2236, 0.40 + 0.2236)        # 5 % of the image area


def _scene(cfg, seed=0, aniso=False, region=None, bg=(0.0, 0.0, 0.0), scale_modifier=1.0):
    c = CONFIGS[cfg]
    cam = R.make_camera(c['W'], c['H'], c['fx'], c['fy'], c['cx'], c['cy'], bg=bg)._replace(scale_modifier=scale_modifier)
    p = R.synthetic_cloud(c['n'], c['W'], c['H'], c['fx'], c['fy'], c['cx'], c['cy'], seed=seed, anisotropic=aniso, region=region)
    return cam, R.cloud_to_rendervar(p)


def _cuda_settings(cam):
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    return Camera(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  bg=cam.bg.cuda(), scale_modifier=cam.scale_modifier, viewmatrix=cam.viewmatrix.cuda(),
                  projmatrix=cam.projmatrix.cuda(), sh_degree=cam.sh_degree, campos=cam.campos.cuda(), prefiltered=cam.prefiltered)


KEYS = ('means3D', 'means2D', 'opacities', 'colors_precomp', 'scales', 'rotations')
GRAD_MAP = [('means3D', 'means3D'), ('means2D', 'means2D'), ('colors_precomp', 'colors'), ('opacities', 'opacities'),
            ('scales', 'scales'), ('rotations', 'rotations')]


def _dropin(cam, rv, gout, path="second"):
    """The call under test is the scene's SECOND one: the first (exact lists, learns the scene's longest list) is what every loop pays
    once; "fast" asserts that the second then ran on group binning + the sorting composite (the "auto" policy's steady state)."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from splatam_amd import rasterizer as rz
    inp = {k: rv[k].detach().cuda().requires_grad_(True) for k in KEYS}
    cs = _cuda_settings(cam)
    if path in ("fast", "second"):
        with torch.no_grad():
            Renderer(raster_settings=cs)(**inp)
    before = dict(rz.fast_path_stats)
    color, radii, depth = Renderer(raster_settings=cs)(**inp)
    took = "fast" if rz.fast_path_stats["fast"] == before["fast"] + 1 else "exact"
    print(f"[dropin] {path}: took the {took} path, scene statistics {list(rz._scene_stats.values())}")
    if path != "second":                # ("second": whatever the policy decides for this scene's second call -- long lists stay exact)
        assert took == path, (path, before, rz.fast_path_stats)
    (color * gout.cuda()).sum().backward()
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(), {k: inp[k].grad.cpu().numpy() for k in KEYS}


def _oracle(cam, rv, gout, precision="f32"):
    cr = c_ref.CRef(precision)
    col, radii, dep = cr.forward(rv['means3D'].numpy(), rv['colors_precomp'].numpy(), rv['opacities'].numpy(), rv['scales'].numpy(),
                                 rv['rotations'].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy,
                                 cam.image_width, cam.image_height, cam.bg.numpy(), scale_modifier=cam.scale_modifier)
    return col, radii, dep, cr.backward(gout.numpy()), cr


def _check_images(gc, gr, gd, oc, orad, od, cam, rv):
    """1e-4 on colour and depth; every pixel beyond it explained by a float32 decision flip the float64 oracle finds at that pixel
    (tests/util.py: assert_outliers_explained), their number bounded.  Returns (flagged pixels, xy, radii) for the gradient check."""
    assert (gr != orad).sum() <= max(2, int(1e-5 * gr.size)), "radii mismatch"
    assert np.abs(gr.astype(np.int64) - orad).max() <= 1
    bound, margin, xy, radii, noise = oracle_flip_bounds(rv, cam)
    C_ = gc.shape[0]
    n = assert_outliers_explained(gc, oc, bound[:C_], 1e-4, rtol=1e-4, noise=noise[:C_], what="color")
    n += assert_outliers_explained(gd, od, bound[C_:C_ + 1], 1e-4, rtol=1e-4, noise=noise[C_:C_ + 1], what="depth")
    assert n <= max(2, int(1e-4 * (gc.size + gd.size))), n
    return flip_pixels(bound, (gc, gd), (oc, od)), xy, radii


def _check_dropin_grads(gg, og32, og64, what, aniso, flips=None):
    for k, ok in GRAD_MAP:
        if ok == 'rotations' and not aniso:
            # isotropic scales: Sigma = s^2 I does not depend on the quaternion; all three evaluations give rounding noise
            continue
        if flips is not None:           # rows beyond 1e-3 of the maximum lie over a float32 decision flip
            assert_grad_outliers_explained(gg[k].reshape(og32[ok].shape), og32[ok], *flips, what=f"{what} grad {k}")
        assert_grad_calibrated(gg[k].reshape(og32[ok].shape), og32[ok], og64[ok], what=f"{what} grad {k}")


@pytest.mark.parametrize("cfg,aniso,path", [('D', False, "fast"), ('D', True, "second"), ('B', True, "second"), ('E', False, "fast"), ('E', True, "second"),
                                            ('D', False, "exact"), ('B', True, "exact")])
def test_dropin_full_size(cfg, aniso, path):
    """Drop-in forward + backward vs the C oracle at the BASELINE configurations round 1 left uncovered, on both paths of the default
    capacity policy (_dropin)."""
    cam, rv = _scene(cfg, seed=7, aniso=aniso)
    gout = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(1))
    gc, gr, gd, gg = _dropin(cam, rv, gout, path)
    oc, orad, od, og, _ = _oracle(cam, rv, gout)
    flips = _check_images(gc, gr, gd, oc, orad, od, cam, rv)
    og64 = _oracle(cam, rv, gout, "f64")[3]
    _check_dropin_grads(gg, og, og64, f"{cfg}{'-aniso' if aniso else ''}", aniso, flips)


@pytest.mark.parametrize("cfg,aniso,bg,mod", [('D', True, (1.0, 1.0, 1.0), 1.0), ('B', False, (0.2, 0.6, 1.0), 1.0), ('D', True, (0.0, 0.0, 0.0), 1.6)])
def test_dropin_full_size_viewer_settings(cfg, aniso, bg, mod):
    """The settings the reference's viewers render with -- a white background (/root/reference/viz_scripts/final_recon.py:110-122: bg
    enters the forward as C + T bg and the backward through dL/dalpha) -- and a scale modifier other than 1 (the settings tuple's
    sixth field, /root/reference/utils/recon_helpers.py:20), at full size: forward and all six gradients vs the C oracle."""
    cam, rv = _scene(cfg, seed=11, aniso=aniso, bg=bg, scale_modifier=mod)
    gout = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3))
    gc, gr, gd, gg = _dropin(cam, rv, gout)
    oc, orad, od, og, _ = _oracle(cam, rv, gout)
    flips = _check_images(gc, gr, gd, oc, orad, od, cam, rv)
    og64 = _oracle(cam, rv, gout, "f64")[3]
    _check_dropin_grads(gg, og, og64, f"{cfg}{'-aniso' if aniso else ''} bg {bg} modifier {mod}", aniso, flips)


def test_dropin_clustered_lists_beyond_lds():
    """Config E, clustered: 1 M Gaussians inside 5 % of the image.  The longest per-tile lists are far beyond what one
    workgroup sorts in LDS; the sorted lists must still be the oracle's, bit for bit."""
    from splatam_amd import rasterizer as rz
    cam, rv = _scene('E', seed=3, region=CLUSTER)
    cs = _cuda_settings(cam)
    empty = torch.empty(0, device="cuda")
    col, radii, dep, pk = rz.rasterize_forward(cs, rv['means3D'].cuda(), rv['colors_precomp'].cuda(), rv['opacities'].cuda().reshape(-1),
                                               rv['scales'].cuda(), rv['rotations'].cuda(), empty, empty)
    torch.cuda.synchronize()
    gout = torch.randn(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(2))
    oc, orad, od, og, cr = _oracle(cam, rv, gout)
    base = cr.ranges()
    longest = int(np.diff(base).max())
    print(f"clustered E: {cr.num_rendered()} instances, longest list {longest}")
    assert longest > 4096, longest
    assert pk.num_rendered == cr.num_rendered()
    assert (pk.tensors['tile_base'].cpu().numpy() == base).all()
    assert (pk.tensors['point_list'].cpu().numpy()[:pk.num_rendered] == cr.point_list()).all()
    flips = _check_images(col.cpu().numpy(), radii.cpu().numpy(), dep.cpu().numpy(), oc, orad, od, cam, rv)
    gc, gr, gd, gg = _dropin(cam, rv, gout)
    og64 = _oracle(cam, rv, gout, "f64")[3]
    _check_dropin_grads(gg, og, og64, "clustered-E", False, flips)


# ---------------------------------------------------------------------------------------------------------------------
# fused iteration vs the oracle's two renders + two backward passes
# ---------------------------------------------------------------------------------------------------------------------
# The comparison is staged, because get_loss is not smooth: |gt - render| has a kink wherever a pixel matches its target
# to float32 rounding and the masks are step functions of the silhouette / depth, so two correct evaluations legitimately
# disagree on the SIGN or the MASK of a few dozen of the ~2.4 M per-pixel loss terms of a full-size frame, and every such
# pixel moves the gradient of the ~50 Gaussians under it by a few per cent.
#   (A) the six rendered planes                       vs the oracle's two renders            1e-4
#   (B) the loss value                                vs the oracle's get_loss               1e-4 relative
#   (C) the per-pixel gradient planes dL/d(render)    vs autograd of the oracle's get_loss   equal except at kink pixels (counted)
#   (D) every parameter / pose gradient               vs the oracle's two BACKWARD passes driven by the SAME gradient planes
#       (float32 and float64 builds), i.e. the whole render-backward + glue adjoint with the kinks taken out: calibrated
#       per-element check (tests/util.py: assert_grad_calibrated)

class _Spy:
    """Wraps c_ref.CRasterizer: keeps the rendered images (with retain_grad) of every call."""
    renders = []

    def __init__(self, raster_settings):
        self.inner = c_ref.CRasterizer(raster_settings)

    def __call__(self, **kw):
        out = self.inner(**kw)
        if out[0].requires_grad:
            out[0].retain_grad()
        _Spy.renders.append(out[0])
        return out


def _cpu_case(params, frame, cam_args, dtype):
    from splatam_amd import slam
    W, H, k = cam_args
    pc = {k_: torch.nn.Parameter(v.detach().cpu().to(dtype).clone()) for k_, v in params.items()}
    cam_c = slam.setup_camera(W, H, k, np.eye(4, dtype=np.float32), device="cpu")
    frame_c = {'cam': cam_c, 'im': frame['im'].cpu().to(dtype), 'depth': frame['depth'].cpu().to(dtype), 'id': 1,
               'w2c': torch.eye(4, dtype=dtype)}
    return pc, frame_c


def _oracle_get_loss(params, frame, variables, cam_args, cfg, tracking, monkeypatch):
    """splatam_amd.slam.get_loss (pinned to /root/reference/scripts/splatam.py:214-347 by tests/golden/) on CPU tensors with the
    C oracle as its Renderer: returns (loss, [rgb render, depth/sil render], their autograd gradients)."""
    from splatam_amd import slam
    monkeypatch.setattr(slam, "Renderer", _Spy)
    _Spy.renders = []
    pc, frame_c = _cpu_case(params, frame, cam_args, torch.float32)
    vc = {k_: v.cpu().clone() for k_, v in variables.items()}
    loss, _, _ = slam.get_loss(pc, frame_c, vc, 1, cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                               cfg['ignore_outlier_depth_loss'], tracking=tracking, mapping=not tracking)
    loss.backward()
    im, ds = _Spy.renders
    zero = torch.zeros_like(ds)
    return float(loss.detach()), [im.detach(), ds.detach()], [im.grad, ds.grad if ds.grad is not None else zero]


def _oracle_backward_from_planes(params, frame, cam_args, planes, tracking, dtype, monkeypatch):
    """The oracle's two renders + two backward passes for GIVEN gradient planes (dL/drgb [3], dL/ddepth [1]) through the
    reference-shaped glue (transform_to_frame, render-variable assembly): gradients of every parameter."""
    from splatam_amd import slam
    monkeypatch.setattr(slam, "Renderer", c_ref.CRasterizer)
    pc, frame_c = _cpu_case(params, frame, cam_args, dtype)
    tg = slam.transform_to_frame(pc, 1, gaussians_grad=not tracking, camera_grad=tracking)
    im, _, _ = slam.Renderer(raster_settings=frame_c['cam'])(**slam.transformed_params2rendervar(pc, tg))
    ds, _, _ = slam.Renderer(raster_settings=frame_c['cam'])(**slam.transformed_params2depthplussilhouette(pc, frame_c['w2c'], tg))
    pl = planes.to(dtype)
    ((im * pl[0:3]).sum() + (ds[0] * pl[3]).sum()).backward()
    return {k_: (None if v.grad is None else v.grad.numpy()) for k_, v in pc.items()}


def _assert_depth_ties_explain(big, params, c, what, time_idx=1, max_pixels=400):
    """Every pixel of the boolean image `big` lies under two Gaussians whose float64 camera-space depths differ by less than
    4 float32 ulps (a depth tie: their order is decided by rounding)."""
    ys, xs = np.nonzero(big)
    print(f"{what}: {ys.size} pixels beyond the one-decision bound")
    if ys.size == 0:
        return 0
    assert ys.size <= max_pixels, (what, ys.size)
    q = params['cam_unnorm_rots'][0, :, time_idx].detach().double().cpu().numpy()
    t = params['cam_trans'][0, :, time_idx].detach().double().cpu().numpy()
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    X = params['means3D'].detach().double().cpu().numpy() @ Rm.T + t
    zc = X[:, 2]
    u, v = c['fx'] * X[:, 0] / zc + c['cx'], c['fy'] * X[:, 1] / zc + c['cy']
    for py, px in zip(ys, xs):
        near = np.nonzero((np.abs(u - px) < 12) & (np.abs(v - py) < 12) & (zc > 0.2))[0]
        zs = np.sort(zc[near])
        gaps = (zs[1:] - zs[:-1]) / zs[1:]
        assert gaps.size and gaps.min() < 4 * 2.0 ** -24, (what, "pixel", int(px), int(py), "smallest relative depth gap", gaps.min() if gaps.size else None)
    return int(ys.size)


def _fused_case(cfg_name, aniso, tracking, monkeypatch, seed=0, region=None):
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    c = CONFIGS[cfg_name]
    n, W, H = c['n'], c['W'], c['H']
    params, variables = slam.synthetic_params(n, W, H, c['fx'], c['fy'], c['cx'], c['cy'], num_frames=3, seed=seed, device="cuda",
                                              anisotropic=aniso, region=region)
    k = [[c['fx'], 0, c['cx']], [0, c['fy'], c['cy']], [0, 0, 1]]
    w2c = torch.eye(4, device="cuda")
    cam = slam.setup_camera(W, H, k, np.eye(4, dtype=np.float32), device="cuda")
    im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.4, trans_m=0.01)
    g = torch.Generator().manual_seed(seed + 1)
    im = (im + 0.03 * torch.randn(im.shape, generator=g).cuda()).clamp(0, 1).contiguous()
    depth = (depth * (1 + 0.01 * torch.randn(depth.shape, generator=g).cuda())).contiguous()
    depth[:, : H // 8, : W // 8] = 0.0
    with torch.no_grad():
        params['cam_unnorm_rots'][0, :, 1] = torch.tensor([0.98, 0.01, -0.02, 0.015], device="cuda") * 1.1
        params['cam_trans'][0, :, 1] = torch.tensor([0.01, -0.02, 0.015], device="cuda")
    frame = {'cam': cam, 'im': im, 'depth': depth, 'id': 1, 'w2c': w2c}
    cfg = slam.REPLICA_TRACKING if tracking else slam.REPLICA_MAPPING          # the SHIPPED thresholds (0.99 / 0.5)
    eng = FusedEngine(params, cam)
    eng.loss_backward(frame, 1, cfg, tracking=tracking)
    torch.cuda.synchronize()
    assert not eng.check_overflow(grow=False)
    what = f"fused {cfg_name}{'-aniso' if aniso else ''}{'-clustered' if region else ''} {'tracking' if tracking else 'mapping'}"
    cam_args = (W, H, k)
    loss_ref, renders, plane_grads = _oracle_get_loss(params, frame, variables, cam_args, cfg, tracking, monkeypatch)
    # (A) rendered planes
    imf, depthf, silf, dsqf = eng.rendered()
    nflip = 2e-4            # the fused glue rounds differently from torch's: a few more alpha >= 1/255 decisions flip than on the drop-in path
    got_im, got_ds = imf.cpu().numpy(), torch.cat([depthf, silf[None], dsqf]).cpu().numpy()
    # every pixel beyond 1e-4 explained by a decision the float64 oracle (float64 glue) finds within rounding of its threshold there
    pc64, frame64 = _cpu_case(params, frame, cam_args, torch.float64)
    with torch.no_grad():
        tg64 = slam.transform_to_frame(pc64, 1, gaussians_grad=False, camera_grad=False)
        b_im, _, _, _, n_im = oracle_flip_bounds(slam.transformed_params2rendervar(pc64, tg64), frame64['cam'])
        b_ds, _, _, _, n_ds = oracle_flip_bounds(slam.transformed_params2depthplussilhouette(pc64, frame64['w2c'], tg64), frame64['cam'])
    nbad = assert_outliers_explained(got_im, renders[0].numpy(), b_im[:3], 1e-4, noise=n_im[:3], what=f"{what} im")
    nbad += assert_outliers_explained(got_ds, renders[1].numpy(), b_ds[:3], 1e-4, rtol=1e-4, noise=n_ds[:3], what=f"{what} depth/sil/depth^2")
    assert nbad <= nflip * (got_im.size + got_ds.size), nbad
    # one alpha >= 1/255 decision moves a pixel by <= ~1/255 |c|; anything larger must be a DEPTH TIE: two overlapping Gaussians
    # whose camera-space depths agree to float32 rounding are ordered by that rounding, and the in-kernel glue (FMA chain) rounds
    # z = (R X + t).z differently from torch's matmul -- a legitimate swap of two list neighbours, verified per pixel
    big = (np.abs(got_im - renders[0].numpy()).max(axis=0) > 0.03) | (np.abs(got_ds - renders[1].numpy()).max(axis=0) > 0.3)
    eng.depth_tie_pixels = _assert_depth_ties_explain(big, params, c, what)
    # (B) loss
    loss_f = eng.loss()
    assert abs(loss_f - loss_ref) <= 1e-4 * abs(loss_ref), (what, loss_f, loss_ref)
    # (C) gradient planes: equal except at the kinks of the L1 terms / mask edges
    planes = eng.buf['dL_dout6'].detach().cpu()
    ref_planes = torch.cat([plane_grads[0], plane_grads[1][0:1]])
    pmax = float(ref_planes.abs().max())
    # (round 3: the printed reports show at most 1.0e-4 of the plane elements at a kink -- D mapping -- hence 1.5e-4, was 1e-3; a kink
    #  flips the sign of one L1 term: the element moves by exactly twice its magnitude, 2 pmax at most -- round 4: 2.05, was 2.5)
    assert_close_outliers(planes[0:4].numpy(), ref_planes.numpy(), 1e-4 * pmax, rtol=1e-3, max_outlier_frac=1.5e-4, outlier_atol=2.05 * pmax,
                          what=f"{what} dL/d(render) planes")
    assert float(planes[4:6].abs().max()) == 0.0 and float(plane_grads[1][1:3].abs().max()) == 0.0
    # (D) parameter / pose gradients for the SAME gradient planes, float32 and float64 oracle
    g32 = _oracle_backward_from_planes(params, frame, cam_args, planes, tracking, torch.float32, monkeypatch)
    g64 = _oracle_backward_from_planes(params, frame, cam_args, planes, tracking, torch.float64, monkeypatch)
    return eng, g32, g64, what


@pytest.mark.parametrize("cfg_name,aniso", [('B', False), ('B', True), ('D', False), ('D', True), ('E', False), ('E', True)])
def test_fused_mapping_vs_oracle(cfg_name, aniso, monkeypatch):
    eng, g32, g64, what = _fused_case(cfg_name, aniso, False, monkeypatch)
    keys = ["means3D", "rgb_colors", "logit_opacities", "log_scales"] + (["unnorm_rotations"] if aniso else [])
    # a depth tie (two list neighbours swapped, verified above) moves the gradients of the Gaussians under it: the 99.99 % quantile
    # of a 150 k-row tensor is 15 elements, so that one swap IS the tail there (D-anisotropic: log_scales 2.3x the oracle's own tail)
    # (round 4: the worst case measured under a verified tie is 2.3x -- D-anisotropic log_scales -- + 30 % = 3.0, was 4.0)
    tail = 2.0 if eng.depth_tie_pixels == 0 else 3.0
    for k in keys:
        assert_grad_calibrated(eng.grads[k].cpu().numpy(), g32[k], g64[k], what=f"{what} grad {k}", tail_factor=tail)
    if not aniso:
        assert float(eng.grads["unnorm_rotations"].abs().max()) == 0.0


def _check_pose_gradient(eng, g32, g64, what):
    d = eng.buf['d_cam'].cpu().numpy().astype(np.float64)
    gq32, gt32 = g32['cam_unnorm_rots'][0, :, 1], g32['cam_trans'][0, :, 1]
    gq64, gt64 = g64['cam_unnorm_rots'][0, :, 1], g64['cam_trans'][0, :, 1]
    print(what, "pose gradient", d[0:7], "oracle f32", gq32, gt32, "oracle f64", gq64, gt64)
    for got, r32, r64 in ((d[0:4], gq32, gq64), (d[4:7], gt32, gt64)):
        tol = max(1e-4 * np.abs(r64).max(), 2.0 * np.abs(r32 - r64).max())
        assert np.abs(got - r64).max() <= tol, (what, got, r64, tol)


@pytest.mark.parametrize("cfg_name,aniso", [('B', False), ('B', True), ('D', False), ('D', True), ('E', False), ('E', True)])
def test_fused_tracking_vs_oracle(cfg_name, aniso, monkeypatch):
    """Tracking at the SHIPPED sil_thres = 0.99 (round 1 compared at a threshold moved into a gap of the silhouette histogram).
    D and E are the configurations that run 200 tracking iterations per frame (/root/reference/configs/tum/splatam.py:14,
    /root/reference/configs/scannetpp/splatam.py:30)."""
    eng, g32, g64, what = _fused_case(cfg_name, aniso, True, monkeypatch)
    _check_pose_gradient(eng, g32, g64, what)


def test_fused_clustered_tracking_vs_oracle(monkeypatch):
    """Tracking on the clustered stress scene: the exact-list + multi-workgroup-sort path feeding the tracking form of the
    backward composite."""
    eng, g32, g64, what = _fused_case('E', False, True, monkeypatch, seed=5, region=CLUSTER)
    _check_pose_gradient(eng, g32, g64, what)


def test_fused_clustered_vs_oracle(monkeypatch):
    """The clustered stress scene through the fused iteration (long lists: exact-list path, multi-workgroup sort)."""
    eng, g32, g64, what = _fused_case('E', False, False, monkeypatch, seed=5, region=CLUSTER)
    for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales"):
        assert_grad_calibrated(eng.grads[k].cpu().numpy(), g32[k], g64[k], what=f"{what} grad {k}")
