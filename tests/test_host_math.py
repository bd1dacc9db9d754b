"""The per-Gaussian arithmetic of the HIP kernels (splatam_amd/csrc/splat_math.h)
compiled for the host with g++ and compared with the oracle -- CPU only, no
kernel launches.  Catches projection / adjoint mistakes without GPU time."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import raster_ref as R
from tests.util import scene, tilted_w2c

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    out = os.path.join(HERE, "_build", "libhost_math_shim.so")
    src = os.path.join(HERE, "host_math_shim.cpp")
    hdrs = [os.path.join(HERE, "..", "splatam_amd", "csrc", h) for h in ("splat_math.h", "fused_math.h")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src])
    return C.CDLL(out)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def _np(t):
    return np.ascontiguousarray(t.detach().numpy(), dtype=np.float32)


@pytest.mark.parametrize("aniso,view", [(False, False), (True, True)])
def test_projection_forward_and_adjoint(shim, aniso, view):
    W, H = 160, 112
    cam, rv = scene(2000, W, H, 150.0, seed=4, anisotropic=aniso, w2c=tilted_w2c() if view else None)
    # push a few Gaussians outside the 1.3x guard band and behind the near plane
    rv['means3D'][:20, 0] += 6.0
    rv['means3D'][20:30, 2] = 0.1
    P = rv['means3D'].shape[0]
    m, s, q = (rv[k].clone().requires_grad_(True) for k in ('means3D', 'scales', 'rotations'))
    m2 = torch.zeros(P, 3, requires_grad=True)
    geom = R.preprocess(m, m2, s, q, None, cam)
    vis = geom.radii > 0
    g = torch.Generator().manual_seed(0)
    wxy = torch.randn(P, 2, generator=g)
    wcon = torch.randn(P, 3, generator=g)
    loss = ((geom.xy * wxy).sum(1) + (geom.conic * wcon).sum(1))[vis].sum()
    loss.backward()

    view_f, proj_f = _np(cam.viewmatrix).reshape(-1), _np(cam.projmatrix).reshape(-1)
    mm, ss, qq = _np(m), _np(s), _np(q)
    depth = np.zeros(P, np.float32); xy = np.zeros((P, 2), np.float32); conic = np.zeros((P, 3), np.float32)
    radii = np.zeros(P, np.int32); rect = np.zeros((P, 4), np.int32)
    shim.hm_forward(P, _p(mm), _p(ss), _p(qq), _p(view_f), _p(proj_f), W, H, C.c_float(cam.tanfovx), C.c_float(cam.tanfovy),
                    C.c_float(1.0), _p(depth), _p(xy), _p(conic), _p(radii, C.c_int), _p(rect, C.c_int))
    v = vis.numpy()
    assert (radii == geom.radii.numpy()).all()
    assert (radii[20:30] == 0).all()
    np.testing.assert_allclose(xy[v], geom.xy.detach().numpy()[v], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(conic[v], geom.conic.detach().numpy()[v], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(depth[v], geom.depth.numpy()[v], rtol=1e-6)
    assert (rect[:, :2] == geom.rect_min.numpy()).all() and (rect[:, 2:] == geom.rect_max.numpy()).all()

    # adjoint: feed dL/dNDC and the true dL/dconic
    g_ndc = np.ascontiguousarray((wxy * torch.tensor([0.5 * W, 0.5 * H])).numpy(), dtype=np.float32)
    g_con = _np(wcon)
    dmean = np.zeros((P, 3), np.float32); dscale = np.zeros((P, 3), np.float32)
    drot = np.zeros((P, 4), np.float32); dcov = np.zeros((P, 6), np.float32)
    shim.hm_backward(P, _p(mm), _p(ss), _p(qq), _p(view_f), _p(proj_f), W, H, C.c_float(cam.tanfovx), C.c_float(cam.tanfovy),
                     C.c_float(1.0), _p(g_ndc), _p(g_con), _p(dmean), _p(dscale), _p(drot), _p(dcov))
    for got, ref, name in ((dmean, m.grad, 'means3D'), (dscale, s.grad, 'scales'), (drot, q.grad, 'rotations')):
        ref = ref.numpy()
        scale = np.abs(ref[v]).max() + 1e-20
        err = np.abs(got[v] - ref[v]).max() / scale
        assert err < 2e-4, (name, err)
    np.testing.assert_allclose(g_ndc[v], (m2.grad[:, :2].numpy())[v], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_basis_and_derivative(shim, deg):
    P, M = 200, 16
    g = torch.Generator().manual_seed(deg)
    d = torch.randn(P, 3, generator=g, dtype=torch.float64)
    d = (d / d.norm(dim=1, keepdim=True)).requires_grad_(True)
    sh = (0.5 * torch.randn(P, M, 3, generator=g, dtype=torch.float64)).requires_grad_(True)
    gcol = torch.randn(P, 3, generator=g, dtype=torch.float64)
    col = R.eval_sh(deg, sh, d)
    (col * gcol).sum().backward()
    dd, shn, gc = (np.ascontiguousarray(t.detach().numpy(), dtype=np.float32) for t in (d, sh, gcol))
    col2 = np.zeros((P, 3), np.float32); dsh = np.zeros((P, M, 3), np.float32); ddir = np.zeros((P, 3), np.float32)
    shim.hm_sh(P, deg, M, _p(dd), _p(shn), _p(gc), _p(col2), _p(dsh), _p(ddir))
    np.testing.assert_allclose(col2, col.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dsh, sh.grad.numpy(), rtol=1e-4, atol=1e-5)
    dref = np.zeros((P, 3)) if d.grad is None else d.grad.numpy()
    np.testing.assert_allclose(ddir, dref, rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------
# fused_math.h: the callers around the rasterizer (splatam_amd/slam.py is the reference-shaped Python, itself
# pinned to /root/reference by tests/golden/) and their hand-derived adjoints against torch autograd.
# ---------------------------------------------------------------------------------------------------------------

def _glue_inputs(P, iso, seed):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(P, 3, generator=g) * 2.0
    urot = torch.randn(P, 4, generator=g)
    if iso:
        urot = torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1) + 0.05 * urot
    logit = torch.randn(P, 1, generator=g)
    ls = torch.randn(P, 1 if iso else 3, generator=g) * 0.4 - 3.0
    q = torch.tensor([[0.9, 0.1, -0.2, 0.05]]) * 1.3          # deliberately un-normalised
    t = torch.tensor([[0.1, -0.2, 0.3]])
    w2c = torch.eye(4)
    w2c[:3, :3] = torch.tensor(tilted_w2c()[:3, :3])
    w2c[:3, 3] = torch.tensor([0.05, 0.02, -0.1])
    return means, urot, logit, ls, q, t, w2c


def _glue_torch(means, urot, logit, ls, q, t, w2c):
    """The reference-shaped Python for one frame (time index 0 of a 1-frame pose tensor)."""
    from splatam_amd import slam
    params = {'means3D': means, 'unnorm_rotations': urot, 'logit_opacities': logit, 'log_scales': ls,
              'rgb_colors': torch.zeros_like(means),
              'cam_unnorm_rots': q.reshape(1, 4, 1), 'cam_trans': t.reshape(1, 3, 1)}
    tg = slam.transform_to_frame(params, 0, gaussians_grad=True, camera_grad=True)
    rv = slam.transformed_params2rendervar(params, tg)
    dv = slam.transformed_params2depthplussilhouette(params, w2c, tg)
    return torch.cat([rv['means3D'], dv['colors_precomp'][:, 0:1], rv['opacities'], rv['scales'], rv['rotations']], dim=1)


@pytest.mark.parametrize("iso", [True, False])
def test_fused_glue_forward_and_adjoint(shim, iso):
    P = 500
    means, urot, logit, ls, q, t, w2c = _glue_inputs(P, iso, seed=3 + int(iso))
    leaves = [x.clone().requires_grad_(True) for x in (means, urot, logit, ls, q, t)]
    out = _glue_torch(*leaves, w2c)
    cot = torch.randn(P, 12, generator=torch.Generator().manual_seed(9))
    (out * cot).sum().backward()

    args = [_np(x) for x in (q.reshape(-1), t.reshape(-1), w2c[2], means, urot, logit.reshape(-1), ls)]
    got = np.zeros((P, 12), np.float32)
    shim.hm_glue_forward(P, int(iso), *[_p(a) for a in args], _p(got))
    np.testing.assert_allclose(got, out.detach().numpy(), rtol=2e-5, atol=2e-6)

    dmeans = np.zeros((P, 3), np.float32); durot = np.zeros((P, 4), np.float32); dlogit = np.zeros(P, np.float32)
    dls = np.zeros((P, 1 if iso else 3), np.float32); dq = np.zeros(4, np.float32); dt = np.zeros(3, np.float32)
    cotn = _np(cot)
    shim.hm_glue_backward(P, int(iso), *[_p(a) for a in args], _p(cotn), _p(dmeans), _p(durot), _p(dlogit), _p(dls), _p(dq), _p(dt))
    for got_g, ref, name in ((dmeans, leaves[0].grad, "means3D"), (durot, leaves[1].grad, "unnorm_rotations"),
                             (dlogit, leaves[2].grad.reshape(-1), "logit_opacities"), (dls, leaves[3].grad, "log_scales"),
                             (dq, leaves[4].grad.reshape(-1), "cam_unnorm_rots"), (dt, leaves[5].grad.reshape(-1), "cam_trans")):
        ref = ref.numpy()
        scale = np.abs(ref).max() + 1e-12
        assert np.abs(got_g - ref).max() <= 2e-4 * scale, (name, np.abs(got_g - ref).max(), scale)


def test_fused_ssim_pixel_against_autograd(shim):
    from splatam_amd import slam
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 24, 31, generator=g).requires_grad_(True)
    y = (x.detach() + 0.1 * torch.randn(1, 3, 24, 31, generator=g)).clamp(0, 1)
    ssim = slam.calc_ssim(x, y)
    ssim.backward()
    # the same statistic through ssim_pixel + the three blurred partial maps (what the kernels do)
    import torch.nn.functional as F
    win = slam._ssim_window(3, 11, x.device, x.dtype)
    blur = lambda t: F.conv2d(t, win, padding=5, groups=3)                      # noqa: E731
    xd = x.detach()
    stats = [_np(blur(t)).reshape(-1) for t in (xd, y, xd * xd, y * y, xd * y)]
    n = stats[0].size
    out = [np.zeros(n, np.float32) for _ in range(4)]
    shim.hm_ssim_pixel(n, *[_p(a) for a in stats], *[_p(a) for a in out])
    assert abs(out[0].mean() - float(ssim.detach())) < 2e-6
    dmu1, de11, de12 = (torch.from_numpy(a).reshape(1, 3, 24, 31) for a in out[1:])
    grad = (blur(dmu1) + 2 * xd * blur(de11) + y * blur(de12)) / n
    ref = x.grad
    assert (grad - ref).abs().max() <= 2e-4 * ref.abs().max()


def test_fused_adam_matches_torch(shim):
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(1000, generator=g)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [p], 'lr': 0.0025}], lr=0.0, eps=1e-15)
    pn, m, v = _np(p0).copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    for step in range(1, 6):
        grad = torch.randn(1000, generator=g) * 10 ** torch.randint(-6, 1, (1000,), generator=g).float()
        p.grad = grad.clone()
        opt.step()
        bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
        shim.hm_adam(1000, _p(pn), _p(_np(grad)), _p(m), _p(v), C.c_float(0.9), C.c_float(0.999), C.c_float(0.0025 / bc1),
                     C.c_float(bc2 ** 0.5), C.c_float(1e-15))
        np.testing.assert_allclose(pn, p.detach().numpy(), rtol=0, atol=3e-7)


@pytest.mark.parametrize("aniso", [False, True])
def test_live_tile_rect_keeps_every_tile_that_blends(shim, aniso):
    """Group binning files a Gaussian only in the tiles of its rectangle that can hold a pixel with alpha >= 1/255
    (splat_math.h live_tile_rect).  The rule must be conservative: every pixel the composite would blend (power <= 0,
    min(0.99, o G) >= 1/255 -- SURVEY Appendix A, forward composite) lies in a kept tile; and it must cut something."""
    W, H = 200, 136
    cam, rv = scene(3000, W, H, 170.0, seed=11, anisotropic=aniso)
    rv['opacities'][:50] = 0.003            # below 1/255: nothing blends, nothing is filed
    rv['opacities'][50:100] = 0.0045        # just above it
    rv['scales'][100:150] *= 4.0            # splats that span several tiles
    geom = R.preprocess(rv['means3D'], None, rv['scales'], rv['rotations'], None, cam)
    P = rv['means3D'].shape[0]
    conic, xy, op = _np(geom.conic), _np(geom.xy), _np(rv['opacities']).reshape(-1)
    rect0 = np.concatenate([geom.rect_min.numpy(), geom.rect_max.numpy()], 1).astype(np.int32)
    rect = rect0.copy()
    shim.hm_live_tile_rect(P, _p(conic), _p(op), _p(xy), _p(rect, C.c_int))
    vis = geom.radii.numpy() > 0
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    kept = dropped = 0
    for i in np.nonzero(vis)[0]:
        x0, y0, x1, y1 = rect0[i]
        sx, sy = slice(16 * x0, min(W, 16 * x1)), slice(16 * y0, min(H, 16 * y1))
        dx, dy = xy[i, 0] - xs[sy, sx], xy[i, 1] - ys[sy, sx]
        power = -0.5 * (conic[i, 0] * dx * dx + conic[i, 2] * dy * dy) - conic[i, 1] * dx * dy
        live = (power <= 0) & (np.minimum(0.99, op[i] * np.exp(power)) >= 1.0 / 255.0)
        ly, lx = np.nonzero(live)
        tx, ty = (lx + 16 * x0) // 16, (ly + 16 * y0) // 16
        a0, b0, a1, b1 = rect[i]
        assert a0 >= x0 and b0 >= y0 and a1 <= x1 and b1 <= y1
        assert ((tx >= a0) & (tx < a1) & (ty >= b0) & (ty < b1)).all(), (i, rect0[i], rect[i])
        kept += (a1 - a0) * (b1 - b0)
        dropped += (x1 - x0) * (y1 - y0) - (a1 - a0) * (b1 - b0)
    assert dropped > 0.05 * (kept + dropped), (kept, dropped)
    assert (rect[:50, 2] == rect[:50, 0]).all() or not vis[:50].any()      # opacity below the blend threshold: empty
    # NaN geometry leaves the rectangle alone
    bad = conic[:4].copy(); bad[:, 0] = np.nan
    r4 = rect0[:4].copy()
    shim.hm_live_tile_rect(4, _p(bad), _p(op[100:104].copy()), _p(xy[:4].copy()), _p(r4, C.c_int))
    assert (r4 == rect0[:4]).all()


def test_live_tile_rect_at_the_blend_threshold(shim):
    """opacity == float32(1 / 255) with the centre ON a pixel centre: alpha == 1/255 there, the composite blends it (alpha >= 1/255), so the
    tile must stay -- decided on the opacity itself, not on log(255 o), which may round below zero; just below the threshold nothing stays."""
    thr = np.float32(1.0) / np.float32(255.0)
    conic = np.array([[1.0, 0.0, 1.0]] * 2, np.float32)
    xy = np.array([[24.0, 40.0]] * 2, np.float32)
    op = np.array([thr, np.nextafter(thr, np.float32(0))], np.float32)
    rect = np.array([[0, 1, 3, 4]] * 2, np.int32)
    shim.hm_live_tile_rect(2, _p(conic), _p(op), _p(xy), _p(rect, C.c_int))
    assert rect[0, 0] <= 1 < rect[0, 2] and rect[0, 1] <= 2 < rect[0, 3], rect[0]
    assert rect[1, 2] == rect[1, 0] and rect[1, 3] == rect[1, 1], rect[1]
