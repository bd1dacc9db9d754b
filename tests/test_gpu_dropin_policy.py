"""The drop-in rasterizer's default capacity policy ("auto", splatam_amd/rasterizer.py): a scene's first call runs the reference's
exact path (one host read, as the CUDA original reads num_rendered at /root/reference/scripts/splatam.py:249's callee) and learns the
longest per-tile list; later calls of a scene with short lists run group binning + the sorting composite with NO host read.  What the
device does when a list outgrows its bucket after all is tested here: the flag in pinned host memory, the repeat on exact lists."""
import warnings

import numpy as np
import pytest
import torch

from oracle import raster_ref as R
from tests.util import scene

pytestmark = pytest.mark.gpu

KEYS = ('means3D', 'means2D', 'opacities', 'colors_precomp', 'scales', 'rotations')


def _settings(cam):
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    return Camera(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  bg=cam.bg.cuda(), scale_modifier=cam.scale_modifier, viewmatrix=cam.viewmatrix.cuda(),
                  projmatrix=cam.projmatrix.cuda(), sh_degree=cam.sh_degree, campos=cam.campos.cuda(), prefiltered=cam.prefiltered)


def _render(cs, rv, gout=None, wait=True):
    """wait: the forward pass has finished when the backward pass is issued (a training loop: the loss sits between them)."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    inp = {k: rv[k].detach().cuda().requires_grad_(gout is not None) for k in KEYS}
    color, radii, depth = Renderer(raster_settings=cs)(**inp)
    grads = None
    if gout is not None:
        g = gout.cuda()
        if wait:
            torch.cuda.synchronize()
        color.backward(g)
        grads = {k: inp[k].grad.clone() for k in KEYS}
    torch.cuda.synchronize()
    return color.detach().clone(), radii.clone(), depth.detach().clone(), grads


def _clustered(n, W, H, f, seed):
    """The same Gaussian count and camera as scene(n, W, H, f), every Gaussian inside 6 % of the image: lists far beyond 1 024."""
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    p = R.synthetic_cloud(n, W, H, f, f, cx, cy, seed=seed, region=(0.4, 0.4, 0.65, 0.65))
    return R.cloud_to_rendervar(p)


def test_steady_state_makes_no_host_read_and_equals_the_exact_path():
    """Second and later calls of a short-list scene: two launches forward, nothing read back, the same lists -- images bit-identical
    to the exact path's (same sorted lists, same composite arithmetic), gradients equal to float-atomic summation order."""
    from splatam_amd import rasterizer as rz
    n, W, H = 20000, 320, 240
    cam, rv = scene(n, W, H, 0.9 * W, seed=5)
    cs = _settings(cam)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4))
    rz.set_sync_mode("exact")
    c0, r0, d0, g0 = _render(cs, rv, gout)
    rz.set_sync_mode("auto")
    rz.reset_scene_stats()
    before = dict(rz.fast_path_stats)
    _render(cs, rv)                                                 # the scene's first call: exact, learns the lists
    assert rz.fast_path_stats["exact"] == before["exact"] + 1 and rz.fast_path_stats["fast"] == before["fast"]
    c1, r1, d1, g1 = _render(cs, rv, gout)
    assert rz.fast_path_stats["fast"] == before["fast"] + 1 and rz.fast_path_stats["flagged"] == before["flagged"]
    assert torch.equal(r0, r1) and torch.equal(c0, c1) and torch.equal(d0, d1)
    for k in KEYS:
        sc = float(g0[k].abs().max()) + 1e-20
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * sc, k
    # every FAST_REFRESH-th call of the scene is an exact one again (the statistics follow a map that grows)
    for _ in range(rz.FAST_REFRESH):
        _render(cs, rv)
    assert rz.fast_path_stats["exact"] >= before["exact"] + 2


def test_flagged_call_is_repeated_on_exact_lists():
    """A scene that looks like the one the statistics were learnt on (same Gaussian count, image, field of view) but whose lists are
    far beyond the learnt buckets: the device raises the pinned flag.  A render without gradients is repeated at once and returns the
    exact image; a render with gradients warns, forms its gradients on exact lists, and the scene is back on the exact path."""
    from splatam_amd import rasterizer as rz
    n, W, H = 30000, 320, 240
    cam, rv = scene(n, W, H, 0.9 * W, seed=6)
    rv_long = _clustered(n, W, H, 0.9 * W, seed=7)
    cs = _settings(cam)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5))
    rz.set_sync_mode("exact")
    c_ref, r_ref, d_ref, g_ref = _render(cs, rv_long, gout)
    rz.set_sync_mode("auto")
    rz.reset_scene_stats()
    _render(cs, rv)                                                 # learns SHORT lists
    before = dict(rz.fast_path_stats)
    c1, r1, d1, _ = _render(cs, rv_long)                            # no gradients: resolved inside the forward
    assert rz.fast_path_stats["flagged"] == before["flagged"] + 1
    assert torch.equal(c1, c_ref) and torch.equal(d1, d_ref) and torch.equal(r1, r_ref)
    # ... and with gradients (the statistics are re-learnt from the short-list scene first)
    rz.reset_scene_stats()
    _render(cs, rv)
    before = dict(rz.fast_path_stats)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c2, r2, d2, g2 = _render(cs, rv_long, gout)
    assert rz.fast_path_stats["flagged"] == before["flagged"] + 1
    assert any("truncated lists" in str(x.message) for x in w), [str(x.message) for x in w]
    for k in KEYS:
        sc = float(g_ref[k].abs().max()) + 1e-20
        assert float((g_ref[k] - g2[k]).abs().max()) <= 2e-5 * sc, k
    # the scene is back on the exact path: the next call is exact and right
    before = dict(rz.fast_path_stats)
    c3, r3, d3, _ = _render(cs, rv_long)
    assert rz.fast_path_stats["fast"] == before["fast"] and torch.equal(c3, c_ref)
    # ... and a backward pass issued BEHIND a forward pass that is still running (nobody waits for the flag): the device poisons the
    # gradients of the flagged call with NaN, and the next look at the flags raises
    rz.reset_scene_stats()
    _render(cs, rv)
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    inp = {k: rv_long[k].detach().cuda().requires_grad_(True) for k in KEYS}
    g = gout.cuda()
    torch.cuda.synchronize()
    color, _, _ = Renderer(raster_settings=cs)(**inp)
    color.backward(g)                                               # (enqueued microseconds after the forward pass: it cannot have finished)
    torch.cuda.synchronize()
    if rz._unchecked:                                               # (a box fast enough to finish the forward pass first took the repair above)
        assert all(bool(torch.isnan(inp[k].grad).all()) for k in KEYS if k != 'means2D')
        assert bool(torch.isnan(inp['means2D'].grad[:, :2]).all())
        with pytest.raises(RuntimeError, match="set to NaN"):
            rz.check_pending()
        assert not rz._unchecked
    # a healthy call in the same situation is untouched and leaves nothing behind
    rz.reset_scene_stats()
    _render(cs, rv)
    _, _, _, g_ok = _render(cs, rv, gout, wait=False)
    rz.check_pending()
    assert all(bool(torch.isfinite(v).all()) for v in g_ok.values()) and not rz._unchecked


def test_upstream_scale_gradient_switch():
    """SplatGrads.flags SPLAT_GRADS_UPSTREAM_SCALE: dL/dscales without the scale_modifier factor -- the numbers the CUDA original's
    computeCov3D adjoint returns -- against the oracle's twin (ref_set_upstream_scale); every other gradient is
    untouched, and at modifier 1 the switch changes nothing."""
    from oracle import c_ref
    from splatam_amd import rasterizer as rz
    n, W, H = 8000, 256, 192
    cam, rv = scene(n, W, H, 0.9 * W, seed=9, anisotropic=True)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(6))
    res = {}
    for mod in (1.0, 1.7):
        cs = _settings(cam._replace(scale_modifier=mod))
        for up in (False, True):
            rz.set_upstream_scale_gradient(up)
            try:
                res[mod, up] = _render(cs, rv, gout)[3]
            finally:
                rz.set_upstream_scale_gradient(False)
    for k in KEYS:
        assert torch.equal(res[1.0, False][k], res[1.0, True][k]) or float((res[1.0, False][k] - res[1.0, True][k]).abs().max()) <= \
            2e-5 * float(res[1.0, False][k].abs().max()), k
    a, b = res[1.7, False], res[1.7, True]
    sc = float(a['scales'].abs().max())
    assert float((a['scales'] - 1.7 * b['scales']).abs().max()) <= 3e-5 * sc
    for k in KEYS:
        if k != 'scales':
            assert float((a[k] - b[k]).abs().max()) <= 2e-5 * (float(a[k].abs().max()) + 1e-20), k
    # against the oracle's twin
    cr = c_ref.CRef()
    cr.forward(rv['means3D'].numpy(), rv['colors_precomp'].numpy(), rv['opacities'].numpy(), rv['scales'].numpy(), rv['rotations'].numpy(),
               cam.viewmatrix.numpy(), cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy, W, H, cam.bg.numpy(), scale_modifier=1.7)
    cr.set_upstream_scale(True)
    og = cr.backward(gout.numpy())
    ref = og['scales']
    got = b['scales'].cpu().numpy()
    err = np.abs(got - ref)
    assert np.quantile(err, 0.9999) <= 1e-3 * np.abs(ref).max(), (err.max(), np.abs(ref).max())
