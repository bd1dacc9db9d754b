"""Parity of the HIP rasterizer (through the reference-shaped Python surface and
the C ABI underneath) against the oracle.  Tolerances are the north star's:
1e-4 on colour / depth, 1e-3 (relative to the tensor's max magnitude) on
gradients; integer / index outputs (radii, tile ranges, sorted lists,
n_contrib) must match exactly up to float32 threshold flips, which are counted
and bounded."""
import numpy as np
import pytest
import torch

from oracle import c_ref
from oracle import raster_ref as R
from tests.util import (assert_close_outliers, assert_grad_calibrated, assert_grad_outliers_explained, assert_outliers_explained,
                        flip_pixels, grad_scale, oracle_flip_bounds, scene, tilted_w2c)

pytestmark = pytest.mark.gpu


def _to_cuda_settings(cam):
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    return Camera(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  bg=cam.bg.cuda(), scale_modifier=cam.scale_modifier, viewmatrix=cam.viewmatrix.cuda(),
                  projmatrix=cam.projmatrix.cuda(), sh_degree=cam.sh_degree, campos=cam.campos.cuda(),
                  prefiltered=cam.prefiltered)


def _gpu_render(cam, rv, grad_out=None, keys=('means3D', 'means2D', 'opacities', 'colors_precomp', 'scales', 'rotations'), path=None):
    """path: None = whatever the default policy does on a scene's first call (the exact path); "exact" / "fast": the call under test
    must take that path of the "auto" policy -- "fast" is preceded by the scene's first (exact, list-learning) call, as in a loop."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from splatam_amd import rasterizer as rz
    cs = _to_cuda_settings(cam)
    inp = {k: rv[k].detach().cuda().requires_grad_(grad_out is not None) for k in keys}
    if path == "fast":
        with torch.no_grad():
            Renderer(raster_settings=cs)(**inp)
    before = dict(rz.fast_path_stats)
    color, radii, depth = Renderer(raster_settings=cs)(**inp)
    if path is not None:
        assert rz.fast_path_stats[path] == before[path] + 1, (path, before, rz.fast_path_stats)
    grads = None
    if grad_out is not None:
        (color * grad_out.cuda()).sum().backward()
        grads = {k: inp[k].grad.cpu().numpy() for k in keys}
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(), grads


def _c_oracle(cam, rv, grad_out=None, precision="f32"):
    cr = c_ref.CRef(precision)
    W, H = cam.image_width, cam.image_height
    col, radii, dep = cr.forward(rv['means3D'].numpy(), rv['colors_precomp'].numpy(), rv['opacities'].numpy(),
                                 rv['scales'].numpy(), rv['rotations'].numpy(), cam.viewmatrix.numpy(),
                                 cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy, W, H, cam.bg.numpy(),
                                 scale_modifier=cam.scale_modifier)
    g = cr.backward(grad_out.numpy()) if grad_out is not None else None
    return col, radii, dep, g, cr


def _check_forward(gc, gr, gd, oc, orad, od, npix, cam=None, rv=None):
    """North star: 1e-4 on colour and depth.  Pixels beyond it must be EXPLAINED by the float64 oracle (tests/util.py:
    assert_outliers_explained): a decision of that pixel -- alpha >= 1/255, T (1 - alpha) >= 1e-4, power <= 0, the order of two equal
    depths -- within float32 rounding of its threshold, and the difference within what those flips can move.  Their number stays
    bounded too (1e-4 of the elements).  Returns (flagged pixels, xy, radii) for the gradient check."""
    assert (gr != orad).sum() <= max(2, int(1e-5 * gr.size)), "radii mismatch"
    assert np.abs(gr.astype(np.int64) - orad).max() <= 1
    if cam is None:                     # (callers without the inputs at hand: the counted form)
        cmax = max(1.0, float(np.abs(oc).max()))
        assert_close_outliers(gc, oc, 1e-4, rtol=1e-4, max_outlier_frac=1e-4, outlier_atol=0.03 * cmax, what="color")
        assert_close_outliers(gd, od, 1e-4, rtol=1e-4, max_outlier_frac=1e-4, outlier_atol=0.1, what="depth")
        return None
    bound, margin, xy, radii, noise = oracle_flip_bounds(rv, cam)
    C_ = gc.shape[0]
    n = assert_outliers_explained(gc, oc, bound[:C_], 1e-4, rtol=1e-4, noise=noise[:C_], what="color")
    n += assert_outliers_explained(gd, od, bound[C_:C_ + 1], 1e-4, rtol=1e-4, noise=noise[C_:C_ + 1], what="depth")
    assert n <= max(2, int(1e-4 * (gc.size + gd.size))), n
    return flip_pixels(bound, (gc, gd), (oc, od)), xy, radii


GRAD_MAP = [('means3D', 'means3D'), ('means2D', 'means2D'), ('colors_precomp', 'colors'), ('opacities', 'opacities'),
            ('scales', 'scales'), ('rotations', 'rotations')]


def _check_grads(gg, og, og64=None, flips=None):
    """North star: 1e-3 of the tensor's maximum.  Rows beyond it must lie over a pixel with a float32 decision flip (``flips`` from
    _check_forward; tests/util.py: assert_grad_outliers_explained), their number bounded.  With the float64 oracle's gradients the
    check is also per element, calibrated against the float32 oracle's own rounding noise (assert_grad_calibrated)."""
    for k, ok in GRAD_MAP:
        ref = og[ok]
        got = gg[k].reshape(ref.shape)
        assert np.isfinite(got).all(), k
        assert_close_outliers(got, ref, 1e-3 * grad_scale(ref), max_outlier_frac=1e-4,
                              outlier_atol=0.05 * grad_scale(ref), what=f"grad {k}")
        if flips is not None:
            assert_grad_outliers_explained(got, ref, *flips, what=f"grad {k}")
        if og64 is not None and float(np.abs(og64[ok]).max()) > 1e-12 * max(1.0, float(np.abs(og64['means3D']).max())):
            assert_grad_calibrated(got, ref, og64[ok], what=f"grad {k}")


@pytest.mark.parametrize("n,W,H,aniso,view,bg", [
    (10000, 320, 240, False, False, (0, 0, 0)),            # BASELINE config A shape
    (10000, 320, 240, True, True, (1.0, 0.5, 0.2)),        # anisotropic, tilted view, non-zero background
    (3000, 203, 117, True, True, (0, 0, 0)),               # image not a multiple of the tile size
    (50, 64, 48, False, False, (0.1, 0.2, 0.3)),           # sparse: most tiles empty
    (20000, 96, 64, False, False, (0, 0, 0)),              # dense: > 1000 Gaussians per tile, early termination
])
@pytest.mark.parametrize("path", ["exact", "fast", "exact + staged records", "fast + staged records"])
def test_forward_backward_parity(n, W, H, aniso, view, bg, path):
    """Both paths of the default capacity policy: a scene's first call (exact lists: scan, scatter, per-tile sort) and its later ones
    (group binning, lists sorted inside the forward composite, nothing read back) -- the dense scene's lists are beyond what the
    composite sorts, so it must stay on the exact path."""
    cam, rv = scene(n, W, H, 0.9 * W, seed=n, anisotropic=aniso, w2c=tilted_w2c() if view else None, bg=bg)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    if path.endswith("staged records"):         # SplatState.tile_recs forced on (default: by the scene's longest list)
        from splatam_amd import rasterizer as rz
        rz.USE_TILE_RECS = True
        path = path.split(" ")[0]
    if path == "fast" and n == 20000:
        path = "exact"
        with torch.no_grad():           # (the scene's first call; the call under test is its second)
            _gpu_render(cam, rv)
    gc, gr, gd, gg = _gpu_render(cam, rv, gout, path=path)
    oc, orad, od, og, _ = _c_oracle(cam, rv, gout)
    flips = _check_forward(gc, gr, gd, oc, orad, od, W * H, cam, rv)
    _check_grads(gg, og, _c_oracle(cam, rv, gout, "f64")[3], flips)


def test_against_autograd_oracle_small():
    """Same comparison against the independent autograd oracle (float32)."""
    W, H = 96, 80
    cam, rv = scene(1500, W, H, 90.0, seed=11, anisotropic=True, w2c=tilted_w2c(), bg=(0.2, 0.4, 0.6))
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2))
    gc, gr, gd, gg = _gpu_render(cam, rv, gout)
    inp = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    col, radii, dep = R.rasterize(inp['means3D'], inp['means2D'], inp['opacities'], inp['colors_precomp'],
                                  inp['scales'], inp['rotations'], cam)
    (col * gout).sum().backward()
    flips = _check_forward(gc, gr, gd, col.detach().numpy(), radii.numpy(), dep.numpy(), W * H, cam, rv)
    og = {ok: inp[k].grad.numpy() for k, ok in GRAD_MAP}
    _check_grads(gg, og, flips=flips)


def test_internal_state_matches_oracle():
    """Binning must reproduce the oracle's per-tile lists exactly: same ranges,
    same ids in the same (depth, index) order; n_contrib / final_T per pixel."""
    import ctypes as C
    from splatam_amd import rasterizer as rz
    W, H = 160, 112
    cam, rv = scene(6000, W, H, 150.0, seed=3)
    # force depth ties (quantised depth, like a uint16 depth map)
    z = (rv['means3D'][:, 2] * 8).round() / 8
    rv['means3D'] = torch.stack([rv['means3D'][:, 0] / rv['means3D'][:, 2] * z, rv['means3D'][:, 1] / rv['means3D'][:, 2] * z, z], -1)
    cs = _to_cuda_settings(cam)
    empty = torch.empty(0, device="cuda")
    col, radii, dep, pk = rz.rasterize_forward(cs, rv['means3D'].cuda(), rv['colors_precomp'].cuda(),
                                               rv['opacities'].cuda().reshape(-1), rv['scales'].cuda(),
                                               rv['rotations'].cuda(), empty, empty)
    torch.cuda.synchronize()
    oc, orad, od, _, cr = _c_oracle(cam, rv)
    assert pk.num_rendered == cr.num_rendered()
    assert (pk.tensors['tile_base'].cpu().numpy() == cr.ranges()).all()
    assert (pk.tensors['point_list'].cpu().numpy()[:pk.num_rendered] == cr.point_list()).all()
    assert (radii.cpu().numpy() == orad).all()
    ncg, nco = pk.tensors['n_contrib'].cpu().numpy(), cr.n_contrib()
    assert (ncg != nco).sum() <= 3
    assert_close_outliers(pk.tensors['final_T'].cpu().numpy(), cr.final_T(), 1e-5, max_outlier_frac=1e-4, outlier_atol=0.02, what="final_T")


@pytest.mark.parametrize("C_", [1, 2, 4, 5, 7, 8])
def test_channel_counts(C_):
    W, H = 112, 80
    cam, rv = scene(2500, W, H, 100.0, seed=C_)
    g = torch.Generator().manual_seed(C_)
    rv['colors_precomp'] = torch.rand(rv['means3D'].shape[0], C_, generator=g)
    cam = cam._replace(bg=torch.rand(C_, generator=g))
    gout = torch.randn(C_, H, W, generator=g)
    gc, gr, gd, gg = _gpu_render(cam, rv, gout)
    oc, orad, od, og, _ = _c_oracle(cam, rv, gout)
    flips = _check_forward(gc, gr, gd, oc, orad, od, W * H, cam, rv)
    _check_grads(gg, og, flips=flips)


def test_depth_silhouette_pass():
    """The reference's second render: colours = [z, 1, z^2], gradient only on channel 0
    (/root/reference/utils/slam_helpers.py:196-213, /root/reference/scripts/splatam.py:253-259)."""
    W, H = 160, 112
    cam, rv = scene(5000, W, H, 150.0, seed=21)
    z = rv['means3D'][:, 2]
    rv['colors_precomp'] = torch.stack([z, torch.ones_like(z), z * z], -1)
    gout = torch.zeros(3, H, W)
    gout[0] = torch.randn(H, W, generator=torch.Generator().manual_seed(5))
    gc, gr, gd, gg = _gpu_render(cam, rv, gout)
    oc, orad, od, og, cr = _c_oracle(cam, rv, gout)
    flips = _check_forward(gc, gr, gd, oc, orad, od, W * H, cam, rv)
    _check_grads(gg, og, flips=flips)
    assert_close_outliers(gc[1], 1.0 - cr.final_T(), 1e-5, max_outlier_frac=1e-4, outlier_atol=0.02, what="silhouette")


def test_empty_and_culled_inputs():
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    W, H = 64, 48
    cam, rv = scene(10, W, H, 60.0, seed=0, bg=(0.3, 0.2, 0.1))
    cs = _to_cuda_settings(cam)
    # P = 0
    z3, z4, z1 = (torch.zeros(0, k, device="cuda") for k in (3, 4, 1))
    col, radii, dep = Renderer(raster_settings=cs)(means3D=z3, means2D=z3, opacities=z1, colors_precomp=z3, scales=z3, rotations=z4)
    assert radii.numel() == 0 and dep.abs().max().item() == 0
    assert torch.allclose(col, cs.bg[:, None, None].expand(3, H, W))
    # everything behind the near plane
    inp = {k: v.cuda().requires_grad_(True) for k, v in rv.items()}
    with torch.no_grad():
        inp['means3D'][:, 2] = 0.1
    col, radii, dep = Renderer(raster_settings=cs)(**inp)
    col.sum().backward()
    assert (radii == 0).all() and torch.allclose(col, cs.bg[:, None, None].expand(3, H, W))
    for k in inp:
        assert inp[k].grad is not None and inp[k].grad.abs().max().item() == 0


def test_scale_modifier_and_cov3d_precomp():
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    W, H = 96, 80
    cam, rv = scene(1500, W, H, 90.0, seed=8, anisotropic=True)
    cam = cam._replace(scale_modifier=1.7)
    gc, gr, gd, _ = _gpu_render(cam, rv)
    oc, orad, od, _, _ = _c_oracle(cam, rv)
    _check_forward(gc, gr, gd, oc, orad, od, W * H, cam, rv)
    # the same scene through a precomputed covariance (modifier folded in) + its gradient
    cov = R.cov3d_from_scale_rot(rv['scales'], rv['rotations'], 1.7)
    cam1 = cam._replace(scale_modifier=1.0)
    cs = _to_cuda_settings(cam1)
    covg = cov.cuda().requires_grad_(True)
    col, radii, dep = Renderer(raster_settings=cs)(means3D=rv['means3D'].cuda(), means2D=rv['means2D'].cuda(),
                                                  opacities=rv['opacities'].cuda(), colors_precomp=rv['colors_precomp'].cuda(),
                                                  cov3D_precomp=covg)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(4))
    (col * gout.cuda()).sum().backward()
    assert_close_outliers(col.detach().cpu().numpy(), oc, 1e-4, max_outlier_frac=2e-5, outlier_atol=0.03, what="cov3D color")
    covr = cov.clone().requires_grad_(True)
    c2, _, _ = R.rasterize(rv['means3D'], rv['means2D'], rv['opacities'], rv['colors_precomp'], None, None,
                           R.Settings(*cam1), cov3D_precomp=covr)
    (c2 * gout).sum().backward()
    ref = covr.grad.numpy()
    assert_close_outliers(covg.grad.cpu().numpy(), ref, 1e-3 * grad_scale(ref), max_outlier_frac=1e-4,
                          outlier_atol=0.05 * grad_scale(ref), what="grad cov3D")


@pytest.mark.parametrize("deg", [0, 3])
def test_spherical_harmonics_path(deg):
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    W, H = 96, 80
    cam, rv = scene(1200, W, H, 90.0, seed=13, anisotropic=True, w2c=tilted_w2c())
    cam = cam._replace(sh_degree=deg)
    g = torch.Generator().manual_seed(deg)
    shs = 0.4 * torch.randn(rv['means3D'].shape[0], 16, 3, generator=g)
    gout = torch.randn(3, H, W, generator=g)
    cs = _to_cuda_settings(cam)
    inp = dict(means3D=rv['means3D'].cuda().requires_grad_(True), means2D=rv['means2D'].cuda(),
               opacities=rv['opacities'].cuda(), shs=shs.cuda().requires_grad_(True), scales=rv['scales'].cuda(),
               rotations=rv['rotations'].cuda())
    col, radii, dep = Renderer(raster_settings=cs)(**inp)
    (col * gout.cuda()).sum().backward()
    m = rv['means3D'].clone().requires_grad_(True)
    s = shs.clone().requires_grad_(True)
    c2, r2, d2 = R.rasterize(m, rv['means2D'], rv['opacities'], None, rv['scales'], rv['rotations'], cam, shs=s)
    (c2 * gout).sum().backward()
    assert_close_outliers(col.detach().cpu().numpy(), c2.detach().numpy(), 1e-4, max_outlier_frac=5e-5, outlier_atol=0.03, what="sh color")
    for got, ref, name in ((inp['shs'].grad, s.grad, 'shs'), (inp['means3D'].grad, m.grad, 'means3D')):
        ref = ref.numpy()
        assert_close_outliers(got.cpu().numpy(), ref, 1e-3 * grad_scale(ref), max_outlier_frac=1e-4,
                              outlier_atol=0.05 * grad_scale(ref), what=f"sh grad {name}")


def test_mark_visible_and_errors():
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    cam, rv = scene(500, 64, 48, 60.0, seed=2)
    rv['means3D'][:50, 2] = 0.1
    cs = _to_cuda_settings(cam)
    r = Renderer(raster_settings=cs)
    m = r.markVisible(rv['means3D'].cuda()).cpu()
    assert (m == R.mark_visible(rv['means3D'], cam)).all()
    inp = {k: v.cuda() for k, v in rv.items()}
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=inp['means3D'], means2D=inp['means2D'], opacities=inp['opacities'], scales=inp['scales'], rotations=inp['rotations'])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=inp['means3D'], means2D=inp['means2D'], opacities=inp['opacities'], colors_precomp=inp['colors_precomp'])


def test_full_size_replica_shape():
    """BASELINE config B shape: 300k Gaussians, 1200x680, against the C oracle."""
    W, H, n = 1200, 680, 300_000
    cx, cy = 599.5, 339.5
    cam = R.make_camera(W, H, 600.0, 600.0, cx, cy)
    p = R.synthetic_cloud(n, W, H, 600.0, 600.0, cx, cy, seed=0)
    rv = R.cloud_to_rendervar(p)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    gc, gr, gd, gg = _gpu_render(cam, rv, gout)
    oc, orad, od, og, cr = _c_oracle(cam, rv, gout)
    flips = _check_forward(gc, gr, gd, oc, orad, od, W * H, cam, rv)
    _check_grads(gg, og, _c_oracle(cam, rv, gout, "f64")[3], flips)
    # size-independent properties: silhouette in [0,1]; rendering is linear in the colours
    rv2 = dict(rv)
    rv2['colors_precomp'] = 2.0 * rv['colors_precomp']
    gc2, _, _, _ = _gpu_render(cam, rv2)
    np.testing.assert_allclose(gc2, 2.0 * gc, rtol=1e-5, atol=1e-6)


def test_composite_kernels_multi_batch_and_config_a():
    """Multi-batch lists with early termination (dense scene) and the config-A shape against the oracle."""
    for n, W, H, bg in ((20000, 96, 64, (0, 0, 0)), (10000, 320, 240, (0.3, 0.1, 0.6))):
        cam, rv = scene(n, W, H, 0.9 * W, seed=n + 1, bg=bg)
        gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(7))
        gc, gr, gd, gg = _gpu_render(cam, rv, gout)
        oc, orad, od, og, _ = _c_oracle(cam, rv, gout)
        flips = _check_forward(gc, gr, gd, oc, orad, od, W * H, cam, rv)
        _check_grads(gg, og, flips=flips)


@pytest.mark.parametrize("path", ["exact", "fast"])
def test_gaussian_exactly_at_the_blend_threshold_is_blended(path):
    """opacity == float32(1 / 255), centre exactly on a pixel centre: alpha == 1/255 there and the composite blends it
    (Appendix A: skip if alpha < 1/255).  The staging cull and the live tile rectangle of group binning decide that case on the
    opacity itself (log(255 o) may round below zero); one step below the threshold nothing is blended."""
    W, H = 33, 17                                    # odd sizes: the optical axis hits the centre of pixel (16, 8)
    cam = R.make_camera(W, H, 40.0, 40.0, W / 2.0, H / 2.0)          # pixel centres at fx X / Z + cx - 0.5
    thr = np.float32(1.0) / np.float32(255.0)
    rv = dict(means3D=torch.tensor([[0.0, 0.0, 2.0], [0.0, 0.0, 3.0]]), means2D=torch.zeros(2, 3),
              opacities=torch.tensor([[float(thr)], [float(np.nextafter(thr, np.float32(0)))]]),
              colors_precomp=torch.tensor([[1.0, 0.5, 0.25], [0.0, 1.0, 0.0]]), scales=torch.full((2, 3), 0.05),
              rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    gc, gr, gd, _ = _gpu_render(cam, rv, path=path)
    oc, orad, od, _, _ = _c_oracle(cam, rv)
    assert oc[0, 8, 16] == thr * np.float32(1.0) and oc[1, 8, 16] == np.float32(thr * np.float32(0.5))      # the oracle blends the first, not the second
    assert np.array_equal(gr, orad)
    assert np.array_equal(gc[:, 8, 16], oc[:, 8, 16]), (gc[:, 8, 16], oc[:, 8, 16])
    assert np.abs(gc - oc).max() <= 1e-6 and np.abs(gd - od).max() <= 1e-6


@pytest.mark.parametrize("seed", [3, 4])
def test_culls_drop_nothing_on_extreme_splats(seed):
    """The fast path of the auto policy files a Gaussian only in the tiles of its live box (splat_math.h live_tile_rect) and both
    paths visit only the 4x4 blocks inside its live disc (render.hip gather()): an entry or a visit they drop must fail the alpha
    test at every pixel, so the image of the fast path equals the exact path's BIT FOR BIT and the gradients to float-atomic
    order -- on splats the SplaTAM maps never hold: needles (anisotropy up to 100), opacities from below the blend threshold to
    1, footprints from a third of a pixel to a quarter of the frame, a tilted camera."""
    W, H = 208, 144
    cam, rv = scene(2500, W, H, 160.0, seed=seed, anisotropic=True, w2c=tilted_w2c())
    g = torch.Generator().manual_seed(seed)
    n = rv['means3D'].shape[0]
    rv['scales'] = rv['scales'][:, :1] * torch.exp(torch.empty(n, 3).uniform_(-1.5, 3.0, generator=g))
    rv['rotations'] = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1)
    rv['opacities'] = torch.exp(torch.empty(n, 1).uniform_(np.log(0.002), 0.0, generator=g))
    gout = torch.randn(3, H, W, generator=g)
    ec, er, ed, eg = _gpu_render(cam, rv, gout)                      # a scene's first call: the exact path
    fc, fr, fd, fg = _gpu_render(cam, rv, gout, path="fast")
    assert np.array_equal(er, fr)
    assert np.array_equal(ec, fc) and np.array_equal(ed, fd), (np.abs(ec - fc).max(), np.abs(ed - fd).max())
    for k in eg:
        assert np.abs(eg[k] - fg[k]).max() <= 2e-4 * grad_scale(eg[k]) + 1e-12, k      # (sums over up to a quarter of the frame, in another order)
    oc, orad, od, _, _ = _c_oracle(cam, rv)
    assert np.array_equal(er, orad)
    assert_close_outliers(ec, oc, 1e-4, max_outlier_frac=2e-3, outlier_atol=0.2, what="extreme splats: colour vs oracle")
