"""Pins the oracle itself: analytic known answers, finite differences (float64),
and the cross-check between the autograd oracle (oracle/raster_ref.py) and the
hand-written-backward C oracle (oracle/raster_ref.c).  CPU only.

The reference holds no golden vectors for this path; these
tests are what stands in for them ("parity unpinned" -- see DESIGN.md)."""
import math

import numpy as np
import pytest
import torch

from oracle import c_ref
from oracle import raster_ref as R
from tests.util import assert_close_outliers, grad_scale, scene, tilted_w2c


def _single(W=32, H=32, f=40.0, z=2.0, scale=0.05, opacity=0.8, color=(0.2, 0.5, 0.9), bg=(0, 0, 0), px=(15, 17)):
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    cam = R.make_camera(W, H, f, f, cx, cy, bg=bg)
    # ndc2Pix puts a point that projects to u at pixel coordinate u - 0.5 (Appendix A step 7)
    X = torch.tensor([[(px[0] + 0.5 - cx) / f * z, (px[1] + 0.5 - cy) / f * z, z]], dtype=torch.float32)
    rv = dict(means3D=X, means2D=torch.zeros(1, 3), opacities=torch.tensor([[opacity]]),
              colors_precomp=torch.tensor([color], dtype=torch.float32),
              scales=torch.full((1, 3), scale), rotations=torch.tensor([[1.0, 0, 0, 0]]))
    return cam, rv


def _render(cam, rv, aux=False):
    return R.rasterize(rv['means3D'], rv['means2D'], rv['opacities'], rv['colors_precomp'],
                       rv['scales'], rv['rotations'], cam, return_aux=aux)


def test_single_gaussian_centre_pixel():
    # centred Gaussian: alpha = min(0.99, o*exp(0)); colour = alpha*c + (1-alpha)*bg
    bg = (1.0, 1.0, 1.0)
    cam, rv = _single(opacity=0.8, bg=bg)
    col, radii, dep, aux = _render(cam, rv, aux=True)
    a = 0.8
    for ch, c in enumerate((0.2, 0.5, 0.9)):
        assert abs(col[ch, 17, 15].item() - (a * c + (1 - a) * 1.0)) < 1e-5
    assert abs(dep[0, 17, 15].item() - a * 2.0) < 1e-5
    assert abs(aux.final_T[17, 15].item() - (1 - a)) < 1e-6
    # sigma_px = f*s/z = 1 -> cov = 1 + 0.3 -> radius = ceil(3*sqrt(1.3)) = 4
    assert radii[0].item() == 4
    # far pixel sees only background
    assert torch.allclose(col[:, 0, 0], torch.tensor(bg))


def test_opacity_clamp_099():
    cam, rv = _single(opacity=1.0)
    col, _, _, aux = _render(cam, rv, aux=True)
    assert abs(aux.final_T[17, 15].item() - 0.01) < 1e-6


def test_gaussian_profile_matches_closed_form():
    cam, rv = _single(opacity=0.5, color=(1, 1, 1))
    col, *_ = _render(cam, rv)
    # conic = 1/1.3 on the diagonal; 2 px to the right: alpha = 0.5*exp(-0.5*4/1.3)
    want = 0.5 * math.exp(-0.5 * 4 / 1.3)
    assert abs(col[0, 17, 17].item() - want) < 1e-5


def test_two_gaussians_order_and_transmittance():
    W = H = 32
    f, cx, cy = 40.0, 15.5, 15.5
    cam = R.make_camera(W, H, f, f, cx, cy)
    zs = [3.0, 1.5]     # index 0 is farther: must be composited second
    X = torch.tensor([[(15.5 - cx) / f * z, (17.5 - cy) / f * z, z] for z in zs], dtype=torch.float32)
    rv = dict(means3D=X, means2D=torch.zeros(2, 3), opacities=torch.tensor([[0.6], [0.5]]),
              colors_precomp=torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]),
              scales=torch.tensor([[0.075] * 3, [0.0375] * 3]), rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    col, _, dep, aux = _render(cam, rv, aux=True)
    assert abs(col[1, 17, 15].item() - 0.5) < 1e-5                 # near one first
    assert abs(col[0, 17, 15].item() - 0.6 * 0.5) < 1e-5           # far one behind T=0.5
    assert abs(dep[0, 17, 15].item() - (0.5 * 1.5 + 0.3 * 3.0)) < 1e-5
    assert abs(aux.final_T[17, 15].item() - 0.5 * 0.4) < 1e-6
    assert aux.n_contrib[17, 15].item() == 2


def test_equal_depth_tie_breaks_by_index():
    W = H = 32
    f, cx, cy = 40.0, 15.5, 15.5
    cam = R.make_camera(W, H, f, f, cx, cy)
    X = torch.tensor([[(15.5 - cx) / f * 2, (17.5 - cy) / f * 2, 2.0]] * 2, dtype=torch.float32)
    rv = dict(means3D=X, means2D=torch.zeros(2, 3), opacities=torch.tensor([[0.5], [0.5]]),
              colors_precomp=torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]),
              scales=torch.full((2, 3), 0.05), rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    col, *_ = _render(cam, rv)
    assert abs(col[0, 17, 15].item() - 0.5) < 1e-6 and abs(col[1, 17, 15].item() - 0.25) < 1e-6
    cr = c_ref.CRef()
    c2, _, _ = cr.forward(X.numpy(), rv['colors_precomp'].numpy(), rv['opacities'].numpy(), rv['scales'].numpy(),
                          rv['rotations'].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                          cam.tanfovx, cam.tanfovy, W, H, cam.bg.numpy())
    assert abs(c2[0, 17, 15] - 0.5) < 1e-6 and abs(c2[1, 17, 15] - 0.25) < 1e-6


def test_near_plane_and_offscreen_cull():
    cam, rv = _single(z=0.2)              # z <= 0.2 culled
    _, radii, _ = _render(cam, rv)
    assert radii[0].item() == 0
    cam, rv = _single(z=0.21)
    _, radii, _ = _render(cam, rv)
    assert radii[0].item() > 0
    cam, rv = _single(px=(400, 17))       # far right of a 32 px image
    col, radii, _ = _render(cam, rv)
    assert radii[0].item() == 0 and col.abs().max().item() == 0


def test_silhouette_is_one_minus_final_T():
    cam, rv = scene(2000, 96, 64, 90.0, seed=5)
    rv['colors_precomp'] = torch.ones_like(rv['colors_precomp'])
    col, _, _, aux = _render(cam, rv, aux=True)
    assert torch.allclose(col[1], 1 - aux.final_T, atol=1e-5)


def test_fronto_parallel_plane_depth():
    W, H, f, z = 64, 48, 60.0, 2.5
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    cam = R.make_camera(W, H, f, f, cx, cy)
    uu, vv = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing='xy')
    X = torch.stack([(uu.reshape(-1) + 0.5 - cx) / f * z, (vv.reshape(-1) + 0.5 - cy) / f * z, torch.full((W * H,), z)], -1)
    n = X.shape[0]
    rv = dict(means3D=X, means2D=torch.zeros(n, 3), opacities=torch.full((n, 1), 0.9),
              colors_precomp=torch.stack([X[:, 2], torch.ones(n), X[:, 2] ** 2], -1),
              scales=torch.full((n, 3), z / f), rotations=torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1))
    col, *_ = _render(cam, rv)
    sil = col[1, 8:-8, 8:-8]
    assert (sil > 0.99).all()
    assert torch.allclose(col[0, 8:-8, 8:-8] / sil, torch.full_like(sil, z), atol=1e-3)


def test_mark_visible():
    cam, rv = scene(500, 64, 48, 60.0, seed=2)
    rv['means3D'][:50, 2] = 0.1
    m = R.mark_visible(rv['means3D'], cam)
    assert (~m[:50]).all() and m[50:].all()
    assert (c_ref.mark_visible(rv['means3D'].numpy(), cam.viewmatrix.numpy()) == m.numpy()).all()


@pytest.mark.parametrize("aniso,view", [(False, False), (True, False), (True, True)])
def test_c_oracle_matches_autograd_oracle(aniso, view):
    W, H = 160, 112
    cam, rv = scene(3000, W, H, 150.0, seed=1, anisotropic=aniso, w2c=tilted_w2c() if view else None,
                    bg=(1.0, 0.5, 0.2) if view else (0, 0, 0))
    rv = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    col, radii, dep, aux = _render(cam, rv, aux=True)
    gout = torch.randn(col.shape, generator=torch.Generator().manual_seed(3))
    (col * gout).sum().backward()
    cr = c_ref.CRef()
    c2, r2, d2 = cr.forward(rv['means3D'].detach().numpy(), rv['colors_precomp'].detach().numpy(),
                            rv['opacities'].detach().numpy(), rv['scales'].detach().numpy(),
                            rv['rotations'].detach().numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                            cam.tanfovx, cam.tanfovy, W, H, cam.bg.numpy())
    g = cr.backward(gout.numpy())
    assert cr.num_rendered() == int(aux.tile_counts.sum())
    assert (r2 != radii.numpy()).sum() == 0
    assert_close_outliers(c2, col.detach().numpy(), 1e-4, max_outlier_frac=1e-4, outlier_atol=0.02, what="color")
    assert_close_outliers(d2, dep.numpy(), 1e-4, max_outlier_frac=1e-4, outlier_atol=0.05, what="depth")
    for k, kk in [('means3D', 'means3D'), ('means2D', 'means2D'), ('colors', 'colors_precomp'),
                  ('opacities', 'opacities'), ('scales', 'scales'), ('rotations', 'rotations')]:
        ref = rv[kk].grad.numpy()
        assert_close_outliers(g[k], ref, 1e-3 * grad_scale(ref), max_outlier_frac=2e-3,
                              outlier_atol=0.05 * grad_scale(ref), what=k)


def test_finite_differences_float64():
    """Central differences on the float64 autograd oracle for all six inputs."""
    torch.manual_seed(0)
    W, H = 32, 32
    cam, rv = scene(40, W, H, 30.0, seed=7, anisotropic=True, w2c=tilted_w2c(0.15, (0.05, 0.02, 0.1)),
                    bg=(0.3, 0.6, 0.1), dtype=torch.float64)
    rv['scales'] = rv['scales'] * 3.0        # bigger footprints: smoother loss, more overlap
    rv = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    gout = torch.randn(3, H, W, dtype=torch.float64, generator=torch.Generator().manual_seed(1))

    def loss():
        col, _, _ = _render(cam, rv)
        return (col * gout).sum()

    L = loss()
    L.backward()
    rng = np.random.default_rng(0)
    eps = 1e-6
    for k in ('means3D', 'opacities', 'colors_precomp', 'scales', 'rotations', 'means2D'):
        g = rv[k].grad
        flat = rv[k].detach().reshape(-1)
        ncheck, nbad = 0, 0
        for idx in rng.choice(flat.numel(), size=min(12, flat.numel()), replace=False):
            if k == 'means2D' and idx % 3 == 2:
                continue
            old = flat[idx].item()
            with torch.no_grad():
                rv[k].reshape(-1)[idx] = old + eps
                lp = loss().item()
                rv[k].reshape(-1)[idx] = old - eps
                lm = loss().item()
                rv[k].reshape(-1)[idx] = old
            fd = (lp - lm) / (2 * eps)
            an = g.reshape(-1)[idx].item()
            ncheck += 1
            if abs(fd - an) > 1e-4 * max(1.0, abs(an)):
                nbad += 1       # a 1/255 or 1e-4 threshold crossed inside the stencil
        assert nbad <= 1, (k, nbad, ncheck)


# ----------------------------------------------------------------------------------------------------------------------------------
# the decision-flip classifier (ref_flip_bounds): what the GPU parity tests use to EXPLAIN every outlier
# ----------------------------------------------------------------------------------------------------------------------------------

def _both_builds(cam, rv, gout):
    from oracle import c_ref
    out = {}
    for prec in ("f32", "f64"):
        cr = c_ref.CRef(prec)
        col, radii, dep = cr.forward(rv['means3D'].numpy(), rv['colors_precomp'].numpy(), rv['opacities'].numpy(), rv['scales'].numpy(),
                                     rv['rotations'].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy,
                                     cam.image_width, cam.image_height, cam.bg.numpy())
        out[prec] = (col, dep, radii, cr.backward(gout.numpy()))
    return out


@pytest.mark.parametrize("n,W,H,f,seed", [(10000, 320, 240, 288.0, 10000), (60000, 640, 480, 517.0, 5)])
def test_flip_classifier_explains_float32_vs_float64(n, W, H, f, seed):
    """The float32 and float64 builds of the oracle are two correct evaluations: every pixel where they differ by more than 1e-4 must
    hold a decision the classifier finds within rounding of its threshold (and be within the bound of those flips), and every
    gradient row they disagree on by more than 1e-3 of the maximum must lie over a pixel where such a flip happened."""
    from tests.util import assert_grad_outliers_explained, assert_outliers_explained, flip_pixels, oracle_flip_bounds, scene
    cam, rv = scene(n, W, H, f, seed=seed)
    gout = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    b = _both_builds(cam, rv, gout)
    bound, margin, xy, radii, noise = oracle_flip_bounds(rv, cam)
    assert_outliers_explained(b['f32'][0], b['f64'][0], bound[:3], 1e-4, rtol=1e-4, noise=noise[:3], what="float32 vs float64 colour")
    assert_outliers_explained(b['f32'][1], b['f64'][1], bound[3:4], 1e-4, rtol=1e-4, noise=noise[3:4], what="float32 vs float64 depth")
    # the rounding-sensitivity model against the two builds: away from decision flips the float32 build is within a few ulps' worth
    quiet = bound[:3] == 0
    ratio = np.abs(b['f32'][0] - b['f64'][0])[quiet] / np.maximum(noise[:3][quiet], 1e-12)
    moved = np.abs(b['f32'][0] - b['f64'][0])[quiet] > 1e-5
    print(f"float32 vs float64 colour, pixels without a near-threshold decision: |difference| / one-ulp sensitivity: median "
          f"{np.median(ratio[moved]):.2f}, 99 % {np.quantile(ratio[moved], 0.99):.2f}, max {ratio[moved].max():.2f}")
    assert np.quantile(ratio[moved], 0.99) < 3.0         # (the tail beyond: decisions just outside FLIP_TOL, all below the 1e-4 tolerance)
    flips = flip_pixels(bound, (b['f32'][0], b['f32'][1]), (b['f64'][0], b['f64'][1]))
    assert 0 < flips.mean() < 2e-3                       # a few pixels in ten thousand
    for k in ('means3D', 'colors', 'opacities', 'scales'):
        assert_grad_outliers_explained(b['f32'][3][k], b['f64'][3][k], flips, xy, radii, what=f"float32 vs float64 dL/d{k}")


def test_flip_classifier_known_answers():
    """One Gaussian, exact answers: the classifier flags the ring of pixels where alpha crosses 1/255 (and only a ring), bounds a flip
    there by alpha (|c| + cmax) ~ 2/255 |c|, and reports no decision anywhere when the thresholds are far."""
    from oracle import c_ref
    W = H = 64
    cam = R.make_camera(W, H, 60.0, 60.0, 31.5, 31.5)
    z = 2.0
    means = np.array([[0.0, 0.0, z]], np.float32)
    sc = np.full((1, 3), 6.0 * z / 60.0, np.float32)            # sigma = 6 px
    rot = np.array([[1.0, 0, 0, 0]], np.float32)
    col = np.array([[0.5, 0.25, 1.0]], np.float32)
    cr = c_ref.CRef("f64")
    cr.forward(means, col, np.array([0.8], np.float32), sc, rot, cam.viewmatrix.numpy(), cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy,
               W, H, cam.bg.numpy())
    # alpha = 0.8 exp(-r^2 / (2 (36 + 0.3))) = 1/255  at  r = sqrt(2 * 36.3 * ln(204)) = 19.65 px: the margin is smallest on that ring
    bound, margin = cr.flip_bounds(tol=0.05)
    ys, xs = np.nonzero(bound[0] > 0)
    cx, cy = cr.geom()['xy'][0]
    r = np.hypot(xs - cx, ys - cy)
    assert ys.size > 20 and np.abs(r - 19.65).max() < 0.12, (ys.size, r.min(), r.max())       # 5 % in alpha = 0.09 px in r
    np.testing.assert_allclose(bound[0][ys, xs], (1 / 255.0) * (0.5 + 0.5), rtol=0.06)       # alpha T (|c| + cmax), alpha within 5 % of 1/255
    np.testing.assert_allclose(bound[2][ys, xs], (1 / 255.0) * (1.0 + 1.0), rtol=0.06)
    assert margin[int(round(cy)), int(round(cx))] > 0.5                                                           # centre pixel: alpha = 0.8, far from everything
    bound, margin = cr.flip_bounds(tol=1e-5)
    assert not (bound > 0).any()
