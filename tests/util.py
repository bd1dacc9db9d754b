"""Shared helpers for the parity tests (CPU side)."""
import numpy as np
import torch

from oracle import raster_ref as R


def scene(n, W, H, f, seed=0, anisotropic=False, w2c=None, bg=(0.0, 0.0, 0.0), dtype=torch.float32):
    """Seeded SplaTAM-like scene + camera; returns (settings, rendervar dict)."""
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    cam = R.make_camera(W, H, f, f, cx, cy, w2c=w2c, bg=bg, dtype=dtype)
    p = R.synthetic_cloud(n, W, H, f, f, cx, cy, seed=seed, anisotropic=anisotropic, dtype=dtype)
    if w2c is not None:
        c2w = torch.inverse(torch.as_tensor(w2c, dtype=dtype))
        p['means3D'] = p['means3D'] @ c2w[:3, :3].T + c2w[:3, 3]
    rv = R.cloud_to_rendervar(p)
    return cam, rv


def tilted_w2c(th=0.3, t=(0.1, -0.05, 0.2)):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, 0, s, t[0]], [0, 1, 0, t[1]], [-s, 0, c, t[2]], [0, 0, 0, 1]], dtype=np.float32)


def assert_close_outliers(got, ref, atol, rtol=0.0, max_outlier_frac=0.0, outlier_atol=None, what=""):
    """|got-ref| <= atol + rtol*|ref| everywhere except for a bounded fraction of
    elements (float32 threshold flips: alpha<1/255, T<1e-4, power>0 decisions taken
    one ulp apart on different machines), each of which must stay within outlier_atol."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref)
    bad = err > (atol + rtol * np.abs(ref))
    nbad = int(bad.sum())
    allowed = int(np.floor(max_outlier_frac * err.size))
    if what:
        print(f"{what}: {nbad} of {err.size} elements beyond atol {atol:.3g} + rtol {rtol:.3g} ({nbad / max(err.size, 1):.2e}; allowed "
              f"{max_outlier_frac:.2e}), max err {err.max() if err.size else 0.0:.3e}")
    assert nbad <= allowed, f"{what}: {nbad} elements out of tolerance (allowed {allowed}), max err {err.max():.3e}"
    if nbad and outlier_atol is not None:
        assert err.max() <= outlier_atol, f"{what}: outlier error {err.max():.3e} > {outlier_atol}"


def grad_scale(ref):
    """Gradient comparisons are relative to the tensor's max magnitude."""
    return float(np.abs(np.asarray(ref)).max()) + 1e-20


def grad_error_stats(got, ref, floor=1e-3):
    """Per-element gradient error normalised by max(|ref_i|, floor * max|ref|): returns (normalised errors, scale)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64).reshape(got.shape)
    scale = float(np.abs(ref).max()) + 1e-30
    den = np.maximum(np.abs(ref), floor * scale)
    return np.abs(got - ref) / den, scale


def assert_grad_close(got, ref, what="", rel=1e-3, floor=1e-3, max_outlier_frac=1e-4, global_rel=1e-4, outlier_rel=0.05):
    """Tightened gradient check:
      * per element: |got_i - ref_i| <= rel * max(|ref_i|, floor * max|ref|)  -- a Gaussian with a small gradient may not be
        wrong by 100 % and pass, as it could under a bound relative to the tensor's maximum only;
      * all but `max_outlier_frac` of the elements; the exceptions (float32 threshold flips: one alpha >= 1/255 or T < 1e-4
        decision taken differently for one (pixel, Gaussian) pair) are bounded by outlier_rel * max|ref|;
      * globally the 1 - max_outlier_frac quantile of |got - ref| stays below global_rel * max|ref|."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64).reshape(got.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite gradient"
    nerr, scale = grad_error_stats(got, ref, floor)
    err = np.abs(got - ref)
    bad = nerr > rel
    nbad, allowed = int(bad.sum()), int(np.floor(max_outlier_frac * err.size))
    report = (f"{what}: scale {scale:.3e}, max err {err.max():.3e} ({err.max() / scale:.2e} of max), normalised err "
              f"p50 {np.quantile(nerr, 0.5):.2e} p99 {np.quantile(nerr, 0.99):.2e} p99.99 {np.quantile(nerr, 0.9999):.2e} "
              f"max {nerr.max():.2e}, {nbad} of {err.size} over {rel:g}")
    print(report)
    assert nbad <= allowed, report
    assert err.max() <= outlier_rel * scale, report
    if err.size >= 10000:
        q = float(np.quantile(err, 1.0 - max_outlier_frac))
        assert q <= global_rel * scale, report + f"; quantile {q:.3e}"


def assert_grad_calibrated(got, ref32, ref64, what="", factor=1.3, tail_factor=2.0, floor=1e-3, max_outlier_frac=1e-4, outlier_rel=0.05):
    """Gradient parity judged against the float32 rounding noise of the algorithm itself.

    Two CORRECT float32 evaluations of the rasterizer's backward differ per element by far more than 1e-3 of the element:
    the projected centres (~1e3 px) carry ~1e-4 px of float32 rounding, and the chain to means3D / scales sums terms that
    cancel.  The float64 build of the oracle (oracle/raster_ref.c, -DREF_DOUBLE) measures that noise: e32 = |ref32 - ref64|
    is what the reference-arithmetic CPU implementation itself is off by.  Checked here:
      (1) quantile by quantile (50 .. 99.99 %) the HIP result is no further from the float64 evaluation than `factor` x the
          float32 oracle is (factor 1.3 since round 3: the printed reports show the kernels AT the oracle's own error level;
          the 99.99 % quantile -- a few dozen elements that sit under a float32 threshold flip or a depth tie, different ones in
          every float32 evaluation -- keeps round 2's factor 2: D-anisotropic means3D measured 1.49) (errors normalised per element by max(|ref64_i|, floor * max|ref64|));
      (2) the fraction of elements off by more than 1e-3 of themselves is at most `factor` x the oracle's own fraction;
      (3) the north star's bound: every element within 1e-3 of the tensor's maximum, except a bounded fraction of float32
          threshold flips (alpha >= 1/255, T < 1e-4 decided differently for one (pixel, Gaussian) pair), each below
          outlier_rel * max."""
    got = np.asarray(got, dtype=np.float64)
    ref32 = np.asarray(ref32, dtype=np.float64).reshape(got.shape)
    ref64 = np.asarray(ref64, dtype=np.float64).reshape(got.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite gradient"
    scale = float(np.abs(ref64).max()) + 1e-30
    den = np.maximum(np.abs(ref64), floor * scale)
    eg, e32 = np.abs(got - ref64) / den, np.abs(ref32 - ref64) / den
    qs = tuple(q for q in (0.5, 0.9, 0.99, 0.999, 0.9999) if (1.0 - q) * eg.size >= 30) or (0.5,)      # quantiles the sample can resolve
    qg, q32 = np.quantile(eg, qs), np.quantile(e32, qs)
    fg, f32_ = float((eg > 1e-3).mean()), float((e32 > 1e-3).mean())
    err = np.abs(got - ref32)
    report = (f"{what}: scale {scale:.3e}; normalised error vs float64 at q={qs}: HIP {np.array2string(qg, precision=2)} "
              f"float32 oracle {np.array2string(q32, precision=2)}; fraction > 1e-3: HIP {fg:.2e} oracle {f32_:.2e}; "
              f"max |HIP - ref32| {err.max():.3e} = {err.max() / scale:.2e} of max")
    print(report)
    fac = np.array([tail_factor if q > 0.9995 else factor for q in qs])
    assert (qg <= fac * q32 + 2e-6).all(), report
    assert fg <= factor * f32_ + 1e-4, report
    nbad = int((err > 1e-3 * scale).sum())
    assert nbad <= int(np.floor(max_outlier_frac * err.size)), report + f"; {nbad} elements beyond 1e-3 of max"
    assert err.max() <= outlier_rel * scale, report


def track_loop_loss_rtol(it):
    """Tolerance on the loss of iteration `it` of a fused tracking LOOP against the reference-shaped loop on the drop-in path.  The two
    loops are separate float32 computations of a pose optimisation: a pixel whose silhouette sits at the 0.99 threshold (or whose
    gradient sign flips) enters one loss and not the other, Adam normalises the step, and the poses -- hence every later loss --
    differ from then on.  Measured on the 12 000-Gaussian scene of these tests over builds of the same kernels (deterministic per
    build; scripts/track_loop_spread.py): 7e-8 until the first such pixel, then 6e-5 .. 1.3e-3 by iteration 5."""
    return 1e-3 if it < 3 else 3e-3


# ----------------------------------------------------------------------------------------------------------------------------------
# every outlier explained: the float64 oracle classifies the float32 decision flips
# ----------------------------------------------------------------------------------------------------------------------------------

FLIP_TOL = 2e-3        # a decision whose margin is within this (relative) may fall either way in a correct float32 evaluation:
#                        pixel coordinates ~1e3 carry ~6e-5 px of float32 rounding, `power` (~ -5.5 at the alpha threshold) moves by
#                        ~co * dx * 6e-5 ~ 2e-4, alpha = o exp(power) by as much relatively; T is a product of up to hundreds of such factors
TIE_TOL = 5e-7         # two depths within 4 float32 ulps (4 * 2^-23 relative): their order is decided by rounding
CENTRE_ULPS = 3.0      # projected centres are trusted to this many float32 ulps of their pixel coordinate (~1e-4 px at x ~ 1e3): between
#                        the float32 and float64 builds of the oracle the pixels move by 0.2-0.5 (median) .. 2.8 (99 %) times the
#                        one-ulp sensitivity (a worst-case sum over the pixel's contributors), profiles/r05_experiments.md 1


def oracle_flip_bounds(rv, cam, tol=FLIP_TOL, tie_tol=TIE_TOL):
    """float64 build of the C oracle on the render variables ``rv`` (numpy / torch, any float dtype): returns
    (bound[(C+1),H,W], margin[H,W], xy[P,2], radii[P], noise[(C+1),H,W]) -- see ref_flip_bounds in oracle/raster_ref.c; ``noise``: the
    pixel's sensitivity to ONE float32 ulp of rounding in the projected centres."""
    from oracle import c_ref
    n = lambda t: t.detach().cpu().double().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float64)      # noqa: E731
    cr = c_ref.CRef("f64")
    _, radii, _ = cr.forward(n(rv['means3D']), n(rv['colors_precomp']), n(rv['opacities']), n(rv['scales']), n(rv['rotations']),
                             n(cam.viewmatrix), n(cam.projmatrix), float(cam.tanfovx), float(cam.tanfovy), int(cam.image_width),
                             int(cam.image_height), n(cam.bg), scale_modifier=float(cam.scale_modifier))
    bound, margin, noise = cr.flip_bounds(tol, tie_tol, ulps=1.0)
    return bound, margin, cr.geom()['xy'], radii, noise


def assert_outliers_explained(got, ref, bound, atol, rtol=0.0, slack=1.1, noise=None, ulps=CENTRE_ULPS, what=""):
    """Every element of ``got`` further from ``ref`` than atol + rtol |ref| must be EXPLAINED by the float64 oracle's account of that
    pixel (oracle_flip_bounds), one of:
      * a decision flip: decisions within FLIP_TOL of their threshold (alpha >= 1/255, T (1 - alpha) >= 1e-4, power <= 0, a depth tie, a
        tile-rectangle edge) whose flips can move that channel by ``bound``;
      * rounding of the projected centres (``noise``, per ulp): weight moving between contributors of different colour / depth when
        their float32 centres are off by up to ``ulps`` ulps -- the planes of large values (depth, depth^2 at |c| ~ 4-16) and the
        right / bottom of a 1752-pixel frame are where this exceeds 1e-4 without any decision flipping.
    The element must be within the tolerance + ``slack`` x bound + ``ulps`` x noise.  Returns the number of outliers."""
    got, ref, bound = (np.asarray(a, dtype=np.float64) for a in (got, ref, bound))
    assert got.shape == ref.shape == bound.shape, (what, got.shape, ref.shape, bound.shape)
    noise = np.zeros_like(bound) if noise is None else np.asarray(noise, dtype=np.float64)
    err = np.abs(got - ref)
    lim = atol + rtol * np.abs(ref)
    bad = err > lim
    by_flip = bad & (err <= lim + slack * bound)
    by_noise = bad & ~by_flip & (err <= lim + slack * bound + ulps * noise)
    unexplained = bad & ~by_flip & ~by_noise
    nbad, nun = int(bad.sum()), int(unexplained.sum())
    fl = by_flip & (bound > 0)
    ratio = float((err[fl] / bound[fl]).max()) if fl.any() else 0.0
    worst_ulps = float(((err[by_noise] - lim[by_noise] - slack * bound[by_noise]) / noise[by_noise]).max()) if by_noise.any() else 0.0
    print(f"{what}: {nbad} of {err.size} elements beyond {atol:g} + {rtol:g}|ref| (max err {err.max() if err.size else 0.0:.3e}); "
          f"{int(by_flip.sum())} explained by float32 decision flips (largest err / flip bound {ratio:.2f}), {int(by_noise.sum())} by "
          f"centre rounding (largest: {worst_ulps:.2f} ulps of {ulps:g} allowed), {nun} unexplained")
    if nun:
        idx = np.argwhere(unexplained)[:5]
        detail = [(tuple(int(v) for v in i), float(err[tuple(i)]), float(bound[tuple(i)]), float(noise[tuple(i)])) for i in idx]
        raise AssertionError(f"{what}: {nun} outliers without explanation: (index, err, flip bound, noise per ulp) {detail}")
    return nbad


def flip_pixels(bound, got_images, ref_images, noise=2e-5):
    """Pixels where a float32 decision flip HAPPENED between two evaluations: the float64 oracle found a near-threshold decision there
    (``bound`` > 0) AND the two forward results differ by more than evaluation-order rounding (``noise``, absolute + relative; without
    a flip the planes agree to ~4e-6 / ~2e-5 |depth|).  A few hundred of the ~1e6 pixels of a full-size frame -- against ~1.7 % that
    merely hold a near-threshold decision -- which is what makes assert_grad_outliers_explained a sharp statement."""
    flagged = np.asarray(bound).sum(axis=0) > 0
    changed = np.zeros_like(flagged)
    for g, r in zip(got_images, ref_images):
        g, r = np.asarray(g, dtype=np.float64), np.asarray(r, dtype=np.float64)
        changed |= (np.abs(g - r) > noise + noise * np.abs(r)).reshape((-1,) + flagged.shape).any(axis=0)
    return flagged & changed


def assert_grad_outliers_explained(got, ref, flagged, xy, radii, rel=1e-3, what=""):
    """Every ROW of a per-Gaussian gradient with an element further than rel * max|ref| from ``ref`` must lie over a pixel where a
    float32 decision flip happened (``flagged`` [H,W] bool from flip_pixels): a flip changes T for every Gaussian behind it at that
    pixel, so it moves the gradients of the Gaussians whose footprint (centre +- radius) covers the pixel."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64).reshape(got.shape)
    scale = float(np.abs(ref).max()) + 1e-30
    rows = np.nonzero((np.abs(got - ref).reshape(got.shape[0], -1) > rel * scale).any(axis=1))[0]
    H, W = flagged.shape
    integral = np.zeros((H + 1, W + 1), dtype=np.int64)
    integral[1:, 1:] = np.cumsum(np.cumsum(flagged.astype(np.int64), axis=0), axis=1)
    missing = []
    for i in rows:
        r = int(radii[i])
        x0, x1 = int(np.floor(xy[i, 0] - r)), int(np.ceil(xy[i, 0] + r)) + 1
        y0, y1 = int(np.floor(xy[i, 1] - r)), int(np.ceil(xy[i, 1] + r)) + 1
        x0, x1, y0, y1 = max(x0, 0), min(x1, W), max(y0, 0), min(y1, H)
        n = integral[y1, x1] - integral[y0, x1] - integral[y1, x0] + integral[y0, x0] if (x1 > x0 and y1 > y0) else 0
        if n == 0:
            missing.append(int(i))
    frac = float(flagged.mean())
    print(f"{what}: {rows.size} of {got.shape[0]} rows beyond {rel:g} of max; {rows.size - len(missing)} lie over a pixel with a decision flip "
          f"({100 * frac:.3f} % of the pixels are flagged), {len(missing)} do not")
    assert not missing, f"{what}: gradient rows {missing[:8]} differ by more than {rel:g} of max with no decision flip under them"
    return int(rows.size)
