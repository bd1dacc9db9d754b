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
