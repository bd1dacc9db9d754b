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
    assert nbad <= allowed, f"{what}: {nbad} elements out of tolerance (allowed {allowed}), max err {err.max():.3e}"
    if nbad and outlier_atol is not None:
        assert err.max() <= outlier_atol, f"{what}: outlier error {err.max():.3e} > {outlier_atol}"


def grad_scale(ref):
    """Gradient comparisons are relative to the tensor's max magnitude."""
    return float(np.abs(np.asarray(ref)).max()) + 1e-20
