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
    hdr = os.path.join(HERE, "..", "splatam_amd", "csrc", "splat_math.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
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
