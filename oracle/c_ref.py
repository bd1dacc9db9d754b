"""ctypes wrapper around the plain-C oracle (oracle/raster_ref.c).

TEST INFRASTRUCTURE ONLY (see the header of raster_ref.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraster_ref.so")
_SO64 = os.path.join(_HERE, "_build", "libraster_ref_f64.so")      # the same statements in float64 (-DREF_DOUBLE)
_libs = {}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "raster_ref.c")
    stale = any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, _SO64))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib(precision: str = "f32"):
    if precision not in ("f32", "f64"):
        raise ValueError(precision)
    if precision not in _libs:
        so = _SO if precision == "f32" else _SO64
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        rt_real = C.c_float if precision == "f32" else C.c_double
        assert L.ref_real_bytes() == C.sizeof(rt_real)
        L.ref_create.restype = C.c_void_p
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_num_rendered.restype = C.c_int64
        L.ref_num_rendered.argtypes = [C.c_void_p]
        L.ref_num_pairs.restype = C.c_int64
        L.ref_num_pairs.argtypes = [C.c_void_p]
        for name, rt in (("ref_ranges", C.c_int), ("ref_list", C.c_int), ("ref_final_T", rt_real),
                         ("ref_n_contrib", C.c_int), ("ref_geom_xy", rt_real),
                         ("ref_geom_conic_op", rt_real), ("ref_geom_depth", rt_real)):
            f = getattr(L, name)
            f.restype = C.POINTER(rt)
            f.argtypes = [C.c_void_p]
        L.ref_forward.restype = C.c_int
        L.ref_flip_bounds.restype = None
        L.ref_backward.restype = C.c_int
        L.ref_set_upstream_scale.restype = None
        L.ref_set_upstream_scale.argtypes = [C.c_void_p, C.c_int]
        _libs[precision] = L
    return _libs[precision]


class CRef:
    """One forward (+ optional backward) of the C oracle on numpy arrays.

    Matrices are passed exactly as the settings tuple holds them
    (/root/reference/utils/recon_helpers.py:8-13): viewmatrix = w2c^T,
    projmatrix = (P w2c)^T, flattened row-major (== column-major of w2c).
    ``precision="f64"``: the float64 build of the same statements (inputs are the float32 values, widened)."""

    def __init__(self, precision: str = "f32"):
        self.L = lib(precision)
        self.dt = np.float32 if precision == "f32" else np.float64
        self.ct = C.c_float if precision == "f32" else C.c_double
        self.ctx = C.c_void_p(self.L.ref_create())

    def __del__(self):
        try:
            self.L.ref_destroy(self.ctx)
        except Exception:
            pass

    def _f(self, a):
        # float32 inputs widen exactly for the f64 build; float64 inputs (a float64 caller-side chain) pass unrounded
        return None if a is None else np.ascontiguousarray(a, dtype=self.dt)

    def _p(self, a, t=None):
        return None if a is None else a.ctypes.data_as(C.POINTER(t or self.ct))

    def forward(self, means3D, colors, opacities, scales, rotations, view, proj, tanfovx, tanfovy,
                W, H, bg, scale_modifier=1.0, cov3D_precomp=None):
        _f, _p = self._f, self._p
        self.a = dict(means3D=_f(means3D), colors=_f(colors), opac=_f(np.reshape(opacities, -1)),
                      scales=_f(scales), rot=_f(rotations), view=_f(np.reshape(view, -1)),
                      proj=_f(np.reshape(proj, -1)), bg=_f(bg), cov=_f(cov3D_precomp))
        a = self.a
        P, Cc = a['means3D'].shape[0], a['colors'].shape[1]
        self.P, self.C, self.W, self.H = P, Cc, W, H
        self.tan = (float(np.float32(tanfovx)), float(np.float32(tanfovy)))
        self.mod = float(np.float32(scale_modifier))
        color = np.zeros((Cc, H, W), self.dt)
        depth = np.zeros((1, H, W), self.dt)
        radii = np.zeros(P, np.int32)
        rc = self.L.ref_forward(self.ctx, P, Cc, W, H, _p(a['bg']), _p(a['means3D']), _p(a['colors']),
                                _p(a['opac']), _p(a['scales']), self.ct(self.mod), _p(a['rot']), _p(a['cov']),
                                _p(a['view']), _p(a['proj']), self.ct(self.tan[0]), self.ct(self.tan[1]),
                                _p(color), _p(depth), _p(radii, C.c_int))
        if rc != 0:
            raise RuntimeError(f"ref_forward failed: {rc}")
        return color, radii, depth

    # saved state accessors -------------------------------------------------
    def num_rendered(self):
        return int(self.L.ref_num_rendered(self.ctx))

    def num_pairs(self):
        """Live (pixel, Gaussian) pairs composited by the last forward (the secondary ceiling of SURVEY.md 8d)."""
        return int(self.L.ref_num_pairs(self.ctx))

    def _arr(self, fn, n, dt):
        ptr = getattr(self.L, fn)(self.ctx)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

    def final_T(self):
        return self._arr("ref_final_T", self.W * self.H, self.dt).reshape(self.H, self.W)

    def n_contrib(self):
        return self._arr("ref_n_contrib", self.W * self.H, np.int32).reshape(self.H, self.W)

    def ranges(self):
        tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        return self._arr("ref_ranges", tiles + 1, np.int32)

    def point_list(self):
        n = self.num_rendered()
        return self._arr("ref_list", n, np.int32) if n else np.zeros(0, np.int32)

    def geom(self):
        return dict(xy=self._arr("ref_geom_xy", 2 * self.P, self.dt).reshape(-1, 2),
                    conic_op=self._arr("ref_geom_conic_op", 4 * self.P, self.dt).reshape(-1, 4),
                    depth=self._arr("ref_geom_depth", self.P, self.dt))

    def flip_bounds(self, tol=2e-3, tie_tol=1e-6, ulps=None):
        """Per pixel and output channel (the C colour channels, then depth): how much float32 decision flips can move the pixel --
        summed over the decisions of the last forward that were within ``tol`` (relative) of their threshold, or depth ties within
        ``tie_tol`` -- and the smallest margin of any decision of the pixel.  With ``ulps``: also the pixel's sensitivity to ``ulps``
        float32 ulps of rounding in the projected centres (third result).  See ref_flip_bounds in raster_ref.c."""
        bound = np.zeros((self.C + 1, self.H, self.W), self.dt)
        margin = np.zeros((self.H, self.W), self.dt)
        noise = np.zeros((self.C + 1, self.H, self.W), self.dt) if ulps is not None else None
        self.L.ref_flip_bounds(self.ctx, self._p(self.a['colors']), self.ct(tol), self.ct(tie_tol), self._p(bound), self._p(margin),
                               self.ct(ulps or 0.0), self._p(noise))
        return (bound, margin) if ulps is None else (bound, margin, noise)

    def set_upstream_scale(self, on: bool):
        """dscales without the scale_modifier factor: the numbers the CUDA original returns (ref_set_upstream_scale)."""
        self.L.ref_set_upstream_scale(self.ctx, int(bool(on)))

    def backward(self, dL_dcolor):
        a = self.a
        _f, _p = self._f, self._p
        P, Cc = self.P, self.C
        g = _f(dL_dcolor)
        z = lambda *shape: np.zeros(shape, self.dt)      # noqa: E731
        out = dict(means3D=z(P, 3), means2D=z(P, 3), colors=z(P, Cc), opacities=z(P, 1), cov3D=z(P, 6))
        use_sr = a['cov'] is None
        if use_sr:
            out['scales'] = z(P, 3)
            out['rotations'] = z(P, 4)
        rc = self.L.ref_backward(self.ctx, _p(a['bg']), _p(a['means3D']), _p(a['colors']),
                                 _p(a['scales']) if use_sr else None, self.ct(self.mod),
                                 _p(a['rot']) if use_sr else None,
                                 _p(a['view']), _p(a['proj']), self.ct(self.tan[0]), self.ct(self.tan[1]),
                                 _p(g), _p(out['means3D']), _p(out['means2D']), _p(out['colors']),
                                 _p(out['opacities']), _p(out.get('scales')), _p(out.get('rotations')),
                                 _p(out['cov3D']))
        if rc != 0:
            raise RuntimeError(f"ref_backward failed: {rc}")
        return out


def mark_visible(means3D, view):
    m = np.ascontiguousarray(means3D, dtype=np.float32)
    v = np.ascontiguousarray(np.reshape(view, -1), dtype=np.float32)
    out = np.zeros(m.shape[0], np.uint8)
    fp = C.POINTER(C.c_float)
    lib().ref_mark_visible(m.shape[0], m.ctypes.data_as(fp), v.ctypes.data_as(fp), out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out.astype(bool)


# --------------------------------------------------------------------------
# The C oracle behind the reference's Renderer call surface, with autograd (CPU tensors).  Lets the tests run the
# reference-shaped get_loss (both renders + loss.backward()) end to end on the oracle at full size, where the dense
# torch oracle (raster_ref.py) is too slow.
# --------------------------------------------------------------------------
try:
    import torch

    class _CRasterize(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, settings):
            cr = CRef("f64" if means3D.dtype == torch.float64 else "f32")      # float64 tensors -> the float64 build
            s = settings
            col, radii, dep = cr.forward(means3D.detach().numpy(), colors.detach().numpy(), opacities.detach().numpy(),
                                         scales.detach().numpy(), rotations.detach().numpy(),
                                         s.viewmatrix.detach().numpy(), s.projmatrix.detach().numpy(), float(s.tanfovx),
                                         float(s.tanfovy), int(s.image_width), int(s.image_height), s.bg.detach().numpy(),
                                         scale_modifier=float(s.scale_modifier))
            ctx.cr = cr
            ctx.opac_shape = tuple(opacities.shape)
            radii_t = torch.from_numpy(radii)
            ctx.mark_non_differentiable(radii_t)
            return torch.from_numpy(col), radii_t, torch.from_numpy(dep)

        @staticmethod
        def backward(ctx, g_color, _g_radii, _g_depth):
            g = ctx.cr.backward(g_color.contiguous().numpy())
            t = torch.from_numpy
            return (t(g['means3D']), t(g['means2D']), t(g['colors']), t(g['opacities']).reshape(ctx.opac_shape),
                    t(g['scales']), t(g['rotations']), None)

    class CRasterizer(torch.nn.Module):
        """``Renderer(raster_settings=cam)(**rendervar)`` (/root/reference/scripts/splatam.py:249) on the C oracle."""

        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, colors_precomp=None, scales=None, rotations=None, shs=None,
                    cov3D_precomp=None):
            if shs is not None or cov3D_precomp is not None or colors_precomp is None or scales is None or rotations is None:
                raise NotImplementedError("the C oracle wrapper takes colors_precomp + scales + rotations")
            return _CRasterize.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self.raster_settings)
except ImportError:                     # numpy-only users of CRef
    pass
