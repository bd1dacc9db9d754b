"""ctypes view of the C ABI in include/splat_hip.h (libsplat_hip.so).

The library is the product: there is NO fallback.  ``lib()`` raises if the
shared object has not been built (``python -m splatam_amd.build`` or
``__graft_entry__.build()``), and every rasterizer entry point goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsplat_hip.so")

SPLAT_TILE = 16
SPLAT_MAX_CHANNELS = 8
SPLAT_GRAD_STRIDE = 16
SPLAT_COUNTER_STRIDE = 32
ABI_VERSION = 1

_fp = C.c_void_p  # device pointers travel as integers


class SplatCamera(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("bg", _fp), ("scale_modifier", C.c_float),
                ("viewmatrix", _fp), ("projmatrix", _fp),
                ("sh_degree", C.c_int32), ("campos", _fp), ("prefiltered", C.c_int32)]


class SplatGaussians(C.Structure):
    _fields_ = [("P", C.c_int32), ("channels", C.c_int32),
                ("means3D", _fp), ("opacities", _fp), ("colors_precomp", _fp),
                ("scales", _fp), ("rotations", _fp), ("cov3D_precomp", _fp),
                ("shs", _fp), ("sh_coeffs", C.c_int32)]


class SplatState(C.Structure):
    _fields_ = [("depth", _fp), ("xy", _fp), ("conic_opacity", _fp), ("rect", _fp), ("radii", _fp),
                ("rgb", _fp), ("clamped", _fp),
                ("tile_count", _fp), ("tile_base", _fp), ("tile_cursor", _fp),
                ("keys", _fp), ("point_list", _fp), ("capacity", C.c_int64), ("max_list_hint", C.c_int32),
                ("final_T", _fp), ("n_contrib", _fp), ("status", _fp)]


class SplatGrads(C.Structure):
    _fields_ = [("dL_dcolor", _fp), ("accum", _fp), ("dL_dmeans3D", _fp), ("dL_dmeans2D", _fp),
                ("dL_dcolors", _fp), ("dL_dopacities", _fp), ("dL_dscales", _fp), ("dL_drotations", _fp),
                ("dL_dcov3D", _fp), ("dL_dshs", _fp)]


EXPORTS = (
    "splat_error_string", "splat_abi_version", "splat_num_tiles",
    "splat_preprocess_forward", "splat_bin_forward", "splat_render_forward", "splat_forward",
    "splat_render_backward", "splat_preprocess_backward", "splat_backward",
    "splat_mark_visible", "splat_time_kernel", "splat_debug_option",
)

_lib = None


def lib():
    """Load libsplat_hip.so (after torch, so that it binds to the HIP runtime
    torch already loaded) and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP rasterizer has not been built. "
            "Run `python -m splatam_amd.build` (needs hipcc); there is no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64 first)
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
    L.splat_error_string.restype = C.c_char_p
    L.splat_error_string.argtypes = [C.c_int]
    L.splat_abi_version.restype = C.c_int
    L.splat_num_tiles.restype = C.c_size_t
    L.splat_num_tiles.argtypes = [C.c_int32, C.c_int32]
    cam, g, st, gr = C.POINTER(SplatCamera), C.POINTER(SplatGaussians), C.POINTER(SplatState), C.POINTER(SplatGrads)
    for name, args in (
        ("splat_preprocess_forward", [cam, g, st, _fp]),
        ("splat_bin_forward", [cam, g, st, _fp]),
        ("splat_render_forward", [cam, g, st, _fp, _fp, _fp]),
        ("splat_forward", [cam, g, st, _fp, _fp, _fp]),
        ("splat_render_backward", [cam, g, st, gr, _fp]),
        ("splat_preprocess_backward", [cam, g, st, gr, _fp]),
        ("splat_backward", [cam, g, st, gr, _fp]),
        ("splat_mark_visible", [C.c_int32, _fp, _fp, _fp, _fp]),
        ("splat_time_kernel", [C.c_int, C.c_int, cam, g, st, gr, _fp, _fp, _fp, C.POINTER(C.c_float)]),
    ):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = args
    L.splat_debug_option.restype = C.c_int
    L.splat_debug_option.argtypes = [C.c_int, C.c_int]
    if hasattr(L, "splat_selftest"):
        L.splat_selftest.restype = C.c_int
        L.splat_selftest.argtypes = [C.c_int, _fp, _fp, C.c_int, _fp]
    if L.splat_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {L.splat_abi_version()} vs binding {ABI_VERSION}")
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().splat_error_string(rc).decode()} (code {rc})")
