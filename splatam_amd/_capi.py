"""ctypes view of the C ABI in include/splat_hip.h (libsplat_hip.so).

The library is the product: there is NO fallback.  ``lib()`` raises if the
shared object has not been built (``python -m splatam_amd.build`` or
``__graft_entry__.build()``), and every rasterizer entry point goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPLAT_HIP_LIB: developer override, A/B timing of two builds in one process launch each: scripts/ab_lib.sh)
LIB_PATH = os.environ.get("SPLAT_HIP_LIB") or os.path.join(_HERE, "lib", "libsplat_hip.so")

SPLAT_TILE = 16
SPLAT_MAX_CHANNELS = 8
SPLAT_GRAD_STRIDE = 16
SPLAT_COUNTER_STRIDE = 32
SPLAT_GROUP_TILES = 2
ABI_VERSION = 10

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
                ("keys", _fp), ("point_list", _fp), ("capacity", C.c_int64), ("tile_recs", _fp), ("keys_alt", _fp), ("long_base", _fp), ("long_items", _fp),
                ("group_count", _fp), ("group_recs", _fp),
                ("max_list_hint", C.c_int32), ("order_hint", C.c_int32), ("sub_bins", C.c_int32), ("tile_stride", C.c_int32),
                ("group_stride", C.c_int32), ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
                ("tile_work", _fp), ("tile_order", _fp),
                ("final_T", _fp), ("n_contrib", _fp), ("status", _fp), ("status_host", _fp), ("accum_to_zero", _fp)]


class SplatGrads(C.Structure):
    _fields_ = [("dL_dcolor", _fp), ("accum", _fp), ("dL_dmeans3D", _fp), ("dL_dmeans2D", _fp),
                ("dL_dcolors", _fp), ("dL_dopacities", _fp), ("dL_dscales", _fp), ("dL_drotations", _fp),
                ("dL_dcov3D", _fp), ("dL_dshs", _fp), ("flags", C.c_int32)]


class SplatMap(C.Structure):
    _fields_ = [("P", C.c_int32), ("isotropic", C.c_int32),
                ("means3D", _fp), ("rgb_colors", _fp), ("unnorm_rotations", _fp), ("logit_opacities", _fp),
                ("log_scales", _fp), ("cam_unnorm_rots", _fp), ("cam_trans", _fp), ("num_frames", C.c_int32)]


class SplatFrameData(C.Structure):
    _fields_ = [("im", _fp), ("depth", _fp), ("w2c", _fp), ("time_idx", C.c_int32)]


class SplatLossConfig(C.Structure):
    _fields_ = [("tracking", C.c_int32), ("camera_grad", C.c_int32), ("gaussians_grad", C.c_int32),
                ("use_sil_for_loss", C.c_int32), ("sil_thres", C.c_float), ("use_l1", C.c_int32),
                ("ignore_outlier_depth_loss", C.c_int32), ("w_im", C.c_float), ("w_depth", C.c_float),
                ("defer_finish", C.c_int32), ("fused_composite", C.c_int32)]


class SplatIterWorkspace(C.Structure):
    _fields_ = [("st", SplatState), ("feat8", _fp), ("out6", _fp), ("dL_dout6", _fp), ("accum", _fp),
                ("ssim_maps", _fp), ("sums", _fp), ("max_2D_radius", _fp),
                ("d_means3D", _fp), ("d_rgb_colors", _fp), ("d_unnorm_rotations", _fp), ("d_logit_opacities", _fp),
                ("d_log_scales", _fp), ("d_cam", _fp), ("outlier_err", _fp), ("outlier_scratch", _fp)]


class SplatAdamMap(C.Structure):
    _fields_ = [("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("bc2_sqrt", C.c_float * 5),
                ("step_size", C.c_float * 5), ("grad", _fp * 5), ("exp_avg", _fp * 5), ("exp_avg_sq", _fp * 5), ("gate", _fp)]


class SplatMapStore(C.Structure):
    _fields_ = [("map", SplatMap), ("capacity", C.c_int32), ("exp_avg", _fp * 5), ("exp_avg_sq", _fp * 5),
                ("max_2D_radius", _fp), ("means2D_gradient_accum", _fp), ("denom", _fp), ("timestep", _fp), ("counts", _fp)]


class SplatAddArgs(C.Structure):
    _fields_ = [("mode", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("im", _fp), ("depth", _fp), ("out6", _fp),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("sil_thres", C.c_float),
                ("time_idx", C.c_int32), ("w2c", _fp), ("err", _fp), ("scratch", _fp)]


class SplatPruneArgs(C.Structure):
    _fields_ = [("removal_opacity_threshold", C.c_float), ("remove_big", C.c_int32), ("big_scale", C.c_float),
                ("to_remove", _fp), ("flags", _fp), ("stage", _fp), ("scratch", _fp)]


class SplatDensifyArgs(C.Structure):
    _fields_ = [("mode", C.c_int32), ("grad_thresh", C.c_float), ("small_scale", C.c_float), ("rows_with_grad", C.c_int32),
                ("num_to_split_into", C.c_int32), ("samples", _fp), ("flags", _fp), ("scratch", _fp)]


class SplatPoseAdam(C.Structure):
    _fields_ = [("state", _fp), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("bc2_sqrt", C.c_float),
                ("step_size_rot", C.c_float), ("step_size_trans", C.c_float)]


class SplatArrayInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("bytes", C.c_size_t), ("offset", C.c_size_t), ("zero_init", C.c_int32)]


MIRRORED_STRUCTS = (SplatCamera, SplatGaussians, SplatState, SplatGrads, SplatMap, SplatFrameData, SplatLossConfig, SplatIterWorkspace,
                    SplatAdamMap, SplatMapStore, SplatAddArgs, SplatPruneArgs, SplatDensifyArgs, SplatPoseAdam, SplatArrayInfo)


SPLAT_ADD_VALID_DEPTH = 0
SPLAT_ADD_NON_PRESENCE = 1
SPLAT_DENSIFY_CLONE = 0
SPLAT_DENSIFY_SPLIT = 1
SPLAT_ITER_SUMS = 32
SPLAT_ITER_SUM_COPIES = 64
SPLAT_POSE_STATE = 24
SPLAT_ITER_DCAM = 32
SPLAT_SLAB_ALIGN = 256
SPLAT_LAYOUT_SH, SPLAT_LAYOUT_LONG_LISTS, SPLAT_LAYOUT_BACKWARD, SPLAT_LAYOUT_SSIM, SPLAT_LAYOUT_OUTLIER = 1, 2, 4, 8, 16
SPLAT_LAYOUT_GROUPS, SPLAT_LAYOUT_TILE_ORDER, SPLAT_LAYOUT_RECS = 32, 64, 128
SPLAT_GRADS_UPSTREAM_SCALE, SPLAT_GRADS_POISON_IF_FLAGGED, SPLAT_GRADS_ACCUM_ZEROED = 1, 2, 4
SPLAT_LAYOUT_MAX_ARRAYS = 48

EXPORTS = (
    "splat_error_string", "splat_abi_version", "splat_sizeof", "splat_num_tiles",
    "splat_preprocess_forward", "splat_bin_forward", "splat_render_forward", "splat_forward",
    "splat_render_backward", "splat_preprocess_backward", "splat_backward",
    "splat_mark_visible", "splat_same_geometry", "splat_time_kernel", "splat_debug_option", "splat_debug_stamps",
    "splat_iter_loss_backward", "splat_iter_adam_map", "splat_iter_adam_pose", "splat_iter_time_kernel",
    "splat_iter_tracking_step", "splat_iter_mapping_step", "splat_iter_finish", "splat_iter_fold_sums", "splat_iter_render", "splat_map_scratch_words", "splat_map_row_floats", "splat_map_add_new_gaussians", "splat_map_prune",
    "splat_iter_means2d_accumulate", "splat_map_densify_select", "splat_map_duplicate",
    "splat_workspace_bytes", "splat_state_layout", "splat_state_bind", "splat_iter_workspace_layout", "splat_iter_workspace_bind",
    "splat_iter_workspace_bytes",
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
        ("splat_same_geometry", [C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
        ("splat_time_kernel", [C.c_int, C.c_int, cam, g, st, gr, _fp, _fp, _fp, C.POINTER(C.c_float)]),
    ):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = args
    L.splat_iter_loss_backward.restype = C.c_int
    L.splat_iter_loss_backward.argtypes = [cam, C.POINTER(SplatMap), C.POINTER(SplatFrameData), C.POINTER(SplatLossConfig),
                                           C.POINTER(SplatIterWorkspace), _fp]
    L.splat_iter_adam_map.restype = C.c_int
    L.splat_iter_adam_map.argtypes = [C.POINTER(SplatMap), C.POINTER(SplatAdamMap), _fp]
    L.splat_iter_adam_pose.restype = C.c_int
    L.splat_iter_adam_pose.argtypes = [C.POINTER(SplatMap), C.c_int32, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, _fp]
    L.splat_iter_time_kernel.restype = C.c_int
    L.splat_iter_time_kernel.argtypes = [C.c_int, C.c_int, cam, C.c_int32, C.POINTER(SplatIterWorkspace), _fp, C.POINTER(C.c_float)]
    L.splat_iter_tracking_step.restype = C.c_int
    L.splat_iter_tracking_step.argtypes = [cam, C.POINTER(SplatMap), C.POINTER(SplatFrameData), C.POINTER(SplatLossConfig),
                                           C.POINTER(SplatIterWorkspace), C.POINTER(SplatPoseAdam), _fp]
    L.splat_iter_mapping_step.restype = C.c_int
    L.splat_iter_mapping_step.argtypes = [cam, C.POINTER(SplatMap), C.POINTER(SplatFrameData), C.POINTER(SplatLossConfig),
                                          C.POINTER(SplatIterWorkspace), C.POINTER(SplatAdamMap), _fp]
    L.splat_iter_finish.restype = C.c_int
    L.splat_iter_finish.argtypes = [cam, C.POINTER(SplatMap), C.POINTER(SplatFrameData), C.POINTER(SplatLossConfig),
                                    C.POINTER(SplatIterWorkspace), C.POINTER(SplatPoseAdam), _fp]
    L.splat_iter_fold_sums.restype = C.c_int
    L.splat_iter_fold_sums.argtypes = [_fp, _fp]
    L.splat_iter_render.restype = C.c_int
    L.splat_iter_render.argtypes = [cam, C.POINTER(SplatMap), C.POINTER(SplatFrameData), C.POINTER(SplatIterWorkspace), _fp]
    L.splat_map_scratch_words.restype = C.c_size_t
    L.splat_map_scratch_words.argtypes = [C.c_int64]
    L.splat_map_row_floats.restype = C.c_int32
    L.splat_map_row_floats.argtypes = [C.POINTER(SplatMapStore)]
    L.splat_map_add_new_gaussians.restype = C.c_int
    L.splat_map_add_new_gaussians.argtypes = [C.POINTER(SplatMapStore), C.POINTER(SplatAddArgs), _fp]
    L.splat_map_prune.restype = C.c_int
    L.splat_map_prune.argtypes = [C.POINTER(SplatMapStore), C.POINTER(SplatPruneArgs), _fp]
    L.splat_iter_means2d_accumulate.restype = C.c_int
    L.splat_iter_means2d_accumulate.argtypes = [cam, C.POINTER(SplatMap), C.POINTER(SplatIterWorkspace), _fp, _fp, _fp, _fp]
    for name in ("splat_map_densify_select", "splat_map_duplicate"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(SplatMapStore), C.POINTER(SplatDensifyArgs), _fp]
    info = C.POINTER(SplatArrayInfo)
    L.splat_workspace_bytes.restype = C.c_size_t
    L.splat_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    L.splat_state_layout.restype = C.c_int
    L.splat_state_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, info, C.c_int32, C.POINTER(C.c_size_t)]
    L.splat_state_bind.restype = C.c_int
    L.splat_state_bind.argtypes = [st, gr, _fp, info, C.c_int32, C.c_int32, C.c_int64]
    L.splat_iter_workspace_layout.restype = C.c_int
    L.splat_iter_workspace_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, info, C.c_int32, C.POINTER(C.c_size_t)]
    L.splat_iter_workspace_bind.restype = C.c_int
    L.splat_iter_workspace_bind.argtypes = [C.POINTER(SplatIterWorkspace), _fp, info, C.c_int32, C.c_int64, C.c_int32]
    L.splat_iter_workspace_bytes.restype = C.c_size_t
    L.splat_iter_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
    L.splat_debug_option.restype = C.c_int
    L.splat_debug_option.argtypes = [C.c_int, C.c_int]
    L.splat_debug_stamps.restype = C.c_int
    L.splat_debug_stamps.argtypes = [_fp]
    if hasattr(L, "splat_selftest"):
        L.splat_selftest.restype = C.c_int
        L.splat_selftest.argtypes = [C.c_int, _fp, _fp, C.c_int, _fp]
    if L.splat_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {L.splat_abi_version()} vs binding {ABI_VERSION}")
    L.splat_sizeof.restype = C.c_size_t
    L.splat_sizeof.argtypes = [C.c_char_p]
    for cls in MIRRORED_STRUCTS:        # the hand-written mirrors against the compiled layout
        if L.splat_sizeof(cls.__name__.encode()) != C.sizeof(cls):
            raise RuntimeError(f"struct layout mismatch: {cls.__name__} is {L.splat_sizeof(cls.__name__.encode())} bytes in the library, "
                               f"{C.sizeof(cls)} in the binding")
    _lib = L
    return L


class Layout:
    """A scratch layout as the library describes it (include/splat_hip.h, "Scratch layouts"): ``arrays`` (the C table, for
    splat_*_bind), ``n``, ``total`` bytes of the slab, and by field name ``bytes`` / ``offset`` / ``zero_init``."""

    def __init__(self, arrays, n, total):
        self.arrays, self.n, self.total = arrays, n, total
        self.names = [arrays[i].name.decode() for i in range(n)]
        self.bytes = {self.names[i]: int(arrays[i].bytes) for i in range(n)}
        self.offset = {self.names[i]: int(arrays[i].offset) for i in range(n)}
        self.zero_init = {self.names[i]: bool(arrays[i].zero_init) for i in range(n)}


_state_layouts: dict = {}


def state_layout(P, width, height, sub_bins, capacity, flags):
    """(memoised: the drop-in path asks twice per rasterizer call, and a layout is a pure function of its arguments)"""
    key = (P, width, height, sub_bins, capacity, flags)
    hit = _state_layouts.get(key)
    if hit is None:
        if len(_state_layouts) > 256:
            _state_layouts.clear()
        hit = _state_layouts[key] = _state_layout(*key)
    return hit


def _state_layout(P, width, height, sub_bins, capacity, flags):
    arrays = (SplatArrayInfo * SPLAT_LAYOUT_MAX_ARRAYS)()
    total = C.c_size_t(0)
    n = lib().splat_state_layout(P, width, height, sub_bins, capacity, flags, arrays, SPLAT_LAYOUT_MAX_ARRAYS, C.byref(total))
    if n < 0 or n > SPLAT_LAYOUT_MAX_ARRAYS:
        raise RuntimeError(f"splat_state_layout({P}, {width}, {height}, {sub_bins}, {capacity}, {flags}) failed: {n}")
    return Layout(arrays, n, int(total.value))


def iter_workspace_layout(P, width, height, capacity, group_stride, flags):
    arrays = (SplatArrayInfo * SPLAT_LAYOUT_MAX_ARRAYS)()
    total = C.c_size_t(0)
    n = lib().splat_iter_workspace_layout(P, width, height, capacity, group_stride, flags, arrays, SPLAT_LAYOUT_MAX_ARRAYS, C.byref(total))
    if n < 0 or n > SPLAT_LAYOUT_MAX_ARRAYS:
        raise RuntimeError(f"splat_iter_workspace_layout({P}, {width}, {height}, {capacity}, {group_stride}, {flags}) failed: {n}")
    return Layout(arrays, n, int(total.value))


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().splat_error_string(rc).decode()} (code {rc})")
