"""Host-side mirror of the reference's rasterizer package.

Same surface as ``diff_gaussian_rasterization`` (the un-vendored dependency
imported at /root/reference/scripts/splatam.py:37 and
/root/reference/utils/recon_helpers.py:2):

* ``GaussianRasterizationSettings`` -- the 11-field NamedTuple built at
  /root/reference/utils/recon_helpers.py:14-26;
* ``GaussianRasterizer(raster_settings=...)(means3D=, means2D=, opacities=,
  colors_precomp=|shs=, scales=+rotations=|cov3D_precomp=)`` ->
  ``(color[C,H,W], radii[P] int32, depth[1,H,W])``
  (/root/reference/scripts/splatam.py:249,253,384);
* gradients for means3D, means2D, shs/colors_precomp, opacities, scales,
  rotations, cov3D_precomp through ``torch.autograd``.

Everything is computed by libsplat_hip.so (include/splat_hip.h) on the current
torch HIP stream.  PyTorch only owns the device memory.  There is no fallback:
CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _capi


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


# ---------------------------------------------------------------------------
# Instance-capacity policy.
#   "exact": after the per-Gaussian pass the host reads num_rendered (one 16-byte
#            D2H copy + stream sync, exactly what the reference extension does to
#            size its sort buffers) and allocates the lists to fit.
#   "lazy" : no host sync.  Lists are sized from the high-water mark of earlier
#            calls (x1.5); status words are checked when they are next needed
#            and an overflow re-runs the binning with a larger capacity.
#   "auto" (default): the FIRST call on a scene (device, Gaussian count, image size, field of view) runs the exact path and
#            learns the scene's longest per-tile list; while that list is short (x1.6 <= 1 024 entries: every SplaTAM configuration)
#            the following calls run the front end of the fused iteration -- group binning in the per-Gaussian kernel, the lists
#            sorted inside the forward composite (include/splat_hip.h "GROUP BINNING behind the reference API"): two launches
#            instead of six and NO host synchronisation.  The device raises a flag in pinned host memory when a list has outgrown
#            its bucket after all; the flag is looked at when the backward pass starts (or at once for a render that will have no
#            backward pass): the pass is then repeated on exact lists, the scene goes back to the exact path and a warning is
#            issued (the image of THAT one forward call had been composited from truncated lists).  Every 64th call of a scene is
#            an exact one and refreshes the statistics.  Scenes with long lists stay on the exact path.
# ---------------------------------------------------------------------------
_SYNC_MODE = "auto"
_capacity_hint: dict = {}

FAST_STRIDE = 1024          # bucket of a tile's published list: all the forward composite sorts itself
FAST_MARGIN = 1.6           # fast path while longest list x margin <= FAST_STRIDE
FAST_REFRESH = 64           # every n-th call of a scene runs the exact path and refreshes its statistics
_scene_stats: dict = {}     # scene key -> {'longest': longest per-tile list of the last exact call, 'since_exact': calls since}
fast_path_stats = {"fast": 0, "exact": 0, "flagged": 0}


def set_sync_mode(mode: str) -> None:
    global _SYNC_MODE
    if mode not in ("exact", "lazy", "auto"):
        raise ValueError(mode)
    _SYNC_MODE = mode


def get_sync_mode() -> str:
    return _SYNC_MODE


_UPSTREAM_SCALE_GRADIENT = False


def set_upstream_scale_gradient(on: bool) -> None:
    """``scales.grad`` as the CUDA original returns it: WITHOUT the factor ``raster_settings.scale_modifier`` (its computeCov3D adjoint
    hands out dL/d(modifier * s) as dL/ds; include/splat_hip.h: SPLAT_GRADS_UPSTREAM_SCALE).  Off (default): the gradient w.r.t. the
    scales the caller passed.  Identical at scale_modifier = 1, i.e. for every call SplaTAM makes."""
    global _UPSTREAM_SCALE_GRADIENT
    _UPSTREAM_SCALE_GRADIENT = bool(on)


def reset_scene_stats() -> None:
    """Forgets what the "auto" mode has learnt about the scenes it has rendered (the next call of every scene is an exact one)."""
    _scene_stats.clear()


class _FlagRing:
    """Pinned host words the device raises when a fast-path call met a list beyond its bucket (SplatState.status_host): one word per
    call in flight, re-used round robin -- a word is handed out again only after the call that held it has finished."""
    SLOTS = 1024

    def __init__(self):
        self.words = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        self.host = self.words.numpy()          # (the same memory: element access without a tensor operation)
        self.base = self.words.data_ptr()
        self.events = [None] * self.SLOTS
        self.next = 0

    def take(self, dev):
        i = self.next
        self.next = (i + 1) % self.SLOTS
        ev = self.events[i]
        if ev is not None:
            ev.synchronize()            # (the call that used this word 1 024 calls ago)
        else:
            ev = self.events[i] = torch.cuda.Event()
        self.host[i] = 0
        return i, self.base + 4 * i, ev


_flag_rings: dict = {}          # one ring per device (an event belongs to the device it was first recorded on)


def _ring(dev_index=None) -> _FlagRing:
    if dev_index is None:
        dev_index = torch.cuda.current_device()
    r = _flag_rings.get(dev_index)
    if r is None:
        r = _flag_rings[dev_index] = _FlagRing()
    return r


_zero_bg: dict = {}


def _bg_is_zero(bg: torch.Tensor) -> bool:
    """Is the background black (every SplaTAM camera: /root/reference/utils/recon_helpers.py:17)?  Read from the device ONCE per
    tensor and version; the library then drops the background term of the backward composite at compile time (SplatCamera.bg NULL)."""
    key = (bg.data_ptr(), bg._version, bg.numel())
    hit = _zero_bg.get(key)
    if hit is not None and hit[0] is bg:
        return hit[1]
    z = not bool(bg.detach().ne(0).any().item())
    if len(_zero_bg) > 64:
        _zero_bg.clear()
    _zero_bg[key] = (bg, z)
    return z


_contig_cache: dict = {}


def _cached_contiguous(t: torch.Tensor) -> torch.Tensor:
    """Settings tensors (viewmatrix is a transposed, non-contiguous view:
    /root/reference/utils/recon_helpers.py:8) are re-used across thousands of
    calls; make them contiguous float32 once."""
    if t.is_contiguous() and t.dtype == torch.float32:
        return t
    key = (t.data_ptr(), t._version, tuple(t.stride()), t.dtype)
    hit = _contig_cache.get(key)
    if hit is not None and hit[0] is t:
        return hit[1]
    c = t.to(torch.float32).contiguous()
    if len(_contig_cache) > 64:
        _contig_cache.clear()
    _contig_cache[key] = (t, c)
    return c


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _check_input(name: str, t: torch.Tensor, device) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the HIP rasterizer has no CPU path")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


class _Pack:
    """The ctypes structs of one call plus the tensors that keep their memory alive."""

    def __init__(self):
        self.keep = []
        self.cam = None                 # (the shared struct of the settings tuple: _camera_struct)
        self.g = _capi.SplatGaussians()
        self.st = _capi.SplatState()


_cam_structs: dict = {}


def _camera_struct(settings, dev):
    """The SplatCamera of a settings tuple and the tensors that keep its pointers alive.  The callers pass ONE settings object thousands
    of times (curr_data['cam'], /root/reference/scripts/splatam.py:249,253): the struct is built once per object and re-used while the
    tuple's tensors are at the version it was built from."""
    hit = _cam_structs.get(id(settings))
    if hit is not None and hit[0] is settings and hit[3] == (settings.bg._version, settings.viewmatrix._version, settings.projmatrix._version,
                                                             settings.campos._version) and hit[4] == dev:
        return hit[1], hit[2]
    bg = _cached_contiguous(settings.bg)
    view = _cached_contiguous(settings.viewmatrix)
    proj = _cached_contiguous(settings.projmatrix)
    campos = _cached_contiguous(settings.campos)
    for name, t in (("bg", bg), ("viewmatrix", view), ("projmatrix", proj), ("campos", campos)):
        if not t.is_cuda or t.device != dev:
            raise RuntimeError(f"raster_settings.{name} must be on {dev} (got {t.device})")
    cam = _capi.SplatCamera()
    cam.image_height, cam.image_width = int(settings.image_height), int(settings.image_width)
    cam.tanfovx, cam.tanfovy = float(settings.tanfovx), float(settings.tanfovy)
    cam.bg, cam.scale_modifier = (None if _bg_is_zero(bg) else bg.data_ptr()), float(settings.scale_modifier)
    cam.viewmatrix, cam.projmatrix = view.data_ptr(), proj.data_ptr()
    cam.sh_degree, cam.campos, cam.prefiltered = int(settings.sh_degree), campos.data_ptr(), int(bool(settings.prefiltered))
    keep = (bg, view, proj, campos)
    if len(_cam_structs) > 64:
        _cam_structs.clear()
    _cam_structs[id(settings)] = (settings, cam, keep, (settings.bg._version, settings.viewmatrix._version, settings.projmatrix._version,
                                                        settings.campos._version), dev)
    return cam, keep


def _build_pack(settings, means3D, colors, opacities, scales, rotations, cov3D, shs) -> _Pack:
    dev = means3D.device
    pk = _Pack()
    P = means3D.shape[0]
    pk.cam, (bg, view, proj, campos) = _camera_struct(settings, dev)
    use_sh = shs.numel() > 0
    channels = 3 if use_sh else colors.shape[1]
    if not (1 <= channels <= _capi.SPLAT_MAX_CHANNELS):
        raise RuntimeError(f"colors_precomp must have 1..{_capi.SPLAT_MAX_CHANNELS} channels, got {channels}")
    if bg.numel() < channels:
        raise RuntimeError(f"raster_settings.bg has {bg.numel()} entries for {channels} channels")
    g = pk.g
    g.P, g.channels = P, channels
    g.means3D, g.opacities = _ptr(means3D), _ptr(opacities)
    g.colors_precomp = None if use_sh else _ptr(colors)
    g.scales, g.rotations, g.cov3D_precomp = _ptr(scales), _ptr(rotations), _ptr(cov3D)
    g.shs = _ptr(shs) if use_sh else None
    g.sh_coeffs = shs.shape[1] if use_sh else 0
    pk.keep += [bg, view, proj, campos, means3D, colors, opacities, scales, rotations, cov3D, shs]      # (rasterize_backward reads [4:11])
    return pk


SUB_BINS_LONG = 16       # counters per tile once per-tile lists get very long (SplatState.sub_bins)
_longest_seen: dict = {}   # (device, P, H, W) -> longest per-tile list of the last call with that shape


def _alloc_state(pk: _Pack, dev, P: int, H: int, W: int, use_sh: bool):
    """Per-call scratch (the reference's geomBuffer / imgBuffer): ONE slab laid out by the library (splat_state_layout /
    splat_state_bind, include/splat_hip.h "Scratch layouts": sizes, offsets and alignment are the C side's, not this file's);
    torch's caching allocator makes it a host-side pointer bump.  The lists (keys / point_list ...) follow in _alloc_lists once the
    number of instances is known; radii / final_T / n_contrib are tensors of their own (an output; replaced when a second call
    shares the first call's geometry)."""
    f32, i32 = torch.float32, torch.int32
    # very long per-tile lists (clustered scenes): spread each tile's count / scatter atomics over several counters
    S = SUB_BINS_LONG if _longest_seen.get((dev.index, P, H, W), 0) > 2048 else 1
    lay = _capi.state_layout(P, W, H, S, 0, _capi.SPLAT_LAYOUT_SH if use_sh else 0)
    slab = torch.empty(lay.total, dtype=torch.uint8, device=dev)
    _capi.check(_capi.lib().splat_state_bind(C.byref(pk.st), None, slab.data_ptr(), lay.arrays, lay.n, S, 0), "splat_state_bind")

    def view(name, dtype):
        o = lay.offset[name]
        return slab[o:o + lay.bytes[name]].view(dtype)
    status, tile_base = view("status", i32), view("tile_base", i32)
    radii = torch.empty(P, dtype=i32, device=dev)
    final_T = torch.empty(H, W, dtype=f32, device=dev)
    n_contrib = torch.empty(H, W, dtype=i32, device=dev)
    st = pk.st
    st.radii, st.final_T, st.n_contrib = radii.data_ptr(), final_T.data_ptr(), n_contrib.data_ptr()
    rgb = view("rgb", f32).view(P, 3) if use_sh else None
    clamped = view("clamped", torch.uint8).view(P, 3) if use_sh else None
    pk.tensors = dict(geom=slab, radii=radii, final_T=final_T, n_contrib=n_contrib, status=status,
                      tile_base=tile_base, rgb=rgb, clamped=clamped)
    pk.num_tiles = int(_capi.lib().splat_num_tiles(W, H))
    pk.shape = (P, H, W, S)
    return radii, status


LONG_LIST = 4096        # per-tile lists beyond this are sorted by the multi-workgroup kernels, which need a second key buffer
LONG_ITEM_TABLE = True   # SplatState.long_items (work-item table of those kernels); False: they binary-search long_base (tests)


# the forward composite leaves its staged records for the backward one (SplatState.tile_recs): pays where a tile's list is several
# 255-entry batches long, costs where it is one (profiles/r06_experiments.md 2).  None: by the scene's longest list; True / False: forced
USE_TILE_RECS = None


def _want_recs(will_backward: bool, longest) -> bool:
    if not will_backward or USE_TILE_RECS is False:
        return False
    return bool(USE_TILE_RECS) or (longest is not None and longest > 400)


def _alloc_lists(pk: _Pack, dev, capacity: int, longest=None, recs=False):
    """``longest``: the longest per-tile list if the host knows it (exact mode); None = unknown (lazy mode).  Sizes from the library's
    layout for this capacity (splat_state_layout)."""
    capacity = max(int(capacity), 1)
    P, H, W, S = pk.shape
    long_lists = longest is None or longest > LONG_LIST
    lay = _capi.state_layout(P, W, H, S, capacity, _capi.SPLAT_LAYOUT_LONG_LISTS if long_lists else 0)
    keys = torch.empty(lay.bytes["keys"] // 8, dtype=torch.int64, device=dev)
    plist = torch.empty(lay.bytes["point_list"] // 4, dtype=torch.int32, device=dev)
    pk.st.keys, pk.st.point_list, pk.st.capacity = keys.data_ptr(), plist.data_ptr(), capacity
    pk.tensors.update(keys=keys, point_list=plist)
    if recs and pk.g.channels <= 3:
        tr = torch.empty(12 * capacity, dtype=torch.float32, device=dev)
        pk.st.tile_recs = tr.data_ptr()
        pk.tensors.update(tile_recs=tr)
    if long_lists:
        alt = torch.empty(lay.bytes["keys_alt"] // 8, dtype=torch.int64, device=dev)
        pk.st.keys_alt = alt.data_ptr()
        pk.tensors.update(keys_alt=alt)
        if LONG_ITEM_TABLE:
            items = torch.empty(lay.bytes["long_items"] // 4, dtype=torch.int32, device=dev)      # SplatState.long_items
            pk.st.long_items = items.data_ptr()
            pk.tensors.update(long_items=items)


def _alloc_state_fast(pk: _Pack, dev, P: int, H: int, W: int, longest: int, recs: bool = True, backward: bool = True):
    """Scratch of a fast-path call ("auto" sync mode): group binning + published lists, ONE slab laid out by the library
    (splat_state_layout with SPLAT_LAYOUT_GROUPS: no key buckets, no sort scratch; the group counters sit in front of the status words
    so that the library zeroes both with one memset)."""
    i32 = torch.int32
    T = ((W + 15) // 16) * ((H + 15) // 16)
    cap = T * FAST_STRIDE
    lay = _capi.state_layout(P, W, H, 1, cap, _capi.SPLAT_LAYOUT_GROUPS | (_capi.SPLAT_LAYOUT_RECS if recs else 0) |
                             (_capi.SPLAT_LAYOUT_BACKWARD if backward else 0))
    slab = torch.empty(lay.total, dtype=torch.uint8, device=dev)
    base = slab.data_ptr()
    _capi.check(_capi.lib().splat_state_bind(C.byref(pk.st), None, base, lay.arrays, lay.n, 1, cap), "splat_state_bind")
    if backward:
        # the backward pass' accumulator lives in the slab and is zeroed by the per-Gaussian kernel of THIS forward pass
        # (SplatState.accum_to_zero): no memset launch per backward pass
        pk.accum_ptr = pk.st.accum_to_zero = base + lay.offset["accum"]
    # (final_T / n_contrib stay where the library's layout put them, inside the slab; radii is an OUTPUT of the call and a tensor of its own,
    #  so that a caller who keeps it does not keep the slab)
    radii = torch.empty(P, dtype=i32, device=dev)
    st = pk.st
    st.radii = radii.data_ptr()
    st.tile_stride = FAST_STRIDE
    st.group_stride = _capi.SPLAT_GROUP_TILES ** 2 * FAST_STRIDE
    st.max_list_hint = max(1, min(int(longest * FAST_MARGIN), FAST_STRIDE * 4 // 5))
    pk.tensors = dict(geom=slab, radii=radii)
    pk.num_tiles = T
    pk.shape = (P, H, W, 1)
    return radii, None


def _stream(dev) -> int:
    # (the raw handle of torch's current stream; torch.cuda.current_stream(dev).cuda_stream builds a Stream object on the way: 13 us)
    return torch._C._cuda_getCurrentRawStream(dev.index)


class _on_device:
    """``with torch.cuda.device(dev)`` only when ``dev`` is not the current device already (the guard costs ~6 us per use, twice per
    rasterizer call; the callers render on the current device)."""
    __slots__ = ("guard",)

    def __init__(self, dev):
        self.guard = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            return self.guard.__exit__(*exc)
        return False


class LazyOverflow(RuntimeError):
    """Lazy sync mode: the (Gaussian, tile) instances did not fit the lists sized from the high-water mark."""


# ---------------------------------------------------------------------------
# Shared geometry between consecutive calls.
# The reference's get_loss renders the SAME Gaussians twice per iteration -- colours, then depth / silhouette / depth^2
# (/root/reference/scripts/splatam.py:249,253) -- and so do add_new_gaussians + the first mapping iteration, the evaluation code
# (/root/reference/utils/eval_helpers.py:219,225) ...  K1-K5 (projection, tile lists, sort) depend on the camera, means3D,
# opacities, scales and rotations only, not on the colours: the second call re-uses the first call's SplatState and runs K6 alone.
# Proof of equality, all of it on what the caller handed over (any doubt falls back to the full pass):
#   * camera: same image size / tanfov / scale_modifier, viewmatrix and projmatrix in the same storage at the same version;
#   * means3D: the tensor THE CALLER PASSED (before any .contiguous(): the reference's transformed_pts is a transposed, non-contiguous
#     view, /root/reference/utils/slam_helpers.py:294-295, and every call copies it) -- same storage, offset, strides and version,
#     while the cache keeps the first call's tensor alive (the two render-variable dicts share one means3D tensor: :131,241);
#   * opacities / scales / rotations: the same tensors, or -- the reference applies sigmoid / exp / normalize twice, so they are
#     distinct tensors with equal values -- compared bit for bit on the device (splat_same_geometry: one kernel + the one status
#     read that an "exact"-mode forward makes anyway).
# Only in "exact" sync mode, with scales + rotations (no cov3D_precomp), without SHs.
# ---------------------------------------------------------------------------
_GEOM_CACHE = True
_geom_last: dict = {}
geometry_cache_stats = {"shared": 0, "verified_on_device": 0, "mismatch": 0, "other_means3D": 0, "other_camera": 0, "other_shapes": 0,
                        "inputs_modified": 0}


def set_geometry_cache(on: bool) -> None:
    global _GEOM_CACHE
    _GEOM_CACHE = bool(on)
    _geom_last.clear()


def clear_geometry_cache() -> None:
    """Drops the cached state (it keeps one call's geometry and lists alive per device)."""
    _geom_last.clear()


def _cam_key(settings, view, proj):
    bg = settings.bg
    return (int(settings.image_height), int(settings.image_width), float(settings.tanfovx), float(settings.tanfovy),
            float(settings.scale_modifier), bool(settings.prefiltered), view.data_ptr(), view._version, proj.data_ptr(), proj._version,
            bg.data_ptr(), bg._version, _stream(view.device))        # (the stream: work queued on another stream is not ordered with the cached call)


class _GeomEntry:
    __slots__ = ("pk", "key", "tensors", "versions", "snapshot")


def _remember_geometry(pk, settings, means3D_src, opac, scales, rots):
    e = _GeomEntry()
    e.pk = pk
    view, proj = _cached_contiguous(settings.viewmatrix), _cached_contiguous(settings.projmatrix)
    e.key = _cam_key(settings, view, proj)
    e.tensors = (means3D_src, opac, scales, rots, view, proj)      # alive: their storage cannot be handed to another tensor
    e.versions = (means3D_src._version, opac._version, scales._version, rots._version)
    # SNAPSHOTS of what the cached geometry was projected from: version counters do not see raw-pointer writes (this library's own
    # Adam / map-edit kernels, `.data` arithmetic), and the second call may pass the very same tensor objects, so it compares the VALUES
    # on the device against these copies (44 bytes per Gaussian: 13 MB at 300 k)
    e.snapshot = tuple(t.detach().contiguous().clone() for t in (means3D_src, opac, scales, rots))
    _geom_last[means3D_src.device.index] = e


def _shared_geometry(settings, means3D, colors, opac, scales, rots, cov3D, shs, means3D_src):
    """The pack of a call that re-uses the cached call's geometry and lists, or None.  ``means3D_src``: the tensor the caller passed."""
    dev = means3D.device
    e = _geom_last.get(dev.index)
    if e is None:
        return None
    m1, o1, s1, r1, _, _ = e.tensors
    P = means3D.shape[0]
    st = geometry_cache_stats
    if P == 0 or m1.shape != means3D_src.shape or m1.stride() != means3D_src.stride() or m1.data_ptr() != means3D_src.data_ptr() \
            or means3D_src._version != e.versions[0] or m1._version != e.versions[0]:
        st["other_means3D"] += 1
        return None
    if e.key != _cam_key(settings, _cached_contiguous(settings.viewmatrix), _cached_contiguous(settings.projmatrix)):
        st["other_camera"] += 1
        return None
    if o1.shape != opac.shape or s1.shape != scales.shape or r1.shape != rots.shape:
        st["other_shapes"] += 1
        return None
    if (o1._version, s1._version, r1._version) != tuple(e.versions[1:]):
        st["inputs_modified"] += 1
        return None                                                 # the first call's inputs were modified in place since
    # ALWAYS compared bit for bit on the device, centres included (identity of the tensors proves nothing about their contents: see
    # _remember_geometry): two launches of the comparison kernel (the centres travel in its [P][3] slot), ONE host read of both flags
    # -- the read an exact-mode forward makes anyway
    flags = torch.empty(2, dtype=torch.int32, device=dev)
    now = means3D_src.detach().contiguous()
    L = _capi.lib()
    with torch.cuda.device(dev):
        m0, o0, s0, r0 = e.snapshot
        _capi.check(L.splat_same_geometry(P, o0.data_ptr(), opac.data_ptr(), s0.data_ptr(), scales.data_ptr(),
                                          r0.data_ptr(), rots.data_ptr(), flags.data_ptr(), _stream(dev)), "splat_same_geometry")
        _capi.check(L.splat_same_geometry(P, o0.data_ptr(), o0.data_ptr(), m0.data_ptr(), now.data_ptr(),
                                          r0.data_ptr(), r0.data_ptr(), flags.data_ptr() + 4, _stream(dev)), "splat_same_geometry")
    geometry_cache_stats["verified_on_device"] += 1
    if any(flags.tolist()):
        geometry_cache_stats["mismatch"] += 1
        return None
    pk1 = e.pk
    pk = _build_pack(settings, means3D, colors, opac, scales, rots, cov3D, shs)
    C.memmove(C.byref(pk.st), C.byref(pk1.st), C.sizeof(_capi.SplatState))
    H, W = int(settings.image_height), int(settings.image_width)
    final_T = torch.empty(H, W, dtype=torch.float32, device=dev)
    n_contrib = torch.empty(H, W, dtype=torch.int32, device=dev)
    pk.st.final_T, pk.st.n_contrib = final_T.data_ptr(), n_contrib.data_ptr()
    pk.tensors = dict(pk1.tensors, final_T=final_T, n_contrib=n_contrib)       # shares the first call's geometry + lists (ref-counted)
    pk.num_tiles, pk.num_rendered, pk.shape = pk1.num_tiles, pk1.num_rendered, pk1.shape
    pk.shared_geometry = True
    geometry_cache_stats["shared"] += 1
    # one re-use per cached call (get_loss renders twice): the entry goes, so that the first call's geometry, keys and lists are freed
    # with the two autograd graphs instead of living until the next render on this device
    _geom_last.pop(dev.index, None)
    return pk


def rasterize_forward(settings, means3D, colors, opacities, scales, rotations, cov3D, shs, will_backward=True, means3D_src=None):
    """One forward through the C ABI.  Returns (color, radii, depth, pack).

    ``will_backward=False`` (no input requires grad: torch.no_grad renders such as add_new_gaussians, evaluation,
    keyframe selection): in lazy sync mode nothing downstream would ever look at the device-side status words, and an
    overflow publishes EMPTY lists (a background-only image).  The status is therefore resolved here (one D2H read, as
    the exact mode does) and the render is repeated with the raised capacity."""
    for _ in range(4):
        out = _rasterize_forward_once(settings, means3D, colors, opacities, scales, rotations, cov3D, shs,
                                      means3D if means3D_src is None else means3D_src, will_backward=will_backward)
        if not will_backward and getattr(out[3], "fast", None) is not None:
            # a fast-path render nobody will differentiate: its flag is resolved here (the wait the exact path makes anyway), and a
            # flagged render is repeated on exact lists -- what is returned was never composited from truncated lists
            if fast_call_flagged(out[3]):
                continue                # (the scene's statistics are gone: the next pass is an exact one)
            return out
        if will_backward or getattr(out[3], "pending_status", None) is None:
            return out
        try:
            resolve_lazy(out[3])
            return out
        except LazyOverflow:
            continue                    # resolve_lazy has raised the capacity hint: render again
    raise RuntimeError("lazy sync mode: the instance lists could not be sized")


def _scene_key(dev, P, settings):
    return (dev.index, P, int(settings.image_height), int(settings.image_width), float(settings.tanfovx), float(settings.tanfovy))


def _fast_eligible(key, channels, use_sh) -> bool:
    if _SYNC_MODE != "auto" or use_sh or channels != 3 or key[1] == 0:
        return False
    if (((key[3] + 15) // 16 + 1) // 2) * (((key[2] + 15) // 16 + 1) // 2) > 8192:      # (one LDS counter per 2 x 2-tile group in K1: frames beyond ~2900 x 2900)
        return False
    stt = _scene_stats.get(key)
    return stt is not None and 0 < stt['longest'] * FAST_MARGIN <= FAST_STRIDE and stt['since_exact'] < FAST_REFRESH


def _rasterize_forward_fast(settings, means3D, colors, opacities, scales, rotations, cov3D, shs, key, will_backward=True):
    """The front end of the fused iteration behind the reference API: K1 with group binning, the forward composite that sorts its
    own lists.  Nothing is read back; the device raises pk.flag_word (pinned host memory) if a list did not fit after all."""
    L = _capi.lib()
    dev = means3D.device
    H, W = int(settings.image_height), int(settings.image_width)
    P = means3D.shape[0]
    pk = _build_pack(settings, means3D, colors, opacities, scales, rotations, cov3D, shs)
    radii, _ = _alloc_state_fast(pk, dev, P, H, W, _scene_stats[key]['longest'], recs=_want_recs(will_backward, _scene_stats[key]['longest']),
                                 backward=will_backward)
    out_color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    out_depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    ring = _ring(dev.index)
    idx, word, ev = ring.take(dev)
    pk.st.status_host = word
    stream = _stream(dev)
    with _on_device(dev):
        # (splat_forward = preprocess + bin (a no-op with group binning) + render: one crossing of the ABI)
        _capi.check(L.splat_forward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), out_color.data_ptr(), out_depth.data_ptr(), stream), "splat_forward")
        ev.record()
    pk.fast = (idx, ev, key)        # (key[0] is the device index: the flag word lives in that device's ring)
    pk.settings = settings
    pk.num_rendered = None
    _scene_stats[key]['since_exact'] += 1
    fast_path_stats["fast"] += 1
    return out_color, radii, out_depth, pk


_unchecked: list = []        # fast-path calls whose backward pass was launched before their flag could be read (oldest first)


def _poll_unchecked(wait: bool = False) -> None:
    """Reads the flags of the fast-path calls that were never waited for, as far as their forward passes have finished (all of them
    with ``wait``).  A flagged one had its gradients poisoned with NaN on the device: that is reported here, loudly."""
    while _unchecked:
        idx, ev, key = _unchecked[0]
        if not ev.query():
            if not wait:
                return
            ev.synchronize()
        _unchecked.pop(0)
        if int(_ring(key[0]).host[idx]) != 0:
            fast_path_stats["flagged"] += 1
            _scene_stats.pop(key, None)
            raise RuntimeError("splatam_amd rasterizer (sync mode 'auto'): a per-tile Gaussian list outgrew the bucket learnt for its scene in an "
                               "EARLIER call whose backward pass had been issued before its forward pass finished; that call's image was composited "
                               "from truncated lists and its gradients were set to NaN on the device.  The scene is back on the exact path; "
                               "repeat the iteration (set_sync_mode('exact') rules this out)")


def check_pending() -> None:
    """Waits for every fast-path call whose flag has not been read yet and raises if one of them was flagged (see _poll_unchecked)."""
    _poll_unchecked(wait=True)


def fast_call_flagged(pk) -> bool:
    """Did the device flag this fast-path call (a list beyond its bucket)?  Waits for the call's forward composite if it is still
    running (in a training loop the loss sits between the forward and the backward pass: it has long finished)."""
    idx, ev, key = pk.fast
    if not ev.query():
        ev.synchronize()
    flagged = int(_ring(key[0]).host[idx]) != 0
    if flagged:
        fast_path_stats["flagged"] += 1
        _scene_stats.pop(key, None)         # the scene goes back to the exact path (and learns its lists again)
    return flagged


def _rasterize_forward_once(settings, means3D, colors, opacities, scales, rotations, cov3D, shs, means3D_src, force_exact=False,
                            will_backward=True):
    L = _capi.lib()
    dev = means3D.device
    H, W = int(settings.image_height), int(settings.image_width)
    P = means3D.shape[0]
    use_sh = shs.numel() > 0
    scene = _scene_key(dev, P, settings)
    if _unchecked:
        _poll_unchecked()
    if not force_exact and _fast_eligible(scene, 3 if use_sh else colors.shape[1], use_sh) and cov3D.numel() == 0:
        return _rasterize_forward_fast(settings, means3D, colors, opacities, scales, rotations, cov3D, shs, scene, will_backward)
    cacheable = _GEOM_CACHE and _SYNC_MODE == "exact" and not use_sh and cov3D.numel() == 0 and scales.numel() > 0 and rotations.numel() > 0
    if cacheable:
        pk = _shared_geometry(settings, means3D, colors, opacities, scales, rotations, cov3D, shs, means3D_src)
        if pk is not None:                  # K6 alone, on the cached call's geometry and sorted lists
            # (the staged records hold THIS call's colours: a buffer of its own, or none -- never the cached call's)
            pk.st.tile_recs = None
            if _want_recs(will_backward, _longest_seen.get((dev.index, P, H, W))) and pk.g.channels <= 3 and pk.st.capacity > 0:
                tr = torch.empty(12 * int(pk.st.capacity), dtype=torch.float32, device=dev)
                pk.st.tile_recs = tr.data_ptr()
                pk.tensors = dict(pk.tensors, tile_recs=tr)
            out_color = torch.empty(pk.g.channels, H, W, dtype=torch.float32, device=dev)
            out_depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _capi.check(L.splat_render_forward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), out_color.data_ptr(),
                                                   out_depth.data_ptr(), _stream(dev)), "splat_render_forward")
            return out_color, pk.tensors['radii'].clone(), out_depth, pk       # (a tensor of its own: the two calls' results do not alias)
    pk = _build_pack(settings, means3D, colors, opacities, scales, rotations, cov3D, shs)
    radii, status = _alloc_state(pk, dev, P, H, W, use_sh)
    Cn = pk.g.channels
    out_color = torch.empty(Cn, H, W, dtype=torch.float32, device=dev)
    out_depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    stream = _stream(dev)
    with torch.cuda.device(dev):
        hint_key = (dev.index, P, H, W)
        cap = _capacity_hint.get(hint_key) if _SYNC_MODE == "lazy" else None
        if cap is None:
            # size the lists from this call's own count: one 16-byte D2H read + stream sync, exactly what the
            # reference extension does.  (The tile scan publishes EMPTY lists when the count exceeds
            # st.capacity, so the capacity is "unbounded" until the buffers exist.)
            pk.st.capacity = 1 << 62
            _capi.check(L.splat_preprocess_forward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), stream), "splat_preprocess_forward")
            stat = status.tolist()
            num_rendered = int(stat[0])
            _longest_seen[hint_key] = int(stat[2])
            _scene_stats[scene] = {'longest': int(stat[2]), 'since_exact': 0}
            fast_path_stats["exact"] += 1
            if _SYNC_MODE != "lazy":
                _alloc_lists(pk, dev, num_rendered, longest=int(stat[2]), recs=_want_recs(will_backward, int(stat[2])))
                pk.st.max_list_hint = int(stat[2])        # lets the library skip the long-list sort kernel
            else:                                         # lazy, first call for this shape: learn the size
                _capacity_hint[hint_key] = int(num_rendered * 1.5) + 1024
                _alloc_lists(pk, dev, _capacity_hint[hint_key], recs=_want_recs(will_backward, int(stat[2])))
            pk.num_rendered = num_rendered
        else:
            _alloc_lists(pk, dev, cap, recs=_want_recs(will_backward, _longest_seen.get(hint_key)))
            _capi.check(L.splat_preprocess_forward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), stream), "splat_preprocess_forward")
            pk.num_rendered = None
        _capi.check(L.splat_bin_forward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), stream), "splat_bin_forward")
        _capi.check(L.splat_render_forward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), out_color.data_ptr(),
                                           out_depth.data_ptr(), stream), "splat_render_forward")
        if _SYNC_MODE == "lazy" and pk.num_rendered is None:
            pk.pending_status = _async_status(status, dev)
            pk.hint_key = hint_key
    if cacheable:
        _remember_geometry(pk, settings, means3D_src, opacities, scales, rotations)
    pk.settings = settings
    return out_color, radii, out_depth, pk


def _async_status(status, dev):
    host = torch.empty(4, dtype=torch.int32, pin_memory=True)
    host.copy_(status, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return host, ev


def resolve_lazy(pk) -> None:
    """Lazy mode: confirm that the instance lists fitted.  Called before the
    backward pass touches them; raises if the forward that was handed out was
    rendered from truncated lists (the caller must re-run it)."""
    pend = getattr(pk, "pending_status", None)
    if pend is None:
        return
    host, ev = pend
    ev.synchronize()
    pk.pending_status = None
    n = int(host[0])
    pk.num_rendered = n
    _capacity_hint[pk.hint_key] = max(_capacity_hint.get(pk.hint_key, 0), int(n * 1.5) + 1024)
    if int(host[1]) != 0 or n > pk.st.capacity:
        raise LazyOverflow(
            f"lazy sync mode: {n} (Gaussian, tile) instances did not fit the {pk.st.capacity}-entry lists; "
            "the capacity hint has been raised -- re-run the forward (or use set_sync_mode('exact'))")


def rasterize_backward(pk: _Pack, grad_color, need_scale_rot: bool, need_cov3D: bool, use_sh: bool, sh_shape):
    L = _capi.lib()
    P, Cn = pk.g.P, pk.g.channels
    dev = grad_color.device
    f32 = torch.float32
    resolve_lazy(pk)
    poison = False
    if getattr(pk, "fast", None) is not None and not pk.fast[1].query():
        # the forward composite is still running (a backward pass issued right behind its forward pass; in a training loop the loss sits
        # between them): nobody waits.  The backward pass is launched as it is and told to POISON its gradients with NaN should the
        # forward pass turn out to have been flagged; the flag is read when a later call finds the event complete (_poll_unchecked)
        poison = True
        _unchecked.append(pk.fast)
    elif getattr(pk, "fast", None) is not None and fast_call_flagged(pk):
        # the forward pass ran on truncated lists: the gradients are formed on exact ones (the same inputs, kept alive by the pack)
        import warnings
        warnings.warn("splatam_amd rasterizer (sync mode 'auto'): a per-tile Gaussian list outgrew the bucket learnt for this scene; the "
                      "backward pass was repeated on exact lists and the scene returns to the exact path, but the image of this ONE "
                      "forward call had been composited from truncated lists (set_sync_mode('exact') rules that out)", RuntimeWarning)
        means3D, colors, opac, scales, rots, cov3D, shs = pk.keep[4:11]
        _, _, _, pk = _rasterize_forward_once(pk.settings, means3D, colors, opac, scales, rots, cov3D, shs, means3D, force_exact=True)
    grad_color = grad_color.contiguous()
    accum_bytes = _capi.state_layout(P, pk.shape[2], pk.shape[1], 1, 0, _capi.SPLAT_LAYOUT_BACKWARD).bytes["accum"]     # (the library's size)
    # the accumulator and every gradient in ONE allocation, cut into views by one split (eight allocator calls were ~30 us of host time
    # per backward pass; rows of 4-float multiples first, so that every piece is 16-byte aligned)
    sh_n = 1
    for d in sh_shape:
        sh_n *= int(d)
    # (the accumulator first: its rows are read as float4; then the quaternion rows; the rest needs no alignment)
    accum_ptr = getattr(pk, "accum_ptr", None)          # (a fast-path forward pass left a zeroed accumulator in its slab)
    if accum_ptr is not None:
        pk.accum_ptr = None                             # (one backward pass per zeroing)
    sizes = [0 if accum_ptr is not None else accum_bytes // 4, 4 * P if need_scale_rot else 0, 3 * P, 3 * P, P, 0 if use_sh else Cn * P, 3 * P if need_scale_rot else 0,
             6 * P if need_cov3D else 0, sh_n if use_sh else 0]
    parts = torch.empty(sum(sizes), dtype=f32, device=dev).split_with_sizes(sizes)
    accum = parts[0]
    d_rots = parts[1].view(P, 4) if need_scale_rot else None
    d_means3D = parts[2].view(P, 3)
    d_means2D = parts[3].view(P, 3)
    d_opac = parts[4].view(P, 1)
    d_colors = None if use_sh else parts[5].view(P, Cn)
    d_scales = parts[6].view(P, 3) if need_scale_rot else None
    d_cov = parts[7].view(P, 6) if need_cov3D else None
    d_sh = parts[8].view(sh_shape) if use_sh else None
    gr = _capi.SplatGrads()
    gr.dL_dcolor, gr.accum = grad_color.data_ptr(), (accum_ptr if accum_ptr is not None else _ptr(accum))
    gr.dL_dmeans3D, gr.dL_dmeans2D = _ptr(d_means3D), _ptr(d_means2D)
    gr.dL_dcolors, gr.dL_dopacities = _ptr(d_colors), _ptr(d_opac)
    gr.dL_dscales, gr.dL_drotations, gr.dL_dcov3D, gr.dL_dshs = _ptr(d_scales), _ptr(d_rots), _ptr(d_cov), _ptr(d_sh)
    gr.flags = ((_capi.SPLAT_GRADS_UPSTREAM_SCALE if _UPSTREAM_SCALE_GRADIENT else 0) | (_capi.SPLAT_GRADS_POISON_IF_FLAGGED if poison else 0) |
                (_capi.SPLAT_GRADS_ACCUM_ZEROED if accum_ptr is not None else 0))
    with _on_device(dev):
        _capi.check(L.splat_backward(C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), C.byref(gr), _stream(dev)), "splat_backward")
    return d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rots, d_cov


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                grad_enabled=True):
        # grad_enabled: torch.is_grad_enabled() as the CALLER saw it (inside Function.forward it is always False, and
        # ctx.needs_input_grad reflects the inputs' requires_grad whatever the grad mode: nn.Parameters rendered under
        # torch.no_grad -- evaluation, add_new_gaussians -- would otherwise count as "a backward will follow")
        dev = means3D.device
        means3D_src = means3D                   # what the caller passed (the geometry cache proves identity on it)
        means3D = _check_input("means3D", means3D, dev)
        P = means3D.shape[0]
        sh = _check_input("shs", sh, dev) if sh.numel() else sh
        colors = _check_input("colors_precomp", colors_precomp, dev) if colors_precomp.numel() else colors_precomp
        opac = _check_input("opacities", opacities, dev).reshape(-1) if opacities.numel() else opacities
        scales_c = _check_input("scales", scales, dev) if scales.numel() else scales
        rots_c = _check_input("rotations", rotations, dev) if rotations.numel() else rotations
        cov_c = _check_input("cov3D_precomp", cov3Ds_precomp, dev) if cov3Ds_precomp.numel() else cov3Ds_precomp
        for name, t, last in (("means3D", means3D, 3), ("scales", scales_c, 3), ("rotations", rots_c, 4), ("cov3D_precomp", cov_c, 6)):
            if t.numel() and (t.dim() != 2 or t.shape[0] != P or t.shape[1] != last):
                raise RuntimeError(f"{name} must be [{P}, {last}], got {tuple(t.shape)}")
        if opac.numel() != P:
            raise RuntimeError(f"opacities must hold {P} values, got {tuple(opacities.shape)}")
        if colors.numel() and (colors.dim() != 2 or colors.shape[0] != P):
            raise RuntimeError(f"colors_precomp must be [{P}, C], got {tuple(colors.shape)}")
        if sh.numel() and (sh.dim() != 3 or sh.shape[0] != P or sh.shape[2] != 3):
            raise RuntimeError(f"shs must be [{P}, M, 3], got {tuple(sh.shape)}")
        color, radii, depth, pk = rasterize_forward(raster_settings, means3D, colors, opac, scales_c, rots_c, cov_c, sh,
                                                    will_backward=bool(grad_enabled) and any(ctx.needs_input_grad), means3D_src=means3D_src)
        ctx.pack = pk
        ctx.set_materialize_grads(False)        # (radii and depth carry no gradient: no zero tensors are built for them per backward)
        ctx.use_sh = sh.numel() > 0
        ctx.sh_shape = tuple(sh.shape)
        ctx.use_cov = cov_c.numel() > 0
        ctx.opac_shape = tuple(opacities.shape)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth):
        # the rendered depth carries no gradient in the reference either
        pk = ctx.pack
        if grad_out_color is None:
            return (None,) * 10
        d_means3D, d_means2D, d_sh, d_colors, d_opac, d_scales, d_rots, d_cov = rasterize_backward(
            pk, grad_out_color, need_scale_rot=not ctx.use_cov, need_cov3D=ctx.use_cov,
            use_sh=ctx.use_sh, sh_shape=ctx.sh_shape)
        return (d_means3D, d_means2D, d_sh, d_colors, d_opac.reshape(ctx.opac_shape), d_scales, d_rots, d_cov, None, None)


_apply_c = super(torch.autograd.Function, _RasterizeGaussians).apply
_empties: dict = {}


def _empty_on(dev):
    e = _empties.get(dev)
    if e is None:
        e = _empties[dev] = torch.empty(0, dtype=torch.float32, device=dev)
    return e


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    # (the C-level apply: torch.autograd.Function.apply first walks its arguments for functorch wrappers, ~8 us per call; no transform
    #  of that kind can differentiate through a ctypes call anyway)
    if torch._C._are_functorch_transforms_active():
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings, torch.is_grad_enabled())
    return _apply_c(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                    torch.is_grad_enabled())


class GaussianRasterizer(nn.Module):
    """``GaussianRasterizer(raster_settings=cam)(**rendervar)`` -- the callers build a FRESH module for every render
    (/root/reference/scripts/splatam.py:249,253,384), so construction and call are kept cheap: the ``nn.Module`` bookkeeping (a dozen
    ordered dicts, ~13 us) is only built when something asks for it (hooks, ``parameters()``, ``repr`` ...), and a module without it is
    called straight through to ``forward``."""

    def __init__(self, raster_settings):
        object.__setattr__(self, "raster_settings", raster_settings)

    def _materialise(self):
        settings = self.__dict__["raster_settings"]
        nn.Module.__init__(self)
        object.__setattr__(self, "raster_settings", settings)

    def __getattr__(self, name):
        if "_parameters" not in self.__dict__:          # nn.Module state asked for the first time
            self._materialise()
            return getattr(self, name)
        return super().__getattr__(name)

    def __call__(self, *args, **kwargs):
        if "_parameters" not in self.__dict__:          # no hooks can be registered on a module without its state
            return self.forward(*args, **kwargs)
        return super().__call__(*args, **kwargs)

    def markVisible(self, positions):
        """Boolean mask of the Gaussians in front of the near plane (view-space z > 0.2)."""
        L = _capi.lib()
        with torch.no_grad():
            pos = _check_input("positions", positions, positions.device)
            view = _cached_contiguous(self.raster_settings.viewmatrix)
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=pos.device)
            with torch.cuda.device(pos.device):
                _capi.check(L.splat_mark_visible(pos.shape[0], _ptr(pos), view.data_ptr(), _ptr(out), _stream(pos.device)),
                            "splat_mark_visible")
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = _empty_on(means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)
