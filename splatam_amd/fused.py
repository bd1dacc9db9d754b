"""Fused SplaTAM iteration: the reference's per-iteration Python (``get_loss`` ->
``loss.backward()`` -> ``optimizer.step()``, /root/reference/scripts/splatam.py:690-711
and :828-869) as ~10 kernel launches of libsplat_hip.so with no host synchronisation.

``FusedEngine`` owns the device scratch (geometry, per-tile lists, the 6-channel
render, loss gradients, Adam moments) and updates the caller's ``params`` tensors
IN PLACE, exactly as ``torch.optim.Adam`` does for the reference.  Results are the
same function of the same inputs as ``splatam_amd.slam.get_loss`` + autograd +
``torch.optim.Adam`` (tests/test_gpu_fused.py), for every argument of ``get_loss``
(incl. ``ignore_outlier_depth_loss``: ``torch.median`` by exact radix selection); only
the colour pass' own ``means2D.grad`` (gradient-based densification) is not produced.

Everything is computed by the C ABI (include/splat_hip.h, "Fused SplaTAM
iteration"); PyTorch only owns the memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
import math

import torch

from . import _capi
from .rasterizer import _cached_contiguous

PARAM_ORDER = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")
VARIABLE_KEYS = ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")
_FLAG_SLOTS = 16        # floats ahead of the flat gradient bucket (one 64-byte line): slot 0 = capacity flag of the exchange


class FusedEngine:
    def __init__(self, params, cam, capacity=None, track_max_radius=None, gaussian_capacity=None, variables=None, row_headroom=0.0):
        """params: the reference's dict of float32 CUDA tensors / Parameters (updated in place);
        cam: a GaussianRasterizationSettings; capacity: (Gaussian, tile) instances the lists can hold;
        gaussian_capacity: rows the map may grow to.  When given, the five Gaussian tensors (and ``variables``' per-Gaussian
        entries) move into capacity-sized backing arrays owned by the engine and ``params[k]`` / ``variables[k]`` become
        views of their first P rows, re-made after every ``add_new_gaussians`` / ``prune_gaussians`` -- the reference
        replaces the dict entries at the same places (/root/reference/scripts/splatam.py:410-411).
        row_headroom (engines that do NOT own the map): the per-Gaussian scratch is allocated for (1 + row_headroom) x P rows, so
        that ``rebind`` to a map the caller has grown (splatam_amd.plugin) need not re-allocate it."""
        self.L = _capi.lib()
        self.params = params
        self.variables = variables
        self.cam_settings = cam
        dev = params['means3D'].device
        if dev.type != "cuda":
            raise RuntimeError("FusedEngine needs CUDA/HIP tensors; the HIP library has no CPU path")
        self.dev = dev
        for k in PARAM_ORDER + ("cam_unnorm_rots", "cam_trans"):
            t = params[k]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise RuntimeError(f"params['{k}'] must be a contiguous float32 tensor on {dev}")
        P = params['means3D'].shape[0]
        self.P = P
        self.iso = params['log_scales'].shape[1] == 1
        self.managed = gaussian_capacity is not None
        self.Pcap = max(int(gaussian_capacity), P) if self.managed else int(P * (1.0 + float(row_headroom)))
        self.store = None
        if self.managed:
            self._adopt(params, variables)
        elif variables is not None and track_max_radius is None:
            track_max_radius = variables.get('max_2D_radius')
        P_alloc = self.Pcap
        self.num_frames = params['cam_unnorm_rots'].shape[-1]
        H, W = int(cam.image_height), int(cam.image_width)
        self.H, self.W = H, W
        T = int(self.L.splat_num_tiles(W, H))
        f32, i32 = torch.float32, torch.int32
        self.capacity = int(capacity) if capacity else 4 * P + 65536
        z = dict(device=dev)
        b = self.buf = {}
        # every array of the iteration's workspace is sized by the LIBRARY (splat_iter_workspace_layout, include/splat_hip.h "Scratch
        # layouts"); they are tensors of their own here because the per-Gaussian ones grow with the map (_grow_rows / rebind)
        self._alloc_fixed(self._layout(P_alloc))
        self._alloc_rows(self._layout(P_alloc))
        GT = _capi.SPLAT_GROUP_TILES
        self.num_groups = (((W + 15) // 16 + GT - 1) // GT) * (((H + 15) // 16 + GT - 1) // GT)
        # launch order of the composites' workgroups (SplatState.tile_work / tile_order): heaviest tiles of every XCD band first.  An entry
        # holds tile + 1; the zero-initialised buffer of the library's layout IS the natural order
        self.tile_order_on = os.environ.get("SPLAT_TILE_ORDER", "1") != "0"
        # ... one order PER VIEW (keyed by the frame's time index, the last 64 views): an iteration leaves the order for the NEXT visit
        # of its view.  Mapping draws a random keyframe per iteration (/root/reference/scripts/splatam.py:831-845): the order the
        # previous iteration left belongs to another view
        self.order_per_view = os.environ.get("SPLAT_TILE_ORDER_PER_VIEW", "1") != "0"
        self._natural_order, self._orders = torch.zeros_like(b['tile_order']), {}
        b['pose_state'] = torch.zeros(_capi.SPLAT_POSE_STATE, dtype=f32, **z)
        self.max_2D_radius = self.store['max_2D_radius'] if self.managed else track_max_radius
        b['counts'] = torch.zeros(8, dtype=i32, **z)
        # map gradients: ONE flat buffer (the all-reduce bucket of the view-sharded mapping step), viewed per parameter
        self._widths = [3, 3, 4, 1, 1 if self.iso else 3]
        self._grad_store = torch.zeros(_FLAG_SLOTS + sum(self._widths) * P_alloc, dtype=f32, **z)
        self._m_store = {k: torch.zeros(P_alloc, w, dtype=f32, **z) for k, w in zip(PARAM_ORDER, self._widths)}
        self._v_store = {k: torch.zeros(P_alloc, w, dtype=f32, **z) for k, w in zip(PARAM_ORDER, self._widths)}
        self._layout_rows()
        self.map_step = 0
        self.pose_step = 0
        self.track_time_idx = None
        self.max_list_hint = 0          # longest tile list seen at the last check_overflow(); 0 = unknown
        self.tile_stride = 0            # > 0: bucketed lists (no scan / scatter pass), learnt by check_overflow()
        self.num_tiles = T
        self.allow_buckets = True
        # group binning (SplatState.group_count): with bucketed lists short enough for the composite's own sort, the per-Gaussian
        # kernel files one record per 2 x 2-tile group (slots through an LDS histogram) and the composite filters its group's
        # records: ~10x fewer global atomics in the per-Gaussian kernel.  Results do not depend on it
        self.group_bins = True
        # the Adam step inside F6 touches moments and parameters row by row (12-byte rows).  Round 2: it lost to the separate, fully
        # coalesced adam_map_kernel once the rows no longer stayed cache resident (96 us fused vs 42 + 42 us at 830 k rows) and maps above
        # 500 k rows took the two-kernel form.  Since F6 requests all its moments in one round (round 3) the fused form is ahead at every
        # size measured (mapping at 816 k rows 1 491 -> 1 505 it/s, at 5 M 383 -> 386): no limit by default (SPLAT_FUSED_ADAM_MAX_ROWS)
        self.fused_adam_max_rows = int(os.environ.get("SPLAT_FUSED_ADAM_MAX_ROWS", 1 << 30))
        self._tile_rows = None          # (begin, end): the band of tile rows the next iteration composites (tile-row-sharded tracking)
        self._stats_partial = False     # the last iteration's list statistics cover a band only: check_overflow() does not learn from them
        self.sub_bins = 1               # counters per tile on the exact-list path (16 once lists get very long: SplatState.sub_bins)
        # rows in creation (pixel-scan) order: true for a map this engine grew itself (add_valid_depth_points / add_new_gaussians
        # append per pixel in scan order); callers that hand over such a map may set it.  Only a speed hint (SplatState.order_hint)
        self.creation_order = bool(self.managed and P == 0)
        self.keep_map_grads = True      # mapping_iteration: store the gradients beside the fused Adam step (False: as the reference's loop, which discards them)
        self.use_recs = {"0": 0, "1": 1}.get(os.environ.get("SPLAT_TILE_RECS", "auto"), 2)      # SplatState.tile_recs: 0 never, 1 always, 2 by list length
        self._alloc_lists(self.capacity)
        self._cam = self._make_cam(cam)
        self._cam_ok = {}
        self._frame_keep = None
        self.track_fused = os.environ.get("SPLAT_TRACK_FUSED", "1") != "0"    # tracking: forward + loss + backward composite in one kernel
        self.track_fused_full = os.environ.get("SPLAT_TRACK_FUSED_FULL", "1") != "0"   # ... also when the map's gradients are wanted (the mapping form inside)
        self.fold_sums = os.environ.get("SPLAT_FOLD_SUMS", "1") != "0"     # tile-row-sharded tracking: exchange 256 B instead of 16 KB
        self.skipped_iterations = 0     # of the last check_overflow() / digest_report(): iterations whose Adam step the device skipped
        self._learnt_P = None           # rows of the map the list statistics were learnt on (rebind keeps them for a similar map)

    # ------------------------------------------------------------------ capacity-managed map
    # workspace arrays by the library's field names: (key in self.buf, dtype, trailing shape); per-Gaussian ones grow with the map
    _ROW_ARRAYS = (("st.conic_opacity", "conic", torch.float32, (4,)), ("st.xy", "xy", torch.float32, (2,)), ("st.rect", "rect", torch.int32, (2,)),
                   ("st.depth", "depth", torch.float32, ()), ("st.radii", "radii", torch.int32, ()), ("feat8", "feat8", torch.float32, (8,)),
                   ("accum", "accum", torch.float32, (_capi.SPLAT_GRAD_STRIDE,)))
    _FIXED_ARRAYS = (("st.tile_count", "tile_count", torch.int32), ("st.tile_base", "tile_base", torch.int32),
                     ("st.tile_cursor", "tile_cursor", torch.int32), ("st.long_base", "long_base", torch.int32),
                     ("st.group_count", "group_count", torch.int32), ("st.status", "status", torch.int32),
                     ("st.tile_work", "tile_work", torch.int32), ("st.tile_order", "tile_order", torch.int32),
                     ("st.final_T", "final_T", torch.float32),
                     ("st.n_contrib", "n_contrib", torch.int32), ("out6", "out6", torch.float32), ("dL_dout6", "dL_dout6", torch.float32),
                     ("ssim_maps", "ssim_maps", torch.float32), ("sums", "sums", torch.float64), ("d_cam", "d_cam", torch.float32))

    def _layout(self, rows, capacity=0, group_stride=0, outlier=False):
        flags = _capi.SPLAT_LAYOUT_SSIM | _capi.SPLAT_LAYOUT_TILE_ORDER | _capi.SPLAT_LAYOUT_RECS | (_capi.SPLAT_LAYOUT_OUTLIER if outlier else 0)
        return _capi.iter_workspace_layout(int(rows), self.W, self.H, int(capacity), int(group_stride), flags)

    def _new(self, lay, name, dtype, tail=()):
        n = lay.bytes[name] // torch.empty((), dtype=dtype).element_size()
        t = (torch.zeros if lay.zero_init[name] else torch.empty)(n, dtype=dtype, device=self.dev)
        return t.view((-1,) + tuple(tail)) if tail else t

    def _alloc_fixed(self, lay):
        for name, key, dtype in self._FIXED_ARRAYS:
            self.buf[key] = self._new(lay, name, dtype)
        H, W = self.H, self.W
        for key, lead in (('final_T', ()), ('n_contrib', ()), ('out6', (6,)), ('dL_dout6', (6,)), ('ssim_maps', (9,))):
            self.buf[key] = self.buf[key].view(lead + (H, W))

    def _alloc_rows(self, lay):
        for name, key, dtype, tail in self._ROW_ARRAYS:
            self.buf[key] = self._new(lay, name, dtype, tail)

    def _adopt(self, params, variables):
        """Move the caller's Gaussian tensors into backing arrays of ``self.Pcap`` rows."""
        dev, P = self.dev, self.P
        self.store = {}
        for k in PARAM_ORDER:
            src = params[k].detach()
            back = torch.zeros((self.Pcap,) + tuple(src.shape[1:]), dtype=torch.float32, device=dev)
            back[:P] = src
            self.store[k] = back
        for k in VARIABLE_KEYS:
            back = torch.zeros(self.Pcap, dtype=torch.float32, device=dev)
            if variables is not None and k in variables:
                back[:P] = variables[k]
            self.store[k] = back
        self._publish()

    def _publish(self):
        """params[k] / variables[k] = views of the first P rows of the backing arrays."""
        P = self.P
        for k in PARAM_ORDER:
            self.params[k] = torch.nn.Parameter(self.store[k][:P], requires_grad=True)
        if self.variables is not None:
            for k in VARIABLE_KEYS:
                self.variables[k] = self.store[k][:P]
        self.max_2D_radius = self.store['max_2D_radius']

    def _layout_rows(self):
        """Views that depend on the number of rows: the flat gradient bucket and the moment views."""
        P = self.P
        # [0, _FLAG_SLOTS): the exchange's header -- slot 0 carries this rank's capacity flag through the SAME collective as the gradients
        # (exchange_gradients); the gradients follow, 64-byte aligned
        self.grad_flat = self._grad_store[_FLAG_SLOTS:_FLAG_SLOTS + sum(self._widths) * P]
        # the rotations go last: for an isotropic map their gradient is exactly zero and the exchange skips them
        order = [k for k in PARAM_ORDER if k != "unnorm_rotations"] + ["unnorm_rotations"]
        widths = dict(zip(PARAM_ORDER, self._widths))
        self.grads, o = {}, 0
        for k in order:
            self.grads[k] = self.grad_flat[o:o + widths[k] * P].view(P, widths[k])
            o += widths[k] * P
        self.reduce_flat = self.grad_flat[:(sum(self._widths) - 4) * P] if self.iso else self.grad_flat
        self._exchange_flat = self._grad_store[:_FLAG_SLOTS + self.reduce_flat.numel()]
        self.exp_avg = {k: self._m_store[k][:P] for k in PARAM_ORDER}
        self.exp_avg_sq = {k: self._v_store[k][:P] for k in PARAM_ORDER}

    def rebind(self, params, track_max_radius=None, keep_lists_within=0.10):
        """The caller replaced its tensors (the reference's add_new_gaussians / remove_points re-create every parameter:
        /root/reference/scripts/splatam.py:410-411, /root/reference/utils/slam_external.py:139-162): point the engine at the new
        ones.  The workspace (per-pixel planes, lists, records), the list statistics and the bucket stride are KEPT when the number
        of rows moved by less than ``keep_lists_within`` of the rows they were learnt on -- an engine built from scratch starts on
        exact lists (scan + scatter + sort launches) and re-learns them through a host read.  Engines that own their map
        (gaussian_capacity) are edited through add_new_gaussians / remove_points instead."""
        if self.managed:
            raise RuntimeError("rebind is for engines on caller-owned tensors")
        dev = self.dev
        for k in PARAM_ORDER + ("cam_unnorm_rots", "cam_trans"):
            t = params[k]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise RuntimeError(f"params['{k}'] must be a contiguous float32 tensor on {dev}")
        if (params['log_scales'].shape[1] == 1) != self.iso or params['cam_unnorm_rots'].shape[-1] != self.num_frames:
            raise RuntimeError("rebind: the map's layout (isotropy, number of frames) differs from the engine's")
        P = int(params['means3D'].shape[0])
        self.params = params
        self.max_2D_radius = track_max_radius
        if P > self.Pcap:
            # per-Gaussian scratch for the grown map (+12.5 %: the next few edits fit); accum must be zero, the others are written
            # by the per-Gaussian kernel before they are read
            cap = P + P // 8 + 1024
            f32, i32, b = torch.float32, torch.int32, self.buf
            self._alloc_rows(self._layout(cap))
            self._grad_store = torch.zeros(_FLAG_SLOTS + sum(self._widths) * cap, dtype=f32, device=dev)
            self._m_store = {k: torch.zeros(cap, w, dtype=f32, device=dev) for k, w in zip(PARAM_ORDER, self._widths)}
            self._v_store = {k: torch.zeros(cap, w, dtype=f32, device=dev) for k, w in zip(PARAM_ORDER, self._widths)}
            self.Pcap = cap
        changed = P != self.P
        self.P = P
        self._layout_rows()
        if changed:
            ref = self._learnt_P
            if ref is None or abs(P - ref) > keep_lists_within * max(ref, 1):
                self.tile_stride = 0
                self.max_list_hint = 0
        return self

    def _grow_rows(self, new_cap):
        """Re-allocate every per-Gaussian array for ``new_cap`` rows (contents of the first P rows kept)."""
        if not self.managed:
            raise RuntimeError("this FusedEngine was built without gaussian_capacity: the map cannot grow")
        dev, P, b = self.dev, self.P, self.buf
        f32, i32 = torch.float32, torch.int32

        def grown(t, rows):
            n = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            n[:P] = t[:P]
            return n
        self.Pcap = int(new_cap)
        for k in list(self.store):
            self.store[k] = grown(self.store[k], self.Pcap)
        for k in PARAM_ORDER:
            self._m_store[k] = grown(self._m_store[k], self.Pcap)
            self._v_store[k] = grown(self._v_store[k], self.Pcap)
        self._grad_store = torch.zeros(_FLAG_SLOTS + sum(self._widths) * self.Pcap, dtype=f32, device=dev)
        self._alloc_rows(self._layout(self.Pcap))
        b.pop('flags', None)
        b.pop('stage', None)
        b.pop('map_scratch', None)
        self._publish()
        self._layout_rows()

    def _store_struct(self, with_moments):
        st = _capi.SplatMapStore()
        st.map = self._map_struct()
        st.capacity = self.Pcap
        for i, k in enumerate(PARAM_ORDER):
            st.exp_avg[i] = self._m_store[k].data_ptr() if with_moments else None
            st.exp_avg_sq[i] = self._v_store[k].data_ptr() if with_moments else None
        if self.managed:
            st.max_2D_radius = self.store['max_2D_radius'].data_ptr()
            st.means2D_gradient_accum = self.store['means2D_gradient_accum'].data_ptr()
            st.denom = self.store['denom'].data_ptr()
            st.timestep = self.store['timestep'].data_ptr()
        st.counts = self.buf['counts'].data_ptr()
        return st

    def _map_scratch(self, n):
        words = int(self.L.splat_map_scratch_words(int(n)))
        sc = self.buf.get('map_scratch')
        if sc is None or sc.numel() < words:
            sc = self.buf['map_scratch'] = torch.zeros(words, dtype=torch.int32, device=self.dev)
        return sc

    def _set_rows(self, P, keep_lists_within=0.10):
        self.P = int(P)
        self._publish()
        self._layout_rows()
        # the per-tile list statistics were learnt for another map: back to exact lists until check_overflow() / relearn_lists()
        # re-learns them -- unless the number of rows moved by less than ``keep_lists_within`` of the rows they were learnt on (the frame
        # loop's edits: +0.7 % per add_new_gaussians, a handful of rows per pruning): the buckets are 1.5x the longest list seen, the
        # statistics are refreshed at the end of every phase, and a list that does outgrow its bucket raises the flag (the skipped
        # iterations are run again), so stale statistics can cost time, never a wrong step
        ref = self._learnt_P
        if ref is None or self.tile_stride == 0 or abs(self.P - ref) > keep_lists_within * max(ref, 1):
            self.tile_stride = 0
            self.max_list_hint = 0

    def render(self, curr_data, time_idx):
        """Forward-only 6-channel render of the map from pose ``time_idx`` (no loss, no gradients): returns
        ``rendered()``.  The render of add_new_gaussians (/root/reference/scripts/splatam.py:381-385)."""
        self._check_cam(curr_data)
        fr = _capi.SplatFrameData()
        w2c = curr_data['w2c'] if curr_data['w2c'].is_contiguous() else curr_data['w2c'].contiguous()
        fr.im, fr.depth, fr.w2c, fr.time_idx = None, None, w2c.data_ptr(), int(time_idx)
        self._frame_keep = (w2c,)
        self._tile_rows, self._stats_partial = None, False         # a whole-frame render: its list statistics are the frame's
        self._select_order(int(time_idx))
        ws = self._workspace(False, with_ssim=False)
        ws.max_2D_radius = None
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_render(C.byref(self._cam), C.byref(m), C.byref(fr), C.byref(ws), self._stream()),
                        "splat_iter_render")
        return self.rendered()

    def lists_known(self):
        """The per-tile list statistics (bucket stride, longest list) are usable for the map as it is: an edit kept them (_set_rows)."""
        return self.tile_stride > 0 and self.max_list_hint > 0

    def relearn_lists(self, curr_data, time_idx):
        """One probe render with exact lists + ``check_overflow()``: sizes the list capacity and the per-tile buckets
        for the map as it is now (call after an edit that did not keep them -- ``lists_known()``; one D2H read)."""
        self.tile_stride = 0
        self.max_list_hint = 0
        for _ in range(3):
            self.render(curr_data, time_idx)
            if not self.check_overflow():
                return
        raise RuntimeError("per-tile lists could not be sized")

    def _append(self, mode, curr_data, time_idx, sil_thres, w2c=None):
        if not self.managed:
            raise RuntimeError("this FusedEngine was built without gaussian_capacity: the map cannot grow")
        H, W = self.H, self.W
        im, depth = curr_data['im'].contiguous(), curr_data['depth'].contiguous()
        if tuple(im.shape) != (3, H, W) or tuple(depth.shape) != (1, H, W):
            raise RuntimeError("frame size differs from the engine's camera")
        k = curr_data['intrinsics']
        fx, fy, cx, cy = float(k[0][0]), float(k[1][1]), float(k[0][2]), float(k[1][2])
        a = _capi.SplatAddArgs()
        a.mode, a.width, a.height = mode, W, H
        a.im, a.depth = im.data_ptr(), depth.data_ptr()
        a.out6 = self.buf['out6'].data_ptr() if mode == _capi.SPLAT_ADD_NON_PRESENCE else None
        a.fx, a.fy, a.cx, a.cy, a.sil_thres, a.time_idx = fx, fy, cx, cy, float(sil_thres), int(time_idx)
        keep = (im, depth)
        if w2c is not None:
            w2c = w2c.to(device=self.dev, dtype=torch.float32).contiguous()
            a.w2c = w2c.data_ptr()
            keep += (w2c,)
        a.err = self.buf['ssim_maps'].data_ptr()
        a.scratch = self._map_scratch(max(H * W, self.Pcap)).data_ptr()
        while True:
            st = self._store_struct(with_moments=True)
            with torch.cuda.device(self.dev):
                _capi.check(self.L.splat_map_add_new_gaussians(C.byref(st), C.byref(a), self._stream()), "splat_map_add_new_gaussians")
            counts = self.buf['counts'].tolist()            # the one host sync of the edit
            if not counts[2]:
                break
            self._grow_rows(int((self.P + counts[1]) * 1.5) + 1024)
            a.scratch = self._map_scratch(max(H * W, self.Pcap)).data_ptr()
        added = counts[0] - self.P
        if added or counts[3]:
            self._set_rows(counts[0])
        return added

    def add_new_gaussians(self, curr_data, sil_thres, time_idx, mean_sq_dist_method="projective", gaussian_distribution=None,
                          depth_sil=None):
        """/root/reference/scripts/splatam.py:378-420 on the device, in place: render at the tracked pose, select the
        pixels the map does not explain, append one Gaussian per selected pixel (pixel order).  Returns the number added.
        ``depth_sil`` ([>=2,H,W]: depth, silhouette) replaces the render (tests)."""
        if mean_sq_dist_method != "projective":
            raise ValueError(f"Unknown mean_sq_dist_method {mean_sq_dist_method}")
        if gaussian_distribution is not None and (gaussian_distribution == "isotropic") != self.iso:
            raise ValueError("gaussian_distribution differs from the map's log_scales layout")
        if depth_sil is None:
            # the densification render must come from complete, sorted lists: a spilled bucket / stale list-length hint would
            # permanently add wrong Gaussians (the status words are overwritten by the next relearn_lists)
            for _ in range(3):
                self.render(curr_data, time_idx)
                if not self.check_overflow():
                    break
            else:
                raise RuntimeError("add_new_gaussians: the per-tile lists could not be sized for the densification render")
        else:
            self.buf['out6'][3] = depth_sil[0]
            self.buf['out6'][4] = depth_sil[1]
        return self._append(_capi.SPLAT_ADD_NON_PRESENCE, curr_data, time_idx, sil_thres)

    def add_valid_depth_points(self, color, depth, intrinsics, w2c, time_idx=0):
        """get_pointcloud(mask = depth > 0) + initialize_params' Gaussian rows (/root/reference/scripts/splatam.py:197-206):
        one Gaussian per valid-depth pixel of the frame, appended to the map."""
        return self._append(_capi.SPLAT_ADD_VALID_DEPTH, {'im': color, 'depth': depth, 'intrinsics': intrinsics}, time_idx, 0.0, w2c=w2c)

    def remove_points(self, to_remove=None, removal_opacity_threshold=0.0, big_scale=None):
        """remove_points (/root/reference/utils/slam_external.py:139-162): stable in-place compaction of parameters,
        Adam moments and per-Gaussian variables.  ``to_remove``: bool[P], or None to form the flags on the device from
        prune_gaussians' two rules.  Returns the number removed."""
        if not self.managed:
            raise RuntimeError("this FusedEngine was built without gaussian_capacity: rows cannot be removed")
        if self.P == 0:
            return 0
        a = _capi.SplatPruneArgs()
        a.removal_opacity_threshold = float(removal_opacity_threshold)
        a.remove_big, a.big_scale = int(big_scale is not None), float(big_scale or 0.0)
        keep = None
        if to_remove is not None:
            keep = to_remove.to(device=self.dev, dtype=torch.uint8).contiguous()
            a.to_remove = keep.data_ptr()
        b = self.buf
        if 'flags' not in b:
            b['flags'] = torch.empty(self.Pcap, dtype=torch.uint8, device=self.dev)
        st = self._store_struct(with_moments=True)
        need = ((self.Pcap + 3) // 4 * 4) * int(self.L.splat_map_row_floats(C.byref(st)))
        if 'stage' not in b or b['stage'].numel() < need:
            b['stage'] = torch.empty(need, dtype=torch.float32, device=self.dev)
        a.flags, a.stage = b['flags'].data_ptr(), b['stage'].data_ptr()
        a.scratch = self._map_scratch(max(self.H * self.W, self.Pcap)).data_ptr()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_map_prune(C.byref(st), C.byref(a), self._stream()), "splat_map_prune")
        counts = b['counts'].tolist()
        if counts[1]:
            self._set_rows(counts[0])
        return counts[1]

    def prune_gaussians(self, iter, prune_dict, scene_radius):
        """prune_gaussians (/root/reference/utils/slam_external.py:169-196) on the schedule of ``prune_dict``."""
        removed = 0
        if iter <= prune_dict['stop_after']:
            if iter >= prune_dict['start_after'] and iter % prune_dict['prune_every'] == 0:
                thr = prune_dict['final_removal_opacity_threshold'] if iter == prune_dict['stop_after'] \
                    else prune_dict['removal_opacity_threshold']
                # 0.1 * variables['scene_radius'] as the reference forms it (a float32 tensor product when given a tensor)
                big = float(0.1 * scene_radius) if iter >= prune_dict['remove_big_after'] else None
                removed = self.remove_points(None, thr, big)
            if iter > 0 and iter % prune_dict['reset_opacities_every'] == 0 and prune_dict['reset_opacities']:
                with torch.no_grad():
                    self._reset_opacities()
        return removed

    def _reset_opacities(self):
        """The reference's opacity reset re-creates the parameter through update_params_and_optimizer
        (/root/reference/utils/slam_external.py:186-190, :236-240): value inverse_sigmoid(0.01), fresh Adam state and NO .grad, so
        the optimizer.step() of that iteration leaves logit_opacities alone.  Same here: moments and this iteration's gradient are
        zeroed (adam_map skips elements whose gradient and both moments are zero)."""
        self.params['logit_opacities'].fill_(math.log(0.01 / (1 - 0.01)))
        self.exp_avg['logit_opacities'].zero_()
        self.exp_avg_sq['logit_opacities'].zero_()
        self.grads['logit_opacities'].zero_()

    # ------------------------------------------------------------------ gradient-based densification
    def accumulate_mean2d_gradient(self, want_grad=False):
        """accumulate_mean2d_gradient (/root/reference/utils/slam_external.py:100-104) for the iteration whose ``loss_backward``
        has just run: variables['means2D_gradient_accum'][seen] += |colour pass' dL/dmeans2D.xy|, variables['denom'][seen] += 1.
        The fused backward sums the RGB and the depth render together, so the colour pass' own screen-space gradient takes one
        more backward composite over the three colour planes (splat_iter_means2d_accumulate).  ``want_grad``: also return that
        gradient ([P, 2], the first two columns of the reference's variables['means2D'].grad)."""
        if not self.managed:
            raise RuntimeError("this FusedEngine was built without gaussian_capacity / variables: there is nothing to accumulate into")
        out = torch.empty(self.P, 2, dtype=torch.float32, device=self.dev) if want_grad else None
        ws = self._workspace(False, with_ssim=False)
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_means2d_accumulate(C.byref(self._cam), C.byref(m), C.byref(ws),
                                                             self.store['means2D_gradient_accum'].data_ptr(), self.store['denom'].data_ptr(),
                                                             out.data_ptr() if out is not None and self.P else None, self._stream()),
                        "splat_iter_means2d_accumulate")
        return out

    def means2d_gradient(self):
        """The colour pass' screen-space gradient of the iteration whose ``loss_backward`` has just run (planes kept: every mapping
        iteration, tracking with ``keep_planes``): [P, 2], the first two columns of the reference's ``variables['means2D'].grad``
        (/root/reference/scripts/splatam.py:248-250).  One more backward composite over the three colour planes; nothing is
        accumulated (the caller's own accumulate_mean2d_gradient statement does that)."""
        out = torch.zeros(self.P, 2, dtype=torch.float32, device=self.dev)
        if self.P == 0:
            return out
        ws = self._workspace(False, with_ssim=False)
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_means2d_accumulate(C.byref(self._cam), C.byref(m), C.byref(ws), None, None, out.data_ptr(),
                                                             self._stream()), "splat_iter_means2d_accumulate")
        return out

    def _densify_args(self, mode, thr, small, rows_with_grad, n=1, samples=None):
        b = self.buf
        if 'flags' not in b or b['flags'].numel() < self.Pcap:
            b['flags'] = torch.empty(self.Pcap, dtype=torch.uint8, device=self.dev)
        a = _capi.SplatDensifyArgs()
        a.mode, a.grad_thresh, a.small_scale = mode, float(thr), float(small)
        a.rows_with_grad, a.num_to_split_into = int(rows_with_grad), int(n)
        a.samples = samples.data_ptr() if samples is not None and samples.numel() else None
        a.flags = b['flags'].data_ptr()
        a.scratch = self._map_scratch(max(self.H * self.W, self.Pcap)).data_ptr()
        return a

    def _select_and_append(self, mode, thr, small, rows_with_grad, n=1):
        """One clone / split step: select on the device, (split: draw the samples with torch's generator, as the reference does),
        append.  Returns the number of selected rows."""
        while True:
            a = self._densify_args(mode, thr, small, rows_with_grad, n)
            st = self._store_struct(with_moments=True)
            with torch.cuda.device(self.dev):
                _capi.check(self.L.splat_map_densify_select(C.byref(st), C.byref(a), self._stream()), "splat_map_densify_select")
            counts = self.buf['counts'].tolist()            # host sync (the reference synchronises on its boolean indexing here)
            if not counts[2]:
                break
            self._grow_rows(int((self.P + counts[1] * n) * 1.5) + 1024)
        S = counts[1]
        if S == 0:
            return 0
        samples = None
        if mode == _capi.SPLAT_DENSIFY_SPLIT:
            sel = self.buf['flags'][:self.P].bool()
            ls = self.store['log_scales'][:self.P]
            stds = torch.exp(ls)[sel].repeat(n, 3) if self.iso else torch.exp(ls)[sel].repeat(n, 1)
            # (the reference's repeat(n, 3) of an [S, 3] anisotropic scale would be [S n, 9]: its densify only works for isotropic maps)
            samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=self.dev), std=stds).contiguous()
        a = self._densify_args(mode, thr, small, rows_with_grad, n, samples)
        st = self._store_struct(with_moments=True)
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_map_duplicate(C.byref(st), C.byref(a), self._stream()), "splat_map_duplicate")
        self._set_rows(counts[0])
        return S

    def densify(self, iter, densify_dict, scene_radius, accumulate=True):
        """densify (/root/reference/utils/slam_external.py:191-240) on the device, in place, called where the reference calls it
        (between backward() and optimizer.step()): accumulate the screen-space gradient; on the schedule clone the small /
        split the large Gaussians whose mean gradient reaches ``grad_thresh``, reset the three per-Gaussian variables, remove
        the split originals, prune by opacity / size; optional opacity reset.  Returns True when the number of rows changed."""
        if iter > densify_dict['stop_after']:
            return False
        if accumulate:
            # re-runs the RGB backward composite over THIS iteration's workspace (lists, radii, feat8 indexed by the rows the
            # render saw): a caller that removes rows between loss_backward() and densify() must accumulate BEFORE it does
            # (accumulate_mean2d_gradient(), then densify(..., accumulate=False)) -- pipeline._map_frame does
            self.accumulate_mean2d_gradient()
        changed = False
        if iter >= densify_dict['start_after'] and iter % densify_dict['densify_every'] == 0:
            thr, small = densify_dict['grad_thresh'], float(0.01 * scene_radius)
            P0 = self.P
            self._select_and_append(_capi.SPLAT_DENSIFY_CLONE, thr, small, P0)
            n = int(densify_dict['num_to_split_into'])
            P1 = self.P
            S = self._select_and_append(_capi.SPLAT_DENSIFY_SPLIT, thr, small, P0, n)
            for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius'):
                self.store[k][:self.P].zero_()
            if S:
                to_remove = torch.zeros(self.P, dtype=torch.uint8, device=self.dev)
                to_remove[:P1] = self.buf['flags'][:P1]
                self.remove_points(to_remove)
            op_thr = densify_dict['final_removal_opacity_threshold'] if iter == densify_dict['stop_after'] \
                else densify_dict['removal_opacity_threshold']
            big = float(0.1 * scene_radius) if iter >= densify_dict['remove_big_after'] else None
            self.remove_points(None, op_thr, big)
            changed = True
        if iter > 0 and iter % densify_dict['reset_opacities_every'] == 0 and densify_dict['reset_opacities']:
            with torch.no_grad():
                self._reset_opacities()
        return changed

    # ------------------------------------------------------------------ plumbing
    def _alloc_lists(self, capacity):
        self.capacity = int(capacity)
        lay = self._layout(0, capacity=self.capacity)
        for name, key, dtype in (("st.keys", "keys", torch.int64), ("st.keys_alt", "keys_alt", torch.int64),        # keys_alt: merge passes of lists beyond LDS
                                 ("st.point_list", "point_list", torch.int32),
                                 # work-item table of the multi-workgroup sort (SplatState.long_items): one word per 1024 keys of a long list
                                 ("st.long_items", "long_items", torch.int32)):
            self.buf[key] = self._new(lay, name, dtype)
        # the staged record of every list entry, handed from the forward to the backward composite (SplatState.tile_recs: 48 bytes per
        # slot; left out beyond 16 GB -- the clustered stress scenes' hundreds of millions of slots -- where the backward composite
        # gathers as before)
        self.buf['tile_recs'] = None
        if self.use_recs and lay.bytes["st.tile_recs"] <= 16 << 30:
            self.buf['tile_recs'] = self._new(lay, "st.tile_recs", torch.float32)

    def _make_cam(self, settings):
        bg = _cached_contiguous(settings.bg)
        view = _cached_contiguous(settings.viewmatrix)
        proj = _cached_contiguous(settings.projmatrix)
        campos = _cached_contiguous(settings.campos)
        if float(bg.abs().max()) != 0.0:
            raise RuntimeError("the fused iteration renders with a zero background (as setup_camera builds it)")
        bg6 = torch.zeros(8, dtype=torch.float32, device=self.dev)
        cam = _capi.SplatCamera()
        cam.image_height, cam.image_width = self.H, self.W
        cam.tanfovx, cam.tanfovy = float(settings.tanfovx), float(settings.tanfovy)
        cam.bg, cam.scale_modifier = bg6.data_ptr(), float(settings.scale_modifier)
        cam.viewmatrix, cam.projmatrix = view.data_ptr(), proj.data_ptr()
        cam.sh_degree, cam.campos, cam.prefiltered = 0, campos.data_ptr(), 0
        self._cam_keep = (bg6, view, proj, campos)
        return cam

    def _check_cam(self, curr_data):
        """The engine bakes the camera into its launch arguments at construction; the reference's get_loss reads
        curr_data['cam'] on every call (/root/reference/scripts/splatam.py:249).  A different camera is an error here, not a
        silently ignored argument."""
        cam = curr_data.get('cam') if hasattr(curr_data, 'get') else None
        if cam is None or cam is self.cam_settings or id(cam) in self._cam_ok:
            return
        ref = self.cam_settings
        same = (int(cam.image_height) == int(ref.image_height) and int(cam.image_width) == int(ref.image_width)
                and float(cam.tanfovx) == float(ref.tanfovx) and float(cam.tanfovy) == float(ref.tanfovy)
                and float(cam.scale_modifier) == float(ref.scale_modifier)
                and all(torch.equal(getattr(cam, f).to(self.dev).float().reshape(-1), getattr(ref, f).to(self.dev).float().reshape(-1))
                        for f in ("viewmatrix", "projmatrix", "bg")))
        if not same:
            raise RuntimeError("curr_data['cam'] differs from the camera this FusedEngine was built for "
                               "(build one engine per camera / resolution)")
        self._cam_ok[id(cam)] = cam          # keeps the tuple alive, so the id stays unique

    def _map_struct(self):
        p = self.params
        if self.managed:          # the backing arrays (a zero-row view has no data pointer)
            p = dict(self.store, cam_unnorm_rots=p['cam_unnorm_rots'], cam_trans=p['cam_trans'])
        m = _capi.SplatMap()
        m.P, m.isotropic = self.P, int(self.iso)
        m.means3D, m.rgb_colors = p['means3D'].data_ptr(), p['rgb_colors'].data_ptr()
        m.unnorm_rotations, m.logit_opacities = p['unnorm_rotations'].data_ptr(), p['logit_opacities'].data_ptr()
        m.log_scales = p['log_scales'].data_ptr()
        m.cam_unnorm_rots, m.cam_trans = p['cam_unnorm_rots'].data_ptr(), p['cam_trans'].data_ptr()
        m.num_frames = self.num_frames
        return m

    def _workspace(self, with_map_grads, with_ssim):
        b = self.buf
        ws = _capi.SplatIterWorkspace()
        st = ws.st
        st.depth, st.xy, st.conic_opacity, st.rect = b['depth'].data_ptr(), b['xy'].data_ptr(), b['conic'].data_ptr(), b['rect'].data_ptr()
        st.radii = b['radii'].data_ptr()
        st.tile_count, st.tile_base, st.tile_cursor = b['tile_count'].data_ptr(), b['tile_base'].data_ptr(), b['tile_cursor'].data_ptr()
        st.keys, st.point_list, st.capacity = b['keys'].data_ptr(), b['point_list'].data_ptr(), self.capacity
        st.keys_alt, st.long_base = b['keys_alt'].data_ptr(), b['long_base'].data_ptr()
        # staged records handed from the forward to the backward composite: pays where a tile's list is several batches long (every batch
        # but the last is re-staged: B-loop, mapping +1.4 %) and costs where it is one (B: the forward composite's 34 MB of extra stores,
        # mapping -1.4 %): profiles/r06_experiments.md 2.  SPLAT_TILE_RECS=1 / 0 forces it on / off
        recs_on = self.use_recs == 1 or (self.use_recs == 2 and self.max_list_hint > 400)
        st.tile_recs = b['tile_recs'].data_ptr() if (recs_on and b.get('tile_recs') is not None) else None
        st.long_items = b['long_items'].data_ptr()
        st.max_list_hint = self.max_list_hint
        st.tile_stride = self.tile_stride
        st.tile_row_begin, st.tile_row_end = self._tile_rows if self._tile_rows else (0, 0)
        st.group_count, st.group_recs, st.group_stride = b['group_count'].data_ptr(), None, 0
        if self.group_bins and self.tile_stride > 0 and 0 < self.max_list_hint and self.max_list_hint * 5 // 4 <= 1024:
            gs = _capi.SPLAT_GROUP_TILES ** 2 * self.tile_stride
            need = self.num_groups * gs * 4
            if need <= 1 << 30:                       # (int32 words; 4 GiB of records)
                if b.get('group_recs') is None or b['group_recs'].numel() < need:
                    b['group_recs'] = self._new(self._layout(0, group_stride=gs), "st.group_recs", torch.int32)
                    assert b['group_recs'].numel() == need
                st.group_recs, st.group_stride = b['group_recs'].data_ptr(), gs
        st.order_hint = int(self.creation_order)
        if self.tile_order_on:
            st.tile_work, st.tile_order = b['tile_work'].data_ptr(), b['tile_order'].data_ptr()
        st.sub_bins = self.sub_bins if self.tile_stride == 0 else 1
        st.final_T, st.n_contrib, st.status = b['final_T'].data_ptr(), b['n_contrib'].data_ptr(), b['status'].data_ptr()
        ws.feat8, ws.out6, ws.dL_dout6, ws.accum = b['feat8'].data_ptr(), b['out6'].data_ptr(), b['dL_dout6'].data_ptr(), b['accum'].data_ptr()
        ws.ssim_maps = b['ssim_maps'].data_ptr() if with_ssim else None
        ws.sums = b['sums'].data_ptr()
        ws.max_2D_radius = self.max_2D_radius.data_ptr() if self.max_2D_radius is not None else None
        if with_map_grads and with_map_grads != "step only":      # ("step only": the fused Adam step takes them from registers, nothing is stored)
            g = self.grads
            ws.d_means3D, ws.d_rgb_colors = g['means3D'].data_ptr(), g['rgb_colors'].data_ptr()
            ws.d_unnorm_rotations, ws.d_logit_opacities = g['unnorm_rotations'].data_ptr(), g['logit_opacities'].data_ptr()
            ws.d_log_scales = g['log_scales'].data_ptr()
        ws.d_cam = b['d_cam'].data_ptr()
        if 'outlier_err' in b:
            ws.outlier_err, ws.outlier_scratch = b['outlier_err'].data_ptr(), b['outlier_scratch'].data_ptr()
        return ws

    def _select_order(self, view):
        """Point ``buf['tile_order']`` at the launch order view ``view`` left at its last visit (the natural order at its first)."""
        if not (self.tile_order_on and self.order_per_view):
            return
        o = self._orders.pop(view, None)
        if o is None:
            if len(self._orders) >= 64:
                self._orders.pop(next(iter(self._orders)))          # the view visited longest ago
            o = self._natural_order.clone()
        self._orders[view] = o                                      # (most recently visited last)
        self.buf['tile_order'] = o

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    @staticmethod
    def loss_config(cfg, tracking, do_ba=False, defer_finish=False, fused_composite=0):
        c = _capi.SplatLossConfig()
        c.defer_finish = int(defer_finish)
        c.fused_composite = int(fused_composite)
        c.tracking = int(tracking)
        c.camera_grad = int(tracking or do_ba)
        c.gaussians_grad = int(not tracking)
        c.use_sil_for_loss, c.sil_thres = int(cfg['use_sil_for_loss']), float(cfg['sil_thres'])
        c.use_l1, c.ignore_outlier_depth_loss = int(cfg['use_l1']), int(cfg['ignore_outlier_depth_loss'])
        c.w_im, c.w_depth = float(cfg['loss_weights']['im']), float(cfg['loss_weights']['depth'])
        return c

    # ------------------------------------------------------------------ one iteration
    def loss_backward(self, curr_data, time_idx, cfg, tracking, map_grads=None, do_ba=False, pose_adam=None, map_adam=None,
                      tile_rows=None, keep_planes=None):
        """get_loss + backward.  Afterwards (stream order): ``self.grads`` (mapping) and
        ``self.buf['d_cam']`` = [dL/dq_raw(4), dL/dt_raw(3), loss].  ``pose_adam`` (a SplatPoseAdam): the pose's Adam step
        rides in the last kernel (splat_iter_tracking_step); ``map_adam`` (a SplatAdamMap): likewise the map's
        (splat_iter_mapping_step).  ``keep_planes`` (tracking): the rendered planes / gradient planes (``rendered()``,
        ``buf['dL_dout6']``) are wanted -- default: yes, unless the pose's Adam step rides along (the loop's own iterations); without
        them the tracking iteration's composites run as ONE kernel that keeps its planes in registers (SplatLossConfig.fused_composite)."""
        if map_grads is None:
            map_grads = not tracking
        self._check_cam(curr_data)
        fr = _capi.SplatFrameData()
        im, depth = curr_data['im'], curr_data['depth']
        w2c = curr_data['w2c']
        if not (im.is_contiguous() and depth.is_contiguous() and w2c.is_contiguous()):
            im, depth, w2c = im.contiguous(), depth.contiguous(), w2c.contiguous()
        fr.im, fr.depth, fr.w2c, fr.time_idx = im.data_ptr(), depth.data_ptr(), w2c.data_ptr(), int(time_idx)
        self._frame_keep = (im, depth, w2c)
        if keep_planes is None:
            keep_planes = pose_adam is None
        # (with the map's gradients the one-kernel form carries the backward composite's mapping form at four workgroups per CU: ahead
        #  where a tile's list is one or two batches -- B: +6.4 % --, behind where it is three -- B-loop: -1.2 %; profiles/r06_experiments.md 4)
        full_ok = self.track_fused_full and 0 < self.max_list_hint <= 400
        one_kernel = (2 if keep_planes else 1) if (tracking and self.track_fused and (not map_grads or full_ok)) else 0
        lc = self.loss_config(cfg, tracking, do_ba, defer_finish=tile_rows is not None, fused_composite=one_kernel)
        self._tile_rows = tile_rows         # a band: the iteration stops before its last kernel (finish_iteration completes it)
        self._stats_partial = tile_rows is not None
        self._lc_keep = lc
        if lc.ignore_outlier_depth_loss and 'outlier_err' not in self.buf:       # scratch of the median selection, on first use
            lay = self._layout(0, outlier=True)
            self.buf['outlier_err'] = self._new(lay, "outlier_err", torch.float32)
            self.buf['outlier_scratch'] = self._new(lay, "outlier_scratch", torch.int32)
        self._select_order(int(time_idx))
        ws = self._workspace(map_grads, with_ssim=not tracking)
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            if pose_adam is not None:
                _capi.check(self.L.splat_iter_tracking_step(C.byref(self._cam), C.byref(m), C.byref(fr), C.byref(lc), C.byref(ws),
                                                            C.byref(pose_adam), self._stream()), "splat_iter_tracking_step")
            elif map_adam is not None:
                _capi.check(self.L.splat_iter_mapping_step(C.byref(self._cam), C.byref(m), C.byref(fr), C.byref(lc), C.byref(ws),
                                                           C.byref(map_adam), self._stream()), "splat_iter_mapping_step")
            else:
                _capi.check(self.L.splat_iter_loss_backward(C.byref(self._cam), C.byref(m), C.byref(fr), C.byref(lc), C.byref(ws),
                                                            self._stream()), "splat_iter_loss_backward")
        self._tile_rows = None
        self._fr_keep = fr

    def finish_iteration(self, pose_adam=None):
        """The last kernel of an iteration that ran on a band of tile rows (``loss_backward(..., tile_rows=...)``), after the caller
        has summed ``self.buf['sums']`` over the ranks: pose gradient, loss value and -- with ``pose_adam`` -- the pose's Adam step
        and the best-candidate bookkeeping (splat_iter_finish)."""
        ws = self._workspace(False, with_ssim=False)
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_finish(C.byref(self._cam), C.byref(m), C.byref(self._fr_keep), C.byref(self._lc_keep), C.byref(ws),
                                                 C.byref(pose_adam) if pose_adam is not None else None, self._stream()), "splat_iter_finish")

    def tile_row_band(self, rank, world):
        """Rows [begin, end) of the 16-pixel tile grid that rank ``rank`` of ``world`` composites in tile-row-sharded tracking."""
        from .dist import tile_row_band
        return tile_row_band((self.H + 15) // 16, rank, world)

    def _adam_map_args(self, lrs, beta1=0.9, beta2=0.999, eps=1e-15, steps=None):
        """The next step of torch.optim.Adam(param_groups, lr=0.0, eps=1e-15) over the five Gaussian groups
        (/root/reference/scripts/splatam.py:160-166).  Bias corrections in double on the host, as torch forms them; ``steps``:
        the step count of each group AFTER this step (torch counts per parameter: a parameter the caller re-created restarts),
        default: the engine's own count for all five.  The step is gated on the iteration's capacity flag (d_cam[12])."""
        if steps is None:
            self.map_step += 1
            steps = (self.map_step,) * 5
        o = _capi.SplatAdamMap()
        o.beta1, o.beta2, o.eps = beta1, beta2, eps
        for k, name in enumerate(PARAM_ORDER):
            t = max(int(steps[k]), 1)
            o.bc2_sqrt[k] = math.sqrt(1.0 - beta2 ** t)
            o.step_size[k] = lrs[name] / (1.0 - beta1 ** t)
            o.grad[k] = self.grads[name].data_ptr()
            o.exp_avg[k] = self.exp_avg[name].data_ptr()
            o.exp_avg_sq[k] = self.exp_avg_sq[name].data_ptr()
        o.gate = self.buf['d_cam'].data_ptr()
        return o

    def adam_map(self, lrs, beta1=0.9, beta2=0.999, eps=1e-15):
        """optimizer.step() of the mapping optimizer on ``self.grads`` (see _adam_map_args)."""
        o = self._adam_map_args(lrs, beta1, beta2, eps)
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_adam_map(C.byref(m), C.byref(o), self._stream()), "splat_iter_adam_map")

    def reset_map_optimizer(self):
        """The reference re-creates the optimizer for every frame's mapping phase (:821)."""
        for k in PARAM_ORDER:
            self.exp_avg[k].zero_()
            self.exp_avg_sq[k].zero_()
        self.map_step = 0

    def begin_tracking(self, time_idx):
        """Fresh Adam state and best-candidate bookkeeping for one frame (:680-684)."""
        st = self.buf['pose_state']
        st.zero_()
        st[14] = 1e20
        st[15:19] = self.params['cam_unnorm_rots'].detach()[0, :, time_idx]
        st[19:22] = self.params['cam_trans'].detach()[0, :, time_idx]
        self.pose_step = 0
        self.track_time_idx = int(time_idx)

    def adam_pose(self, lr_rot, lr_trans, beta1=0.9, beta2=0.999, eps=1e-8):
        self.pose_step += 1
        t = self.pose_step
        bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_adam_pose(C.byref(m), self.track_time_idx, self.buf['d_cam'].data_ptr(),
                                                    self.buf['pose_state'].data_ptr(), beta1, beta2, eps, math.sqrt(bc2),
                                                    lr_rot / bc1, lr_trans / bc1, self._stream()), "splat_iter_adam_pose")

    def end_tracking(self):
        """Copy the best candidate back (:741-744)."""
        st, t = self.buf['pose_state'], self.track_time_idx
        with torch.no_grad():
            self.params['cam_unnorm_rots'][0, :, t] = st[15:19]
            self.params['cam_trans'][0, :, t] = st[19:22]

    def _pose_adam_args(self, cfg):
        """The next step of the tracking optimizer (default betas / eps of torch.optim.Adam) as the C ABI takes it."""
        self.pose_step += 1
        t, beta1, beta2 = self.pose_step, 0.9, 0.999
        bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
        pa = _capi.SplatPoseAdam()
        pa.state, pa.beta1, pa.beta2, pa.eps, pa.bc2_sqrt = self.buf['pose_state'].data_ptr(), beta1, beta2, 1e-8, math.sqrt(bc2)
        pa.step_size_rot, pa.step_size_trans = cfg['lrs']['cam_unnorm_rots'] / bc1, cfg['lrs']['cam_trans'] / bc1
        return pa

    def tracking_iteration(self, curr_data, cfg, shard=None, allreduce_sums=None):
        """Loop body of /root/reference/scripts/splatam.py:690-711 for frame ``begin_tracking`` named.

        ``shard = (rank, world)``: tile-row-sharded tracking over ``world`` processes holding the same map and pose -- this rank
        composites its band of tile rows only (forward, loss, backward: every term of the loss and of the pose gradient is a sum
        over pixels), ``allreduce_sums`` sums the 16 KB of partial sums over the ranks, and every rank takes the SAME Adam step on
        the pose (no broadcast needed).  Not with ``ignore_outlier_depth_loss`` (its median sees the whole render)."""
        pa = self._pose_adam_args(cfg)
        if shard is None or shard[1] <= 1:
            self.loss_backward(curr_data, self.track_time_idx, cfg, tracking=True, pose_adam=pa)
            return
        if cfg['ignore_outlier_depth_loss']:
            raise RuntimeError("tile-row-sharded tracking needs a pixel-local loss: not with ignore_outlier_depth_loss")
        self.loss_backward(curr_data, self.track_time_idx, cfg, tracking=True, tile_rows=self.tile_row_band(*shard), keep_planes=False)
        if self.fold_sums:
            # the 64 copies of the partial sums folded into the first (one tiny launch): the exchange carries 256 bytes, not 16 KB
            with torch.cuda.device(self.dev):
                _capi.check(self.L.splat_iter_fold_sums(self.buf['sums'].data_ptr(), self._stream()), "splat_iter_fold_sums")
            allreduce_sums(self.buf['sums'][:_capi.SPLAT_ITER_SUMS])
        else:
            allreduce_sums(self.buf['sums'])
        self.finish_iteration(pa)

    def mapping_iteration(self, iter_data, iter_time_idx, cfg, bucket_allreduce=None, keep_grads=None):
        """Loop body of /root/reference/scripts/splatam.py:828-869 (without pruning / densification).  Without a gradient
        exchange the whole iteration is one C call (the Adam step rides in the last kernel: splat_iter_mapping_step).
        ``keep_grads`` (default ``self.keep_map_grads``): also STORE the gradients the step was taken on (``self.grads``); the
        reference's loop discards them right after the step (optimizer.zero_grad(set_to_none=True), :860-861), and a loop that does
        the same saves 32 - 48 bytes of stores per Gaussian and iteration."""
        if bucket_allreduce is None and self.P <= self.fused_adam_max_rows:
            keep = self.keep_map_grads if keep_grads is None else keep_grads
            self.loss_backward(iter_data, iter_time_idx, cfg, tracking=False, map_adam=self._adam_map_args(cfg['lrs']),
                               map_grads=True if keep else "step only")
            return
        self.loss_backward(iter_data, iter_time_idx, cfg, tracking=False)
        if bucket_allreduce is not None:
            self.exchange_gradients(bucket_allreduce)      # one collective: 8 (isotropic) or 14 floats per Gaussian (+ the flag header)
        self.adam_map(cfg['lrs'])

    def mapping_batch(self, views, cfg, total_views=None, allreduce_sum=None):
        """One mapping step over SEVERAL keyframe views (the view-sharded form of the loop body: BASELINE config 3): the
        gradients of the map over ``views`` = [(iter_data, iter_time_idx), ...] are accumulated in the flat bucket, summed over
        the ranks by ``allreduce_sum`` (one collective on ``reduce_flat``), divided by ``total_views`` (views of ALL ranks) and
        applied by ONE Adam step -- what a single process gets by accumulating the same views."""
        acc = self._acc_flat()
        for i, (data, t) in enumerate(views):
            self.loss_backward(data, t, cfg, tracking=False)
            if i == 0:
                acc.copy_(self.grad_flat)
            else:
                acc.add_(self.grad_flat)
        red = acc[:self.reduce_flat.numel()]
        if allreduce_sum is not None:           # (the capacity flag of any rank's views travels in the header: exchange_gradients)
            self.exchange_gradients(allreduce_sum, self._acc_store[:_FLAG_SLOTS + self.reduce_flat.numel()])
        n = float(total_views if total_views is not None else len(views))
        if n != 1.0:
            red.mul_(1.0 / n)
        self.grad_flat.copy_(acc)
        self.adam_map(cfg['lrs'])

    def exchange_gradients(self, all_reduce, flat=None):
        """The gradient exchange of a multi-rank mapping step: ``all_reduce`` (sum or mean, in place) over the flat gradient bucket WITH
        this rank's capacity flag in its header.  Ranks render different views, so typically only some overflow their lists; the
        flag of ANY rank comes back non-zero on EVERY rank and is made this rank's sticky flag (``d_cam[12]``) before the Adam step
        that follows, so the replicas skip the same steps and stay bit-identical (the reduced gradient of such an iteration holds a
        truncated-list contribution: nobody may step on it).  ``flat``: another buffer laid out like ``_exchange_flat`` (mapping_batch's
        accumulator)."""
        flat = self._exchange_flat if flat is None else flat
        flag = self.buf['d_cam'][12:13]
        flat[0:1].copy_(flag)
        all_reduce(flat)
        torch.maximum(flag, (flat[0:1] != 0.0).to(flag.dtype), out=flag)

    def _acc_flat(self):
        a = getattr(self, "_acc_store", None)
        if a is None or a.numel() < _FLAG_SLOTS + self.grad_flat.numel():
            a = self._acc_store = torch.zeros(self._grad_store.numel(), dtype=torch.float32, device=self.dev)
        return a[_FLAG_SLOTS:_FLAG_SLOTS + self.grad_flat.numel()]

    # ------------------------------------------------------------------ read-backs (host sync)
    def loss(self):
        return float(self.buf['d_cam'][7])

    def check_overflow(self, grow=True):
        """Lists are fixed-size; an iteration whose instances did not fit rendered truncated / empty lists and flagged
        it (stickily; from then on the device skips every Adam step: nothing moves on bad lists).  Call at frame end (two small
        D2H reads): returns True when iterations since the last call were flagged -- ``self.skipped_iterations`` says how many
        took no step.  Also learns the list statistics: from then on the per-tile lists are BUCKETED at 1.5x the longest
        list seen (the per-Gaussian kernel writes instances straight into their tile's bucket: no scan kernel, no
        scatter pass) and the long-list sort launch is skipped while lists stay short."""
        stat = self.buf['status'].tolist()
        rep = self.buf['d_cam'].cpu()
        return self._digest(stat, float(rep[12]) != 0.0, int(rep.view(torch.int32)[21]), grow)

    def digest_report(self, report, grow=True):
        """check_overflow() from a HOST copy of an iteration's report (``buf['d_cam']``, SPLAT_ITER_DCAM floats: the status words
        the iteration left are in [16..19]) -- for callers that fetch the report asynchronously (splatam_amd.plugin): no
        blocking read here.  Only valid for reports of whole iterations (their last kernel writes the snapshot)."""
        ints = report.view(torch.int32)
        return self._digest(ints[16:20].tolist(), float(report[12]) != 0.0, int(ints[21]), grow, hysteresis=True)

    def _digest(self, stat, sticky, skipped, grow, hysteresis=False):
        bad = sticky or stat[1] != 0 or stat[3] != 0 or (self.tile_stride == 0 and stat[0] > self.capacity)
        self.skipped_iterations = 0
        if bad:
            self.skipped_iterations = max(int(skipped), 1)
            self.buf['d_cam'][12] = 0.0
            self.buf['d_cam'][21] = 0.0             # (an int32 counter: the bit pattern of 0.0 is 0)
            self.buf['status'].zero_()
            self.buf['tile_count'].zero_()
            self.buf['group_count'].zero_()
            self.buf['accum'].zero_()
            self.buf['sums'].zero_()
            self.max_list_hint = 0
            if grow:
                if self.tile_stride > 0:            # a bucket overflowed: back to exact lists, re-learn
                    self.tile_stride = 0
                if stat[0] > self.capacity:
                    self._alloc_lists(int(stat[0] * 1.5) + 65536)
            return True
        if self._stats_partial:                 # the last iteration composited a band of tile rows: its statistics are not the frame's
            return False
        longest = int(stat[2])
        self.max_list_hint = longest            # short lists: sorted inside the composite, no sort launch
        self._learnt_P = self.P
        self._set_sub_bins(16 if longest > 2048 else 1)
        if grow and self.allow_buckets and longest > 0:
            stride = max(256, (int(longest * 1.5) + 63) // 64 * 64)
            # buckets cost 20 bytes per slot (keys, their merge partner, sorted ids): up to ~15 GB of the 288 GB for the clustered
            # stress scenes (337 M slots at 5 M Gaussians) -- one returning atomic per instance instead of count + scan + scatter
            # (reports digested every iteration: the stride only moves when the margin has become thin or the buckets far too wide --
            #  a stride that follows every fluctuation of the longest list would re-lay the buckets iteration by iteration)
            if hysteresis and self.tile_stride > 0 and longest * 5 // 4 <= self.tile_stride <= 2 * stride:
                stride = self.tile_stride
            if stride != self.tile_stride and stride * self.num_tiles <= 768 * 1024 * 1024:
                if stride * self.num_tiles > self.capacity:
                    self._alloc_lists(stride * self.num_tiles)
                self.tile_stride = stride
        return False

    def _set_sub_bins(self, S):
        """Very long per-tile lists stay on the exact-list path (their buckets would not fit); their count / scatter atomics
        are then spread over S counters per tile (same-address serialisation otherwise: 0.9 ms per pass at 11 M instances)."""
        if S == self.sub_bins:
            return
        CS = _capi.SPLAT_COUNTER_STRIDE
        self.sub_bins = S
        self.buf['tile_count'] = torch.zeros(self.num_tiles * CS * S, dtype=torch.int32, device=self.dev)
        self.buf['tile_cursor'] = torch.zeros(self.num_tiles * CS * S, dtype=torch.int32, device=self.dev)

    @property
    def seen(self):
        """variables['seen'] of the last iteration ([P], /root/reference/scripts/splatam.py:343)."""
        return self.buf['radii'][:self.P] > 0

    def rendered(self):
        """(im[3,H,W], depth[1,H,W], silhouette[H,W], depth_sq[1,H,W]) of the last iteration."""
        o = self.buf['out6']
        return o[0:3], o[3:4], o[4], o[5:6]
