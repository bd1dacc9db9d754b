"""Fused SplaTAM iteration: the reference's per-iteration Python (``get_loss`` ->
``loss.backward()`` -> ``optimizer.step()``, /root/reference/scripts/splatam.py:690-711
and :828-869) as ~10 kernel launches of libsplat_hip.so with no host synchronisation.

``FusedEngine`` owns the device scratch (geometry, per-tile lists, the 6-channel
render, loss gradients, Adam moments) and updates the caller's ``params`` tensors
IN PLACE, exactly as ``torch.optim.Adam`` does for the reference.  Results are the
same function of the same inputs as ``splatam_amd.slam.get_loss`` + autograd +
``torch.optim.Adam`` (tests/test_gpu_fused.py); what no shipped config uses
(``ignore_outlier_depth_loss``, densification from ``means2D.grad``) is not fused
and raises, so that the caller keeps the two-call path for it.

Everything is computed by the C ABI (include/splat_hip.h, "Fused SplaTAM
iteration"); PyTorch only owns the memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _capi
from .rasterizer import _cached_contiguous

PARAM_ORDER = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")


class FusedEngine:
    def __init__(self, params, cam, capacity=None, track_max_radius=None):
        """params: the reference's dict of float32 CUDA tensors / Parameters (updated in place);
        cam: a GaussianRasterizationSettings; capacity: (Gaussian, tile) instances the lists can hold."""
        self.L = _capi.lib()
        self.params = params
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
        self.num_frames = params['cam_unnorm_rots'].shape[-1]
        H, W = int(cam.image_height), int(cam.image_width)
        self.H, self.W = H, W
        T = ((W + 15) // 16) * ((H + 15) // 16)
        f32, i32 = torch.float32, torch.int32
        CS = _capi.SPLAT_COUNTER_STRIDE
        self.capacity = int(capacity) if capacity else 4 * P + 65536
        z = dict(device=dev)
        b = self.buf = {}
        b['conic'] = torch.empty(P, 4, dtype=f32, **z)
        b['xy'] = torch.empty(P, 2, dtype=f32, **z)
        b['rect'] = torch.empty(P, 2, dtype=i32, **z)
        b['depth'] = torch.empty(P, dtype=f32, **z)
        b['radii'] = torch.zeros(P, dtype=i32, **z)
        b['tile_count'] = torch.zeros(T * CS, dtype=i32, **z)
        b['tile_base'] = torch.empty(T + 1, dtype=i32, **z)
        b['tile_cursor'] = torch.empty(T * CS, dtype=i32, **z)
        b['status'] = torch.zeros(4, dtype=i32, **z)
        b['final_T'] = torch.empty(H, W, dtype=f32, **z)
        b['n_contrib'] = torch.empty(H, W, dtype=i32, **z)
        b['feat8'] = torch.empty(P, 8, dtype=f32, **z)
        b['out6'] = torch.empty(6, H, W, dtype=f32, **z)
        b['dL_dout6'] = torch.zeros(6, H, W, dtype=f32, **z)
        b['accum'] = torch.zeros(P, _capi.SPLAT_GRAD_STRIDE, dtype=f32, **z)
        b['ssim_maps'] = torch.empty(9, H, W, dtype=f32, **z)
        b['sums'] = torch.zeros(_capi.SPLAT_ITER_SUM_COPIES * _capi.SPLAT_ITER_SUMS, dtype=torch.float64, **z)
        b['d_cam'] = torch.zeros(16, dtype=f32, **z)
        b['pose_state'] = torch.zeros(_capi.SPLAT_POSE_STATE, dtype=f32, **z)
        self.max_2D_radius = track_max_radius
        # map gradients: ONE flat buffer (the all-reduce bucket of the view-sharded mapping step), viewed per parameter
        sizes = [params[k].numel() for k in PARAM_ORDER]
        self.grad_flat = torch.zeros(sum(sizes), dtype=f32, **z)
        self.grads, o = {}, 0
        for k, n in zip(PARAM_ORDER, sizes):
            self.grads[k] = self.grad_flat[o:o + n].view_as(params[k])
            o += n
        self.exp_avg = {k: torch.zeros_like(params[k].detach()) for k in PARAM_ORDER}
        self.exp_avg_sq = {k: torch.zeros_like(params[k].detach()) for k in PARAM_ORDER}
        self.map_step = 0
        self.pose_step = 0
        self.track_time_idx = None
        self.max_list_hint = 0          # longest tile list seen at the last check_overflow(); 0 = unknown
        self.tile_stride = 0            # > 0: bucketed lists (no scan / scatter pass), learnt by check_overflow()
        self.num_tiles = T
        self.allow_buckets = True
        self._alloc_lists(self.capacity)
        self._cam = self._make_cam(cam)
        self._frame_keep = None

    # ------------------------------------------------------------------ plumbing
    def _alloc_lists(self, capacity):
        self.capacity = int(capacity)
        self.buf['keys'] = torch.empty(self.capacity, dtype=torch.int64, device=self.dev)
        self.buf['point_list'] = torch.empty(self.capacity, dtype=torch.int32, device=self.dev)

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

    def _map_struct(self):
        p = self.params
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
        st.max_list_hint = self.max_list_hint
        st.tile_stride = self.tile_stride
        st.final_T, st.n_contrib, st.status = b['final_T'].data_ptr(), b['n_contrib'].data_ptr(), b['status'].data_ptr()
        ws.feat8, ws.out6, ws.dL_dout6, ws.accum = b['feat8'].data_ptr(), b['out6'].data_ptr(), b['dL_dout6'].data_ptr(), b['accum'].data_ptr()
        ws.ssim_maps = b['ssim_maps'].data_ptr() if with_ssim else None
        ws.sums = b['sums'].data_ptr()
        ws.max_2D_radius = self.max_2D_radius.data_ptr() if self.max_2D_radius is not None else None
        if with_map_grads:
            g = self.grads
            ws.d_means3D, ws.d_rgb_colors = g['means3D'].data_ptr(), g['rgb_colors'].data_ptr()
            ws.d_unnorm_rotations, ws.d_logit_opacities = g['unnorm_rotations'].data_ptr(), g['logit_opacities'].data_ptr()
            ws.d_log_scales = g['log_scales'].data_ptr()
        ws.d_cam = b['d_cam'].data_ptr()
        return ws

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    @staticmethod
    def loss_config(cfg, tracking, do_ba=False):
        c = _capi.SplatLossConfig()
        c.tracking = int(tracking)
        c.camera_grad = int(tracking or do_ba)
        c.gaussians_grad = int(not tracking)
        c.use_sil_for_loss, c.sil_thres = int(cfg['use_sil_for_loss']), float(cfg['sil_thres'])
        c.use_l1, c.ignore_outlier_depth_loss = int(cfg['use_l1']), int(cfg['ignore_outlier_depth_loss'])
        c.w_im, c.w_depth = float(cfg['loss_weights']['im']), float(cfg['loss_weights']['depth'])
        return c

    # ------------------------------------------------------------------ one iteration
    def loss_backward(self, curr_data, time_idx, cfg, tracking, map_grads=None, do_ba=False):
        """get_loss + backward.  Afterwards (stream order): ``self.grads`` (mapping) and
        ``self.buf['d_cam']`` = [dL/dq_raw(4), dL/dt_raw(3), loss]."""
        if map_grads is None:
            map_grads = not tracking
        fr = _capi.SplatFrameData()
        im, depth = curr_data['im'], curr_data['depth']
        w2c = curr_data['w2c']
        if not (im.is_contiguous() and depth.is_contiguous() and w2c.is_contiguous()):
            im, depth, w2c = im.contiguous(), depth.contiguous(), w2c.contiguous()
        fr.im, fr.depth, fr.w2c, fr.time_idx = im.data_ptr(), depth.data_ptr(), w2c.data_ptr(), int(time_idx)
        self._frame_keep = (im, depth, w2c)
        lc = self.loss_config(cfg, tracking, do_ba)
        ws = self._workspace(map_grads, with_ssim=not tracking)
        m = self._map_struct()
        with torch.cuda.device(self.dev):
            _capi.check(self.L.splat_iter_loss_backward(C.byref(self._cam), C.byref(m), C.byref(fr), C.byref(lc), C.byref(ws),
                                                        self._stream()), "splat_iter_loss_backward")

    def adam_map(self, lrs, beta1=0.9, beta2=0.999, eps=1e-15):
        """torch.optim.Adam(param_groups, lr=0.0, eps=1e-15).step() over the five Gaussian groups
        (/root/reference/scripts/splatam.py:160-166).  Bias corrections in double on the host, as torch forms them."""
        self.map_step += 1
        t = self.map_step
        bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
        o = _capi.SplatAdamMap()
        o.beta1, o.beta2, o.eps, o.bc2_sqrt = beta1, beta2, eps, math.sqrt(bc2)
        for k, name in enumerate(PARAM_ORDER):
            o.step_size[k] = lrs[name] / bc1
            o.grad[k] = self.grads[name].data_ptr()
            o.exp_avg[k] = self.exp_avg[name].data_ptr()
            o.exp_avg_sq[k] = self.exp_avg_sq[name].data_ptr()
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

    def tracking_iteration(self, curr_data, cfg):
        """Loop body of /root/reference/scripts/splatam.py:690-711 for frame ``begin_tracking`` named."""
        self.loss_backward(curr_data, self.track_time_idx, cfg, tracking=True)
        self.adam_pose(cfg['lrs']['cam_unnorm_rots'], cfg['lrs']['cam_trans'])

    def mapping_iteration(self, iter_data, iter_time_idx, cfg, bucket_allreduce=None):
        """Loop body of /root/reference/scripts/splatam.py:828-869 (without pruning / densification)."""
        self.loss_backward(iter_data, iter_time_idx, cfg, tracking=False)
        if bucket_allreduce is not None:
            bucket_allreduce(self.grad_flat)
        self.adam_map(cfg['lrs'])

    # ------------------------------------------------------------------ read-backs (host sync)
    def loss(self):
        return float(self.buf['d_cam'][7])

    def check_overflow(self, grow=True):
        """Lists are fixed-size; an iteration whose instances did not fit rendered truncated / empty lists and flagged
        it (stickily).  Call at frame end (one D2H read): returns True when an iteration since the last call has to
        be repeated.  Also learns the list statistics: from then on the per-tile lists are BUCKETED at 1.5x the longest
        list seen (the per-Gaussian kernel writes instances straight into their tile's bucket: no scan kernel, no
        scatter pass) and the long-list sort launch is skipped while lists stay short."""
        stat = self.buf['status'].tolist()
        sticky = float(self.buf['d_cam'][12]) != 0.0
        bad = sticky or stat[1] != 0 or stat[3] != 0 or (self.tile_stride == 0 and stat[0] > self.capacity)
        if bad:
            self.buf['d_cam'][12] = 0.0
            self.buf['status'].zero_()
            self.buf['tile_count'].zero_()
            self.buf['accum'].zero_()
            self.buf['sums'].zero_()
            self.max_list_hint = 0
            if grow:
                if self.tile_stride > 0:            # a bucket overflowed: back to exact lists, re-learn
                    self.tile_stride = 0
                if stat[0] > self.capacity:
                    self._alloc_lists(int(stat[0] * 1.5) + 65536)
            return True
        longest = int(stat[2])
        self.max_list_hint = longest            # short lists: sorted inside the composite, no sort launch
        if grow and self.allow_buckets and longest > 0:
            stride = max(256, (int(longest * 1.5) + 63) // 64 * 64)
            if stride != self.tile_stride and stride * self.num_tiles <= 64 * 1024 * 1024:
                if stride * self.num_tiles > self.capacity:
                    self._alloc_lists(stride * self.num_tiles)
                self.tile_stride = stride
        return False

    @property
    def seen(self):
        return self.buf['radii'] > 0

    def rendered(self):
        """(im[3,H,W], depth[1,H,W], silhouette[H,W], depth_sq[1,H,W]) of the last iteration."""
        o = self.buf['out6']
        return o[0:3], o[3:4], o[4], o[5:6]
