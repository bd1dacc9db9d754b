"""The per-frame SplaTAM loop on the HIP engine: what ``rgbd_slam`` does between loading a frame and
storing a keyframe (/root/reference/scripts/splatam.py:654-905), without the reference's I/O, logging and evaluation.

=============================================  =============================================
here                                           reference
=============================================  =============================================
``keyframe_selection_overlap``                 utils/keyframe_selection.py:44-103 (+ its get_pointcloud :10-41)
``replica_config``                             configs/replica/splatam.py (the values the loop reads)
``rgbd_slam``                                  scripts/splatam.py:455-905 (frame loop: pose initialisation, tracking with
                                               best-candidate bookkeeping and the depth-loss retry, densification,
                                               keyframe selection, mapping with pruning, keyframe list)
``save_params`` / ``load_params``              utils/common_utils.py:25-52 (``params.npz``)
``SyntheticRGBDSequence``                      stands in for datasets/gradslam_datasets/* (no datasets offline)
=============================================  =============================================

``engine="fused"`` runs every iteration through ``FusedEngine`` (C ABI ``splat_iter_*`` / ``splat_map_*``: ~8 kernel
launches per iteration, map edits in place on the device); ``engine="dropin"`` runs the reference-shaped PyTorch loop
(``splatam_amd.slam``) on the drop-in rasterizer.  Both follow the reference's control flow line by line, including the
places where it is surprising: pruning happens between ``backward()`` and ``optimizer.step()``, and ``remove_points``
re-creates the parameters, so the iterations on the pruning schedule take NO Adam step
(scripts/splatam.py:857-868, utils/slam_external.py:139-162).
"""
from __future__ import annotations

import copy
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import slam


# --------------------------------------------------------------------------
# keyframe selection
# --------------------------------------------------------------------------

def _sampled_cloud(z, intrinsics, w2c, sampled_indices):
    """World-frame points of the sampled pixels (``z``: their depths); points that coincide after rounding to 1e-4 (duplicated
    samples, or the camera origin) are dropped, all copies of them, as the reference's unique/isin construction does."""
    fx, fy, cx, cy = intrinsics[0][0], intrinsics[1][1], intrinsics[0][2], intrinsics[1][2]
    v, u = sampled_indices[:, 0], sampled_indices[:, 1]
    pts_cam = torch.stack(((u - cx) / fx * z, (v - cy) / fy * z, z), dim=-1)
    c2w = torch.inverse(w2c)
    pts = pts_cam @ c2w[:3, :3].t() + c2w[:3, 3]
    # |round(p, 4)| as integers: round(p, 4) is round(p * 1e4) / 1e4, and distinct integers below 2^24 give distinct float32 quotients, so
    # two points coincide after rounding exactly when their integer triples do.  Three 21-bit fields make ONE int64 key per point:
    # a 1-d unique instead of unique(dim=0) (3 ms -> 0.1 ms for the 1600 samples on the host)
    q = torch.abs(torch.round(pts * 1e4)).to(torch.int64)
    if pts.dtype == torch.float32 and int(q.max()) < (1 << 21):
        key = torch.cat(((q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2], torch.zeros(1, dtype=torch.int64, device=pts.device)))
        _, inverse, counts = key.unique(return_inverse=True, return_counts=True)
    else:
        key = torch.cat((torch.abs(torch.round(pts, decimals=4)), torch.zeros(1, 3, device=pts.device, dtype=pts.dtype)), dim=0)
        _, inverse, counts = key.unique(dim=0, return_inverse=True, return_counts=True)
    keep = (counts[inverse] == 1)[:pts.shape[0]]
    return pts[keep]


def keyframe_selection_overlap(gt_depth, w2c, intrinsics, keyframe_list, k, pixels=1600):
    """Indices (into ``keyframe_list``) of up to ``k`` keyframes that see part of the current frame: 1600 valid-depth
    pixels are back-projected and re-projected into every keyframe (one batched product over all keyframes); keyframes with
    a non-zero share of points inside the image (20 px border) are kept, ordered by that share, then shuffled."""
    # The frame stays where it is: the valid-pixel list (one nonzero: row-major order, as torch.where gives it) and the gather of the
    # 1600 sampled depths run on the depth image's device, and only those samples travel to the host -- copying the frame and scanning
    # it there cost ~5 ms per 1200x680 frame.  The few thousand points against a few dozen keyframes that follow are host-side work
    # (on the GPU the unique(dim=0) / tolist chain cost 86 ms per frame in synchronisations and tiny launches).  Random numbers are
    # drawn where the reference draws them: torch.randint on the CPU generator, then numpy's permutation.
    H, W = gt_depth.shape[1], gt_depth.shape[2]
    valid = torch.nonzero(gt_depth[0] > 0)
    pick = torch.randint(valid.shape[0], (pixels,))
    sampled = valid[pick.to(valid.device)]
    z = gt_depth[0, sampled[:, 0], sampled[:, 1]].cpu()
    sampled, w2c, intrinsics = sampled.cpu(), w2c.cpu(), intrinsics.cpu()
    # (a few thousand floats per operation: on a 256-core host every one of these CPU operators otherwise wakes the whole thread pool)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _select_from_samples(z, sampled, w2c, intrinsics, keyframe_list, k, H, W)
    finally:
        torch.set_num_threads(threads)


def _select_from_samples(z, sampled, w2c, intrinsics, keyframe_list, k, H, W):
    pts = _sampled_cloud(z, intrinsics, w2c, sampled)
    if len(keyframe_list) == 0:
        return []
    est = torch.stack([kf['est_w2c'] for kf in keyframe_list]).cpu()               # [K,4,4]
    cam = torch.einsum('kij,nj->kni', est[:, :3, :3], pts) + est[:, None, :3, 3]     # [K,N,3]
    proj = torch.einsum('ij,knj->kni', intrinsics.to(cam.dtype), cam)
    zc = proj[..., 2] + 1e-5
    px, py = proj[..., 0] / zc, proj[..., 1] / zc
    edge = 20
    inside = (px < W - edge) & (px > edge) & (py < H - edge) & (py > edge) & (zc > 0)
    share = (inside.sum(dim=1) / max(pts.shape[0], 1)).tolist()
    order = sorted(range(len(keyframe_list)), key=lambda i: share[i], reverse=True)      # stable, like the reference's sorted()
    chosen = [i for i in order if share[i] > 0.0]
    return list(np.random.permutation(np.array(chosen))[:k])


# --------------------------------------------------------------------------
# configuration, I/O
# --------------------------------------------------------------------------

def replica_config(tracking_iters=40, mapping_iters=60, map_every=1, keyframe_every=5, mapping_window_size=24):
    """The entries of /root/reference/configs/replica/splatam.py that the frame loop reads."""
    return dict(
        seed=0, map_every=map_every, keyframe_every=keyframe_every, mapping_window_size=mapping_window_size,
        scene_radius_depth_ratio=3, mean_sq_dist_method="projective", gaussian_distribution="isotropic",
        tracking=dict(use_gt_poses=False, forward_prop=True, num_iters=tracking_iters, use_sil_for_loss=True, sil_thres=0.99,
                      use_l1=True, ignore_outlier_depth_loss=False, use_depth_loss_thres=False, depth_loss_thres=100000,
                      loss_weights=dict(im=0.5, depth=1.0), lrs=dict(slam.REPLICA_TRACKING['lrs'])),
        mapping=dict(num_iters=mapping_iters, add_new_gaussians=True, sil_thres=0.5, use_l1=True, use_sil_for_loss=False,
                     ignore_outlier_depth_loss=False, loss_weights=dict(im=0.5, depth=1.0), lrs=dict(slam.REPLICA_MAPPING['lrs']),
                     prune_gaussians=True, pruning_dict=dict(slam.REPLICA_PRUNE), use_gaussian_splatting_densification=False))


def save_params(output_params, output_dir, time_idx=None):
    """``params.npz`` (or ``params<time_idx>.npz``): one array per entry of the params dict."""
    os.makedirs(output_dir, exist_ok=True)
    name = "params.npz" if time_idx is None else f"params{time_idx}.npz"
    path = os.path.join(output_dir, name)
    np.savez(path, **{k: (v.detach().cpu().contiguous().numpy() if isinstance(v, torch.Tensor) else v) for k, v in output_params.items()})
    return path


def load_params(path, device="cuda"):
    """Inverse of ``save_params`` the way the reference's checkpoint loader reads it (scripts/splatam.py:611-613)."""
    raw = dict(np.load(path, allow_pickle=True))
    return {k: torch.tensor(v).to(device).float().requires_grad_(True) for k, v in raw.items()}


class SyntheticRGBDSequence:
    """RGB-D frames rendered from a seeded synthetic scene along a smooth trajectory (no datasets offline).  Items look like
    the reference's gradslam datasets: ``(color[H,W,3] in 0..255, depth[H,W,1], intrinsics[4,4], pose[4,4])`` with the pose
    (camera-to-world) relative to the first frame.  The scene is a smooth, opaque, textured surface (``n_gaussians`` splats
    on z = z0 + ripples, colours a sum of low- and mid-frequency waves) so that frame-to-model tracking is well posed --
    the random per-pixel depth layers of the kernel workloads (bench.build_scene) are not."""

    def __init__(self, n_gaussians, width, height, fx, fy, cx, cy, num_frames, seed=0, device="cuda", step_m=0.01, step_deg=0.3):
        self.W, self.H, self.num_frames, self.device = width, height, num_frames, device
        self.k = torch.tensor([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32, device=device)
        g = torch.Generator().manual_seed(seed)
        n = int(n_gaussians)
        m = 0.15                                            # the surface extends past the first view: later frames see new parts
        u = (torch.rand(n, generator=g, dtype=torch.float64) * (1 + 2 * m) - m) * width - 0.5
        v = (torch.rand(n, generator=g, dtype=torch.float64) * (1 + 2 * m) - m) * height - 0.5
        su, sv = u / width * 2 * math.pi, v / height * 2 * math.pi
        z = 2.5 + 0.35 * torch.sin(1.3 * su + 0.4) * torch.cos(0.9 * sv) + 0.15 * torch.sin(2.7 * sv + 1.0)
        means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], dim=-1)
        rgb = torch.stack([0.5 + 0.25 * torch.sin(3 * su) * torch.cos(2 * sv) + 0.2 * torch.sin(17 * su + 0.3) * torch.sin(13 * sv),
                           0.5 + 0.25 * torch.cos(2 * su + 1.0) * torch.sin(3 * sv) + 0.2 * torch.sin(11 * su) * torch.cos(19 * sv + 0.7),
                           0.5 + 0.25 * torch.sin(4 * su + 2.0) + 0.2 * torch.cos(15 * su + 9 * sv)], dim=-1).clamp(0.02, 0.98)
        spacing = math.sqrt((1 + 2 * m) ** 2 * width * height / n)        # mean distance between neighbouring splats in pixels
        log_scales = torch.log(1.1 * spacing * z / ((fx + fy) / 2))[:, None]
        rots = torch.zeros(n, 4, dtype=torch.float64)
        rots[:, 0] = 1.0
        cam_rots = torch.zeros(1, 4, num_frames)
        cam_trans = torch.zeros(1, 3, num_frames)
        for t in range(num_frames):
            ang = math.radians(step_deg * t)
            cam_rots[0, :, t] = torch.tensor([math.cos(ang / 2), 0.0, math.sin(ang / 2), 0.0])
            cam_trans[0, :, t] = torch.tensor([step_m * t, -0.3 * step_m * t, 0.5 * step_m * t])
        raw = dict(means3D=means, rgb_colors=rgb, unnorm_rotations=rots, logit_opacities=torch.full((n, 1), 4.0, dtype=torch.float64),
                   log_scales=log_scales, cam_unnorm_rots=cam_rots, cam_trans=cam_trans)
        self._scene = {k: t.to(device=device, dtype=torch.float32).contiguous() for k, t in raw.items()}
        self._cam = slam.setup_camera(width, height, self.k[:3, :3].cpu().numpy(), np.eye(4, dtype=np.float32), device=device)
        self._w2c0 = torch.eye(4, device=device)

    def __len__(self):
        return self.num_frames

    def preload(self):
        """Render every frame once and keep it (a real dataset hands over loaded images; without this every ``dataset[t]`` renders)."""
        self._cache = {t: self._item(t) for t in range(self.num_frames)}
        return self

    def gt_w2c(self, t):
        q = F.normalize(self._scene['cam_unnorm_rots'][..., t].detach())
        w2c = torch.eye(4, device=self.device)
        w2c[:3, :3] = slam.build_rotation(q)[0]
        w2c[:3, 3] = self._scene['cam_trans'][0, :, t].detach()
        return w2c

    def __getitem__(self, t):
        cache = getattr(self, "_cache", None)
        return cache[t] if cache is not None and t in cache else self._item(t)

    def _item(self, t):
        im, depth = self._render(t)
        pose = torch.inverse(self.gt_w2c(t))
        return (im.permute(1, 2, 0) * 255.0).contiguous(), depth.permute(1, 2, 0).contiguous(), self.k, pose

    def _render(self, t):
        with torch.no_grad():
            p = self._scene
            tg = slam.transform_to_frame(p, t, gaussians_grad=False, camera_grad=False)
            rv = slam.transformed_params2rendervar(p, tg)
            im, _, _ = slam.Renderer(raster_settings=self._cam)(**{k: v.detach() for k, v in rv.items()})
            dv = slam.transformed_params2depthplussilhouette(p, self._w2c0, tg)
            ds, _, _ = slam.Renderer(raster_settings=self._cam)(**{k: v.detach() for k, v in dv.items()})
            sil = ds[1:2]
            depth = torch.where(sil > 0.5, ds[0:1] / sil.clamp_min(1e-6), torch.zeros_like(sil))
        return im.clamp(0, 1).contiguous(), depth.contiguous()


# --------------------------------------------------------------------------
# the frame loop
# --------------------------------------------------------------------------

def _est_w2c(params, time_idx):
    q = F.normalize(params['cam_unnorm_rots'][..., time_idx].detach())
    w2c = torch.eye(4, device=q.device)
    w2c[:3, :3] = slam.build_rotation(q)[0]
    w2c[:3, 3] = params['cam_trans'][0, :, time_idx].detach()
    return w2c


def initialize_first_timestep(dataset, num_frames, scene_radius_depth_ratio, mean_sq_dist_method, gaussian_distribution, device="cuda"):
    """Map, variables, intrinsics, first-frame world-to-camera and camera from frame 0."""
    color, depth, intrinsics, pose = dataset[0]
    color = color.permute(2, 0, 1) / 255
    depth = depth.permute(2, 0, 1)
    intrinsics = intrinsics[:3, :3]
    w2c = torch.linalg.inv(pose)
    cam = slam.setup_camera(color.shape[2], color.shape[1], intrinsics.cpu().numpy(), w2c.detach().cpu().numpy(), device=device)
    mask = (depth > 0).reshape(-1)
    cloud, msd = slam.get_pointcloud(color, depth, intrinsics, w2c, mask=mask, compute_mean_sq_dist=True,
                                     mean_sq_dist_method=mean_sq_dist_method)
    params, variables = slam.initialize_params(cloud, num_frames, msd, gaussian_distribution)
    variables['scene_radius'] = torch.max(depth) / scene_radius_depth_ratio
    return params, variables, intrinsics, w2c, cam


class _PhaseTimer:
    """Wall time per named phase of the frame being processed (a device synchronisation on either side: ~10 per frame)."""

    def __init__(self, dev):
        self.dev, self.frame, self._open = dev, {}, []

    def _sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def __call__(self, name):
        self._name = name
        return self

    def __enter__(self):
        self._sync()
        self._open.append((self._name, time.perf_counter()))

    def __exit__(self, *exc):
        self._sync()
        name, t0 = self._open.pop()
        self.frame[name] = self.frame.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)

    def next_frame(self):
        done, self.frame = self.frame, {}
        return {k: round(v, 3) for k, v in done.items()}


def rgbd_slam(dataset, config, engine="fused", num_frames=None, gaussian_capacity=None, verbose=False):
    """Runs the SplaTAM frame loop over ``dataset``; returns ``(params, variables, stats)`` with
    ``stats = {keyframe_time_indices, tracking_iters, mapping_iters, tracking_s, mapping_s, mapping_loop_s, num_gaussians,
    redone_iterations, phase_ms}`` (``mapping_loop_s``: the iterations alone, where the reference's own mapping timer runs,
    scripts/splatam.py:825-891; ``mapping_s`` also holds densification, keyframe selection and list re-learning; ``phase_ms``: one
    dict per frame -- tracking, add_new_gaussians, keyframe_selection, relearn_lists, mapping_iterations, prune, keyframe_store).

    ``engine``: "fused" (FusedEngine: every iteration one C call, the map edited in place on the device), "dropin" (the
    reference-shaped PyTorch loop of splatam_amd.slam on the drop-in rasterizer), or "plugin": the SAME reference-shaped loop -- its
    ``add_new_gaussians`` / ``prune_gaussians`` / ``remove_points`` re-create every tensor, its statements are get_loss -> backward ->
    prune -> step -- with ``splatam_amd.plugin`` installed into the module that holds it, i.e. what
    /root/reference/scripts/splatam.py:654-905 executes after ``plugin.install``; "plugin_map_edits": the same with
    ``plugin.install(map_edits=True)`` (the two map edits are adapters too: the engine owns the map and edits it in place).

    Per-tile list overflow (fused): a flagged iteration and every iteration after it take NO Adam step on the device
    (include/splat_hip.h, d_cam[12]); at the end of a phase ``check_overflow()`` says how many, the lists are re-sized and exactly
    that many iterations are run again -- the pose / the map never saw a bad gradient, so nothing has to be restored.

    With ``torch.distributed`` initialised (one process per GPU, splatam_amd.dist.init_from_env) the loop runs on every rank
    over the REPLICATED map:
      * tracking: the frame's tile rows are sharded over the ranks (every rank takes the same Adam step); with the outlier-rejecting
        loss every rank tracks the whole frame and rank 0's pose is then broadcast (7 floats), because the float
        atomics of the backward composite leave the replicas' poses different in the last bits and the densification that follows
        thresholds a render at that pose;
      * mapping: every iteration draws ``world`` keyframe views instead of one (the same random stream on every rank, the
        reference's one-random-keyframe-per-iteration rule /root/reference/scripts/splatam.py:831-845 applied ``world`` times),
        rank r renders the r-th, ONE gradient all-reduce (mean) follows, and every rank takes the identical Adam step;
      * after every edit of the map (densification, pruning) the row counts of the replicas are compared (all-reduce of min / max)."""
    from . import dist as sdist
    if engine not in ("fused", "dropin", "plugin", "plugin_map_edits"):
        raise ValueError(engine)
    fused = engine == "fused"
    plugged = engine in ("plugin", "plugin_map_edits")
    world, rank = sdist.world_size(), sdist.get_rank()
    if plugged and world > 1:
        raise NotImplementedError(f"engine='{engine}' runs the reference's single-process loop")
    num_frames = len(dataset) if num_frames is None else min(num_frames, len(dataset))
    tcfg, mcfg = config['tracking'], config['mapping']
    if mcfg.get('use_gaussian_splatting_densification') and not fused:
        # the reference's own densify cannot run inside its SLAM loop either: it never extends variables['timestep'], and the
        # remove_points that follows indexes it with the longer mask (utils/slam_external.py:206-227, 139-162).  The fused engine
        # carries `timestep` along with the duplicated rows (FusedEngine.densify).
        raise NotImplementedError("gradient-based densification inside the frame loop needs engine='fused'")
    dist_kind = config.get('gaussian_distribution', 'isotropic')
    eng = None
    if fused:
        # first frame on the device: an empty capacity-managed map + one append of every valid-depth pixel
        # (splat_map_add_new_gaussians, SPLAT_ADD_VALID_DEPTH) = get_pointcloud + initialize_params of the reference
        from .fused import FusedEngine
        color0, depth0, intr0, pose0 = dataset[0]
        color0 = (color0.permute(2, 0, 1) / 255).contiguous()
        depth0 = depth0.permute(2, 0, 1).contiguous()
        dev = depth0.device
        intrinsics = intr0[:3, :3]
        first_frame_w2c = torch.linalg.inv(pose0).to(dev).float().contiguous()
        H, W = color0.shape[1], color0.shape[2]
        cam = slam.setup_camera(W, H, intrinsics.cpu().numpy(), first_frame_w2c.detach().cpu().numpy(), device=dev)
        cols = 1 if dist_kind == "isotropic" else 3
        if dist_kind not in ("isotropic", "anisotropic"):
            raise ValueError(f"Unknown gaussian_distribution {dist_kind}")
        rots = torch.zeros(1, 4, num_frames, device=dev)
        rots[:, 0, :] = 1.0
        z = lambda *shape: torch.nn.Parameter(torch.zeros(*shape, device=dev))      # noqa: E731
        params = {'means3D': z(0, 3), 'rgb_colors': z(0, 3), 'unnorm_rotations': z(0, 4), 'logit_opacities': z(0, 1),
                  'log_scales': z(0, cols), 'cam_unnorm_rots': torch.nn.Parameter(rots), 'cam_trans': z(1, 3, num_frames)}
        variables = {k: torch.zeros(0, device=dev) for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'timestep')}
        variables['scene_radius'] = torch.max(depth0) / config['scene_radius_depth_ratio']
        cap = gaussian_capacity or int(H * W * 2.5) + 65536
        eng = FusedEngine(params, cam, gaussian_capacity=cap, variables=variables)
        eng.keep_map_grads = False      # (a mapping iteration's gradients are discarded after its step: /root/reference/scripts/splatam.py:860-861)
        if config['mean_sq_dist_method'] != "projective":
            raise ValueError(f"Unknown mean_sq_dist_method {config['mean_sq_dist_method']}")
        eng.add_valid_depth_points(color0, depth0, intrinsics, first_frame_w2c)
        scene_radius = variables['scene_radius']
    else:
        params, variables, intrinsics, first_frame_w2c, cam = initialize_first_timestep(
            dataset, num_frames, config['scene_radius_depth_ratio'], config['mean_sq_dist_method'], dist_kind,
            device=dataset[0][1].device)
        dev = params['means3D'].device
        first_frame_w2c = first_frame_w2c.to(dev).float().contiguous()
    keyframe_list, keyframe_time_indices = [], []
    stats = dict(tracking_iters=0, mapping_iters=0, tracking_s=0.0, mapping_s=0.0, mapping_loop_s=0.0, redone_iterations=0,
                 num_gaussians=[], phase_ms=[], frame_s=[], decisions=[])
    phase = _PhaseTimer(dev)
    installed = None
    if plugged:
        from . import plugin
        installed = plugin.install(slam, map_edits=engine == "plugin_map_edits")
    try:
        for time_idx in range(num_frames):
            t_frame = time.perf_counter()
            color, depth, _, gt_pose = dataset[time_idx]
            color = (color.permute(2, 0, 1) / 255).contiguous()
            depth = depth.permute(2, 0, 1).contiguous()
            curr_data = {'cam': cam, 'im': color, 'depth': depth, 'id': time_idx, 'intrinsics': intrinsics, 'w2c': first_frame_w2c}
            # what the loop DECIDED on this frame, engine independent (host integers only; tests/loop_trace.py derives the same table
            # from a recording of the reference's own rgbd_slam)
            decided = dict(time_idx=time_idx, tracking_iters=0, rows_after_add=None, selected=None, views=[], prunes=[], rows_end=None,
                           keyframe=False)
            stats['decisions'].append(decided)
            if time_idx > 0:
                slam.initialize_camera_pose(params, time_idx, forward_prop=tcfg['forward_prop'])

            # ---------------- tracking (scripts/splatam.py:676-744)
            with phase("tracking"):
                t0 = time.perf_counter()
                if time_idx > 0 and not tcfg['use_gt_poses']:
                    if plugged:
                        n_track, variables = _track_frame_statements(params, variables, curr_data, time_idx, tcfg)
                    else:
                        n_track = _track_frame(params, variables, curr_data, time_idx, tcfg, eng, stats)
                    stats['tracking_iters'] += n_track
                    decided['tracking_iters'] = n_track
                    sdist.broadcast_pose(params, time_idx)              # replicas: one pose for the map edits that follow
                elif time_idx > 0:
                    with torch.no_grad():
                        rel = torch.linalg.inv(gt_pose).to(dev)
                        params['cam_unnorm_rots'][..., time_idx] = _matrix_to_quaternion(rel[:3, :3])
                        params['cam_trans'][..., time_idx] = rel[:3, 3]
            stats['tracking_s'] += time.perf_counter() - t0

            # ---------------- densification + keyframe mapping (scripts/splatam.py:768-891)
            if time_idx == 0 or (time_idx + 1) % config['map_every'] == 0:
                t0 = time.perf_counter()
                if mcfg['add_new_gaussians'] and time_idx > 0:
                    with phase("add_new_gaussians"):
                        if fused:
                            eng.add_new_gaussians(curr_data, mcfg['sil_thres'], time_idx, config['mean_sq_dist_method'], dist_kind)
                        else:
                            params, variables = slam.add_new_gaussians(params, variables, curr_data, mcfg['sil_thres'], time_idx,
                                                                       config['mean_sq_dist_method'], dist_kind)
                    decided['rows_after_add'] = int(params['means3D'].shape[0])
                    sdist.assert_replicated_count(int(params['means3D'].shape[0]), f"add_new_gaussians (frame {time_idx})", dev)
                with phase("keyframe_selection"), torch.no_grad():
                    curr_w2c = _est_w2c(params, time_idx)
                    selected = keyframe_selection_overlap(depth, curr_w2c, intrinsics.to(dev), keyframe_list[:-1],
                                                          config['mapping_window_size'] - 2)
                    if len(keyframe_list) > 0:
                        selected.append(len(keyframe_list) - 1)
                    selected.append(-1)
                    decided['selected'] = [int(x) for x in selected[:-1 - (1 if len(keyframe_list) > 0 else 0)]]
                if fused and not eng.lists_known():
                    with phase("relearn_lists"):
                        eng.relearn_lists(curr_data, time_idx)
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                t_loop = time.perf_counter()                        # the reference's mapping timer starts here (scripts/splatam.py:825)
                t_prune0 = phase.frame.get("prune", 0.0)
                with phase("mapping_iterations"):
                    _map_frame(params, variables, curr_data, time_idx, selected, keyframe_list, mcfg, eng,
                               scene_radius if fused else None, stats, phase, decided)
                # (the prune phase is timed inside the loop: report the iterations without it)
                phase.frame["mapping_iterations"] -= phase.frame.get("prune", 0.0) - t_prune0
                stats['mapping_iters'] += mcfg['num_iters']
                stats['mapping_s'] += time.perf_counter() - t0
                stats['mapping_loop_s'] += time.perf_counter() - t_loop

            # ---------------- keyframe list (scripts/splatam.py:893-905)
            if time_idx == 0 or (time_idx + 1) % config['keyframe_every'] == 0 or time_idx == num_frames - 2:
                with phase("keyframe_store"), torch.no_grad():
                    keyframe_list.append({'id': time_idx, 'est_w2c': _est_w2c(params, time_idx), 'color': color, 'depth': depth})
                    keyframe_time_indices.append(time_idx)
                    decided['keyframe'] = True
            decided['rows_end'] = int(params['means3D'].shape[0])
            stats['num_gaussians'].append(int(params['means3D'].shape[0]))
            stats['phase_ms'].append(phase.next_frame())
            stats['frame_s'].append(time.perf_counter() - t_frame)
            if verbose:
                print(f"frame {time_idx}: {stats['num_gaussians'][-1]} Gaussians, keyframes {keyframe_time_indices}", flush=True)
    finally:
        if installed is not None:
            from . import plugin
            stats['plugin'] = plugin.session_stats()
            installed.uninstall()
    stats['keyframe_time_indices'] = keyframe_time_indices
    return params, variables, stats


def _settle_lists(eng, dev, stats, what):
    """End of a phase on the fused engine: were iterations flagged (on any rank)?  Returns how many took no Adam step -- the lists
    have been re-sized; the caller runs that many again."""
    from . import dist as sdist
    if not sdist.any_rank(eng.check_overflow(), dev):
        return 0
    lost = sdist.max_int(eng.skipped_iterations, dev)
    stats['redone_iterations'] += lost
    if stats['redone_iterations'] > 100000:
        raise RuntimeError(f"{what}: the per-tile lists keep overflowing")
    return max(lost, 1)


def _track_frame(params, variables, curr_data, time_idx, tcfg, eng, stats):
    """Tracking iterations of one frame incl. the reference's doubling of the budget when the depth loss stays above
    ``depth_loss_thres`` (scripts/splatam.py:727-735).  Returns the number of iterations run."""
    num_iters = tcfg['num_iters']
    if eng is not None:
        eng.begin_tracking(time_idx)
    else:
        optimizer = slam.initialize_optimizer(params, tcfg['lrs'], tracking=True)
        state = slam.TrackingState(params, time_idx)
    # several ranks (replicated map and pose): each composites its band of tile rows, the partial sums are all-reduced, every rank
    # takes the same Adam step on the pose (FusedEngine.tracking_iteration); the outlier-rejecting loss needs the whole render
    from . import dist as sdist
    world, rank = sdist.world_size(), sdist.get_rank()
    dev = params['cam_trans'].device
    shard = (rank, world) if (eng is not None and world > 1 and not tcfg['ignore_outlier_depth_loss']) else None
    it, todo, doubled, rounds = 0, num_iters, False, 0
    while todo:
        for _ in range(todo):
            if eng is not None:
                eng.tracking_iteration(curr_data, tcfg, shard=shard, allreduce_sums=sdist.all_reduce_sum_flat)
            else:
                loss, _ = slam.tracking_iteration(params, curr_data, variables, time_idx, optimizer, state, tcfg)
        it += todo
        todo = 0
        if eng is not None:
            lost = _settle_lists(eng, dev, stats, f"frame {time_idx} tracking")
            if lost:
                rounds += 1
                if rounds > 3:
                    raise RuntimeError(f"frame {time_idx}: the per-tile lists overflowed three times in a row")
                eng.pose_step = max(eng.pose_step - lost, 0)     # (the skipped steps never happened: their bias corrections are taken again)
                it -= lost
                todo = lost
                continue
        if it == num_iters and tcfg.get('use_depth_loss_thres', False) and not doubled:
            # the value the reference compares is the LAST iteration's weighted depth loss (scripts/splatam.py:728): no extra evaluation
            depth_loss = float(eng.buf['d_cam'][14]) if eng is not None else float(state.last_losses['depth'].detach())
            if depth_loss >= tcfg['depth_loss_thres']:
                doubled, todo = True, num_iters
    if eng is not None:
        eng.end_tracking()
    else:
        state.commit(params)
    return it


def _track_frame_statements(params, variables, curr_data, time_idx, tcfg):
    """The tracking phase in the reference's own statements (scripts/splatam.py:680-744: optimizer per frame, get_loss -> backward ->
    step -> zero_grad, the host-side `if loss < current_min_loss`, the doubled budget) -- every name resolved in ``slam`` at call
    time, so that ``plugin.install(slam)`` is what runs."""
    optimizer = slam.initialize_optimizer(params, tcfg['lrs'], tracking=True)
    candidate_rot = params['cam_unnorm_rots'][..., time_idx].detach().clone()
    candidate_tran = params['cam_trans'][..., time_idx].detach().clone()
    current_min_loss = float(1e20)
    it, extended, budget = 0, False, tcfg['num_iters']
    while True:
        loss, variables, losses = slam.get_loss(params, curr_data, variables, time_idx, tcfg['loss_weights'], tcfg['use_sil_for_loss'],
                                                tcfg['sil_thres'], tcfg['use_l1'], tcfg['ignore_outlier_depth_loss'], tracking=True)
        loss.backward()
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        with torch.no_grad():
            if loss < current_min_loss:
                current_min_loss = loss
                candidate_rot = params['cam_unnorm_rots'][..., time_idx].detach().clone()
                candidate_tran = params['cam_trans'][..., time_idx].detach().clone()
        it += 1
        if it == budget:
            use_thres = tcfg.get('use_depth_loss_thres', False)
            if use_thres and losses['depth'] < tcfg['depth_loss_thres']:
                break
            if use_thres and not extended:
                extended, budget = True, 2 * budget
            else:
                break
    with torch.no_grad():
        params['cam_unnorm_rots'][..., time_idx] = candidate_rot
        params['cam_trans'][..., time_idx] = candidate_tran
    return it, variables


def _map_frame(params, variables, curr_data, time_idx, selected, keyframe_list, mcfg, eng, scene_radius, stats, phase, decided):
    """The mapping iterations of one frame over the selected keyframes + the current frame.  With ``world`` ranks every
    iteration draws ``world`` views from the SAME random stream on every rank; this rank renders its own one, the gradients are
    averaged by one all-reduce, every rank takes the same Adam step and the same pruning decisions.
    (Gradient-based densification accumulates the screen-space gradient of every iteration it sees, a flagged one included: its
    statistic is then one truncated-list sample short -- the map itself never moves on a flagged iteration.)"""
    from . import dist as sdist
    world, rank = sdist.world_size(), sdist.get_rank()
    cam, intrinsics, w2c0 = curr_data['cam'], curr_data['intrinsics'], curr_data['w2c']
    prune, pd = mcfg['prune_gaussians'], mcfg['pruning_dict']
    dev = params['means3D'].device
    if world > 1 and mcfg.get('use_gaussian_splatting_densification'):
        # every rank renders another view: means2D_gradient_accum / denom / max_2D_radius would differ between the replicas and
        # so would the clone / split selection -- the replicated-map invariant of the multi-rank loop does not hold
        raise NotImplementedError("gradient-based densification is not supported in the multi-rank frame loop")
    bucket = None                   # drop-in path: flat gradient bucket, re-made when an edit changes the number of rows
    if eng is not None:
        eng.reset_map_optimizer()
    else:
        optimizer = slam.initialize_optimizer(params, mcfg['lrs'], tracking=False)

    def iteration(it):
        nonlocal bucket
        if world == 1:
            sel = selected[np.random.randint(0, len(selected))]
        else:
            sel = selected[int(np.random.randint(0, len(selected), size=world)[rank])]
        if sel == -1:
            iter_time_idx, iter_color, iter_depth = time_idx, curr_data['im'], curr_data['depth']
        else:
            kf = keyframe_list[sel]
            iter_time_idx, iter_color, iter_depth = kf['id'], kf['color'], kf['depth']
        iter_data = {'cam': cam, 'im': iter_color, 'depth': iter_depth, 'id': iter_time_idx, 'intrinsics': intrinsics, 'w2c': w2c0}
        decided['views'].append(int(iter_time_idx))
        # the reference prunes between backward() and step(); remove_points re-creates the parameters without their
        # gradients, so an iteration on the pruning schedule takes no Adam step
        on_schedule = prune and it <= pd['stop_after'] and it >= pd['start_after'] and it % pd['prune_every'] == 0
        # prune_gaussians also resets the opacities on its own schedule (utils/slam_external.py:186-190)
        resets = prune and it <= pd['stop_after'] and it > 0 and it % pd['reset_opacities_every'] == 0 and pd['reset_opacities']
        if eng is not None and world == 1 and not on_schedule and not resets and not mcfg.get('use_gaussian_splatting_densification'):
            # nothing between backward() and step() in this iteration: one call, the Adam step rides in the last kernel
            eng.mapping_iteration(iter_data, iter_time_idx, mcfg)
        elif eng is not None:
            eng.loss_backward(iter_data, iter_time_idx, mcfg, tracking=False)
            if world > 1 and not on_schedule:                   # (an iteration on the pruning schedule takes no Adam step)
                # (with every rank's capacity flag in the bucket's header: a rank whose lists overflowed stops ALL replicas' steps)
                eng.exchange_gradients(sdist.all_reduce_mean_flat)
            edited = False
            densifying = bool(mcfg.get('use_gaussian_splatting_densification'))
            if densifying and it <= mcfg['densify_dict']['stop_after']:
                # the colour pass' screen-space gradient is accumulated from THIS iteration's workspace (lists, radii, features
                # indexed by the rows the render saw): before any row is removed
                eng.accumulate_mean2d_gradient()
            if prune and (on_schedule or resets):
                rows = eng.P
                with phase("prune"):
                    edited = bool(eng.prune_gaussians(it, pd, scene_radius))
                if on_schedule:
                    decided['prunes'].append((it, rows, eng.P))
            if densifying:                                               # scripts/splatam.py:864-867
                dd = mcfg['densify_dict']
                dens_sched = it <= dd['stop_after'] and it >= dd['start_after'] and it % dd['densify_every'] == 0
                edited = bool(eng.densify(it, dd, scene_radius, accumulate=False)) or edited
                on_schedule = on_schedule or dens_sched                   # re-created parameters carry no gradient: no Adam step
            if edited:
                sdist.assert_replicated_count(eng.P, f"map edit (frame {time_idx}, iteration {it})", dev)
                if not eng.lists_known():
                    with phase("relearn_lists"):
                        eng.relearn_lists(curr_data, time_idx)
            if not on_schedule:
                eng.adam_map(mcfg['lrs'])
        else:
            loss, _, _ = slam.get_loss(params, iter_data, variables, iter_time_idx, mcfg['loss_weights'], mcfg['use_sil_for_loss'],
                                       mcfg['sil_thres'], mcfg['use_l1'], mcfg['ignore_outlier_depth_loss'], mapping=True)
            loss.backward()
            if world > 1 and not on_schedule:
                if bucket is None or bucket.sizes != [params[k].numel() for k in bucket.keys]:
                    bucket = sdist.GradBucket(params)
                bucket.all_reduce_mean(params)
            with torch.no_grad():
                if prune:
                    n_before = params['means3D'].shape[0]
                    if on_schedule or resets:
                        with phase("prune"):
                            slam.prune_gaussians(params, variables, optimizer, it, pd)
                    else:                                       # (the reference calls it every iteration; off schedule it does nothing)
                        slam.prune_gaussians(params, variables, optimizer, it, pd)
                    if on_schedule:
                        decided['prunes'].append((it, n_before, int(params['means3D'].shape[0])))
                    if params['means3D'].shape[0] != n_before:
                        sdist.assert_replicated_count(int(params['means3D'].shape[0]), f"prune_gaussians (frame {time_idx}, iteration {it})", dev)
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)

    it, todo, rounds = 0, mcfg['num_iters'], 0
    while todo:
        for _ in range(todo):
            iteration(it)
            it += 1
        todo = 0
        if eng is not None:
            todo = _settle_lists(eng, dev, stats, f"frame {time_idx} mapping")
            if todo:
                rounds += 1
                if rounds > 3:
                    raise RuntimeError(f"frame {time_idx}: the per-tile lists overflowed three times in a row")
                eng.map_step = max(eng.map_step - todo, 0)


def _matrix_to_quaternion(R):
    """Rotation matrix -> (w, x, y, z), the branch-free form the reference imports from pytorch3d-style helpers
    (utils/slam_external.py: matrix_to_quaternion) for ``use_gt_poses``."""
    m = R.reshape(3, 3).double()
    t = torch.stack([1 + m[0, 0] + m[1, 1] + m[2, 2], 1 + m[0, 0] - m[1, 1] - m[2, 2],
                     1 - m[0, 0] + m[1, 1] - m[2, 2], 1 - m[0, 0] - m[1, 1] + m[2, 2]]).clamp_min(0).sqrt()
    cand = torch.stack([
        torch.stack([t[0] ** 2, m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]]),
        torch.stack([m[2, 1] - m[1, 2], t[1] ** 2, m[1, 0] + m[0, 1], m[0, 2] + m[2, 0]]),
        torch.stack([m[0, 2] - m[2, 0], m[1, 0] + m[0, 1], t[2] ** 2, m[2, 1] + m[1, 2]]),
        torch.stack([m[1, 0] - m[0, 1], m[2, 0] + m[0, 2], m[2, 1] + m[1, 2], t[3] ** 2])])
    best = int(torch.argmax(t))
    q = cand[best] / (2.0 * t[best].clamp_min(0.1))
    return q.float().reshape(1, 4)
