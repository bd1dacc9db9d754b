"""Host-side mirror of the callers on either side of the rasterizer boundary.

SplaTAM's tracking / mapping iteration is: world->camera transform of the
Gaussian centres, render-variable assembly, two rasterizer calls, masked
losses, ``backward()``, Adam.  The functions below keep the reference's names,
argument meaning and results so that ``bench.py`` and the tests can run the
reference's iteration on a box where /root/reference does not exist:

=============================================  =============================================
here                                           reference
=============================================  =============================================
``build_rotation``                             utils/slam_external.py:25-42
``setup_camera``                               utils/recon_helpers.py:4-27
``transform_to_frame``                         utils/slam_helpers.py:252-304
``transformed_params2rendervar``               utils/slam_helpers.py:124-139
``get_depth_and_silhouette``                   utils/slam_helpers.py:196-213
``transformed_params2depthplussilhouette``     utils/slam_helpers.py:234-249
``l1_loss_v1`` / ``calc_ssim``                 utils/slam_helpers.py:5-6 / utils/slam_external.py:54-97
``get_loss``                                   scripts/splatam.py:214-347
``initialize_optimizer``                       scripts/splatam.py:160-166
``tracking_iteration`` / ``mapping_iteration``  scripts/splatam.py:690-711 / 828-869 (loop bodies)
``get_pointcloud``                             scripts/splatam.py:67-116
``initialize_params`` / ``initialize_new_params``  scripts/splatam.py:119-157 / 350-376
``add_new_gaussians``                          scripts/splatam.py:378-420
``initialize_camera_pose``                     scripts/splatam.py:423-441
``remove_points`` / ``prune_gaussians``        utils/slam_external.py:139-188
``accumulate_mean2d_gradient`` / ``densify``   utils/slam_external.py:100-104 / 191-240 (gradient-based densification of the
                                               gaussian_splatting / post_splatam_opt configs; torch formulation only)
=============================================  =============================================

Differences are host-side only and do not change results: masked sums are
written as ``where(mask, x, 0).sum()`` instead of boolean-index compaction
(no host-visible sizes), and the tracking loop keeps its best-pose candidate
with a device-side select instead of ``if loss < current_min_loss`` (no
per-iteration host sync).  The rasterizer is always the HIP one
(``diff_gaussian_rasterization`` in this repository).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .rasterizer import GaussianRasterizationSettings as Camera
from .rasterizer import GaussianRasterizer as Renderer


# --------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------

def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """[B,4] quaternion (w,x,y,z), normalised here, -> [B,3,3]."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(dim=1)
    rows = (1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y))
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def quat_mult(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


def setup_camera(w, h, k, w2c, near=0.01, far=100, device="cuda"):
    """Settings tuple for an intrinsics matrix ``k`` and a world-to-camera ``w2c``;
    viewmatrix = w2c^T (kept as the non-contiguous transposed view the reference
    produces), projmatrix = (P w2c)^T with the reference's OpenGL-style P."""
    fx, fy, cx, cy = float(k[0][0]), float(k[1][1]), float(k[0][2]), float(k[1][2])
    w2c = torch.as_tensor(w2c, dtype=torch.float32, device=device)
    cam_center = torch.inverse(w2c)[:3, 3]
    view = w2c.unsqueeze(0).transpose(1, 2)
    P = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                      [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                      [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                      [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32, device=device)
    full_proj = view.bmm(P.unsqueeze(0).transpose(1, 2))
    return Camera(image_height=h, image_width=w, tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
                  bg=torch.zeros(3, dtype=torch.float32, device=device), scale_modifier=1.0,
                  viewmatrix=view, projmatrix=full_proj, sh_degree=0, campos=cam_center, prefiltered=False)


def transform_to_frame(params, time_idx, gaussians_grad, camera_grad):
    """Gaussian centres (and, for anisotropic maps, rotations) of the world-frame
    map expressed in the camera frame of ``time_idx``; the two flags choose which
    side of the product receives gradient."""
    cam_q = params['cam_unnorm_rots'][..., time_idx]
    cam_t = params['cam_trans'][..., time_idx]
    if not camera_grad:
        cam_q, cam_t = cam_q.detach(), cam_t.detach()
    cam_q = F.normalize(cam_q)
    pts, rots = params['means3D'], params['unnorm_rotations']
    if not gaussians_grad:
        pts, rots = pts.detach(), rots.detach()
    Rm = build_rotation(cam_q)[0]                       # [3,3]
    out = {'means3D': pts @ Rm.t() + cam_t.reshape(1, 3)}
    if params['log_scales'].shape[1] == 1:              # isotropic: orientation is irrelevant
        out['unnorm_rotations'] = rots
    else:
        out['unnorm_rotations'] = quat_mult(cam_q, F.normalize(rots))
    return out


def _scales3(params):
    ls = params['log_scales']
    return torch.exp(ls.expand(-1, 3) if ls.shape[1] == 1 else ls)


def transformed_params2rendervar(params, transformed_gaussians):
    return {
        'means3D': transformed_gaussians['means3D'],
        'colors_precomp': params['rgb_colors'],
        'rotations': F.normalize(transformed_gaussians['unnorm_rotations']),
        'opacities': torch.sigmoid(params['logit_opacities']),
        'scales': _scales3(params),
        # non-leaf zero tensor: the caller calls retain_grad() on it and reads the screen-space gradient
        'means2D': torch.zeros_like(params['means3D'], requires_grad=True) + 0,
    }


def get_depth_and_silhouette(pts_3D, w2c):
    """Per-Gaussian 'colours' of the second render: [z_cam, 1, z_cam^2]."""
    z = pts_3D @ w2c[2, :3] + w2c[2, 3]
    return torch.stack([z, torch.ones_like(z), z * z], dim=-1)


def transformed_params2depthplussilhouette(params, w2c, transformed_gaussians):
    return {
        'means3D': transformed_gaussians['means3D'],
        'colors_precomp': get_depth_and_silhouette(transformed_gaussians['means3D'], w2c),
        'rotations': F.normalize(transformed_gaussians['unnorm_rotations']),
        'opacities': torch.sigmoid(params['logit_opacities']),
        'scales': _scales3(params),
        'means2D': torch.zeros_like(params['means3D'], requires_grad=True) + 0,
    }


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------

def l1_loss_v1(x, y):
    return (x - y).abs().mean()


_ssim_windows: dict = {}


def _ssim_window(channel, size, device, dtype):
    key = (channel, size, str(device), dtype)
    w = _ssim_windows.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(i - size // 2) ** 2 / (2 * 1.5 ** 2)) for i in range(size)])
        g = (g / g.sum()).unsqueeze(1)
        w = (g @ g.t()).float().expand(channel, 1, size, size).contiguous().to(device=device, dtype=dtype)
        _ssim_windows[key] = w
    return w


def calc_ssim(img1, img2, window_size=11, size_average=True):
    """Mean SSIM with an 11x11 sigma-1.5 Gaussian window, zero padding, per-channel."""
    ch = img1.size(-3)
    win = _ssim_window(ch, window_size, img1.device, img1.dtype)
    pad = window_size // 2

    def blur(t):
        return F.conv2d(t, win, padding=pad, groups=ch)
    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = blur(img1 * img1) - mu1_sq
    s2 = blur(img2 * img2) - mu2_sq
    s12 = blur(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return ssim_map.mean() if size_average else ssim_map.mean(-1).mean(-1).mean(-1)


def _masked_sum(x, mask):
    return torch.where(mask, x, torch.zeros((), dtype=x.dtype, device=x.device)).sum()


def get_loss(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss,
             sil_thres, use_l1, ignore_outlier_depth_loss, tracking=False, mapping=False, do_ba=False):
    """One SplaTAM objective evaluation: RGB render + depth/silhouette render + masked losses.
    Returns (loss, variables, weighted_losses) like the reference."""
    if tracking:
        tg = transform_to_frame(params, iter_time_idx, gaussians_grad=False, camera_grad=True)
    elif mapping and do_ba:
        tg = transform_to_frame(params, iter_time_idx, gaussians_grad=True, camera_grad=True)
    else:
        tg = transform_to_frame(params, iter_time_idx, gaussians_grad=True, camera_grad=False)

    rendervar = transformed_params2rendervar(params, tg)
    depth_sil_rendervar = transformed_params2depthplussilhouette(params, curr_data['w2c'], tg)

    rendervar['means2D'].retain_grad()
    im, radius, _ = Renderer(raster_settings=curr_data['cam'])(**rendervar)
    variables['means2D'] = rendervar['means2D']          # densification reads the colour pass' screen gradient

    depth_sil, _, _ = Renderer(raster_settings=curr_data['cam'])(**depth_sil_rendervar)
    depth = depth_sil[0:1]
    silhouette = depth_sil[1]
    presence_sil_mask = silhouette > sil_thres
    uncertainty = (depth_sil[2:3] - depth ** 2).detach()

    gt_depth = curr_data['depth']
    nan_mask = (~torch.isnan(depth)) & (~torch.isnan(uncertainty))
    if ignore_outlier_depth_loss:
        depth_error = torch.abs(gt_depth - depth) * (gt_depth > 0)
        mask = (depth_error < 10 * depth_error.median()) & (gt_depth > 0)
    else:
        mask = gt_depth > 0
    mask = mask & nan_mask
    if tracking and use_sil_for_loss:
        mask = mask & presence_sil_mask
    mask = mask.detach()

    losses = {}
    if use_l1:
        d_abs = torch.abs(gt_depth - depth)
        if tracking:
            losses['depth'] = _masked_sum(d_abs, mask)
        else:
            losses['depth'] = _masked_sum(d_abs, mask) / mask.sum()

    if tracking and (use_sil_for_loss or ignore_outlier_depth_loss):
        losses['im'] = _masked_sum(torch.abs(curr_data['im'] - im), mask.expand(3, -1, -1))
    elif tracking:
        losses['im'] = torch.abs(curr_data['im'] - im).sum()
    else:
        losses['im'] = 0.8 * l1_loss_v1(im, curr_data['im']) + 0.2 * (1.0 - calc_ssim(im, curr_data['im']))

    weighted_losses = {k: v * loss_weights[k] for k, v in losses.items()}
    loss = sum(weighted_losses.values())

    seen = radius > 0
    variables['max_2D_radius'] = torch.where(seen, torch.max(radius.to(variables['max_2D_radius'].dtype),
                                                             variables['max_2D_radius']), variables['max_2D_radius'])
    variables['seen'] = seen
    weighted_losses['loss'] = loss
    return loss, variables, weighted_losses


# --------------------------------------------------------------------------
# optimiser + loop bodies
# --------------------------------------------------------------------------

def initialize_optimizer(params, lrs_dict, tracking):
    groups = [{'params': [v], 'name': k, 'lr': lrs_dict[k]} for k, v in params.items()]
    if tracking:
        return torch.optim.Adam(groups)
    return torch.optim.Adam(groups, lr=0.0, eps=1e-15)


REPLICA_TRACKING = dict(
    use_sil_for_loss=True, sil_thres=0.99, use_l1=True, ignore_outlier_depth_loss=False,
    loss_weights=dict(im=0.5, depth=1.0),
    lrs=dict(means3D=0.0, rgb_colors=0.0, unnorm_rotations=0.0, logit_opacities=0.0, log_scales=0.0,
             cam_unnorm_rots=0.0004, cam_trans=0.002))
REPLICA_MAPPING = dict(
    use_sil_for_loss=False, sil_thres=0.5, use_l1=True, ignore_outlier_depth_loss=False,
    loss_weights=dict(im=0.5, depth=1.0),
    lrs=dict(means3D=0.0001, rgb_colors=0.0025, unnorm_rotations=0.001, logit_opacities=0.05, log_scales=0.001,
             cam_unnorm_rots=0.0, cam_trans=0.0))
"""Values of /root/reference/configs/replica/splatam.py:60-100."""


class TrackingState:
    """Best-candidate bookkeeping of the tracking loop kept on the device."""

    def __init__(self, params, time_idx):
        self.time_idx = time_idx
        self.min_loss = torch.full((), 1e20, device=params['cam_trans'].device)
        self.best_rot = params['cam_unnorm_rots'][..., time_idx].detach().clone()
        self.best_tran = params['cam_trans'][..., time_idx].detach().clone()
        self.last_losses = None       # the weighted losses of the last iteration, as the loop reads them (scripts/splatam.py:728)

    def update(self, params, loss):
        with torch.no_grad():
            better = loss.detach() < self.min_loss
            self.min_loss = torch.where(better, loss.detach(), self.min_loss)
            self.best_rot = torch.where(better, params['cam_unnorm_rots'][..., self.time_idx], self.best_rot)
            self.best_tran = torch.where(better, params['cam_trans'][..., self.time_idx], self.best_tran)

    def commit(self, params):
        with torch.no_grad():
            params['cam_unnorm_rots'][..., self.time_idx] = self.best_rot
            params['cam_trans'][..., self.time_idx] = self.best_tran


def tracking_iteration(params, curr_data, variables, time_idx, optimizer, state: TrackingState, cfg=REPLICA_TRACKING):
    loss, variables, losses = get_loss(params, curr_data, variables, time_idx, cfg['loss_weights'],
                                       cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                       cfg['ignore_outlier_depth_loss'], tracking=True)
    loss.backward()
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    state.update(params, loss)
    state.last_losses = losses
    return loss, variables


def mapping_iteration(params, iter_data, variables, iter_time_idx, optimizer, cfg=REPLICA_MAPPING):
    loss, variables, losses = get_loss(params, iter_data, variables, iter_time_idx, cfg['loss_weights'],
                                       cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                       cfg['ignore_outlier_depth_loss'], mapping=True)
    loss.backward()
    with torch.no_grad():
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss, variables


# --------------------------------------------------------------------------
# map growth and maintenance (the torch formulation; FusedEngine does the same edits in place on the device)
# --------------------------------------------------------------------------

GAUSSIAN_KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales')


def get_pointcloud(color, depth, intrinsics, w2c, transform_pts=True, mask=None, compute_mean_sq_dist=False,
                   mean_sq_dist_method="projective"):
    """Back-projection of an RGB-D frame: rows [x, y, z, r, g, b] in pixel order (world frame when
    ``transform_pts``), optionally with the projective scale estimate (z / mean focal)^2 per point."""
    H, W = color.shape[1], color.shape[2]
    fx, fy, cx, cy = intrinsics[0][0], intrinsics[1][1], intrinsics[0][2], intrinsics[1][2]
    dev = depth.device
    u = torch.arange(W, device=dev, dtype=torch.float32)
    v = torch.arange(H, device=dev, dtype=torch.float32)
    xx = ((u - cx) / fx).unsqueeze(0).expand(H, W).reshape(-1)
    yy = ((v - cy) / fy).unsqueeze(1).expand(H, W).reshape(-1)
    z = depth[0].reshape(-1)
    pts = torch.stack((xx * z, yy * z, z), dim=-1)
    if transform_pts:
        c2w = torch.inverse(w2c)
        pts = pts @ c2w[:3, :3].t() + c2w[:3, 3]
    msd = None
    if compute_mean_sq_dist:
        if mean_sq_dist_method != "projective":
            raise ValueError(f"Unknown mean_sq_dist_method {mean_sq_dist_method}")
        msd = (z / ((fx + fy) / 2)) ** 2
    cloud = torch.cat((pts, color.permute(1, 2, 0).reshape(-1, 3)), dim=-1)
    if mask is not None:
        cloud = cloud[mask]
        if msd is not None:
            msd = msd[mask]
    return (cloud, msd) if compute_mean_sq_dist else cloud


def initialize_new_params(new_pt_cld, mean3_sq_dist, gaussian_distribution):
    """Parameter rows of freshly back-projected Gaussians: identity rotation, logit opacity 0,
    log scale log(sqrt(mean3_sq_dist)) in one (isotropic) or three (anisotropic) columns."""
    if gaussian_distribution not in ("isotropic", "anisotropic"):
        raise ValueError(f"Unknown gaussian_distribution {gaussian_distribution}")
    n, dev = new_pt_cld.shape[0], new_pt_cld.device
    rots = torch.zeros(n, 4, device=dev)
    rots[:, 0] = 1.0
    ls = torch.log(torch.sqrt(mean3_sq_dist)).unsqueeze(-1)
    raw = {'means3D': new_pt_cld[:, :3], 'rgb_colors': new_pt_cld[:, 3:6], 'unnorm_rotations': rots,
           'logit_opacities': torch.zeros(n, 1, device=dev),
           'log_scales': ls if gaussian_distribution == "isotropic" else ls.expand(-1, 3)}
    return {k: torch.nn.Parameter(t.float().contiguous().requires_grad_(True)) for k, t in raw.items()}


def initialize_params(init_pt_cld, num_frames, mean3_sq_dist, gaussian_distribution):
    """First-frame map + an identity camera trajectory of ``num_frames`` poses + the per-Gaussian variables."""
    params = initialize_new_params(init_pt_cld, mean3_sq_dist, gaussian_distribution)
    dev = init_pt_cld.device
    rots = torch.zeros(1, 4, num_frames, device=dev)
    rots[:, 0, :] = 1.0
    params['cam_unnorm_rots'] = torch.nn.Parameter(rots.requires_grad_(True))
    params['cam_trans'] = torch.nn.Parameter(torch.zeros(1, 3, num_frames, device=dev).requires_grad_(True))
    n = params['means3D'].shape[0]
    variables = {k: torch.zeros(n, device=dev) for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'timestep')}
    return params, variables


def add_new_gaussians(params, variables, curr_data, sil_thres, time_idx, mean_sq_dist_method, gaussian_distribution):
    """Densification of frame ``time_idx``: pixels the map does not explain yet (low silhouette, or rendered
    depth behind the measurement by more than 50x the median depth error) become new Gaussians."""
    with torch.no_grad():
        tg = transform_to_frame(params, time_idx, gaussians_grad=False, camera_grad=False)
        dv = transformed_params2depthplussilhouette(params, curr_data['w2c'], tg)
        depth_sil, _, _ = Renderer(raster_settings=curr_data['cam'])(**{k: v.detach() for k, v in dv.items()})
    return _add_from_render(params, variables, curr_data, depth_sil, sil_thres, time_idx, mean_sq_dist_method, gaussian_distribution)


def _add_from_render(params, variables, curr_data, depth_sil, sil_thres, time_idx, mean_sq_dist_method, gaussian_distribution):
    silhouette, render_depth = depth_sil[1], depth_sil[0]
    gt_depth = curr_data['depth'][0]
    depth_error = torch.abs(gt_depth - render_depth) * (gt_depth > 0)
    behind = (render_depth > gt_depth) & (depth_error > 50 * depth_error.median())
    non_presence = ((silhouette < sil_thres) | behind).reshape(-1)
    if int(non_presence.sum()) > 0:
        q = F.normalize(params['cam_unnorm_rots'][..., time_idx].detach())
        w2c = torch.eye(4, device=q.device)
        w2c[:3, :3] = build_rotation(q)[0]
        w2c[:3, 3] = params['cam_trans'][0, :, time_idx].detach()
        pick = non_presence & (gt_depth > 0).reshape(-1)
        cloud, msd = get_pointcloud(curr_data['im'], curr_data['depth'], curr_data['intrinsics'], w2c, mask=pick,
                                    compute_mean_sq_dist=True, mean_sq_dist_method=mean_sq_dist_method)
        new = initialize_new_params(cloud, msd, gaussian_distribution)
        for k, v in new.items():
            params[k] = torch.nn.Parameter(torch.cat((params[k].detach(), v.detach()), dim=0).requires_grad_(True))
        n = params['means3D'].shape[0]
        dev = params['means3D'].device
        for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius'):
            variables[k] = torch.zeros(n, device=dev)
        variables['timestep'] = torch.cat((variables['timestep'], torch.full((cloud.shape[0],), float(time_idx), device=dev)))
    return params, variables


def initialize_camera_pose(params, curr_time_idx, forward_prop):
    """Pose of the new frame before tracking: constant-velocity extrapolation of the last two poses, or a copy of
    the previous one."""
    with torch.no_grad():
        q, t = params['cam_unnorm_rots'], params['cam_trans']
        if curr_time_idx > 1 and forward_prop:
            q1 = F.normalize(q[..., curr_time_idx - 1].detach())
            q2 = F.normalize(q[..., curr_time_idx - 2].detach())
            q[..., curr_time_idx] = F.normalize(q1 + (q1 - q2))
            t1, t2 = t[..., curr_time_idx - 1].detach(), t[..., curr_time_idx - 2].detach()
            t[..., curr_time_idx] = t1 + (t1 - t2)
        else:
            q[..., curr_time_idx] = q[..., curr_time_idx - 1].detach()
            t[..., curr_time_idx] = t[..., curr_time_idx - 1].detach()
    return params


def remove_points(to_remove, params, variables, optimizer):
    """Drop the flagged Gaussians from the parameters, the optimizer's moments and the per-Gaussian variables."""
    keep = ~to_remove
    for k in GAUSSIAN_KEYS:
        group = next(g for g in optimizer.param_groups if g['name'] == k)
        old = group['params'][0]
        state = optimizer.state.pop(old, None)
        new = torch.nn.Parameter(old.detach()[keep].requires_grad_(True))
        if state:
            state['exp_avg'] = state['exp_avg'][keep]
            state['exp_avg_sq'] = state['exp_avg_sq'][keep]
            optimizer.state[new] = state
        group['params'][0] = new
        params[k] = new
    for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius', 'timestep'):
        if k in variables:
            variables[k] = variables[k][keep]
    return params, variables


def prune_gaussians(params, variables, optimizer, iter, prune_dict):
    """Opacity / size pruning on the schedule of ``prune_dict`` (+ the optional opacity reset)."""
    if iter <= prune_dict['stop_after']:
        if iter >= prune_dict['start_after'] and iter % prune_dict['prune_every'] == 0:
            thr = prune_dict['final_removal_opacity_threshold'] if iter == prune_dict['stop_after'] \
                else prune_dict['removal_opacity_threshold']
            to_remove = (torch.sigmoid(params['logit_opacities']) < thr).squeeze(-1)
            if iter >= prune_dict['remove_big_after']:
                to_remove = to_remove | (torch.exp(params['log_scales']).max(dim=1).values > 0.1 * variables['scene_radius'])
            params, variables = remove_points(to_remove, params, variables, optimizer)
        if iter > 0 and iter % prune_dict['reset_opacities_every'] == 0 and prune_dict['reset_opacities']:
            params = _reset_opacities(params, optimizer)
    return params, variables


def accumulate_mean2d_gradient(variables):
    """Running sum of the screen-space gradient norm of the Gaussians seen by the last colour render."""
    seen = variables['seen']
    variables['means2D_gradient_accum'][seen] += torch.norm(variables['means2D'].grad[seen, :2], dim=-1)
    variables['denom'][seen] += 1
    return variables


def _cat_rows(new_rows, params, optimizer):
    """Append rows to the Gaussian parameters (and zero moments to the optimizer's state, where it has one)."""
    for k, v in new_rows.items():
        group = next(g for g in optimizer.param_groups if g['name'] == k)
        old = group['params'][0]
        state = optimizer.state.pop(old, None)
        new = torch.nn.Parameter(torch.cat((old.detach(), v.detach()), dim=0).requires_grad_(True))
        if state:
            state['exp_avg'] = torch.cat((state['exp_avg'], torch.zeros_like(v)), dim=0)
            state['exp_avg_sq'] = torch.cat((state['exp_avg_sq'], torch.zeros_like(v)), dim=0)
            optimizer.state[new] = state
        group['params'][0] = new
        params[k] = new
    return params


def densify(params, variables, optimizer, iter, densify_dict):
    """Gradient-based densification (3D Gaussian Splatting's): clone the small Gaussians with a large accumulated
    screen-space gradient, split the large ones into ``num_to_split_into`` samples of themselves, then prune by opacity
    and size; optional opacity reset.  Draws from the global torch RNG exactly where the reference does."""
    if iter > densify_dict['stop_after']:
        return params, variables
    variables = accumulate_mean2d_gradient(variables)
    thr = densify_dict['grad_thresh']
    if iter >= densify_dict['start_after'] and iter % densify_dict['densify_every'] == 0:
        grads = variables['means2D_gradient_accum'] / variables['denom']
        grads[grads.isnan()] = 0.0
        small = torch.exp(params['log_scales']).max(dim=1).values <= 0.01 * variables['scene_radius']
        to_clone = (grads >= thr) & small
        params = _cat_rows({k: params[k].detach()[to_clone] for k in GAUSSIAN_KEYS}, params, optimizer)
        n_pts = params['means3D'].shape[0]
        dev = params['means3D'].device
        padded = torch.zeros(n_pts, device=dev)
        padded[:grads.shape[0]] = grads
        to_split = (padded >= thr) & (torch.exp(params['log_scales']).max(dim=1).values > 0.01 * variables['scene_radius'])
        n = densify_dict['num_to_split_into']
        rows = {k: params[k].detach()[to_split].repeat(n, 1) for k in GAUSSIAN_KEYS}
        stds = torch.exp(params['log_scales'].detach())[to_split].repeat(n, 3)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = build_rotation(params['unnorm_rotations'].detach()[to_split]).repeat(n, 1, 1)
        rows['means3D'] = rows['means3D'] + torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1)
        rows['log_scales'] = torch.log(torch.exp(rows['log_scales']) / (0.8 * n))
        params = _cat_rows(rows, params, optimizer)
        n_pts = params['means3D'].shape[0]
        for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius'):
            variables[k] = torch.zeros(n_pts, device=dev)
        to_remove = torch.cat((to_split, torch.zeros(n * int(to_split.sum()), dtype=torch.bool, device=dev)))
        params, variables = remove_points(to_remove, params, variables, optimizer)
        op_thr = densify_dict['final_removal_opacity_threshold'] if iter == densify_dict['stop_after'] \
            else densify_dict['removal_opacity_threshold']
        to_remove = (torch.sigmoid(params['logit_opacities']) < op_thr).squeeze(-1)
        if iter >= densify_dict['remove_big_after']:
            to_remove = to_remove | (torch.exp(params['log_scales']).max(dim=1).values > 0.1 * variables['scene_radius'])
        params, variables = remove_points(to_remove, params, variables, optimizer)
    if iter > 0 and iter % densify_dict['reset_opacities_every'] == 0 and densify_dict['reset_opacities']:
        params = _reset_opacities(params, optimizer)
    return params, variables


def _reset_opacities(params, optimizer):
    group = next(g for g in optimizer.param_groups if g['name'] == 'logit_opacities')
    old = group['params'][0]
    state = optimizer.state.pop(old, None)
    v = torch.full_like(old.detach(), math.log(0.01 / (1 - 0.01)))
    new = torch.nn.Parameter(v.requires_grad_(True))
    if state is not None:
        state['exp_avg'] = torch.zeros_like(v)
        state['exp_avg_sq'] = torch.zeros_like(v)
        optimizer.state[new] = state
    group['params'][0] = new
    params['logit_opacities'] = new
    return params


REPLICA_PRUNE = dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
                     final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500)
"""/root/reference/configs/replica/splatam.py:102-111."""


# --------------------------------------------------------------------------
# synthetic RGB-D scene (datasets are not available offline)
# --------------------------------------------------------------------------

def synthetic_params(n, width, height, fx, fy, cx, cy, num_frames=2, seed=0, device="cuda", anisotropic=False, region=None):
    """Seeded SplaTAM-like map: one Gaussian per random sub-pixel, back-projected at
    z~U[1,4] with the reference's projective scale init (scripts/splatam.py:76-99,120-157).
    ``region`` = (u0, v0, u1, v1) as fractions of the image: every Gaussian projects inside that window (the clustered
    stress workload: per-tile lists far longer than LDS)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, generator=g, dtype=torch.float64) * width - 0.5
    v = torch.rand(n, generator=g, dtype=torch.float64) * height - 0.5
    if region is not None:
        u0, v0, u1, v1 = region
        u = (u + 0.5) * (u1 - u0) + u0 * width - 0.5
        v = (v + 0.5) * (v1 - v0) + v0 * height - 0.5
    z = 1.0 + 3.0 * torch.rand(n, generator=g, dtype=torch.float64)
    means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], dim=-1)
    log_s = torch.log(z / ((fx + fy) / 2)) + 0.3 * torch.randn(n, generator=g, dtype=torch.float64)
    if anisotropic:
        log_scales = log_s[:, None] + 0.3 * torch.randn(n, 3, generator=g, dtype=torch.float64)
        rots = torch.randn(n, 4, generator=g, dtype=torch.float64)
    else:
        log_scales = log_s[:, None]
        rots = torch.zeros(n, 4, dtype=torch.float64)
        rots[:, 0] = 1.0
    logit_op = 2.0 + torch.randn(n, 1, generator=g, dtype=torch.float64)
    rgb = torch.rand(n, 3, generator=g, dtype=torch.float64)
    cam_rots = torch.zeros(1, 4, num_frames)
    cam_rots[:, 0, :] = 1.0
    raw = dict(means3D=means, rgb_colors=rgb, unnorm_rotations=rots, logit_opacities=logit_op, log_scales=log_scales,
               cam_unnorm_rots=cam_rots, cam_trans=torch.zeros(1, 3, num_frames))
    params = {k: torch.nn.Parameter(t.to(device=device, dtype=torch.float32).contiguous().requires_grad_(True))
              for k, t in raw.items()}
    variables = {'max_2D_radius': torch.zeros(n, device=device), 'means2D_gradient_accum': torch.zeros(n, device=device),
                 'denom': torch.zeros(n, device=device), 'timestep': torch.zeros(n, device=device)}
    return params, variables


def synthetic_frame(params, cam, w2c_first, time_idx, rot_deg=0.5, trans_m=0.01):
    """'Ground-truth' RGB-D frame = render of the map from the pose of ``time_idx``
    perturbed by rot_deg / trans_m, so that tracking has a real gradient."""
    with torch.no_grad():
        dev = params['means3D'].device
        ang = math.radians(rot_deg)
        q = torch.tensor([[math.cos(ang / 2), 0.0, math.sin(ang / 2), 0.0]], device=dev)
        t = torch.tensor([[trans_m, -trans_m / 2, trans_m / 2]], device=dev)
        fake = dict(params)
        rots = params['cam_unnorm_rots'].detach().clone()
        trans = params['cam_trans'].detach().clone()
        rots[..., time_idx] = q
        trans[..., time_idx] = t
        fake['cam_unnorm_rots'], fake['cam_trans'] = rots, trans
        tg = transform_to_frame(fake, time_idx, gaussians_grad=False, camera_grad=False)
        rv = transformed_params2rendervar(fake, tg)
        im, _, _ = Renderer(raster_settings=cam)(**{k: v.detach() for k, v in rv.items()})
        dv = transformed_params2depthplussilhouette(fake, w2c_first, tg)
        ds, _, _ = Renderer(raster_settings=cam)(**{k: v.detach() for k, v in dv.items()})
        sil = ds[1:2]
        depth = torch.where(sil > 0.5, ds[0:1] / sil.clamp_min(1e-6), torch.zeros_like(sil))
    return im.contiguous(), depth.contiguous()
