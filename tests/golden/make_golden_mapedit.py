"""Generates tests/golden/mapedit_reference.npz by running the REFERENCE's own map-editing code on CPU in this
container: get_pointcloud / initialize_params / initialize_new_params / add_new_gaussians / initialize_camera_pose
(sources cut out of /root/reference/scripts/splatam.py) and prune_gaussians / remove_points
(/root/reference/utils/slam_external.py, imported as is, driving a real torch.optim.Adam).

``.cuda()`` / ``device="cuda"`` are redirected to the CPU; the reference's rasterizer call inside add_new_gaussians is
bound to a stub that returns the depth/silhouette image stored in the fixture, so the fixture pins everything the
function does AROUND the render (masks, median rule, back-projection, parameter initialisation, variable resets).

Run:  python tests/golden/make_golden_mapedit.py     (needs /root/reference; not needed on the GPU box)
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self
for _name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "arange"):
    _orig = getattr(torch, _name)

    def _wrap(*a, _o=_orig, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return _o(*a, **k)
    setattr(torch, _name, _wrap)

from utils import slam_external as ref_ext  # noqa: E402
from utils import slam_helpers as ref_h  # noqa: E402


class _StubRenderer:
    """Stands in for the un-vendored rasterizer: returns the stored depth/silhouette image."""
    image = None

    def __init__(self, raster_settings=None):
        pass

    def __call__(self, **kw):
        return _StubRenderer.image, None, None


def reference_functions():
    src = open(os.path.join(REF, "scripts", "splatam.py")).read()
    tree = ast.parse(src)
    wanted = ("get_pointcloud", "initialize_params", "initialize_new_params", "add_new_gaussians", "initialize_camera_pose")
    ns = dict(torch=torch, np=np, F=F, Renderer=_StubRenderer, transform_to_frame=ref_h.transform_to_frame,
              transformed_params2depthplussilhouette=ref_h.transformed_params2depthplussilhouette,
              build_rotation=ref_ext.build_rotation)
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in wanted:
            exec(compile(ast.get_source_segment(src, n), "reference_" + n.name, "exec"), ns)
    return ns


def make_frame(W, H, f, seed, nan_pixel=False):
    g = torch.Generator().manual_seed(seed)
    im = torch.rand(3, H, W, generator=g)
    depth = 1.0 + 3.0 * torch.rand(1, H, W, generator=g)
    depth[0, :3, :] = 0.0                                     # invalid-depth rows
    depth[0, torch.rand(H, W, generator=g) < 0.05] = 0.0
    sil = 0.3 + 0.7 * torch.rand(H, W, generator=g)
    sil[H // 2:, : W // 2] = 0.9995                           # a well-covered region
    rd = depth[0] + 0.002 * torch.randn(H, W, generator=g)
    far = torch.rand(H, W, generator=g) < 0.03                # rendered surface far behind the measurement
    rd = torch.where(far, rd + 1.5, rd)
    near = torch.rand(H, W, generator=g) < 0.03               # in front: never selected by the depth rule
    rd = torch.where(near, rd - 0.8, rd)
    if nan_pixel:
        rd[5, 7] = float('nan')
    depth_sil = torch.stack([rd, sil, rd * rd + 0.01], dim=0)
    k = torch.tensor([[f, 0.0, W / 2 - 0.5], [0.0, f * 1.02, H / 2 - 0.5], [0.0, 0.0, 1.0]])
    return im, depth, depth_sil, k


def case_add(name, ns, W, H, f, n0, dist, seed, out, nan_pixel=False, nothing=False):
    g = torch.Generator().manual_seed(seed + 7)
    im, depth, depth_sil, k = make_frame(W, H, f, seed, nan_pixel)
    if nothing:                                               # fully explained frame: the map is left alone
        depth_sil[1] = 1.0
        depth_sil[0] = depth[0]
    T = 4
    cols = 1 if dist == "isotropic" else 3
    params = {
        'means3D': torch.randn(n0, 3, generator=g), 'rgb_colors': torch.rand(n0, 3, generator=g),
        'unnorm_rotations': torch.randn(n0, 4, generator=g), 'logit_opacities': torch.randn(n0, 1, generator=g),
        'log_scales': -3.0 + 0.3 * torch.randn(n0, cols, generator=g),
        'cam_unnorm_rots': torch.tensor([1.0, 0, 0, 0]).reshape(1, 4, 1).repeat(1, 1, T) + 0.05 * torch.randn(1, 4, T, generator=g),
        'cam_trans': 0.1 * torch.randn(1, 3, T, generator=g),
    }
    variables = {'max_2D_radius': torch.rand(n0, generator=g) * 9, 'means2D_gradient_accum': torch.rand(n0, generator=g),
                 'denom': torch.rand(n0, generator=g) * 5, 'timestep': torch.zeros(n0)}
    time_idx = 2
    for kk, v in params.items():
        out[f"{name}/in/param/{kk}"] = v.numpy().copy()
    for kk, v in variables.items():
        out[f"{name}/in/var/{kk}"] = v.numpy().copy()
    out[f"{name}/in/im"], out[f"{name}/in/depth"], out[f"{name}/in/depth_sil"] = im.numpy(), depth.numpy(), depth_sil.numpy()
    out[f"{name}/in/intrinsics"] = k.numpy()
    out[f"{name}/in/meta"] = np.array([W, H, time_idx, 0.5, 1.0 if dist == "isotropic" else 0.0])
    P = {kk: torch.nn.Parameter(v.clone()) for kk, v in params.items()}
    V = {kk: v.clone() for kk, v in variables.items()}
    _StubRenderer.image = depth_sil
    curr = {'cam': None, 'im': im, 'depth': depth, 'id': time_idx, 'intrinsics': k, 'w2c': torch.eye(4)}
    P, V = ns['add_new_gaussians'](P, V, curr, 0.5, time_idx, "projective", dist)
    for kk, v in P.items():
        out[f"{name}/out/param/{kk}"] = v.detach().numpy()
    for kk, v in V.items():
        out[f"{name}/out/var/{kk}"] = v.detach().numpy()
    gt = depth[0]
    err = torch.abs(gt - depth_sil[0]) * (gt > 0)
    out[f"{name}/median"] = np.array(err.median().item())


def case_init(name, ns, W, H, f, dist, seed, out):
    im, depth, _, k = make_frame(W, H, f, seed)
    th = 0.2
    w2c = torch.tensor([[np.cos(th), 0, np.sin(th), 0.1], [0, 1, 0, -0.05], [-np.sin(th), 0, np.cos(th), 0.2], [0, 0, 0, 1]],
                       dtype=torch.float32)
    mask = (depth > 0).reshape(-1)
    cloud, msd = ns['get_pointcloud'](im, depth, k, w2c, mask=mask, compute_mean_sq_dist=True, mean_sq_dist_method="projective")
    P, V = ns['initialize_params'](cloud, 5, msd, dist)
    out[f"{name}/in/im"], out[f"{name}/in/depth"], out[f"{name}/in/intrinsics"], out[f"{name}/in/w2c"] = \
        im.numpy(), depth.numpy(), k.numpy(), w2c.numpy()
    out[f"{name}/cloud"], out[f"{name}/msd"] = cloud.numpy(), msd.numpy()
    for kk, v in P.items():
        out[f"{name}/out/param/{kk}"] = v.detach().numpy()
    for kk, v in V.items():
        out[f"{name}/out/var/{kk}"] = v.detach().numpy()
    # unmasked, untransformed variant
    out[f"{name}/cloud_cam"] = ns['get_pointcloud'](im, depth, k, w2c, transform_pts=False).numpy()


def case_prune(name, n, dist, seed, out):
    g = torch.Generator().manual_seed(seed)
    cols = 1 if dist == "isotropic" else 3
    params = {
        'means3D': torch.randn(n, 3, generator=g), 'rgb_colors': torch.rand(n, 3, generator=g),
        'unnorm_rotations': torch.randn(n, 4, generator=g), 'logit_opacities': -3.0 + 3.0 * torch.randn(n, 1, generator=g),
        'log_scales': -3.0 + 1.0 * torch.randn(n, cols, generator=g),
        'cam_unnorm_rots': torch.randn(1, 4, 3, generator=g), 'cam_trans': torch.randn(1, 3, 3, generator=g),
    }
    variables = {'max_2D_radius': torch.rand(n, generator=g), 'means2D_gradient_accum': torch.rand(n, generator=g),
                 'denom': torch.rand(n, generator=g), 'timestep': torch.floor(torch.rand(n, generator=g) * 5),
                 'scene_radius': torch.tensor(1.2)}
    P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    lrs = dict(means3D=0.0001, rgb_colors=0.0025, unnorm_rotations=0.001, logit_opacities=0.05, log_scales=0.001,
               cam_unnorm_rots=0.0, cam_trans=0.0)
    opt = torch.optim.Adam([{'params': [v], 'name': k, 'lr': lrs[k]} for k, v in P.items()], lr=0.0, eps=1e-15)
    for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales'):   # moments become non-trivial
        P[k].grad = torch.randn(P[k].shape, generator=g)
    opt.step()
    for k, v in P.items():
        out[f"{name}/in/param/{k}"] = v.detach().numpy().copy()
        st = opt.state.get(v)
        if st:
            out[f"{name}/in/exp_avg/{k}"] = st['exp_avg'].numpy().copy()
            out[f"{name}/in/exp_avg_sq/{k}"] = st['exp_avg_sq'].numpy().copy()
    for k, v in variables.items():
        out[f"{name}/in/var/{k}"] = v.numpy().copy()
    prune_dict = dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
                      final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500)
    V = {k: v.clone() for k, v in variables.items()}
    P, V = ref_ext.prune_gaussians(P, V, opt, 0, prune_dict)
    for k, v in P.items():
        out[f"{name}/out/param/{k}"] = v.detach().numpy()
        st = opt.state.get(v)
        if st:
            out[f"{name}/out/exp_avg/{k}"] = st['exp_avg'].numpy()
            out[f"{name}/out/exp_avg_sq/{k}"] = st['exp_avg_sq'].numpy()
    for k, v in V.items():
        out[f"{name}/out/var/{k}"] = v.numpy()
    # an iteration off the schedule leaves everything alone
    n_before = P['means3D'].shape[0]
    P, V = ref_ext.prune_gaussians(P, V, opt, 7, prune_dict)
    assert P['means3D'].shape[0] == n_before


def case_pose(ns, out):
    g = torch.Generator().manual_seed(3)
    params = {'cam_unnorm_rots': torch.nn.Parameter(torch.randn(1, 4, 6, generator=g)),
              'cam_trans': torch.nn.Parameter(torch.randn(1, 3, 6, generator=g))}
    out["pose/in/cam_unnorm_rots"], out["pose/in/cam_trans"] = params['cam_unnorm_rots'].detach().numpy().copy(), \
        params['cam_trans'].detach().numpy().copy()
    ns['initialize_camera_pose'](params, 3, True)
    ns['initialize_camera_pose'](params, 1, True)       # curr_time_idx <= 1: copies the previous pose
    ns['initialize_camera_pose'](params, 5, False)
    out["pose/out/cam_unnorm_rots"], out["pose/out/cam_trans"] = params['cam_unnorm_rots'].detach().numpy(), \
        params['cam_trans'].detach().numpy()


def case_keyframe_selection(out):
    """utils/keyframe_selection.py as is (CPU), seeded like the caller seeds everything (utils/common_utils.py:8-22)."""
    from utils import keyframe_selection as ref_kf
    g = torch.Generator().manual_seed(11)
    H, W, f = 120, 160, 110.0
    depth = 1.0 + 2.0 * torch.rand(1, H, W, generator=g)
    depth[0, :10, :] = 0.0
    k = torch.tensor([[f, 0.0, W / 2 - 0.5], [0.0, f, H / 2 - 0.5], [0.0, 0.0, 1.0]])
    w2c = torch.eye(4)
    kfs = []
    for i, (ang, tx) in enumerate([(0.0, 0.0), (0.15, 0.1), (0.6, 0.5), (2.5, 0.0), (-0.3, -0.2), (1.2, 1.0), (0.05, 0.02)]):
        m = torch.eye(4)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
        m[0, 3] = tx
        kfs.append({'id': i * 5, 'est_w2c': m})
    out["kfsel/depth"], out["kfsel/intrinsics"], out["kfsel/w2c"] = depth.numpy(), k.numpy(), w2c.numpy()
    out["kfsel/est_w2c"] = torch.stack([kf['est_w2c'] for kf in kfs]).numpy()
    for kk in (3, 10):
        torch.manual_seed(5)
        np.random.seed(5)
        sel = ref_kf.keyframe_selection_overlap(depth, w2c, k, kfs, kk)
        out[f"kfsel/selected_k{kk}"] = np.array([int(x) for x in sel], dtype=np.int64)
    torch.manual_seed(5)
    np.random.seed(5)
    out["kfsel/selected_empty"] = np.array(ref_kf.keyframe_selection_overlap(depth, w2c, k, [], 3), dtype=np.int64)


def case_densify(name, n, dist, seed, out):
    """utils/slam_external.py densify (+ accumulate_mean2d_gradient, cat_params_to_optimizer, remove_points,
    update_params_and_optimizer) as is, on CPU, seeded; two calls: a densification iteration and an opacity-reset iteration."""
    g = torch.Generator().manual_seed(seed)
    cols = 1 if dist == "isotropic" else 3
    params = {
        'means3D': torch.randn(n, 3, generator=g), 'rgb_colors': torch.rand(n, 3, generator=g),
        'unnorm_rotations': torch.randn(n, 4, generator=g), 'logit_opacities': -2.0 + 3.0 * torch.randn(n, 1, generator=g),
        'log_scales': -4.4 + 0.8 * torch.randn(n, cols, generator=g),
        'cam_unnorm_rots': torch.randn(1, 4, 3, generator=g), 'cam_trans': torch.randn(1, 3, 3, generator=g),
    }
    variables = {'max_2D_radius': torch.rand(n, generator=g), 'means2D_gradient_accum': 3e-4 * torch.rand(n, generator=g),
                 'denom': torch.floor(torch.rand(n, generator=g) * 3),
                 # no 'timestep': the reference's densify never extends it, so remove_points raises IndexError when it is present
                 'scene_radius': torch.tensor(1.2), 'seen': torch.rand(n, generator=g) < 0.7}
    means2D = torch.zeros(n, 3, requires_grad=True)
    means2D.grad = 4e-4 * torch.randn(n, 3, generator=g)
    variables['means2D'] = means2D
    P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    lrs = dict(means3D=0.0001, rgb_colors=0.0025, unnorm_rotations=0.001, logit_opacities=0.05, log_scales=0.001,
               cam_unnorm_rots=0.0, cam_trans=0.0)
    opt = torch.optim.Adam([{'params': [v], 'name': k, 'lr': lrs[k]} for k, v in P.items()], lr=0.0, eps=1e-15)
    for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales'):
        P[k].grad = torch.randn(P[k].shape, generator=g)
    opt.step()
    for k, v in P.items():
        out[f"{name}/in/param/{k}"] = v.detach().numpy().copy()
        st = opt.state.get(v)
        if st:
            out[f"{name}/in/exp_avg/{k}"] = st['exp_avg'].numpy().copy()
            out[f"{name}/in/exp_avg_sq/{k}"] = st['exp_avg_sq'].numpy().copy()
    for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'scene_radius', 'seen'):
        out[f"{name}/in/var/{k}"] = variables[k].numpy().copy()
    out[f"{name}/in/means2D_grad"] = means2D.grad.numpy().copy()
    dd = dict(start_after=0, remove_big_after=0, stop_after=5000, densify_every=100, grad_thresh=0.0002, num_to_split_into=2,
              removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=True, reset_opacities_every=300)
    V = dict(variables)
    torch.manual_seed(1234)
    P, V = ref_ext.densify(P, V, opt, 100, dd)                 # a densification iteration (clone + split + prune)
    n_mid = P['means3D'].shape[0]
    # the next call only accumulates the gradient statistics and resets the opacities (iter 300: densify too, so use 600 % 100 ... keep simple)
    for k, v in P.items():
        out[f"{name}/out/param/{k}"] = v.detach().numpy().copy()
        st = opt.state.get(v)
        if st:
            out[f"{name}/out/exp_avg/{k}"] = st['exp_avg'].numpy().copy()
            out[f"{name}/out/exp_avg_sq/{k}"] = st['exp_avg_sq'].numpy().copy()
    for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom'):
        out[f"{name}/out/var/{k}"] = V[k].numpy().copy()
    out[f"{name}/counts"] = np.array([n, n_mid])


if __name__ == "__main__":
    ns = reference_functions()
    out = {}
    case_keyframe_selection(out)
    case_densify("densify_iso", 900, "isotropic", 21, out)
    # (anisotropic maps make the reference's densify raise: stds.repeat(n, 3) of [N,3] scales is [nN,9])
    case_add("add_iso", ns, 56, 40, 60.0, 400, "isotropic", 0, out)
    case_add("add_aniso", ns, 40, 33, 45.0, 300, "anisotropic", 1, out)
    case_add("add_nan", ns, 40, 32, 45.0, 100, "isotropic", 2, out, nan_pixel=True)
    case_add("add_nothing", ns, 32, 24, 40.0, 50, "isotropic", 3, out, nothing=True)
    case_init("init_iso", ns, 48, 36, 50.0, "isotropic", 4, out)
    case_init("init_aniso", ns, 32, 24, 35.0, "anisotropic", 5, out)
    case_prune("prune_iso", 1200, "isotropic", 6, out)
    case_prune("prune_aniso", 700, "anisotropic", 7, out)
    case_pose(ns, out)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mapedit_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")
    for k in ("add_iso", "add_aniso", "add_nan", "add_nothing"):
        print(k, out[f"{k}/in/param/means3D"].shape[0], "->", out[f"{k}/out/param/means3D"].shape[0], "median", out[f"{k}/median"])
    for k in ("prune_iso", "prune_aniso"):
        print(k, out[f"{k}/in/param/means3D"].shape[0], "->", out[f"{k}/out/param/means3D"].shape[0])
