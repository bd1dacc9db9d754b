"""Generates tests/golden/slam_reference.npz by running the REFERENCE's own Python
(/root/reference/utils/slam_helpers.py, utils/slam_external.py and the source of
get_loss cut out of scripts/splatam.py:214-347) on CPU in this container.

The reference hard-codes ``.cuda()`` / ``device="cuda"``; a small shim redirects
those to the CPU.  Its rasterizer dependency is not vendored, so ``Renderer`` is
bound to the repository's CPU oracle (oracle/raster_ref.py) -- the fixture
therefore pins everything AROUND the rasterizer boundary (transform, render-var
assembly, masks, losses, gradient wiring) to the reference's code, and the
rasterizer itself to the oracle.

Run:  python tests/golden/make_golden.py     (needs /root/reference; not needed on the GPU box)
"""
import ast
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

# ---- device shim -----------------------------------------------------------
torch.Tensor.cuda = lambda self, *a, **k: self
for _name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "arange"):
    _orig = getattr(torch, _name)

    def _wrap(*a, _o=_orig, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return _o(*a, **k)
    setattr(torch, _name, _wrap)

from oracle import raster_ref as R  # noqa: E402
from utils import slam_external as ref_ext  # noqa: E402
from utils import slam_helpers as ref_h  # noqa: E402


def reference_get_loss():
    """exec() the reference's get_loss source with its own helpers and the oracle renderer."""
    src = open(os.path.join(REF, "scripts", "splatam.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_loss")
    code = ast.get_source_segment(src, fn)
    ns = dict(torch=torch, np=np, Renderer=R.OracleRasterizer,
              transform_to_frame=ref_h.transform_to_frame,
              transformed_params2rendervar=ref_h.transformed_params2rendervar,
              transformed_params2depthplussilhouette=ref_h.transformed_params2depthplussilhouette,
              l1_loss_v1=ref_h.l1_loss_v1, calc_ssim=ref_ext.calc_ssim)
    exec(compile(code, "reference_get_loss", "exec"), ns)
    return ns["get_loss"]


def make_scene(n, W, H, f, anisotropic, seed):
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    p = R.synthetic_cloud(n, W, H, f, f, cx, cy, seed=seed, anisotropic=anisotropic)
    T = 3
    g = torch.Generator().manual_seed(seed + 100)
    cam_rots = torch.zeros(1, 4, T)
    cam_rots[:, 0, :] = 1.0
    cam_rots += 0.02 * torch.randn(1, 4, T, generator=g)
    cam_trans = 0.03 * torch.randn(1, 3, T, generator=g)
    params = dict(p)
    params['cam_unnorm_rots'] = cam_rots
    params['cam_trans'] = cam_trans
    return params, (W, H, f, cx, cy)


def run_case(name, n, W, H, f, anisotropic, seed, out):
    params, (W, H, f, cx, cy) = make_scene(n, W, H, f, anisotropic, seed)
    for k, v in params.items():
        out[f"{name}/param/{k}"] = v.numpy()
    k_mat = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], dtype=np.float32)
    cam = R.make_camera(W, H, f, f, cx, cy)
    out[f"{name}/meta"] = np.array([n, W, H, f, cx, cy], dtype=np.float64)
    time_idx = 1

    # (1) helper functions
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    tg = ref_h.transform_to_frame(P, time_idx, gaussians_grad=True, camera_grad=True)
    rv = ref_h.transformed_params2rendervar(P, tg)
    dv = ref_h.transformed_params2depthplussilhouette(P, torch.eye(4), tg)
    out[f"{name}/tg/means3D"] = tg['means3D'].detach().numpy()
    out[f"{name}/tg/unnorm_rotations"] = tg['unnorm_rotations'].detach().numpy()
    for k in ('rotations', 'opacities', 'scales', 'colors_precomp'):
        out[f"{name}/rv/{k}"] = rv[k].detach().numpy()
    out[f"{name}/dv/colors_precomp"] = dv['colors_precomp'].detach().numpy()
    out[f"{name}/build_rotation"] = ref_ext.build_rotation(params['cam_unnorm_rots'][..., 1]).numpy()

    # (2) ground truth frame (a render from a perturbed pose) and both loss modes
    with torch.no_grad():
        P2 = {k: v.clone() for k, v in params.items()}
        P2['cam_trans'][..., time_idx] += torch.tensor([[0.01, -0.005, 0.005]])
        tg2 = ref_h.transform_to_frame(P2, time_idx, False, False)
        rv2 = ref_h.transformed_params2rendervar(P2, tg2)
        gt_im, _, _ = R.OracleRasterizer(cam)(**rv2)
        dv2 = ref_h.transformed_params2depthplussilhouette(P2, torch.eye(4), tg2)
        ds, _, _ = R.OracleRasterizer(cam)(**dv2)
        gt_depth = torch.where(ds[1:2] > 0.5, ds[0:1] / ds[1:2].clamp_min(1e-6), torch.zeros_like(ds[0:1]))
    out[f"{name}/gt_im"] = gt_im.numpy()
    out[f"{name}/gt_depth"] = gt_depth.numpy()
    ssim = ref_ext.calc_ssim(gt_im, gt_im * 0.9 + 0.02)
    out[f"{name}/ssim"] = np.array(ssim.item())

    get_loss = reference_get_loss()
    for mode in ("tracking", "mapping"):
        P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
        variables = {'max_2D_radius': torch.zeros(n), 'means2D_gradient_accum': torch.zeros(n), 'denom': torch.zeros(n),
                     'timestep': torch.zeros(n)}
        curr = {'cam': cam, 'im': gt_im, 'depth': gt_depth, 'id': time_idx, 'w2c': torch.eye(4)}
        kw = dict(tracking=True) if mode == "tracking" else dict(mapping=True)
        loss, variables, wl = get_loss(P, curr, variables, time_idx, dict(im=0.5, depth=1.0),
                                       mode == "tracking", 0.99 if mode == "tracking" else 0.5, True, False, **kw)
        loss.backward()
        out[f"{name}/{mode}/loss"] = np.array([loss.item(), wl['im'].item(), wl['depth'].item()])
        for k, v in P.items():
            out[f"{name}/{mode}/grad/{k}"] = (torch.zeros_like(v) if v.grad is None else v.grad).numpy()
        out[f"{name}/{mode}/max_2D_radius"] = variables['max_2D_radius'].numpy()
        out[f"{name}/{mode}/means2D_grad"] = variables['means2D'].grad.numpy()


if __name__ == "__main__":
    out = {}
    run_case("iso", 1500, 96, 64, 80.0, False, 0, out)
    run_case("aniso", 1200, 80, 64, 70.0, True, 1, out)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slam_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")
