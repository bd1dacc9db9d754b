"""Generates tests/golden/caller_reference.npz: the CALL SEQUENCE of the reference's own get_loss
(/root/reference/scripts/splatam.py:214-347, exec'd from its source with its own helpers) at the rasterizer boundary.

For one small scene with config B's aspect and intrinsics scaled down, tracking and mapping mode, a recording Renderer stores
  * the settings tuple and the exact kwargs of BOTH Renderer(raster_settings=...)(**rendervar) calls (:249, :253),
  * what the call returned (the C oracle renders them: color, radii, depth),
  * after loss.backward(): the gradient autograd delivered to each render (dL/d im, dL/d depth_sil), the gradient each
    call's inputs received, variables['means2D'].grad (the retained non-leaf of :248), and the max_2D_radius / seen
    bookkeeping of :341-345.
tests/test_gpu_caller_replay.py replays exactly these kwargs through diff_gaussian_rasterization on the GPU: the unmodified
caller's contract, checked on the HIP path itself rather than on the repository's mirror of the caller.

Run:  python tests/golden/make_golden_caller.py     (needs /root/reference; not needed on the GPU box)
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

torch.Tensor.cuda = lambda self, *a, **k: self
for _name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "arange"):
    _orig = getattr(torch, _name)

    def _wrap(*a, _o=_orig, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return _o(*a, **k)
    setattr(torch, _name, _wrap)

from oracle import c_ref  # noqa: E402
from oracle import raster_ref as R  # noqa: E402
from utils import slam_external as ref_ext  # noqa: E402
from utils import slam_helpers as ref_h  # noqa: E402

CALLS = []


class RecordingRenderer:
    """`Renderer` as the reference constructs and calls it; renders with the C oracle and records the call."""

    def __init__(self, raster_settings):
        self.settings = raster_settings
        self.inner = c_ref.CRasterizer(raster_settings)

    def __call__(self, **kwargs):
        assert set(kwargs) == {'means3D', 'colors_precomp', 'rotations', 'opacities', 'scales', 'means2D'}, sorted(kwargs)
        out = self.inner(**kwargs)
        out[0].retain_grad()
        for v in kwargs.values():
            if v.requires_grad and not v.is_leaf:
                v.retain_grad()
        CALLS.append((self.settings, kwargs, out))
        return out


def reference_get_loss():
    src = open(os.path.join(REF, "scripts", "splatam.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_loss")
    ns = dict(torch=torch, np=np, Renderer=RecordingRenderer, transform_to_frame=ref_h.transform_to_frame,
              transformed_params2rendervar=ref_h.transformed_params2rendervar,
              transformed_params2depthplussilhouette=ref_h.transformed_params2depthplussilhouette,
              l1_loss_v1=ref_h.l1_loss_v1, calc_ssim=ref_ext.calc_ssim)
    exec(compile(ast.get_source_segment(src, fn), "reference_get_loss", "exec"), ns)
    return ns["get_loss"]


def main():
    n, W, H = 5000, 200, 112                       # config B's aspect; fx = fy = 600 scaled by 1/6
    f, cx, cy = 100.0, W / 2 - 0.5, H / 2 - 0.5
    p = R.synthetic_cloud(n, W, H, f, f, cx, cy, seed=11)
    T = 3
    g = torch.Generator().manual_seed(5)
    cam_rots = torch.zeros(1, 4, T)
    cam_rots[:, 0, :] = 1.0
    cam_rots += 0.02 * torch.randn(1, 4, T, generator=g)
    params = dict(p, cam_unnorm_rots=cam_rots, cam_trans=0.03 * torch.randn(1, 3, T, generator=g))
    cam = R.make_camera(W, H, f, f, cx, cy)
    out = {"meta": np.array([n, W, H, f, cx, cy], dtype=np.float64)}
    for k, v in params.items():
        out[f"param/{k}"] = v.numpy()
    for fld in ("tanfovx", "tanfovy", "scale_modifier"):
        out[f"cam/{fld}"] = np.array(getattr(cam, fld), dtype=np.float64)
    for fld in ("bg", "viewmatrix", "projmatrix", "campos"):
        out[f"cam/{fld}"] = getattr(cam, fld).numpy()
    # ground truth: a render from a perturbed pose + sensor noise
    with torch.no_grad():
        P2 = {k: v.clone() for k, v in params.items()}
        P2['cam_trans'][..., 1] += torch.tensor([[0.01, -0.005, 0.005]])
        tg2 = ref_h.transform_to_frame(P2, 1, False, False)
        gt_im, _, _ = c_ref.CRasterizer(cam)(**ref_h.transformed_params2rendervar(P2, tg2))
        ds, _, _ = c_ref.CRasterizer(cam)(**ref_h.transformed_params2depthplussilhouette(P2, torch.eye(4), tg2))
        gt_depth = torch.where(ds[1:2] > 0.5, ds[0:1] / ds[1:2].clamp_min(1e-6), torch.zeros_like(ds[0:1]))
        gt_im = (gt_im + 0.03 * torch.randn(gt_im.shape, generator=g)).clamp(0, 1)
    out["gt_im"], out["gt_depth"] = gt_im.numpy(), gt_depth.numpy()
    get_loss = reference_get_loss()
    for mode in ("tracking", "mapping"):
        CALLS.clear()
        P = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
        variables = {'max_2D_radius': torch.zeros(n), 'means2D_gradient_accum': torch.zeros(n), 'denom': torch.zeros(n), 'timestep': torch.zeros(n)}
        curr = {'cam': cam, 'im': gt_im, 'depth': gt_depth, 'id': 1, 'w2c': torch.eye(4)}
        kw = dict(tracking=True) if mode == "tracking" else dict(mapping=True)
        loss, variables, wl = get_loss(P, curr, variables, 1, dict(im=0.5, depth=1.0), mode == "tracking", 0.99 if mode == "tracking" else 0.5,
                                       True, False, **kw)
        loss.backward()
        assert len(CALLS) == 2 and CALLS[0][0] is cam and CALLS[1][0] is cam
        out[f"{mode}/loss"] = np.array(loss.item())
        for ci, (settings, kwargs, res) in enumerate(CALLS):
            for k, v in kwargs.items():
                # both modes and both calls see the same Gaussians at the same pose: inputs and outputs are stored once
                # (call 1 differs from call 0 in colors_precomp only -- asserted here, relied on by the replay test)
                key = f"call{ci}/in/{k}"
                val = v.detach().numpy()
                if ci == 1 and k != 'colors_precomp':
                    assert np.array_equal(val, out[f"call0/in/{k}"]), k
                elif key in out:
                    assert np.array_equal(val, out[key]), key
                else:
                    out[key] = val
                if v.grad is not None:
                    out[f"{mode}/call{ci}/grad_in/{k}"] = v.grad.numpy()
            for name, val in (("color", res[0].detach().numpy()), ("radii", res[1].numpy()), ("depth", res[2].detach().numpy())):
                key = f"call{ci}/out/{name}"
                if key in out:
                    assert np.array_equal(val, out[key]), key
                else:
                    out[key] = val
            out[f"{mode}/call{ci}/grad_out/color"] = (torch.zeros_like(res[0]) if res[0].grad is None else res[0].grad).numpy()
        assert variables['means2D'] is CALLS[0][1]['means2D']
        out[f"{mode}/means2D_grad"] = variables['means2D'].grad.numpy()
        out[f"{mode}/max_2D_radius"] = variables['max_2D_radius'].numpy()
        out[f"{mode}/seen"] = variables['seen'].numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "caller_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")


if __name__ == "__main__":
    main()
