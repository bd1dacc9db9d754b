"""bench.py's ``cpu_baseline`` leg through the REFERENCE's own Python, where the reference is present (this container; not the GPU box).

The hot path as `north_star` names it for the CPU baseline -- "the reference's pure-PyTorch/CPU transformed_params2rendervar path" --
is /root/reference/utils/slam_helpers.py (transform_to_frame :252-304, transformed_params2rendervar :124-139,
transformed_params2depthplussilhouette :234-249), /root/reference/utils/slam_external.py (build_rotation, calc_ssim) and get_loss /
initialize_optimizer of /root/reference/scripts/splatam.py (:214-347, :160-166).  This script IMPORTS those modules as they are (a
device shim redirects their hard-coded ``.cuda()`` / ``device="cuda"`` to the CPU; the two functions of scripts/splatam.py, whose
module cannot be imported for its absent third-party imports, are executed from its source text), binds the un-vendored rasterizer
name ``Renderer`` to the C oracle (oracle/c_ref.py: the checker -- this is the cpu_baseline leg, where it may be timed), and times one
tracking iteration of the scene bench.py hands over: get_loss -> backward -> optimizer.step.

It runs in a process of its own (the shim patches torch): bench.py writes the scene to an .npz and reads one JSON line back.
usage: python scripts/cpu_baseline_reference.py <scene.npz> [budget seconds] [max timed iterations]"""
import ast
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SPLAT_REFERENCE_DIR", "/root/reference")


def main():
    path = sys.argv[1]
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
    max_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    if not os.path.isdir(os.path.join(REF, "utils")):
        print(json.dumps({"error": f"no reference at {REF}"}))
        return 2
    sys.path.insert(0, REPO)
    sys.path.insert(0, REF)
    # ---- device shim (as tests/golden/make_golden.py): the reference hard-codes .cuda() / device="cuda"
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "arange"):
        orig = getattr(torch, name)

        def wrap(*a, _o=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return _o(*a, **k)
        setattr(torch, name, wrap)
    from utils import slam_external as ref_ext  # noqa: E402  (the reference's own modules)
    from utils import slam_helpers as ref_h  # noqa: E402
    from oracle import c_ref  # noqa: E402
    from splatam_amd.rasterizer import GaussianRasterizationSettings as Camera  # noqa: E402  (the 11-field tuple: utils/recon_helpers.py:14-26)

    src = open(os.path.join(REF, "scripts", "splatam.py")).read()
    tree = ast.parse(src)
    ns = dict(torch=torch, np=np, Renderer=c_ref.CRasterizer, transform_to_frame=ref_h.transform_to_frame,
              transformed_params2rendervar=ref_h.transformed_params2rendervar,
              transformed_params2depthplussilhouette=ref_h.transformed_params2depthplussilhouette,
              l1_loss_v1=ref_h.l1_loss_v1, calc_ssim=ref_ext.calc_ssim)
    for fn_name in ("get_loss", "initialize_optimizer"):
        fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fn_name)
        exec(compile(ast.get_source_segment(src, fn), f"reference_{fn_name}", "exec"), ns)
    get_loss, initialize_optimizer = ns["get_loss"], ns["initialize_optimizer"]

    z = np.load(path)
    params = {k[6:]: torch.nn.Parameter(torch.tensor(z[k])) for k in z.files if k.startswith("param/")}
    cam = Camera(image_height=int(z["H"]), image_width=int(z["W"]), tanfovx=float(z["tanfovx"]), tanfovy=float(z["tanfovy"]),
                 bg=torch.tensor(z["bg"]), scale_modifier=1.0, viewmatrix=torch.tensor(z["viewmatrix"]), projmatrix=torch.tensor(z["projmatrix"]),
                 sh_degree=0, campos=torch.tensor(z["campos"]), prefiltered=False)
    data = {'cam': cam, 'im': torch.tensor(z["im"]), 'depth': torch.tensor(z["depth"]), 'id': int(z["time_idx"]), 'w2c': torch.eye(4),
            'intrinsics': None, 'iter_gt_w2c_list': None}
    n = params['means3D'].shape[0]
    variables = {'max_2D_radius': torch.zeros(n), 'means2D_gradient_accum': torch.zeros(n), 'denom': torch.zeros(n)}
    lrs = {k[3:]: float(z[k]) for k in z.files if k.startswith("lr/")}
    opt = initialize_optimizer(params, lrs, tracking=True)
    weights = {'im': float(z["w_im"]), 'depth': float(z["w_depth"])}
    t_idx = int(z["time_idx"])
    losses = []

    def one():
        loss, _, _ = get_loss(params, data, variables, t_idx, weights, bool(z["use_sil_for_loss"]), float(z["sil_thres"]), bool(z["use_l1"]),
                              bool(z["ignore_outlier_depth_loss"]), tracking=True)
        loss.backward()
        with torch.no_grad():
            opt.step()
            opt.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    one()                                       # warm-up (first-use costs of the torch operators, the oracle's thread pool)
    times = []
    t_all = time.perf_counter()
    while len(times) < max_iters and (time.perf_counter() - t_all) < budget:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    print(json.dumps({"value": round(1.0 / med, 4), "unit": "iters/s", "median_s": round(med, 4), "iterations": len(times),
                      "threads": torch.get_num_threads(), "first_loss": losses[0], "losses": losses[:3],
                      "reference": REF}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
