"""Generates tests/golden/loop_reference.npz by EXECUTING THE REFERENCE'S OWN FRAME LOOP -- ``rgbd_slam`` of
/root/reference/scripts/splatam.py:455-990, the file imported as it is -- on CPU in this container, on a small synthetic RGB-D
sequence, and recording what the loop did (tests/loop_trace.py): every get_loss / optimizer / add_new_gaussians / prune_gaussians /
keyframe_selection_overlap / initialize_camera_pose call in order, the loss of every iteration, the final parameters, the keyframe
list.  tests/test_loop_golden.py (CPU, on the oracle) and tests/test_gpu_loop_golden.py (HIP: drop-in, plug-in and fused engines)
hold ``splatam_amd.pipeline.rgbd_slam`` to this recording.

What is the reference's and what is not:
  * the module is the reference's file, loaded with ``importlib`` under its own ``sys.path`` entry.  Its third-party imports that do
    not exist offline (cv2, wandb, imageio, natsort, kornia, lpips, open3d, torchmetrics, pytorch_msssim, plyfile, faiss) and its
    dataset package (whose loaders need them) are served as empty stand-in modules; ``utils/*`` are the reference's own files;
  * ``.cuda()`` / ``device="cuda"`` are redirected to the CPU (the shim of tests/golden/make_golden.py);
  * ``diff_gaussian_rasterization`` -- un-vendored in the reference (requirements.txt:15) -- is bound to this repository's C ORACLE
    (oracle/c_ref.CRasterizer): the recording pins the LOOP (everything scripts/splatam.py itself does) to the reference's code and the
    rasterizer to the oracle, like the function-level fixtures;
  * ``get_dataset`` returns the recorded synthetic sequence; ``report_progress`` / ``eval`` (plots, PSNR / LPIPS, files) are no-ops;
    ``save_params`` hands the final dict to the recorder instead of writing ``params.npz``;
  * the configuration is the reference's own configs/replica/splatam.py, loaded with SourceFileLoader as ``__main__`` does
    (:999-1001), with the sizes of the loop reduced (iterations, cadence, prune schedule) -- every override is listed in ``CASES``.

Run:  python tests/golden/make_golden_loop.py [log file]      (needs /root/reference; not needed on the GPU box)
"""
import copy
import importlib.machinery
import importlib.util
import json
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

STAND_INS = ("cv2", "wandb", "imageio", "natsort", "kornia", "lpips", "open3d", "torchmetrics", "pytorch_msssim", "plyfile", "faiss",
             "datasets")      # `datasets`: the reference's package has no __init__.py (datasets/_init_.py), so the name would resolve to
#                               the unrelated HuggingFace package of this image; its loaders need cv2 / imageio / natsort anyway


# ---- device shim (as tests/golden/make_golden.py) -------------------------------------------------------------------------------
def install_device_shim():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for name in ("zeros", "ones", "eye", "zeros_like", "ones_like", "tensor", "arange", "empty", "full"):
        orig = getattr(torch, name)

        def wrap(*a, _o=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return _o(*a, **k)
        setattr(torch, name, wrap)


class _StandInFinder:
    """Serves ``mock.MagicMock`` modules for the third-party names the reference imports and this image does not have."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in STAND_INS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__, m.__path__, m.__spec__, m.__loader__ = spec.name, [], spec, self
        return m

    def exec_module(self, module):
        pass


def load_reference_module(renderer_module):
    """The reference's scripts/splatam.py as a module object named ``ref_splatam`` (its ``__main__`` block does not run)."""
    for name in list(sys.modules):
        if name.split(".")[0] in STAND_INS:
            del sys.modules[name]
    sys.meta_path.insert(0, _StandInFinder())
    sys.modules["diff_gaussian_rasterization"] = renderer_module
    sys.path.insert(0, REF)                                     # `utils.*`: the reference's own files
    spec = importlib.util.spec_from_file_location("ref_splatam", os.path.join(REF, "scripts", "splatam.py"))
    module = importlib.util.module_from_spec(spec)
    with open(os.devnull, "w") as null:                         # (the file prints sys.path at import)
        out, sys.stdout = sys.stdout, null
        try:
            spec.loader.exec_module(module)
        finally:
            sys.stdout = out
    return module


def oracle_renderer_module():
    from oracle import c_ref
    from splatam_amd.rasterizer import GaussianRasterizationSettings
    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizer = c_ref.CRasterizer
    m.GaussianRasterizationSettings = GaussianRasterizationSettings
    return m


def reference_config(overrides):
    """configs/replica/splatam.py as scripts/splatam.py's __main__ loads it, then the overrides of one case."""
    path = os.path.join(REF, "configs", "replica", "splatam.py")
    cfg = copy.deepcopy(importlib.machinery.SourceFileLoader("ref_replica_config", path).load_module().config)

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v
    merge(cfg, overrides)
    return cfg


# ---- the cases -------------------------------------------------------------------------------------------------------------------
COMMON = dict(primary_device="cpu", use_wandb=False, workdir="/tmp/splatam_golden_loop", report_global_progress_every=500,
              data=dict(gradslam_data_cfg=None, basedir="unused", sequence="synthetic"))
CASES = {
    # the shipped Replica flow in small: mapping on every frame, keyframe every 2nd frame, constant-velocity pose initialisation,
    # pruning at mapping iterations 0 / 3 / 6 with a size bound that removes rows (scene radius / 8)
    "base": dict(
        scene=dict(n_gaussians=5000, W=96, H=64, f=80.0, frames=5, seed=3, step_m=0.012, step_deg=0.4),
        config=dict(run_name="base", map_every=1, keyframe_every=2, mapping_window_size=4, scene_radius_depth_ratio=8.1,
                    data=dict(desired_image_height=64, desired_image_width=96, num_frames=-1),
                    tracking=dict(num_iters=8),
                    mapping=dict(num_iters=8, pruning_dict=dict(start_after=0, remove_big_after=0, stop_after=6, prune_every=3)))),
    # the branches the shipped flow leaves alone: mapping every 2nd frame (frames in between are tracked only), keyframe every 3rd
    # frame + the forced keyframe at num_frames - 2, a window that keeps ONE overlapping keyframe, the depth-loss retry
    # (use_depth_loss_thres: the budget doubles on frames whose last depth loss is above the threshold), no forward propagation,
    # anisotropic Gaussians
    "variant": dict(
        scene=dict(n_gaussians=4000, W=80, H=64, f=70.0, frames=6, seed=7, step_m=0.015, step_deg=0.5),
        config=dict(run_name="variant", map_every=2, keyframe_every=3, mapping_window_size=3, scene_radius_depth_ratio=3,
                    gaussian_distribution="anisotropic",
                    data=dict(desired_image_height=64, desired_image_width=80, num_frames=-1),
                    tracking=dict(num_iters=5, forward_prop=False, use_depth_loss_thres=True, depth_loss_thres=None),
                    mapping=dict(num_iters=6, pruning_dict=dict(start_after=1, remove_big_after=2, stop_after=4, prune_every=2,
                                                                removal_opacity_threshold=0.515, final_removal_opacity_threshold=0.52)))),
    # ground-truth poses instead of tracking (scripts/splatam.py:745-754: matrix_to_quaternion of the dataset's relative pose), no
    # pruning, a keyframe on every frame, every keyframe in the window
    "gtposes": dict(
        scene=dict(n_gaussians=4000, W=80, H=64, f=70.0, frames=4, seed=11, step_m=0.02, step_deg=0.8),
        config=dict(run_name="gtposes", map_every=1, keyframe_every=1, mapping_window_size=8, scene_radius_depth_ratio=3,
                    data=dict(desired_image_height=64, desired_image_width=80, num_frames=-1),
                    tracking=dict(num_iters=4, use_gt_poses=True),
                    mapping=dict(num_iters=5, prune_gaussians=False))),
}


def make_sequence(scene):
    """The synthetic sequence of splatam_amd.pipeline rendered ONCE, on the oracle, and kept as arrays (the tests read them back)."""
    from oracle import c_ref
    from splatam_amd import pipeline, slam
    saved = slam.Renderer
    slam.Renderer = c_ref.CRasterizer
    try:
        W, H, f = scene['W'], scene['H'], scene['f']
        ds = pipeline.SyntheticRGBDSequence(scene['n_gaussians'], W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, num_frames=scene['frames'],
                                            seed=scene['seed'], device="cpu", step_m=scene['step_m'], step_deg=scene['step_deg'])
        items = [ds[t] for t in range(len(ds))]
    finally:
        slam.Renderer = saved
    return dict(color=np.stack([i[0].numpy() for i in items]).astype(np.float32),
                depth=np.stack([i[1].numpy() for i in items]).astype(np.float32),
                intrinsics=items[0][2].numpy().astype(np.float32),
                poses=np.stack([i[3].numpy() for i in items]).astype(np.float32))


def run_case(S, name, case, out, log):
    from loop_trace import KIND_NAMES, LOSS, LoopRecorder, RecordedRGBDSequence
    frames = make_sequence(case['scene'])
    for k, v in frames.items():
        out[f"{name}/frames/{k}"] = v
    dataset = RecordedRGBDSequence({f"{name}/frames/{k}": v for k, v in frames.items()}, name)
    overrides = copy.deepcopy(COMMON)
    for k, v in case['config'].items():
        if isinstance(v, dict) and isinstance(overrides.get(k), dict):
            overrides[k].update(v)
        else:
            overrides[k] = v
    config = reference_config(overrides)
    config['data'].pop('gradslam_data_cfg')                                  # (:489: the dataset name then comes from the config)
    config['data']['dataset_name'] = "synthetic"

    def run(cfg):
        final = {}
        PROBE['last'].clear()
        rec = LoopRecorder().wrap(S)
        saved = {k: getattr(S, k) for k in ("get_dataset", "report_progress", "eval", "save_params", "tqdm")}
        S.get_dataset = lambda **kw: dataset
        S.report_progress = lambda *a, **k: None
        S.eval = lambda *a, **k: None
        S.save_params = lambda params, output_dir: final.update(params, __last_depth_losses=np.array(
            [PROBE['last'][t] for t in sorted(PROBE['last'])]))
        S.tqdm = lambda it=None, *a, **k: mock.MagicMock() if it is None else _Quiet(it)
        try:
            S.seed_everything(seed=cfg['seed'])                               # scripts/splatam.py:1004
            S.rgbd_slam(copy.deepcopy(cfg))
        finally:
            for k, v in saved.items():
                setattr(S, k, v)
            rec.restore()
        return rec, final

    if config['tracking'].get('depth_loss_thres', 0) is None:
        # the retry threshold of this case: half way between the median of the frames' last depth losses (in a run without the retry)
        # and the next larger one, so that some frames double their budget and some do not -- and none sits on the threshold
        probe = copy.deepcopy(config)
        probe['tracking']['use_depth_loss_thres'] = False
        probe['tracking']['depth_loss_thres'] = 100000
        _, final = run(probe)
        last = final.pop('__last_depth_losses')
        ordered = np.sort(last)
        config['tracking']['depth_loss_thres'] = float(0.5 * (ordered[len(ordered) // 2] + ordered[len(ordered) // 2 + 1]))
        print(f"[{name}] last depth losses per tracked frame {np.round(last, 4).tolist()} -> depth_loss_thres "
              f"{config['tracking']['depth_loss_thres']:.6f}", file=log)
    rec, final = run(config)
    final.pop('__last_depth_losses', None)
    events, values, selected = rec.arrays()
    out[f"{name}/config"] = np.array(json.dumps(config))
    out[f"{name}/events"], out[f"{name}/values"], out[f"{name}/selected"] = events, values, selected
    for k, v in final.items():
        out[f"{name}/final/{k}"] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    counts = {KIND_NAMES[k]: int((events[:, 0] == k).sum()) for k in range(len(KIND_NAMES))}
    print(f"[{name}] {len(events)} events {counts}; keyframes {final['keyframe_time_indices'].tolist()}; "
          f"{final['means3D'].shape[0]} Gaussians at the end", file=log)
    for i in range(len(events)):
        k, a, b, c, d = (int(x) for x in events[i])
        extra = f" loss {values[i]:.6f}" if k == LOSS else (f" -> {selected[c:c + d].tolist()}" if KIND_NAMES[k] == "KFSEL" else "")
        print(f"[{name}]   {KIND_NAMES[k]:6s} {a:5d} {b:5d} {c:5d} {d:3d}{extra}", file=log)


class _Quiet:
    """tqdm's surface as the loop uses it (iteration, update, close) without the bars."""

    def __init__(self, it):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def update(self, n=1):
        pass

    def close(self):
        pass


def install_depth_loss_probe(S):
    """The probe run of a case needs the LAST tracking depth loss of every frame: read from the recorder would do, but the value the
    loop itself compares (:728) is ``losses['depth']``; keep them as the loop sees them."""
    orig = S.get_loss
    state = {'last': {}, 'orig': orig}

    def get_loss(params, curr_data, variables, iter_time_idx, *a, **k):
        out = orig(params, curr_data, variables, iter_time_idx, *a, **k)
        if k.get('tracking'):
            state['last'][int(iter_time_idx)] = float(out[2]['depth'])
        return out
    S.get_loss = get_loss
    return state


PROBE = {}


def main():
    log_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "profiles", "r05_reference_loop.log")
    install_device_shim()
    S = load_reference_module(oracle_renderer_module())
    PROBE.update(install_depth_loss_probe(S))
    out = {}
    with open(log_path, "w") as log:
        print(f"reference module: {S.__file__}; rgbd_slam at line {S.rgbd_slam.__code__.co_firstlineno}; Renderer = "
              f"{S.Renderer.__module__}.{S.Renderer.__name__} (C oracle); torch {torch.__version__}", file=log)
        for name, case in CASES.items():
            run_case(S, name, case, out, log)
    path = os.path.join(HERE, "loop_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays; log", log_path)


if __name__ == "__main__":
    main()
