"""world_size-2 gloo test of the multi-rank FRAME LOOP (splatam_amd/pipeline.py: rgbd_slam with torch.distributed
initialised): tracking replicas + pose broadcast, view-sharded mapping with one gradient all-reduce per iteration, replicated
map edits.  CPU only: the reference-shaped loop (engine="dropin") with the C oracle as its Renderer.  Checked: the replicas end
bit-identical (map, poses, row counts), the loop tracks the synthetic trajectory as well as one process does, and -- against a single-process run of the
same seeds -- that the ranks really rendered DIFFERENT views (the two-rank map differs from the one-rank map)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_ref
from splatam_amd import dist as sdist
from splatam_amd import pipeline, slam

W, H, F_ = 64, 48, 60.0
FRAMES = 3


def _run(out_path):
    slam.Renderer = c_ref.CRasterizer
    torch.manual_seed(0)
    np.random.seed(0)
    ds = pipeline.SyntheticRGBDSequence(2500, W, H, F_, F_, W / 2 - 0.5, H / 2 - 0.5, num_frames=FRAMES, seed=3, device="cpu")
    cfg = pipeline.replica_config(tracking_iters=6, mapping_iters=5, keyframe_every=1)
    params, variables, st = pipeline.rgbd_slam(ds, cfg, engine="dropin")
    err = max(float((pipeline._est_w2c(params, t)[:3, 3] - ds.gt_w2c(t)[:3, 3]).norm()) for t in range(FRAMES))
    np.savez(out_path, err=err, n=np.array(st['num_gaussians']), **{k: v.detach().numpy() for k, v in params.items()})


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    os.environ["OMP_NUM_THREADS"] = "2"
    sdist.init_from_env(backend="gloo")
    _run(os.path.join(out_dir, f"rank{rank}.npz"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_loop_keeps_replicas_identical(tmp_path, monkeypatch):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{i}.npz") for i in (0, 1))
    assert r0['n'].tolist() == r1['n'].tolist()
    for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales', 'cam_unnorm_rots', 'cam_trans'):
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)           # replicas bit-identical
    # single process, same seeds: one view per mapping iteration instead of two -> a different map
    monkeypatch.setattr(slam, "Renderer", c_ref.CRasterizer)
    _run(str(tmp_path / "single.npz"))
    single = np.load(tmp_path / "single.npz")
    # the loop tracks as well as the single-process loop does (6 + 5 iterations per frame on a 64x48 image: coarse either way)
    assert float(r0['err']) <= 2.0 * float(single['err']) + 2e-3, (float(r0['err']), float(single['err']))
    assert not np.array_equal(single['rgb_colors'], r0['rgb_colors']) or single['rgb_colors'].shape != r0['rgb_colors'].shape
