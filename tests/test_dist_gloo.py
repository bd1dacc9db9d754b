"""world_size-2 gloo test of the view-sharded mapping step: the mean
of the per-rank gradients after GradBucket.all_reduce_mean equals single-process
gradient accumulation over the same two views.  CPU only; the rasterizer inside
get_loss is the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import raster_ref as R
from splatam_amd import dist as sdist
from splatam_amd import slam

KEYS = sdist.GAUSSIAN_KEYS


def _scene():
    W, H, f = 64, 48, 60.0
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    params, variables = slam.synthetic_params(600, W, H, f, f, cx, cy, num_frames=4, seed=3, device="cpu")
    cam = R.make_camera(W, H, f, f, cx, cy)
    w2c = torch.eye(4)
    frames = {}
    g = torch.Generator().manual_seed(0)
    for t in (2, 3):
        im, depth = slam.synthetic_frame(params, cam, w2c, t, rot_deg=0.4 * t, trans_m=0.02 * t)
        im = (im + 0.05 * torch.randn(im.shape, generator=g)).clamp(0, 1)
        frames[t] = {'cam': cam, 'im': im, 'depth': depth, 'id': t, 'w2c': w2c}
    return params, variables, frames


def _view_grads(params, variables, frame, t):
    for p in params.values():
        p.grad = None
    loss, _, _ = slam.get_loss(params, frame, variables, t, slam.REPLICA_MAPPING['loss_weights'], False, 0.5, True, False, mapping=True)
    loss.backward()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    slam.Renderer = R.OracleRasterizer
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    params, variables, frames = _scene()
    views = [2, 3]
    mine = [views[i] for i in sdist.shard_views(len(views), rank, world)]
    assert len(mine) == 1
    _view_grads(params, variables, frames[mine[0]], mine[0])
    # the fused engine's exchange step: the same mean taken directly on a flat gradient buffer (no packing)
    flat = torch.cat([params[k].grad.reshape(-1) for k in KEYS]).clone()
    sdist.all_reduce_mean_flat(flat)
    bucket = sdist.GradBucket(params)
    bucket.all_reduce_mean(params)
    assert torch.equal(flat, torch.cat([params[k].grad.reshape(-1) for k in KEYS]))
    opt = slam.initialize_optimizer(params, slam.REPLICA_MAPPING['lrs'], tracking=False)
    opt.step()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             **{f"g_{k}": params[k].grad.numpy() for k in KEYS}, **{f"p_{k}": params[k].detach().numpy() for k in KEYS})
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharded_mapping_step_equals_gradient_accumulation(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{i}.npz") for i in (0, 1))
    # single process: accumulate both views, average
    slam_renderer = slam.Renderer
    slam.Renderer = R.OracleRasterizer
    try:
        params, variables, frames = _scene()
        acc = {k: torch.zeros_like(params[k]) for k in KEYS}
        for t in (2, 3):
            _view_grads(params, variables, frames[t], t)
            for k in KEYS:
                acc[k] += params[k].grad
    finally:
        slam.Renderer = slam_renderer
    for k in KEYS:
        want = (acc[k] / 2).numpy()
        np.testing.assert_allclose(r0[f"g_{k}"], want, rtol=1e-5, atol=1e-7 * (np.abs(want).max() + 1))
        np.testing.assert_array_equal(r0[f"g_{k}"], r1[f"g_{k}"])      # replicas stay bit-identical
        np.testing.assert_array_equal(r0[f"p_{k}"], r1[f"p_{k}"])


def test_single_process_bucket_is_a_noop(monkeypatch):
    monkeypatch.setattr(slam, "Renderer", R.OracleRasterizer)
    params, _, _ = _scene()
    for k in KEYS:
        params[k].grad = torch.ones_like(params[k])
    b = sdist.GradBucket(params)
    b.all_reduce_mean(params)
    assert all(torch.equal(params[k].grad, torch.ones_like(params[k])) for k in KEYS)
    assert sdist.shard_views(8, 3, 8) == [3] and sdist.shard_views(8, 1, 2) == [1, 3, 5, 7]


@pytest.mark.parametrize("rows,world", [(43, 8), (43, 1), (16, 3), (73, 8), (5, 5)])
def test_tile_row_bands_partition_the_grid(rows, world):
    """The bands of tile-row-sharded tracking: contiguous, disjoint, covering, balanced to one row."""
    bands = [sdist.tile_row_band(rows, r, world) for r in range(world)]
    assert bands[0][0] == 0 and bands[-1][1] == rows
    assert all(bands[r][1] == bands[r + 1][0] for r in range(world - 1))
    sizes = [e - b for b, e in bands]
    assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sdist.tile_row_band(rows, 0, rows + 1)


def _any_rank_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sdist.init_from_env(backend="gloo")
    got = [sdist.any_rank(rank == 1), sdist.any_rank(False), sdist.any_rank(True)]
    sums = torch.full((4,), float(rank + 1), dtype=torch.float64)
    sdist.all_reduce_sum_flat(sums)                              # the exchange of the tracking iteration's partial sums
    np.save(os.path.join(out_dir, f"any{rank}.npy"), np.array(got + [float(sums[0])]))
    dist.barrier()
    dist.destroy_process_group()


def test_any_rank_and_partial_sum_exchange(tmp_path):
    assert sdist.any_rank(True) is True and sdist.any_rank(False) is False      # single process: no collective
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_any_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        assert np.load(tmp_path / f"any{r}.npy").tolist() == [1.0, 0.0, 1.0, 3.0]
