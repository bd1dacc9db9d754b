"""world_size-8 gloo tests (CPU): BASELINE configs 3 and 5 are 8-rank configurations, so the multi-rank paths are exercised at 8 ranks,
not only at 2 (tests/test_dist_gloo.py, tests/test_dist_pipeline_gloo.py):

  * view-sharded mapping, 8 keyframe views over 8 ranks, one each (config C, bench.py `--workload C`): the all-reduced mean equals
    single-process accumulation over the same 8 views, every replica takes the same Adam step;
  * tile-row-sharded tracking with 8 bands on the 43-row grid of a 680-pixel frame and the 73-row grid of a 1168-pixel frame (bands
    of 5-6 and 9-10 tile rows): the all-reduced partial sums -- loss and pose gradient -- equal the whole frame's;
  * the frame loop with 8 ranks: collective decisions (pose broadcast, replicated row counts, 8 views per mapping iteration from one
    random stream), replicas bit-identical at the end.

The rasterizer inside get_loss is the C oracle; each rank runs single-threaded (this container has 8 cores)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_ref
from oracle import raster_ref as R
from splatam_amd import dist as sdist
from splatam_amd import pipeline, slam

WORLD = 8
KEYS = sdist.GAUSSIAN_KEYS


def _spawn(fn, tmp_path, *args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(fn, args=(WORLD, port, str(tmp_path)) + args, nprocs=WORLD, join=True)


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    slam.Renderer = c_ref.CRasterizer
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)


# ---------------------------------------------------------------------------------------------------------------- mapping, 8 views
def _mapping_scene():
    W, H, f = 64, 48, 60.0
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    params, variables = slam.synthetic_params(800, W, H, f, f, cx, cy, num_frames=WORLD + 1, seed=3, device="cpu")
    cam = R.make_camera(W, H, f, f, cx, cy)
    w2c = torch.eye(4)
    g = torch.Generator().manual_seed(0)
    frames = {}
    for t in range(1, WORLD + 1):                                # 8 keyframe poses on an arc
        im, depth = slam.synthetic_frame(params, cam, w2c, t, rot_deg=0.3 * t, trans_m=0.01 * t)
        im = (im + 0.05 * torch.randn(im.shape, generator=g)).clamp(0, 1)
        frames[t] = {'cam': cam, 'im': im, 'depth': depth, 'id': t, 'w2c': w2c}
    return params, variables, frames


def _view_grads(params, variables, frame, t):
    for p in params.values():
        p.grad = None
    loss, _, _ = slam.get_loss(params, frame, variables, t, slam.REPLICA_MAPPING['loss_weights'], False, 0.5, True, False, mapping=True)
    loss.backward()


def _mapping_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    params, variables, frames = _mapping_scene()
    views = list(range(1, WORLD + 1))
    mine = [views[i] for i in sdist.shard_views(len(views), rank, world)]
    assert mine == [rank + 1]
    _view_grads(params, variables, frames[mine[0]], mine[0])
    flat = torch.cat([params[k].grad.reshape(-1) for k in KEYS]).clone()
    sdist.all_reduce_mean_flat(flat)                              # the fused engine's exchange: one flat buffer
    sdist.GradBucket(params).all_reduce_mean(params)             # the drop-in path's bucket
    assert torch.equal(flat, torch.cat([params[k].grad.reshape(-1) for k in KEYS]))
    opt = slam.initialize_optimizer(params, slam.REPLICA_MAPPING['lrs'], tracking=False)
    opt.step()
    np.savez(os.path.join(out_dir, f"map{rank}.npz"), **{f"g_{k}": params[k].grad.numpy() for k in KEYS},
             **{f"p_{k}": params[k].detach().numpy() for k in KEYS})
    dist.barrier()
    dist.destroy_process_group()


def test_eight_views_over_eight_ranks_equal_single_process_accumulation(tmp_path, monkeypatch):
    _spawn(_mapping_worker, tmp_path)
    ranks = [np.load(tmp_path / f"map{i}.npz") for i in range(WORLD)]
    monkeypatch.setattr(slam, "Renderer", c_ref.CRasterizer)
    params, variables, frames = _mapping_scene()
    acc = {k: torch.zeros_like(params[k]) for k in KEYS}
    for t in range(1, WORLD + 1):
        _view_grads(params, variables, frames[t], t)
        for k in KEYS:
            acc[k] += params[k].grad
    for k in KEYS:
        want = (acc[k] / WORLD).numpy()
        np.testing.assert_allclose(ranks[0][f"g_{k}"], want, rtol=2e-5, atol=2e-7 * (np.abs(want).max() + 1))
        for r in ranks[1:]:
            np.testing.assert_array_equal(ranks[0][f"g_{k}"], r[f"g_{k}"])      # replicas bit-identical
            np.testing.assert_array_equal(ranks[0][f"p_{k}"], r[f"p_{k}"])


# ---------------------------------------------------------------------------------------------------------------- tracking, 8 bands
def _tracking_scene(H):
    W, f = 48, 90.0
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    params, variables = slam.synthetic_params(3000, W, H, f, f, cx, cy, num_frames=3, seed=5, device="cpu")
    cam = R.make_camera(W, H, f, f, cx, cy)
    w2c = torch.eye(4)
    im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.4, trans_m=0.01)
    return params, variables, {'cam': cam, 'im': im, 'depth': depth, 'id': 1, 'w2c': w2c}


def _band_sums(params, variables, frame, rows):
    """Tracking loss and pose gradient of the pixels of tile rows [rows[0], rows[1]): every term of the tracking loss is a sum over
    the pixels with a valid measured depth (scripts/splatam.py:256-288), so zeroing the measured depth outside the band restricts it."""
    cfg = slam.REPLICA_TRACKING
    d = torch.zeros_like(frame['depth'])
    d[:, rows[0] * 16:rows[1] * 16] = frame['depth'][:, rows[0] * 16:rows[1] * 16]
    for p in params.values():
        p.grad = None
    loss, _, _ = slam.get_loss(params, dict(frame, depth=d), dict(variables), 1, cfg['loss_weights'], cfg['use_sil_for_loss'],
                               cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'], tracking=True)
    loss.backward()
    return torch.cat((loss.detach().reshape(1).double(), params['cam_unnorm_rots'].grad[0, :, 1].double(),
                      params['cam_trans'].grad[0, :, 1].double()))


def _tracking_worker(rank, world, port, out_dir, H):
    _init(rank, world, port)
    params, variables, frame = _tracking_scene(H)
    band = sdist.tile_row_band((H + 15) // 16, rank, world)
    sums = _band_sums(params, variables, frame, band)
    sdist.all_reduce_sum_flat(sums)                               # the exchange of the sharded tracking iteration
    np.save(os.path.join(out_dir, f"track{rank}.npy"), np.concatenate((np.array(band, dtype=np.float64), sums.numpy())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,rows,sizes", [(680, 43, (5, 6)), (1168, 73, (9, 10))])
def test_eight_bands_of_tile_rows_sum_to_the_whole_frame(tmp_path, monkeypatch, H, rows, sizes):
    _spawn(_tracking_worker, tmp_path, H)
    got = [np.load(tmp_path / f"track{i}.npy") for i in range(WORLD)]
    bands = [(int(g[0]), int(g[1])) for g in got]
    assert bands[0][0] == 0 and bands[-1][1] == rows and all(bands[i][1] == bands[i + 1][0] for i in range(WORLD - 1))
    assert set(e - b for b, e in bands) == set(sizes)
    monkeypatch.setattr(slam, "Renderer", c_ref.CRasterizer)
    params, variables, frame = _tracking_scene(H)
    whole = _band_sums(params, variables, frame, (0, rows)).numpy()
    for g in got:
        np.testing.assert_array_equal(g[2:], got[0][2:])                        # every rank holds the same sums: the same Adam step
    np.testing.assert_allclose(got[0][2:], whole, rtol=2e-5, atol=1e-6 * np.abs(whole).max())
    assert np.abs(whole[1:]).max() > 0


# ---------------------------------------------------------------------------------------------------------------- the frame loop
def _loop_run(out_path):
    torch.manual_seed(0)
    np.random.seed(0)
    W, H, f = 64, 48, 60.0
    ds = pipeline.SyntheticRGBDSequence(2500, W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, num_frames=3, seed=3, device="cpu")
    cfg = pipeline.replica_config(tracking_iters=4, mapping_iters=3, keyframe_every=1)
    cfg['mapping']['pruning_dict'] = dict(cfg['mapping']['pruning_dict'], stop_after=2, prune_every=2)
    params, variables, st = pipeline.rgbd_slam(ds, cfg, engine="dropin")
    np.savez(out_path, n=np.array(st['num_gaussians']), views=np.array([d['views'] for d in st['decisions']]),
             **{k: v.detach().numpy() for k, v in params.items()})


def _loop_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    _loop_run(os.path.join(out_dir, f"loop{rank}.npz"))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_frame_loop_keeps_replicas_identical(tmp_path):
    _spawn(_loop_worker, tmp_path)
    ranks = [np.load(tmp_path / f"loop{i}.npz") for i in range(WORLD)]
    for r in ranks[1:]:
        assert ranks[0]['n'].tolist() == r['n'].tolist()
        for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales', 'cam_unnorm_rots', 'cam_trans'):
            np.testing.assert_array_equal(ranks[0][k], r[k], err_msg=k)
    # every mapping iteration drew 8 views from ONE random stream: rank r rendered the r-th
    views = np.stack([r['views'] for r in ranks])                # [rank, frame, iteration]
    assert views.shape == (WORLD, 3, 3)
    assert len({tuple(v.reshape(-1)) for v in views}) > 1         # the ranks did render different views
    assert (views[:, 0] == 0).all()                               # frame 0: only itself to choose from
