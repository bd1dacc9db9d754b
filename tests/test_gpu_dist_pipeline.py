"""Two processes sharing ONE GPU over gloo: the multi-rank frame loop on the fused engine (splatam_amd/pipeline.py with
torch.distributed initialised) -- tracking sharded over tile rows (each rank composites its band, the 16 KB of partial sums are
all-reduced, every rank takes the same Adam step on the pose), view-sharded mapping with one gradient all-reduce per iteration,
replicated map edits.  Checked: the replicas stay identical in row counts and (to float-atomic summation order) in map and poses, and
the loop tracks the synthetic trajectory as well as the single-process loop."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

W, H, F_ = 160, 112, 140.0
FRAMES = 3


def _run(out_path):
    from splatam_amd import pipeline
    torch.manual_seed(0)
    np.random.seed(0)
    ds = pipeline.SyntheticRGBDSequence(6000, W, H, F_, F_, W / 2 - 0.5, H / 2 - 0.5, num_frames=FRAMES, seed=2, step_m=0.012, step_deg=0.4)
    cfg = pipeline.replica_config(tracking_iters=12, mapping_iters=12, keyframe_every=1)
    params, variables, st = pipeline.rgbd_slam(ds, cfg, engine="fused")
    torch.cuda.synchronize()
    err = max(float((pipeline._est_w2c(params, t)[:3, 3] - ds.gt_w2c(t)[:3, 3]).norm()) for t in range(FRAMES))
    np.savez(out_path, err=err, n=np.array(st['num_gaussians']), redone=st['redone_iterations'],
             **{k: v.detach().cpu().numpy() for k, v in params.items()})


def _worker(rank, world, port, out_dir, backend="gloo"):
    from splatam_amd import dist as sdist
    local = rank if backend == "nccl" else 0                    # RCCL: one GPU per rank; gloo: both ranks share GPU 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sdist.init_from_env(backend=backend)
    torch.cuda.set_device(local)
    assert dist.get_backend() == backend
    _run(os.path.join(out_dir, f"rank{rank}.npz"))
    dist.barrier()
    dist.destroy_process_group()


def _check_two_ranks(tmp_path, backend):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), backend), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{i}.npz") for i in (0, 1))
    assert r0['n'].tolist() == r1['n'].tolist()
    # the pose is the same on both ranks bit for bit (same summed partial sums -> same Adam step; broadcast after tracking)
    np.testing.assert_array_equal(r0['cam_unnorm_rots'], r1['cam_unnorm_rots'])
    np.testing.assert_array_equal(r0['cam_trans'], r1['cam_trans'])
    for k in ('means3D', 'rgb_colors', 'logit_opacities', 'log_scales'):
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)           # one all-reduced gradient, one Adam step
    _run(str(tmp_path / "single.npz"))
    single = np.load(tmp_path / "single.npz")
    # 12 tracking + 12 mapping iterations per frame at 160x112: a sanity bound either way (bench.py slam_loop: sub-millimetre at size)
    assert float(r0['err']) <= 2.0 * float(single['err']) + 2e-3, (float(r0['err']), float(single['err']))
    assert float(r0['err']) < 0.02


def test_two_rank_fused_frame_loop_with_sharded_tracking(tmp_path):
    _check_two_ranks(tmp_path, "gloo")


def test_rccl_two_ranks(tmp_path):
    """The same frame loop over RCCL (backend "nccl": ReduceOp.AVG inside the collective, one GPU per rank, init_from_env's
    set_device path) -- runs by itself wherever two GPUs are visible (the driver's multi-GPU node); the 1-GPU development boxes skip it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _check_two_ranks(tmp_path, "nccl")


def _rccl_ops_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    assert dist.get_backend() == "nccl"
    # every (collective, reduction, dtype) the multi-rank paths issue through torch.distributed -- splatam_amd/dist.py, bench.py:
    bucket = torch.arange(16 + 3_600_000, dtype=torch.float32, device=dev)     # the gradient bucket with its 16-float header: AVG
    want = bucket.clone()
    dist.all_reduce(bucket, op=dist.ReduceOp.AVG)
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    sums = torch.full((64 * 32,), 1.5, dtype=torch.float64, device=dev)         # tracking's partial sums (doubles): SUM
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    folded = sums[:32]                                                           # ... folded: 256 bytes, a view of the same buffer
    dist.all_reduce(folded, op=dist.ReduceOp.SUM)
    flag = torch.tensor([1], dtype=torch.int32, device=dev)                     # any_rank / all_agree: MAX / MIN of int32
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    rows = torch.tensor([816000, -816000], dtype=torch.int64, device=dev)       # assert_replicated_count / max_int: int64
    dist.all_reduce(rows, op=dist.ReduceOp.MIN)
    dist.all_reduce(rows, op=dist.ReduceOp.MAX)
    t64 = torch.tensor([1.25, 2.5], dtype=torch.float64, device=dev)            # bench.py: elapsed times, MAX of doubles
    dist.all_reduce(t64, op=dist.ReduceOp.MAX)
    pose = torch.arange(7, dtype=torch.float32, device=dev)                     # broadcast_pose
    dist.broadcast(pose, src=0)
    box = [b"x" * 128]                                                          # the in-stream communicator's id
    dist.broadcast_object_list(box, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    ok = (torch.equal(bucket, want) and float(sums.min()) == 1.5 and int(flag[0]) == 1 and rows.tolist() == [816000, -816000]
          and t64.tolist() == [1.25, 2.5] and torch.equal(pose, torch.arange(7, dtype=torch.float32, device=dev)) and box[0] == b"x" * 128)
    # ... and the library's own entry points on top of that group (what a multi-rank job calls them with)
    from splatam_amd import dist as sdist
    assert sdist.world_size() == 1 and sdist.get_rank() == 0
    sdist.all_reduce_mean_flat(bucket)
    sdist.all_reduce_sum_flat(sums)
    assert sdist.any_rank(True, dev) and not sdist.any_rank(False, dev) and sdist.max_int(7, dev) == 7
    sdist.assert_replicated_count(5, "test", dev)
    np.savez(os.path.join(out_dir, "rccl_ops.npz"), ok=ok)
    dist.destroy_process_group()


def test_rccl_accepts_every_collective_the_multi_rank_paths_issue(tmp_path):
    """No development box has two GPUs, so RCCL has never carried the multi-rank job.  What ONE GPU can settle: that this torch /
    RCCL build accepts every collective, reduction and dtype the multi-rank paths issue (ReduceOp.AVG of floats, SUM of doubles, MIN /
    MAX of int32 / int64, broadcast, object broadcast, barrier) on the "nccl" backend -- a one-rank group runs them through RCCL -- and
    returns the buffers unchanged.  In a child process: the process group must not leak into the other tests."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_ops_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert bool(np.load(tmp_path / "rccl_ops.npz")['ok'])


def test_instream_rccl_single_rank():
    """splatam_amd.dist.InStreamRccl (ncclAllReduce on the caller's stream, own communicator): what one GPU can show -- the
    communicator initialises, the collective is ordered with the kernels queued before and after it on the SAME non-default stream
    without any synchronisation, sums of doubles (the tracking partial sums) and averages of floats (the gradient bucket) come back
    unchanged with one rank."""
    from splatam_amd.dist import InStreamRccl
    dev = torch.device("cuda", 0)
    comm = InStreamRccl(0, 1)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        sums = torch.zeros(64 * 32, dtype=torch.float64, device=dev)
        for k in range(50):                         # queued back to back: add, all-reduce, add, ...
            sums += 1.0
            comm.all_reduce(sums, InStreamRccl.SUM)
        grads = torch.arange(3_600_000, dtype=torch.float32, device=dev)
        grads *= 2.0
        comm.all_reduce(grads, InStreamRccl.AVG)
        grads += 1.0
    side.synchronize()
    assert float(sums.min()) == 50.0 and float(sums.max()) == 50.0
    assert torch.equal(grads, torch.arange(3_600_000, dtype=torch.float32, device=dev) * 2.0 + 1.0)
    with pytest.raises(ValueError):
        comm.all_reduce(torch.zeros(4, 4, device=dev).t())
    with pytest.raises(ValueError):
        comm.all_reduce(torch.zeros(4, dtype=torch.float16, device=dev))
    comm.close()


def _overflow_worker(rank, world, port, out_dir):
    """Engine-level multi-rank mapping steps where ONLY rank 1's bucketed lists overflow (its bucket stride is cut below its longest
    list before step 2): the flag travels in the header of the gradient exchange (FusedEngine.exchange_gradients), so BOTH ranks skip
    steps 2 and 3."""
    from splatam_amd import dist as sdist
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sdist.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    n, f = 20000, 140.0
    params, variables = slam.synthetic_params(n, W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, num_frames=3, seed=3, device="cuda")
    w2c = torch.eye(4, device="cuda")
    cam = slam.setup_camera(W, H, [[f, 0, W / 2 - 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]], np.eye(4, dtype=np.float32), device="cuda")
    frames = []
    for t in (1, 2):                                           # two views: rank r renders view r
        im, depth = slam.synthetic_frame(params, cam, w2c, t, rot_deg=0.4 * t, trans_m=0.01 * t)
        frames.append({'cam': cam, 'im': im, 'depth': depth, 'id': t, 'w2c': w2c})
    cfg = slam.REPLICA_MAPPING
    eng = FusedEngine(params, cam)
    snaps, flags = [], []
    for step in range(4):
        if step == 1:
            assert not sdist.any_rank(eng.check_overflow(), eng.dev)         # learns the lists: bucketed from now on
            assert eng.tile_stride > 0
        if step == 2 and rank == 1:
            assert eng.max_list_hint > 256, eng.max_list_hint
            eng.tile_stride = 256                                 # rank 1's buckets no longer hold its longest list
        eng.mapping_iteration(frames[rank], frames[rank]['id'], cfg, bucket_allreduce=sdist.all_reduce_mean_flat)
        torch.cuda.synchronize()
        snaps.append({k: params[k].detach().cpu().numpy().copy() for k in ('means3D', 'rgb_colors', 'logit_opacities', 'log_scales')})
        flags.append(float(eng.buf['d_cam'][12]))
    local_bad = eng.check_overflow()
    np.savez(os.path.join(out_dir, f"ov{rank}.npz"), flags=np.array(flags), local_bad=local_bad, skipped=eng.skipped_iterations,
             **{f"s{i}/{k}": v for i, s in enumerate(snaps) for k, v in s.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_overflow_stops_every_replica(tmp_path):
    """the skip-on-overflow gate must be global in multi-rank mapping -- ranks render different views, so
    typically only some overflow; the others must not step on the all-reduced gradient that holds a truncated-list contribution."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_overflow_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"ov{i}.npz") for i in (0, 1))
    assert r0['flags'].tolist() == r1['flags'].tolist() == [0.0, 0.0, 1.0, 1.0]        # rank 0 never overflowed itself
    assert bool(r0['local_bad']) and bool(r1['local_bad'])
    assert int(r1['skipped']) == 2 and int(r0['skipped']) >= 1
    for k in ('means3D', 'rgb_colors', 'logit_opacities', 'log_scales'):
        for i in range(4):
            np.testing.assert_array_equal(r0[f"s{i}/{k}"], r1[f"s{i}/{k}"], err_msg=f"step {i} {k}")     # replicas bit-identical
        assert not np.array_equal(r0[f"s0/{k}"], r0[f"s1/{k}"]), k                                     # steps 0, 1 moved the map
        np.testing.assert_array_equal(r0[f"s1/{k}"], r0[f"s3/{k}"], err_msg=k)                         # steps 2, 3 moved nothing
