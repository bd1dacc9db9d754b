#!/usr/bin/env python3
"""Benchmark of the SplaTAM hot path on MI355X.

A *step* is one optimisation iteration of the reference's per-frame loops
(/root/reference/scripts/splatam.py:690-711 tracking, :828-869 mapping):
``get_loss`` (two rasterizer forwards) -> ``backward`` (two rasterizer backwards)
-> Adam step, at BASELINE.json config B: 300 000 Gaussians, 1200x680, Replica
intrinsics, synthetic data.  Steps follow the reference's 40:60 tracking:mapping
mix (2 tracking + 3 mapping per 5 steps).  With N > 1 ranks every rank runs the
same schedule (weak scaling in the mapping views): mapping steps render one view per
rank and exchange Gaussian gradients with one RCCL all-reduce; a tracking step is ONE
iteration on one frame whose tile rows are sharded over the ranks (counted once).

Three ways through the same iteration (same loss, same gradients, same Adam update: tests/test_gpu_fused.py, tests/test_gpu_plugin.py):
  * ``fused``  (default, the headline ``value``): splatam_amd.fused.FusedEngine -- the whole iteration as 4 (tracking) / 7 (mapping)
    kernel launches of libsplat_hip.so (shared geometry, ONE 6-channel composite for the RGB and depth/silhouette renders, the
    tracking iteration's forward composite + loss + backward composite as one kernel, fused SSIM / pose-gradient / Adam kernels, no host
    synchronisation);
  * ``plugin_iters_per_s``: the reference's own loop statements with ``splatam_amd.plugin.install()`` (get_loss / initialize_optimizer
    of the caller's module replaced at run time by adapters over the fused engine; no host read per iteration);
  * ``dropin``: the reference's own Python glue (splatam_amd.slam, PyTorch autograd + torch.optim.Adam) around the
    drop-in ``GaussianRasterizer`` -- what an unmodified scripts/splatam.py gets; reported as ``dropin_iters_per_s`` with the
    rasterizer's own share of the iteration beside it.
Beside them: ``slam_loop`` / ``slam_loop_plugin`` (the whole frame loop, 13 frames, two runs, per-phase milliseconds), ``roofline`` (the
dominant kernel live + counters of the committed profile set), ``cpu_baseline`` (the C oracle + PyTorch-CPU glue on the host cores).
``python bench.py --gpus N`` started without a launcher starts its own N ranks (torch.distributed.run, 127.0.0.1).
Timing: W untimed warm-up steps, then EXACTLY K steps between barrier + torch.cuda.synchronize() on both sides, maximum over the ranks.  In
front of the W warm-up steps the same schedule runs for >= --prewarm-s seconds (state restored afterwards; ``prewarm`` in the JSON line): an
MI355X that idled through the set-up runs its first ~25 ms of work 10-15 % slow, and K = 20 steps are 5 ms (profiles/r05_experiments.md 10).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (N, W, H, fx, fy, cx, cy)
    "A": (10_000, 320, 240, 300.0, 300.0, 159.5, 119.5),
    "B": (300_000, 1200, 680, 600.0, 600.0, 599.5, 339.5),       # /root/reference/configs/data/replica.yaml:3-8
    "D": (150_000, 640, 480, 517.3, 516.5, 318.6, 255.3),        # /root/reference/configs/data/TUM/freiburg1_desk.yaml:3-8
    "E": (1_000_000, 1752, 1168, 1200.0, 1200.0, 875.5, 583.5),  # /root/reference/datasets/gradslam_datasets/scannetpp.py:28-29
    # SURVEY.md 8(d) stress variants of E: every Gaussian inside 5 % of the image -> per-tile lists of ~5 000 (1 M) / ~25 000 (5 M)
    # entries, far beyond the 4 096 keys one workgroup sorts in LDS ("per-tile Gaussian list spilling HBM", BASELINE config 5)
    # B at four times the frame and the Gaussians (same density: the same per-tile work over 4x the tiles): what launch ramp /
    # tail effects cost the composites at B (profiles/r02_experiments.md)
    "B-4x": (1_200_000, 2400, 1360, 600.0, 600.0, 1199.5, 679.5),
    # the map SplaTAM really builds at B's frame size: one Gaussian per valid first-frame pixel, in pixel (creation) order, opacity 0.5,
    # scale from the projective mean squared distance (initialize_first_timestep, /root/reference/scripts/splatam.py:169-211)
    "B-loop": (816_000, 1200, 680, 600.0, 600.0, 599.5, 339.5),
    "E-clustered": (1_000_000, 1752, 1168, 1200.0, 1200.0, 875.5, 583.5),
    "E-clustered-5M": (5_000_000, 1752, 1168, 1200.0, 1200.0, 875.5, 583.5),
}
REGIONS = {"E-clustered": (0.40, 0.40, 0.6236, 0.6236), "E-clustered-5M": (0.40, 0.40, 0.6236, 0.6236)}
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_scene_loop(name, dev, n_views, seed=3):
    """Workload B-loop: the first frame of a synthetic RGB-D sequence (a smooth textured surface, splatam_amd.pipeline) turned into the
    map exactly as the frame loop does it (initialize_first_timestep); frame 1 is tracked, frames 2.. are keyframes at their true poses."""
    from splatam_amd import pipeline
    N, W, H, fx, fy, cx, cy = WORKLOADS[name]
    num_frames = 2 + n_views
    ds = pipeline.SyntheticRGBDSequence(300_000, W, H, fx, fy, cx, cy, num_frames=num_frames, seed=seed, device=dev)
    params, variables, intrinsics, w2c0, cam = pipeline.initialize_first_timestep(ds, num_frames, 3.0, "projective", "isotropic", device=dev)
    frames = {}
    for t in range(1, num_frames):
        color, depth, _, pose = ds[t]
        frames[t] = {'cam': cam, 'im': (color.permute(2, 0, 1) / 255).contiguous(), 'depth': depth.permute(2, 0, 1).contiguous(), 'id': t, 'w2c': w2c0}
        if t > 1:
            rel = torch.linalg.inv(pose)            # world (first camera) -> camera t
            with torch.no_grad():
                params['cam_unnorm_rots'][..., t] = pipeline._matrix_to_quaternion(rel[:3, :3]).to(dev)
                params['cam_trans'][..., t] = rel[:3, 3].reshape(1, 3)
    return params, variables, frames, (int(params['means3D'].shape[0]), W, H)


def build_scene(name, dev, n_views, seed=0):
    from splatam_amd import slam
    if name == "B-loop":
        return build_scene_loop(name, dev, n_views)
    N, W, H, fx, fy, cx, cy = WORKLOADS[name]
    num_frames = 2 + n_views
    params, variables = slam.synthetic_params(N, W, H, fx, fy, cx, cy, num_frames=num_frames, seed=seed, device=dev,
                                              region=REGIONS.get(name))
    k = [[fx, 0, cx], [0, fy, cy], [0, 0, 1]]
    first_w2c = torch.eye(4, device=dev)
    cam = slam.setup_camera(W, H, k, first_w2c.cpu().numpy(), device=dev)
    g = torch.Generator().manual_seed(seed + 7)
    frames = {}
    # frame 1 is the frame being tracked; frames 2.. are keyframe views on a small arc
    for t in range(1, num_frames):
        rot = 0.5 if t == 1 else 0.3 * (t - 1)
        tr = 0.01 if t == 1 else 0.02 * (t - 1)
        im, depth = slam.synthetic_frame(params, cam, first_w2c, t, rot_deg=rot, trans_m=tr)
        if t > 1:   # keyframes: the pose is known (set it), add sensor noise so that mapping has a gradient
            import math
            with torch.no_grad():
                ang = math.radians(rot)
                params['cam_unnorm_rots'][..., t] = torch.tensor([[math.cos(ang / 2), 0.0, math.sin(ang / 2), 0.0]], device=dev)
                params['cam_trans'][..., t] = torch.tensor([[tr, -tr / 2, tr / 2]], device=dev)
            im = (im + 0.05 * torch.randn(im.shape, generator=g).to(dev)).clamp(0, 1)
            depth = depth * (1 + 0.01 * torch.randn(depth.shape, generator=g).to(dev))
        frames[t] = {'cam': cam, 'im': im, 'depth': depth, 'id': t, 'w2c': first_w2c}
    return params, variables, frames, (N, W, H)


def run_steps(params, variables, frames, bucket, rank, world, nsteps, opt_track, opt_map, track_state, start=0):
    from splatam_amd import slam
    n_views = len(frames) - 1
    for i in range(start, start + nsteps):
        if i % 5 < 2:
            slam.tracking_iteration(params, frames[1], variables, 1, opt_track, track_state)
        else:
            view = 2 + (rank + i * world) % n_views
            data = frames[view]
            loss, variables, _ = slam.get_loss(params, data, variables, view, slam.REPLICA_MAPPING['loss_weights'],
                                               False, 0.5, True, False, mapping=True)
            loss.backward()
            if world > 1:
                bucket.all_reduce_mean(params)
            with torch.no_grad():
                opt_map.step()
                opt_map.zero_grad(set_to_none=True)
    return variables


SHARD_TRACKING = True       # --replicated-tracking clears it


def run_steps_fused(eng, frames, rank, world, nsteps, start=0):
    from splatam_amd import slam
    n_views = len(frames) - 1

    from splatam_amd.dist import all_reduce_mean_flat as allreduce
    from splatam_amd.dist import all_reduce_sum_flat as allreduce_sums
    for i in range(start, start + nsteps):
        if i % 5 < 2:
            # several ranks: the frame's tile rows are sharded over them (one 16 KB all-reduce of the partial sums per iteration)
            eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING, shard=(rank, world) if (world > 1 and SHARD_TRACKING) else None,
                                   allreduce_sums=allreduce_sums)
        else:
            view = 2 + (rank + i * world) % n_views
            eng.mapping_iteration(frames[view], view, slam.REPLICA_MAPPING, allreduce if world > 1 else None)


def run_steps_views(eng, frames, rank, world, nsteps, batch_views=8):
    """BASELINE config 3 ("mapping-only, 8 keyframe views sharded over the GPUs"): ONE step = one mapping step over a fixed batch
    of `batch_views` keyframe views; rank r renders views r, r + world, ... of the batch and accumulates their gradients, one
    all-reduce (sum) follows, every rank divides by the batch size and takes the identical Adam step.  Strong scaling: N = 1
    accumulates all eight views, N = 8 renders one each."""
    from splatam_amd import slam
    from splatam_amd.dist import all_reduce_sum_flat
    views = [(frames[2 + v], 2 + v) for v in range(rank, batch_views, world)]
    for _ in range(nsteps):
        eng.mapping_batch(views, slam.REPLICA_MAPPING, total_views=batch_views, allreduce_sum=all_reduce_sum_flat if world > 1 else None)


def phase_rate(fn, n, dev, repeat=3):
    """Calls per second of `fn` over n calls; the MEDIAN of `repeat` such windows (the secondary figures of the JSON line are windows of
    5 - 25 ms: one stall of the box -- 37 ms inside the 24 ms window of b_loop's mapping phase in one round-end check, 650 it/s where every
    other run reads 1 650 -- must not become the figure.  `value` itself is the contract's K steps, timed once)."""
    rates = []
    for _ in range(repeat):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        rates.append(n / (time.perf_counter() - t0))
    return sorted(rates)[len(rates) // 2]


def kernel_roofline(params, frames, shape, dev):
    """Live HIP-event timing of the two composite kernels on the stream they are
    launched on (splat_time_kernel), priced against the algorithmic bytes of
    SURVEY.md 8(d): K6 = R*44 + HW*24, K7 = R*40 + HW*20 + N*44."""
    import ctypes as C
    from splatam_amd import _capi, slam
    from splatam_amd import rasterizer as rz
    N, W, H = shape
    with torch.no_grad():
        tg = slam.transform_to_frame(params, 1, False, False)
        rv = slam.transformed_params2rendervar(params, tg)
    empty = torch.empty(0, device=dev)
    args = (frames[1]['cam'], rv['means3D'].contiguous(), rv['colors_precomp'].detach().contiguous(),
            rv['opacities'].detach().reshape(-1).contiguous(), rv['scales'].detach().contiguous(),
            rv['rotations'].detach().contiguous(), empty, empty)
    mode = rz.get_sync_mode()
    rz.set_sync_mode("exact")
    color, radii, depth, pk = rz.rasterize_forward(*args)
    rz.set_sync_mode(mode)
    R = pk.num_rendered
    gcol = torch.randn_like(color)
    f32 = torch.float32
    bufs = dict(accum=torch.empty(N, _capi.SPLAT_GRAD_STRIDE, dtype=f32, device=dev), m3=torch.empty(N, 3, device=dev),
                m2=torch.empty(N, 3, device=dev), col=torch.empty(N, 3, device=dev), op=torch.empty(N, device=dev),
                sc=torch.empty(N, 3, device=dev), ro=torch.empty(N, 4, device=dev))
    gr = _capi.SplatGrads()
    gr.dL_dcolor, gr.accum = gcol.data_ptr(), bufs['accum'].data_ptr()
    gr.dL_dmeans3D, gr.dL_dmeans2D, gr.dL_dcolors = bufs['m3'].data_ptr(), bufs['m2'].data_ptr(), bufs['col'].data_ptr()
    gr.dL_dopacities, gr.dL_dscales, gr.dL_drotations = bufs['op'].data_ptr(), bufs['sc'].data_ptr(), bufs['ro'].data_ptr()
    L = _capi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for fn, name in ((0, "render_forward"), (1, "render_backward")):
        ms = C.c_float(0)
        for iters in (3, 20):     # warm-up, then measure
            _capi.check(L.splat_time_kernel(fn, iters, C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st), C.byref(gr),
                                            color.data_ptr(), depth.data_ptr(), stream, C.byref(ms)), "splat_time_kernel")
        out[name] = ms.value
    HW = W * H
    bytes_fwd = R * 44 + HW * 24
    bytes_bwd = R * 40 + HW * 20 + N * 44
    gbs_f = bytes_fwd / (out["render_forward"] * 1e-3) / 1e9
    gbs_b = bytes_bwd / (out["render_backward"] * 1e-3) / 1e9
    dominant = "render_backward" if out["render_backward"] >= out["render_forward"] else "render_forward"
    ach = gbs_b if dominant == "render_backward" else gbs_f
    roof = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": None, "kernel": dominant, "kernel_ms": round(out[dominant], 4),
            "algorithmic_bytes": bytes_bwd if dominant == "render_backward" else bytes_fwd,
            "other": {"render_forward_ms": round(out["render_forward"], 4), "render_forward_GBps": round(gbs_f, 2),
                      "render_backward_ms": round(out["render_backward"], 4), "render_backward_GBps": round(gbs_b, 2),
                      "num_rendered": R, "pairs_per_launch": None}}
    return roof, pk


def reference_instances(eng):
    """R of SURVEY.md 8(d)'s algorithmic bytes: the (Gaussian, tile) instances of the REFERENCE's binning rule (tile rectangle of
    ceil(3 sigma_max), Appendix A step 8), counted from the rectangles the per-Gaussian kernel stores.  The lists the fused iteration
    composites are shorter since round 6 (group binning files only the tiles that can hold a pixel with alpha >= 1/255: status[0]); the
    size of the problem is the reference's."""
    rect = eng.buf['rect'][:eng.P].view(-1, 2)
    vis = eng.buf['radii'][:eng.P] > 0
    w = (rect[:, 1] & 0xFFFF) - (rect[:, 0] & 0xFFFF)
    h = ((rect[:, 1] >> 16) & 0xFFFF) - ((rect[:, 0] >> 16) & 0xFFFF)
    return int((w * h * vis).sum())


def fused_roofline(eng, frames, shape, dev, workload="B"):
    """Live HIP-event timing of the two 6-channel composite kernels of the fused iteration on the stream they are
    launched on (splat_iter_time_kernel), in the learnt list state the loop runs in.
    Algorithmic bytes: per instance 4 (id) + 8 (xy) + 16 (conic, opacity) + 24 (six colours) = 52 B;
    K6: R*52 + HW*32 (six planes + final_T + n_contrib);  K7: R*52 + HW*32 (six gradient planes + final_T + n_contrib)
    + P*48 (twelve partial sums per Gaussian)."""
    import ctypes as C
    from splatam_amd import _capi, slam
    N, W, H = shape
    L = _capi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    eng.begin_tracking(1)
    # learn the list statistics first (bucketed lists, group binning, the composite sorts its own list): the state the loop runs in
    for _ in range(3):
        eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
        torch.cuda.synchronize(dev)
        eng.check_overflow()
    # steady clocks first: the kernels below are timed over a few milliseconds each, and a GPU that idled through the scene set-up runs its
    # first ~25 ms of work 10-15 % slow (profiles/r05_experiments.md 10): >= 0.1 s of mapping iterations (gradients formed, no Adam step:
    # the map does not move)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.1:
        for _ in range(25):
            eng.loss_backward(frames[2 % len(frames)], 2 % len(frames), slam.REPLICA_MAPPING, tracking=False)
        torch.cuda.synchronize(dev)
    # the state the mapping loop runs in: learnt lists, the gradient planes of a mapping iteration
    eng.loss_backward(frames[2 % len(frames)], 2 % len(frames), slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize(dev)
    R, list_entries = reference_instances(eng), int(eng.buf['status'][0])
    ws = eng._workspace(False, False)
    # one kernel launched 30x in a row between two events on the iteration's stream (splat_iter_time_kernel): 0 the list-reading forward
    # composite (what the iteration launches on long lists), 1 the backward composite (mapping form), 2 the sorting forward composite
    # (what the iteration launches at this workload).  In this state the figures agree with rocprofv3's per-kernel averages of the
    # bench loop to ~2 % (round 2 timed them on the exact lists of a first iteration: K7 read 13 % long)
    b2b = {}
    for fn, name in ((0, "render_forward_list_reading"), (1, "render_backward"), (2, "render_forward_sorting"), (3, "forward_backward_pair")):
        ms = C.c_float(0)
        rc = 0
        for iters in (5, 30):     # warm-up, then measure
            rc = L.splat_iter_time_kernel(fn, iters, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms))
        b2b[name] = ms.value if rc == 0 else None
    if b2b["render_backward"] is None or b2b["render_forward_list_reading"] is None:
        raise RuntimeError("splat_iter_time_kernel failed")
    out = {"render_forward": b2b["render_forward_sorting"] or b2b["render_forward_list_reading"]}
    # the backward composite between other kernels, as in the loop: the alternating pair minus the forward composite alone
    out["render_backward"] = (b2b["forward_backward_pair"] - out["render_forward"]) if b2b["forward_backward_pair"] else b2b["render_backward"]
    HW = W * H
    bytes_fwd = R * 52 + HW * 32
    bytes_bwd = R * 52 + HW * 32 + N * 48
    gbs_f = bytes_fwd / (out["render_forward"] * 1e-3) / 1e9
    gbs_b = bytes_bwd / (out["render_backward"] * 1e-3) / 1e9
    dominant = "render_backward" if out["render_backward"] >= out["render_forward"] else "render_forward"
    ach = gbs_b if dominant == "render_backward" else gbs_f
    # counters of the same kernels from the newest committed rocprofv3 --pmc passes (separate passes, corrected as
    # /opt/skills/guides/MI355X_MICROARCH.md prescribes); the file names its git head
    pmc = load_pmc(workload)
    k7 = pmc_kernel(pmc, "render_backward_kernel", "<6, 8, 15u, 15u")
    k6 = pmc_kernel(pmc, "render_forward_kernel", "<6, 8, false, true, false>") or pmc_kernel(pmc, "render_forward_kernel", "<6, 8")
    dom = k7 if dominant == "render_backward" else k6
    rows = {}
    trace = load_kernel_trace(workload)          # the committed kernel TRACE of the bench command: a cross-check of the live durations
    live = live_kernel_times(eng, frames, dev)   # per-kernel average durations of THIS run
    # counters may only describe the code they were taken on: the counter file and the trace must name the same git head (the profile
    # script writes <tag>_meta.json beside a set); otherwise the traffic columns are withheld
    trace_meta = load_profile_meta(trace[0]) if trace else None
    pmc_head = pmc[1].get("git_head") if pmc else None
    trace_head = trace_meta.get("git_head") if trace_meta else None
    counters_ok = bool(pmc_head) and pmc_head == trace_head
    counters_note = None if counters_ok else (f"counter file {pmc[0] if pmc else None} @ {pmc_head} and trace {trace[0] if trace else None} @ "
                                              f"{trace_head} are not one profile set: traffic / valu columns withheld")
    # (the tracking iteration's composites are ONE kernel: the list is gathered once -- R x 52 --, the frame is read -- 16 B per pixel --, the
    #  planes stay in registers, six partial sums per Gaussian leave it)
    # (the SSIM kernels are templates since the column-first form: a counter file taken on the row-first kernels does not describe them)
    per_unit = {"render_track_fused_kernel": R * 52 + HW * 16 + N * 24, "fused_preprocess_kernel": N * (48 + 87), "ssim_forward_kernel<": HW * (2 * 12 + 36 + 16), "map_loss_backward_kernel<": HW * (36 + 24 + 16),
                # F6, mapping step: accumulator row 64 + geometry 40 + gradients out 48 + (Adam inside) parameters, two moments in and out 12 x 4 x 5;
                # F6, tracking: accumulator row + geometry in, only the camera sums out
                "fused_backward_kernel<true, true": N * (64 + 40 + 48 + 12 * 4 * 5), "fused_backward_kernel<false, false": N * (64 + 40),
                "adam_map_kernel": N * 12 * 4 * 6}
    for kname, abytes in per_unit.items():
        d = (pmc_kernel(pmc, kname) or {}) if counters_ok else {}
        us_live, us_trace = live_avg_us(live, kname), trace_avg_us(trace, kname)
        us = us_live or us_trace
        kname = {"fused_backward_kernel<true, true": "fused_backward_kernel (mapping, Adam inside)",
                 "fused_backward_kernel<false, false": "fused_backward_kernel (tracking)"}.get(kname, kname.rstrip("<"))
        if us:
            rows[kname] = {"avg_us": round(us, 1), "avg_us_source": "live" if us_live else "committed trace",
                           "avg_us_committed_trace": None if us_trace is None else round(us_trace, 1),
                           "algorithmic_bytes": abytes, "GBps": round(abytes / us / 1e3, 1),
                           "frac_of_hbm_peak": round(abytes / us / 1e3 / HBM_PEAK_GBS, 4), "traffic_bytes": d.get("traffic_bytes"),
                           "valu_cycles_frac": d.get("valu_cycles_frac")}
    r4 = lambda v: None if v is None else round(v, 4)          # noqa: E731
    other = {"render_forward_ms": round(out["render_forward"], 4), "render_forward_GBps": round(gbs_f, 2),
             "render_backward_ms": round(out["render_backward"], 4), "render_backward_GBps": round(gbs_b, 2),
             "render_forward_is": "sorting form" if b2b["render_forward_sorting"] else "list-reading form",
             "render_forward_list_reading_ms": r4(b2b["render_forward_list_reading"]),
             "render_forward_sorting_ms": r4(b2b["render_forward_sorting"]),
             "render_backward_30_in_a_row_ms": r4(b2b["render_backward"]), "forward_backward_pair_ms": r4(b2b["forward_backward_pair"]),
             "num_rendered": R, "list_entries": list_entries,
             # secondary ceiling: live (pixel, Gaussian) pairs; filled from the oracle's count by the cpu_baseline leg
             "pairs_per_launch": None, "pair_evals_per_s": None,
             # filled with pairs_per_launch (cpu_baseline leg): vector lane-operations (64 x SQ_INSTS_VALU) per live pair, and the pair
             # rate against SURVEY.md 8(d)'s secondary ceiling (4.9e12 pair evaluations / s)
             "lane_ops_per_live_pair": None, "pair_evals_frac_of_survey_ceiling": None,
             "valu_cycles_frac": dom.get("valu_cycles_frac") if dom else None,
             "valu_insts_per_launch": dom.get("SQ_INSTS_VALU") if dom else None,
             "pmc_source": (f"profiles/{pmc[0]} @ {pmc[1].get('git_head')}" if pmc else None),
             "kernel_trace_source": (f"profiles/{trace[0]} @ {trace_head}" if trace else None),
             "counters_withheld": counters_note,
             "per_kernel_durations": "live (torch.profiler over this run's iterations)" if live else "committed trace (no tracer in this process)",
             "kernels": rows,
             "note": "K6 / K7 times are live HIP-event measurements of this run (30 launches in a row on the iteration's stream, learnt list "
                     "state, after a mapping iteration); counters (traffic = 2 x FETCH_SIZE + WRITE_SIZE as the microarchitecture guide prescribes: "
                     "an UPPER bound here -- the factor 2 is for wide streaming reads, these kernels mostly gather 8-16 bytes; raw FETCH + WRITE "
                     "and where they come from: profiles/r05_experiments.md 5) and the "
                     "per-kernel rows come from the committed rocprofv3 passes named in pmc_source.  valu_cycles_frac = vector-pipe cycles "
                     "of the kernel's instruction mix / (1024 SIMDs x kernel cycles) with the MEASURED issue costs of gfx950 "
                     "(profiles/r03_valu_issue_bench.txt, r03_visit_replay.txt: plain VALU 2 cycles per wave64 instruction, compares / "
                     "selects / DPP 4, exp / rcp ~20 in the visit's mix; round 2 priced every instruction at 4).  The composites are "
                     "bound by the vector pipe at ~21 % live lanes, not by HBM (DESIGN.md 5)"}
    return {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": dom.get("traffic_bytes") if (dom and pmc_head) else None, "kernel": dominant + "_kernel<6,8> (fused iteration)",
            "kernel_ms": round(out[dominant], 4), "algorithmic_bytes": bytes_bwd if dominant == "render_backward" else bytes_fwd,
            "other": other}


def live_kernel_times(eng, frames, dev, iters=20):
    """Per-kernel average durations of the fused iteration measured IN THIS RUN: torch.profiler (kineto over the ROCm tracer) around
    `iters` tracking and `iters` mapping iterations -- every kernel launched in the process is recorded, libsplat_hip.so's included.
    Returns {kernel name: average us} or None when the tracer is not available (the committed trace is then the only source)."""
    from splatam_amd import slam
    try:
        from torch.profiler import ProfilerActivity, profile
        eng.begin_tracking(1)
        torch.cuda.synchronize(dev)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(iters):
                eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
            for _ in range(iters):
                eng.mapping_iteration(frames[2 % len(frames)], 2 % len(frames), slam.REPLICA_MAPPING)
            torch.cuda.synchronize(dev)
        acc = {}
        for ev in prof.events():
            dt = getattr(ev, "device_time", None) or getattr(ev, "cuda_time", 0.0)
            if dt and "splat" in ev.name:
                acc.setdefault(ev.name, []).append(float(dt))
        return {k: sum(v) / len(v) for k, v in acc.items()} or None
    except Exception as exc:            # noqa: BLE001 (a missing tracer must not cost the bench line)
        print(f"live kernel timing unavailable: {exc!r}", file=sys.stderr)
        return None


def live_avg_us(live, needle):
    if not live:
        return None
    for name, us in live.items():
        if needle in name:
            return us
    return None


def load_profile_meta(path_basename):
    """profiles/<tag>_meta.json written by the profile script beside a set: {"git_head": ..., "files": [...]} -- the head a trace /
    counter file was taken at (counter files also carry their own)."""
    import glob
    for meta in glob.glob(os.path.join(ROOT, "profiles", "*_meta.json")):
        try:
            d = json.load(open(meta))
        except Exception:
            continue
        if path_basename in d.get("files", []):
            return d
    return None


def load_pmc(workload):
    """The newest profiles/*_pmc.json (scripts/pmc.sh -> scripts/pmc_to_json.py: rocprofv3 --pmc passes of the fused iteration,
    one file per profiled round, with the git head it was taken at) for `workload`, or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload:
            best = (os.path.basename(path), d)
    return best


def load_kernel_trace(workload):
    """The newest committed rocprofv3 --kernel-trace --stats CSV of the bench command at `workload`
    (profiles/<tag>_bench_kernel_stats.csv for B, profiles/<tag>_<workload>_kernel_stats.csv otherwise), or None."""
    import csv
    import glob
    pat = "*_bench_kernel_stats.csv" if workload == "B" else f"*_{workload.replace('B-loop', 'Bloop')}_kernel_stats.csv"
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", pat)) if "cluster" not in os.path.basename(p) or "cluster" in workload)
    if not paths:
        return None
    try:
        return os.path.basename(paths[-1]), list(csv.DictReader(open(paths[-1])))
    except Exception:
        return None


def trace_avg_us(trace, needle):
    if trace is None:
        return None
    for r in trace[1]:
        if needle in r.get("Name", ""):
            return float(r["AverageNs"]) / 1e3
    return None


def pmc_kernel(pmc, *needles):
    if pmc is None:
        return None
    for name, d in pmc[1]["kernels"].items():
        if all(n in name for n in needles):
            return dict(d, name=name)
    return None


def render_mpix(params, frames, shape, dev, reps=10):
    """Forward + backward of ONE 3-channel rasterizer call through the autograd surface."""
    from splatam_amd import slam
    from splatam_amd.rasterizer import GaussianRasterizer as Renderer
    N, W, H = shape
    with torch.no_grad():
        tg = slam.transform_to_frame(params, 1, False, False)
        rv = {k: v.detach() for k, v in slam.transformed_params2rendervar(params, tg).items()}
    inp = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    gout = torch.randn(3, H, W, device=dev)

    def once():
        im, _, _ = Renderer(raster_settings=frames[1]['cam'])(**inp)
        im.backward(gout)
        for v in inp.values():
            v.grad = None
    # warm-up: the scene's first call is the exact (list-learning) one, the next ones pay the first uses of the fast path (layouts, pinned
    # flag ring, allocator) and the clock ramp of an idle GPU (profiles/r05_experiments.md 10) -- 60 calls are 12 ms, all of it inside
    # that ramp: the same call read 0.248 ms here and 0.198 - 0.209 ms in scripts/r06_dropin.py on the same box (r06_v5).  The timed calls
    # include the policy's own exact call (every 64th)
    for _ in range(48):
        once()
    rate = phase_rate(once, reps, dev)
    return W * H * rate / 1e6, 1e3 / rate


REFERENCE_DIR = os.environ.get("SPLAT_REFERENCE_DIR", "/root/reference")


def cpu_baseline_reference(name, params, frames, budget_s=25.0, max_iters=5):
    """The cpu_baseline leg through the REFERENCE's own modules (utils/slam_helpers.py, utils/slam_external.py, get_loss /
    initialize_optimizer of scripts/splatam.py) around the C oracle: scripts/cpu_baseline_reference.py in a process of its own (its
    device shim patches torch), on the scene written to a temporary .npz.  None where the reference is not present (the GPU box)."""
    import subprocess
    import tempfile
    from splatam_amd import slam
    if not os.path.isdir(os.path.join(REFERENCE_DIR, "utils")):
        return None
    f = frames[1]
    cam = f['cam']
    cfg = slam.REPLICA_TRACKING
    blob = {f"param/{k}": v.detach().cpu().numpy() for k, v in params.items()}
    blob.update({f"lr/{k}": np.float64(v) for k, v in cfg['lrs'].items()})
    blob.update(H=cam.image_height, W=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=cam.bg.cpu().numpy(),
                viewmatrix=cam.viewmatrix.cpu().numpy(), projmatrix=cam.projmatrix.cpu().numpy(), campos=cam.campos.cpu().numpy(),
                im=f['im'].cpu().numpy(), depth=f['depth'].cpu().numpy(), time_idx=1, w_im=cfg['loss_weights']['im'],
                w_depth=cfg['loss_weights']['depth'], use_sil_for_loss=cfg['use_sil_for_loss'], sil_thres=cfg['sil_thres'],
                use_l1=cfg['use_l1'], ignore_outlier_depth_loss=cfg['ignore_outlier_depth_loss'])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "scene.npz")
        np.savez(path, **blob)
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "cpu_baseline_reference.py")
        env = dict(os.environ, OMP_NUM_THREADS=str(os.cpu_count() or 1))
        res = subprocess.run([sys.executable, script, path, str(budget_s), str(max_iters)], capture_output=True, text=True, env=env,
                             timeout=20 * budget_s + 600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        sys.stderr.write("cpu_baseline through the reference failed; falling back to the mirror\n" + res.stderr[-2000:] + "\n")
        return None
    d = json.loads(lines[-1])
    if "error" in d:
        return None
    return {"value": d["value"], "unit": "iters/s", "cores": os.cpu_count(), "kind": "reference glue + oracle",
            "sample": f"{d['iterations']} tracking iterations (2 fwd + 2 bwd rasterizations + host glue + Adam) of workload {name} at full size; the "
                      f"REFERENCE's own utils/slam_helpers.py, utils/slam_external.py, get_loss and initialize_optimizer (scripts/splatam.py) imported "
                      f"from {d['reference']} with its .cuda() calls redirected to the CPU, around the C oracle (OpenMP, {os.cpu_count()} threads; "
                      f"torch {d['threads']} threads); median {d['median_s']:.3f} s/iter",
            "first_loss": d["first_loss"]}


def cpu_baseline(name, params, frames, budget_s=25.0):
    """The reference's CPU plumbing (same host-side code on CPU tensors) around the
    C oracle rasterizer (oracle/raster_ref.c, 'port'), one tracking iteration of the
    same workload, timed on this box's host cores.  Where the reference itself is present (this container, not the GPU box) its own
    modules are timed instead (cpu_baseline_reference: kind "reference glue + oracle"); the mirror's figure rides along."""
    from oracle import c_ref
    from splatam_amd import slam
    import numpy as np
    via_ref = cpu_baseline_reference(name, params, frames, budget_s)

    class _CpuRaster(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, colors, opac, scales, rots, cam):
            cr = c_ref.CRef()
            col, radii, dep = cr.forward(means3D.detach().numpy(), colors.detach().numpy(), opac.detach().numpy(),
                                         scales.detach().numpy(), rots.detach().numpy(), cam.viewmatrix.numpy(),
                                         cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy, cam.image_width,
                                         cam.image_height, cam.bg.numpy())
            ctx.cr = cr
            _CpuRaster.last = cr
            return torch.from_numpy(col), torch.from_numpy(radii), torch.from_numpy(dep)

        @staticmethod
        def backward(ctx, gcol, _r, _d):
            g = ctx.cr.backward(gcol.contiguous().numpy())
            t = torch.from_numpy
            return t(g['means3D']), t(g['means2D']), t(g['colors']), t(g['opacities']), t(g['scales']), t(g['rotations']), None

    class _CpuRenderer:
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, means3D, means2D, opacities, colors_precomp, scales, rotations):
            return _CpuRaster.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self.s)

    cpu_params = {k: torch.nn.Parameter(v.detach().cpu().clone()) for k, v in params.items()}
    f = frames[1]
    cam = f['cam']
    cam_cpu = type(cam)(*[(x.cpu() if torch.is_tensor(x) else x) for x in cam])
    data = {'cam': cam_cpu, 'im': f['im'].cpu(), 'depth': f['depth'].cpu(), 'id': 1, 'w2c': torch.eye(4)}
    n = cpu_params['means3D'].shape[0]
    variables = {'max_2D_radius': torch.zeros(n)}
    opt = slam.initialize_optimizer(cpu_params, slam.REPLICA_TRACKING['lrs'], tracking=True)
    saved = slam.Renderer
    slam.Renderer = _CpuRenderer
    try:
        def one():
            cfg = slam.REPLICA_TRACKING
            loss, _, _ = slam.get_loss(cpu_params, data, variables, 1, cfg['loss_weights'], cfg['use_sil_for_loss'],
                                       cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'], tracking=True)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        one()                                   # warm-up
        times = []
        t_all = time.perf_counter()
        while len(times) < 5 and (time.perf_counter() - t_all) < budget_s:
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
    finally:
        slam.Renderer = saved
    med = float(np.median(times))
    port = {"value": round(1.0 / med, 4), "unit": "iters/s", "cores": os.cpu_count(), "kind": "port", "pairs_per_render": _CpuRaster.last.num_pairs(),
            "sample": f"{len(times)} tracking iterations (2 fwd + 2 bwd rasterizations + host glue + Adam) of workload {name} "
                      f"at full size; C oracle (OpenMP, {os.cpu_count()} threads) + PyTorch-CPU glue (the mirror splatam_amd/slam.py: the "
                      f"reference is not on this box); median {med:.3f} s/iter"}
    if via_ref is None:
        return port
    via_ref["pairs_per_render"] = port["pairs_per_render"]
    via_ref["mirror_glue_iters_per_s"] = port["value"]      # (the same iteration on the repository's mirror of the glue, for comparison)
    return via_ref


def b_loop_headline(dev, steps=40):
    """The SECOND headline: the same figures at the map SplaTAM really builds at B's frame size -- workload B-loop,
    ~816 k Gaussians in creation order (one per valid first-frame pixel, initialize_first_timestep) -- from this run: the 2:3 mix of
    tracking and mapping iterations of the fused engine, and the backward composite's launch time and roofline fraction there."""
    import ctypes as C
    from splatam_amd import _capi, slam
    from splatam_amd.fused import FusedEngine
    params, variables, frames, shape = build_scene("B-loop", dev, 2)
    N, W, H = shape
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'], track_max_radius=variables['max_2D_radius'])
    eng.keep_map_grads = False
    eng.begin_tracking(1)
    for _ in range(3):
        eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
        torch.cuda.synchronize(dev)
        eng.check_overflow()
    snap = {k: v.detach().clone() for k, v in eng.params.items()}

    def restore():
        with torch.no_grad():
            for k, v in eng.params.items():
                v.copy_(snap[k])
        eng.reset_map_optimizer()
        eng.begin_tracking(1)
    for _ in range(20):
        eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING)
    map_rate = phase_rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING), steps, dev)
    restore()
    for _ in range(10):
        eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
    track_rate = phase_rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING), steps, dev)
    restore()
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize(dev)
    R, list_entries = reference_instances(eng), int(eng.buf['status'][0])
    ws = eng._workspace(False, False)
    L = _capi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    t = {}
    for fn in (2, 3):
        ms = C.c_float(0)
        rc = 0
        for iters in (5, 30):
            rc = L.splat_iter_time_kernel(fn, iters, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms))
        t[fn] = ms.value if rc == 0 else None
    ok = not eng.check_overflow(grow=False)
    k6 = t[2]
    k7 = (t[3] - t[2]) if (t[2] and t[3]) else None
    bytes_bwd = R * 52 + W * H * 32 + N * 48
    out = {"workload": f"B-loop: {N} Gaussians in creation order, {W}x{H} (the map the frame loop builds at B's frame size)",
           "iters_per_s": round(5.0 / (2.0 / track_rate + 3.0 / map_rate), 3), "tracking_iters_per_s": round(track_rate, 3),
           "mapping_iters_per_s": round(map_rate, 3), "render_forward_ms": None if k6 is None else round(k6, 4),
           "render_backward_ms": None if k7 is None else round(k7, 4), "num_rendered": R, "list_entries": list_entries,
           "algorithmic_bytes": bytes_bwd,
           "roofline_frac": None if not k7 else round(bytes_bwd / (k7 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "valid": ok}
    del eng
    return out


def slam_loop_figure(name, dev, frames=13, engine="fused", runs=2):
    """The WHOLE frame loop of scripts/splatam.py:654-905 (pose initialisation, tracking, densification, keyframe selection, mapping
    with pruning, keyframe list; splatam_amd/pipeline.py) on a synthetic RGB-D sequence of the workload's image size with the
    reference's Replica iteration counts.  The map is what the loop builds (one Gaussian per valid first-frame pixel, then
    densification), not the workload's fixed 300k.  `frames` frames, the FIRST EXCLUDED from every rate (it pays the allocations and has no
    tracking phase); run `runs` times in this process, every run reported; per-phase milliseconds per frame (means over the counted
    frames) from synchronised timers around the phases.  engine: "fused", or "plugin" (the reference-shaped loop -- add_new_gaussians /
    prune_gaussians re-create every tensor -- with splatam_amd.plugin installed), or "plugin_map_edits" (plugin.install(map_edits=True):
    the two map edits are adapters as well and the engine owns the map)."""
    from splatam_amd import pipeline
    N, W, H, fx, fy, cx, cy = WORKLOADS[name]
    ds = pipeline.SyntheticRGBDSequence(N, W, H, fx, fy, cx, cy, num_frames=frames, seed=3, device=dev).preload()
    cfg = pipeline.replica_config()
    out_runs = []
    for _ in range(runs):
        torch.manual_seed(0)
        np.random.seed(0)
        torch.cuda.synchronize(dev)
        params, _, st = pipeline.rgbd_slam(ds, cfg, engine=engine)
        torch.cuda.synchronize(dev)
        counted = st['frame_s'][1:]
        ate = max(float((pipeline._est_w2c(params, t)[:3, 3] - ds.gt_w2c(t)[:3, 3]).norm()) for t in range(frames))
        phases = {}
        for fr in st['phase_ms'][1:]:
            for k, v in fr.items():
                phases[k] = phases.get(k, 0.0) + v
        n_counted = max(len(counted), 1)
        track_ms = sum(fr.get('tracking', 0.0) for fr in st['phase_ms'][1:])
        map_ms = sum(fr.get('mapping_iterations', 0.0) for fr in st['phase_ms'][1:])
        tcfg, mcfg = cfg['tracking'], cfg['mapping']
        run = {"frames_per_s": round(n_counted / max(sum(counted), 1e-9), 3),
               # (the iterations that RAN on the counted frames: the doubled budget and redone iterations included)
               "tracking_iters_per_s": round(1e3 * sum(d['tracking_iters'] for d in st['decisions'][1:]) / max(track_ms, 1e-9), 1),
               "mapping_iters_per_s": round(1e3 * mcfg['num_iters'] * n_counted / max(map_ms, 1e-9), 1),     # the reference's timer: iterations only
               "ms_per_frame": round(1e3 * sum(counted) / n_counted, 2),
               "phase_ms_per_frame": {k: round(v / n_counted, 3) for k, v in sorted(phases.items())},
               "first_frame_s": round(st['frame_s'][0], 3),
               "gaussians_first_last": [st['num_gaussians'][0], st['num_gaussians'][-1]],
               "redone_iterations": st['redone_iterations'], "max_translation_error_m": round(ate, 5)}
        if 'plugin' in st:
            run["plugin_session"] = st['plugin']
        out_runs.append(run)
        del params
    fps = [r["frames_per_s"] for r in out_runs]
    return {"frames": frames, "frames_counted": frames - 1, "image": f"{W}x{H}", "engine": engine,
            "tracking_iters_per_frame": cfg['tracking']['num_iters'], "mapping_iters_per_frame": cfg['mapping']['num_iters'],
            "frames_per_s": round(sum(fps) / len(fps), 3), "runs_agree_within": round(abs(max(fps) - min(fps)) / max(fps), 4),
            "runs": out_runs}


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell (no torchrun around it): re-run this command under torch.distributed.run with one
    rank per GPU on this node; rank 0 of the child job prints the JSON line.  With fewer GPUs than ranks (development boxes) the ranks
    share what is there over gloo."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # torchrun pins OMP_NUM_THREADS=1 when it is unset: give every rank its share of the host cores instead (host-side legs)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ngpu < args.gpus:
        env.setdefault("SPLAT_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def launch_check():
    """`--launch-check`: the N-rank launch path without a GPU -- rendezvous, one all-reduce, rank 0 prints a JSON line (tests/test_bench_launch.py)."""
    from splatam_amd import dist as sdist
    rank, world, _ = sdist.init_from_env(backend="gloo" if not torch.cuda.is_available() else None)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "sum_of_ranks_plus_one": float(t[0]),
                          "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="B", choices=sorted(WORKLOADS) + ["C"],
                    help="C = BASELINE config 3: workload B's map, mapping-only, a fixed batch of 8 keyframe views per step "
                         "split over the ranks (strong scaling)")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--sync-mode", default="auto", choices=["auto", "exact", "lazy"],
                    help="capacity policy of the drop-in rasterizer (splatam_amd/rasterizer.py): auto = exact on a scene's first call, then no host read")
    ap.add_argument("--engine", default="fused", choices=["fused", "dropin"])
    ap.add_argument("--sustain-s", type=float, default=6.5, help="length of the sustained region in seconds")
    ap.add_argument("--prewarm-s", type=float, default=0.15,
                    help="run the step schedule for this long BEFORE the W warm-up steps so that the GPU's clocks have ramped (0: off)")
    ap.add_argument("--instream-rccl", action="store_true",
                    help="N > 1 over RCCL: issue the per-iteration all-reduces on the iteration's own stream (splatam_amd.dist.InStreamRccl)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-b-loop", action="store_true", help="skip the second headline (the same figures at workload B-loop, the frame loop's map size)")
    ap.add_argument("--no-slam-loop", action="store_true", help="skip the end-to-end frame-loop figures (slam_loop, slam_loop_plugin)")
    ap.add_argument("--slam-frames", type=int, default=13, help="frames of the frame-loop figures (the first is not counted)")
    ap.add_argument("--launch-check", action="store_true", help="only exercise the N-rank launch (no GPU needed)")
    ap.add_argument("--replicated-tracking", action="store_true",
                    help="--gpus N > 1: every rank tracks the whole frame (no exchange) instead of sharding the frame's tile rows over the ranks")
    args = ap.parse_args()
    global SHARD_TRACKING
    SHARD_TRACKING = not args.replicated_tracking
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.launch_check:
        return launch_check()

    from splatam_amd import dist as sdist
    from splatam_amd import rasterizer as rz
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    if args.instream_rccl:
        os.environ["SPLAT_INSTREAM_RCCL"] = "1"
    rank, world, local_rank = sdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP rasterizer has no CPU path)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())     # several ranks may share a GPU on a development box
    torch.cuda.set_device(dev)
    rz.set_sync_mode(args.sync_mode)

    mode_c = args.workload == "C"
    if mode_c:
        args.workload = "B"
        args.views = max(args.views, 8)
        if args.engine != "fused" or 8 % world != 0:
            raise SystemExit("--workload C runs on the fused engine with 1, 2, 4 or 8 ranks")
    params, variables, frames, shape = build_scene(args.workload, dev, args.views)
    N, W, H = shape
    fused = args.engine == "fused"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the drop-in path works on its own copy of the map so that both engines start from the seeded scene
    params_d = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    bucket = sdist.GradBucket(params_d) if world > 1 else None
    opt_track = slam.initialize_optimizer(params_d, slam.REPLICA_TRACKING['lrs'], tracking=True)
    opt_map = slam.initialize_optimizer(params_d, slam.REPLICA_MAPPING['lrs'], tracking=False)
    tstate = slam.TrackingState(params_d, 1)

    # kernel-level figures are taken on the seeded initial map (before Adam moves it), so that they are
    # comparable from run to run and with scripts/kernel_driver.py
    roof = None
    mpix = ms_call = None
    if rank == 0 and not args.no_roofline:
        # (BASELINE metric (i) first: the roofline leg below runs torch.profiler, and a process that has profiled once pays for it in every
        #  later autograd call -- the drop-in call is host-bound, it read 0.23 ms behind the profiler against 0.21 ms in a fresh process)
        mpix, ms_call = render_mpix(params_d, frames, shape, dev, reps=128)
        if fused:
            probe = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
            roof = fused_roofline(probe, frames, shape, dev, args.workload)
            del probe
        else:
            roof, _ = kernel_roofline(params_d, frames, shape, dev)

    if fused:
        eparams = {k: v.detach().clone() for k, v in params.items()}
        eng = FusedEngine(eparams, frames[1]['cam'], track_max_radius=variables['max_2D_radius'])
        eng.keep_map_grads = False      # (the loop discards a mapping iteration's gradients after the step, as the reference's zero_grad does)
        eng.begin_tracking(1)

        def steps(n, start):
            if mode_c:
                run_steps_views(eng, frames, rank, world, n)
            else:
                run_steps_fused(eng, frames, rank, world, n, start)
        # clock pre-warm, before the W warm-up steps: an MI355X that has idled for >= 20 ms (scene set-up, the host-side legs above) runs
        # its first ~25 ms of work 10-15 % slower than steady state (power management ramps the clocks: scripts/r05_timed_region.py,
        # profiles/r05_experiments.md 10) -- with the driver's K = 20 the whole timed region (5 ms) sat inside that ramp.  The same step
        # schedule runs here in blocks of 50 until >= --prewarm-s seconds have passed (decided jointly by the ranks), the map, the
        # poses and the Adam state going back to the seeded values after every block; the W warm-up steps and the K timed steps then
        # follow without an idle gap, on the seeded scene, exactly as before.  Reported as "prewarm" in the JSON line
        prewarm = {"seconds": 0.0, "steps": 0}
        if args.prewarm_s > 0:
            seeded = {k: v.detach().clone() for k, v in eparams.items()}
            barrier()
            tp = time.perf_counter()
            while True:
                steps(50, prewarm["steps"])
                prewarm["steps"] += 50
                if prewarm["steps"] == 50 and eng.check_overflow():
                    raise SystemExit("instance lists overflowed during the clock pre-warm")
                with torch.no_grad():
                    for k, v in eparams.items():
                        v.copy_(seeded[k])
                eng.reset_map_optimizer()
                eng.begin_tracking(1)
                barrier()
                el = torch.tensor([time.perf_counter() - tp], device=dev, dtype=torch.float64)
                if world > 1:
                    dist.all_reduce(el, op=dist.ReduceOp.MAX)
                prewarm["seconds"] = round(float(el[0]), 3)
                if prewarm["seconds"] >= args.prewarm_s or prewarm["steps"] >= 5000:
                    break
            if eng.check_overflow():
                raise SystemExit("instance lists overflowed during the clock pre-warm")
            del seeded
        # warm-up: the list statistics are learnt from its first steps (check_overflow: bucketed lists, group records, no long-list
        # sort launch), the last ones already run the learnt configuration -- whose buffers are allocated there, not in the timed region
        w0 = max(args.warmup - 3, 0)
        steps(w0, 0)
        if world > 1 and not mode_c:        # (the list statistics are learnt from a whole-frame iteration; a sharded tracking step covers a band)
            eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
        if eng.check_overflow():
            raise SystemExit("instance lists overflowed during warm-up")
        steps(args.warmup - w0, w0)
        if eng.check_overflow(grow=False):
            raise SystemExit("instance lists overflowed during warm-up")
        barrier()
        t0 = time.perf_counter()
        steps(args.steps, args.warmup)
        barrier()
        elapsed = time.perf_counter() - t0
        if eng.check_overflow(grow=False):
            raise SystemExit("instance lists overflowed during the timed region: the result would be invalid")
        if world > 1:                       # every rank must agree on the length of the sustained region (it holds collectives)
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        # a sustained figure beside the K-step one: the same schedule for >= 6.5 s in ONE block (multiples of the 5-step mix period; the
        # driver samples GPU utilisation every 5 s: at least one sample falls inside).  Tens of thousands of Adam steps on one synthetic
        # view set would move the map until its lists outgrow the learnt buckets, so every 250 steps the map, the poses and the Adam
        # state go back to their values at the start of the region (a 14 MB device copy: < 0.1 % of the block)
        n_sus = max(args.steps, int(math.ceil(args.sustain_s * args.steps / max(elapsed, 1e-6) / 5.0)) * 5)
        snapshot = {k: v.detach().clone() for k, v in eparams.items()}
        barrier()
        t1 = time.perf_counter()
        done = 0
        while done < n_sus:
            nblk = min(250, n_sus - done)
            steps(nblk, args.warmup + args.steps + done)
            done += nblk
            with torch.no_grad():
                for k, v in eparams.items():
                    v.copy_(snapshot[k])
            eng.reset_map_optimizer()
            eng.begin_tracking(1)
        barrier()
        sustained_s = time.perf_counter() - t1
        sustained_ok = not eng.check_overflow(grow=False)
        # the exchange step alone (rank-local average over 20 collectives of the flat gradient bucket)
        allreduce_ms = allreduce_small_ms = allreduce_small_folded_ms = None
        if world > 1:
            from splatam_amd.dist import all_reduce_sum_flat
            for _ in range(3):
                all_reduce_sum_flat(eng.reduce_flat)
            barrier()
            t2 = time.perf_counter()
            for _ in range(20):
                all_reduce_sum_flat(eng.reduce_flat)
            torch.cuda.synchronize(dev)
            allreduce_ms = 1e3 * (time.perf_counter() - t2) / 20
            eng.grad_flat.zero_()
            # the small exchange of tile-row-sharded tracking (the 16 KB record of partial sums), likewise
            for _ in range(3):
                all_reduce_sum_flat(eng.buf['sums'])
            barrier()
            t3 = time.perf_counter()
            for _ in range(20):
                all_reduce_sum_flat(eng.buf['sums'])
            torch.cuda.synchronize(dev)
            allreduce_small_ms = 1e3 * (time.perf_counter() - t3) / 20
            eng.buf['sums'].zero_()
            # ... and in the form the loop uses by default: the 64 copies folded on the device first (splat_iter_fold_sums), 256 bytes on the wire
            import ctypes as C
            from splatam_amd import _capi

            def folded():
                _capi.check(eng.L.splat_iter_fold_sums(eng.buf['sums'].data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "splat_iter_fold_sums")
                all_reduce_sum_flat(eng.buf['sums'][:_capi.SPLAT_ITER_SUMS])
            for _ in range(3):
                folded()
            barrier()
            t4 = time.perf_counter()
            for _ in range(20):
                folded()
            torch.cuda.synchronize(dev)
            allreduce_small_folded_ms = 1e3 * (time.perf_counter() - t4) / 20
            eng.buf['sums'].zero_()
    else:
        variables = run_steps(params_d, variables, frames, bucket, rank, world, args.warmup, opt_track, opt_map, tstate, 0)
        barrier()
        t0 = time.perf_counter()
        variables = run_steps(params_d, variables, frames, bucket, rank, world, args.steps, opt_track, opt_map, tstate, args.warmup)
        barrier()
        elapsed = time.perf_counter() - t0
    if not fused:
        n_sus, sustained_s, allreduce_ms, allreduce_small_ms, allreduce_small_folded_ms, sustained_ok = args.steps, elapsed, None, None, None, True
    if world > 1:
        t = torch.tensor([elapsed, sustained_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, sustained_s = float(t[0]), float(t[1])

    def units(nsteps):
        """Work units of `nsteps` steps over all ranks.  Mix: a mapping step renders one view PER RANK (world iterations' worth
        of views), a tracking step is the same iteration on every rank (replicas: counted once).  Config C: 8 views per step."""
        if mode_c:
            return 8 * nsteps
        n_track = sum(1 for i in range(nsteps) if i % 5 < 2)        # (the schedule's phase is a multiple of 5 at every region start
        return n_track + (nsteps - n_track) * world                # when warmup and steps are; otherwise off by at most 2)

    # per-phase rates (rank-local, informational)
    n_phase = max(5, min(40, args.steps))
    if fused:
        track_rate = phase_rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING, shard=(rank, world) if (world > 1 and SHARD_TRACKING) else None,
                                                               allreduce_sums=sdist.all_reduce_sum_flat), n_phase, dev)
        map_rate = phase_rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING), n_phase, dev)

        # the tracking iteration with EVERY gradient the reference's backward() forms (dL/d rgb, opacity, scale too: the reference steps
        # them with learning rate 0, configs/replica/splatam.py:71-79): the headline's tracking iterations leave them out
        def track_full():
            eng.loss_backward(frames[1], eng.track_time_idx, slam.REPLICA_TRACKING, tracking=True, map_grads=True,
                              pose_adam=eng._pose_adam_args(slam.REPLICA_TRACKING))
        track_rate_full = phase_rate(track_full, n_phase, dev) if world == 1 else None
        # several ranks: the same tracking iteration replicated (every rank composites the whole frame, no exchange), and the mapping
        # iteration with its gradient exchange -- beside the sharded / local rates above, so that a reader of the N-GPU line sees what
        # each phase gains
        track_rate_repl = map_rate_exch = None
        if world > 1:
            track_rate_repl = phase_rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING), n_phase, dev)
            map_rate_exch = phase_rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING, sdist.all_reduce_mean_flat), n_phase, dev)
    n_drop = max(5, min(15, args.steps))
    if fused:       # the drop-in path has not run yet: MIOpen / rocBLAS pick their kernels on the first calls
        for _ in range(3):
            slam.tracking_iteration(params_d, frames[1], variables, 1, opt_track, tstate)
            slam.mapping_iteration(params_d, frames[2], variables, 2, opt_map)
    track_rate_d = phase_rate(lambda: slam.tracking_iteration(params_d, frames[1], variables, 1, opt_track, tstate), n_drop, dev)

    def map_once():
        loss, _, _ = slam.get_loss(params_d, frames[2], variables, 2, slam.REPLICA_MAPPING['loss_weights'], False, 0.5, True, False, mapping=True)
        loss.backward()
        with torch.no_grad():
            opt_map.step()
            opt_map.zero_grad(set_to_none=True)
    map_rate_d = phase_rate(map_once, n_drop, dev)
    dropin_rate = 5.0 / (2.0 / track_rate_d + 3.0 / map_rate_d)
    # the rasterizer's own share of a drop-in iteration (two forwards + two backwards through the autograd surface): HIP events around
    # its C calls on the iteration's stream, summed over a few mapping iterations -- the rest is the reference's PyTorch glue
    # (~100 small launches, MIOpen convolutions, boolean-mask indexing, torch.optim.Adam), which the library cannot shorten
    spans = []

    def timed(fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            spans.append((e0, e1))
            return out
        return wrapper
    saved_f, saved_b = rz.rasterize_forward, rz.rasterize_backward
    rz.rasterize_forward, rz.rasterize_backward = timed(saved_f), timed(saved_b)
    try:
        torch.cuda.synchronize(dev)
        for _ in range(5):
            map_once()
        torch.cuda.synchronize(dev)
    finally:
        rz.rasterize_forward, rz.rasterize_backward = saved_f, saved_b
    dropin_raster_ms = sum(a.elapsed_time(b) for a, b in spans) / 5.0
    rz.set_geometry_cache(False)
    map_rate_d_nocache = phase_rate(map_once, n_drop, dev)
    rz.set_geometry_cache(True)
    # the SAME statements with splatam_amd.plugin installed (get_loss / initialize_optimizer of the module replaced at run time): what an
    # unmodified scripts/splatam.py gets from the plug-in -- the fused iteration behind the reference's own loop statements
    from splatam_amd import plugin
    params_p = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    vars_p = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in variables.items()}
    with plugin.install(slam):
        opt_track_p = slam.initialize_optimizer(params_p, slam.REPLICA_TRACKING['lrs'], tracking=True)
        tstate_p = slam.TrackingState(params_p, 1)
        for _ in range(3):
            slam.tracking_iteration(params_p, frames[1], vars_p, 1, opt_track_p, tstate_p)
        track_rate_p = phase_rate(lambda: slam.tracking_iteration(params_p, frames[1], vars_p, 1, opt_track_p, tstate_p), n_phase, dev)
        opt_map_p = slam.initialize_optimizer(params_p, slam.REPLICA_MAPPING['lrs'], tracking=False)
        for _ in range(3):
            slam.mapping_iteration(params_p, frames[2], vars_p, 2, opt_map_p)
        map_rate_p = phase_rate(lambda: slam.mapping_iteration(params_p, frames[2], vars_p, 2, opt_map_p), n_phase, dev)
    plugin_rate = 5.0 / (2.0 / track_rate_p + 3.0 / map_rate_p)
    if not fused:
        track_rate, map_rate = track_rate_d, map_rate_d

    result = None
    if rank == 0:
        if mpix is None:
            mpix, ms_call = render_mpix(params_d, frames, shape, dev, reps=128)
        wl = (f"C: {N} Gaussians, {W}x{H}, mapping-only, a batch of 8 keyframe views per step sharded over the ranks, one gradient all-reduce"
              if mode_c else f"{args.workload}: {N} Gaussians, {W}x{H}, SplaTAM tracking+mapping loop (2:3 mix), isotropic map"
                   + ("; fused tracking iterations do not form dL/d(rgb, opacity, scale): the reference computes them, steps them with learning "
                      "rate 0 and discards the optimizer after the frame (parameters after any number of iterations are identical); mapping iterations "
                      "take their Adam step on gradients held in registers and do not store them (the reference discards them after the step: "
                      "zero_grad(set_to_none=True))" if fused else ""))
        result = {
            "metric": ("mapping view-iterations/sec @300k Gaussians, 8 keyframe views per step" if mode_c
                       else "track+map iters/sec @300k Gaussians (render+backward Mpix/s alongside)"),
            "value": round(units(args.steps) / elapsed, 3), "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if mode_c else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl,
                       "gaussians": N, "width": W, "height": H, "views": args.views, "sync_mode": args.sync_mode, "engine": args.engine,
                       "parallelism": ("1 process/GPU; 8 views per step sharded over the ranks, gradients accumulated per rank, one all-reduce (sum), identical Adam step"
                                       if mode_c else ("1 process/GPU; mapping: one view per rank per step, one gradient all-reduce (mean); tracking: the frame's tile rows sharded over the "
                                             "ranks, one all-reduce of the partial sums per iteration, counted once" if (world > 1 and SHARD_TRACKING) else
                                            ("1 process/GPU; mapping: one view per rank per step, one gradient all-reduce (mean); tracking: replicas, counted once"
                                             if world > 1 else "1 process/GPU")))},
            # untimed work in front of the W warm-up steps (fused engine): the same schedule, state restored, so that the K timed steps do not
            # run inside the clock ramp of a GPU that idled through the set-up
            "prewarm": (prewarm if fused else None),
            "sustained": ({"steps": n_sus, "seconds": round(sustained_s, 3), "iters_per_s": round(units(n_sus) / sustained_s, 3)} if sustained_ok
                          else {"steps": n_sus, "invalid": "a per-tile list outgrew its bucket during the sustained region"}),
            "collectives": ("in-stream RCCL (splatam_amd.dist.InStreamRccl)" if sdist._instream is not None else
                            ("torch.distributed " + torch.distributed.get_backend() if world > 1 else None)),
            "allreduce_ms": None if allreduce_ms is None else round(allreduce_ms, 4),
            "allreduce_small_ms": None if allreduce_small_ms is None else round(allreduce_small_ms, 4),
            "allreduce_small_folded_ms": None if allreduce_small_folded_ms is None else round(allreduce_small_folded_ms, 4),
            "tracking_replicated_iters_per_s": None if not (fused and world > 1) else round(track_rate_repl, 3),
            "mapping_with_exchange_iters_per_s": None if not (fused and world > 1) else round(map_rate_exch, 3),
            "tracking_iters_per_s": round(track_rate, 3), "mapping_iters_per_s": round(map_rate, 3),
            # tracking with all of backward()'s gradients formed (see config.workload), and the 2:3 mix with it
            "tracking_full_gradients_iters_per_s": (round(track_rate_full, 3) if fused and track_rate_full else None),
            "mix_with_full_gradient_tracking_iters_per_s": (round(5.0 / (2.0 / track_rate_full + 3.0 / map_rate), 3) if fused and track_rate_full else None),
            # what answers the north star's "scripts/splatam.py's loops run unmodified": the reference's statements on the drop-in package
            # alone (no change at all), and the same statements after the ONE line `splatam_amd.plugin.install(module)`
            "reference_loop_unmodified": {"dropin_package_only_iters_per_s": round(dropin_rate, 3),
                                          "after_plugin_install_iters_per_s": round(plugin_rate, 3)},
            "plugin_iters_per_s": round(plugin_rate, 3), "plugin_tracking_iters_per_s": round(track_rate_p, 3),
            "plugin_mapping_iters_per_s": round(map_rate_p, 3),
            "dropin_iters_per_s": round(dropin_rate, 3), "dropin_tracking_iters_per_s": round(track_rate_d, 3),
            "dropin_mapping_iters_per_s": round(map_rate_d, 3),
            "dropin_mapping_iters_per_s_without_geometry_cache": round(map_rate_d_nocache, 3),
            "dropin_mapping_ms_per_iter": round(1e3 / map_rate_d, 3), "dropin_mapping_rasterizer_ms_per_iter": round(dropin_raster_ms, 3),
            "render_fwd_bwd_mpix_per_s": round(mpix, 2), "render_fwd_bwd_ms": round(ms_call, 4),
            "host_cores": os.cpu_count(),
        }
        if roof is not None:
            result["roofline"] = roof
        if world == 1 and fused and args.workload == "B" and not mode_c and not args.no_b_loop:
            # the second headline, in the same JSON line: the representative map size (2.5x lower than B's 300 k: say so where B is quoted)
            result["b_loop"] = b_loop_headline(dev)
        if world == 1 and fused and not args.no_slam_loop:
            result["slam_loop"] = slam_loop_figure(args.workload, dev, frames=args.slam_frames)
            result["slam_loop_plugin"] = slam_loop_figure(args.workload, dev, frames=args.slam_frames, engine="plugin")
            result["slam_loop_plugin_map_edits"] = slam_loop_figure(args.workload, dev, frames=args.slam_frames, engine="plugin_map_edits", runs=1)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload, params_d, frames)
            pairs = result["cpu_baseline"].get("pairs_per_render")
            if roof is not None and pairs:
                roof["other"]["pairs_per_launch"] = pairs
                roof["other"]["pair_evals_per_s"] = round(pairs / (roof["other"]["render_backward_ms"] * 1e-3), 1)
                roof["other"]["pair_evals_frac_of_survey_ceiling"] = round(roof["other"]["pair_evals_per_s"] / 4.9e12, 4)
                if roof["other"].get("valu_insts_per_launch"):
                    roof["other"]["lane_ops_per_live_pair"] = round(64.0 * roof["other"]["valu_insts_per_launch"] / pairs, 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
