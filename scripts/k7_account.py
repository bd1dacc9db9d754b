#!/usr/bin/env python3
"""Where the time of the fused backward composite (K7) goes at a workload: the kernel in its measurement builds
(splat_debug_option(4, bits): stage only / phase 1 only / no atomics) timed like bench.py times the product kernel, and the
per-workgroup wall-clock stamps of one launch (duration distribution, ramp and tail of the launch).  Developer tool, run through
gpurun; prints a markdown fragment (profiles/r04_k7_account.md is assembled from it).   usage: scripts/k7_account.py [workload]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import _capi, slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene(wl, dev, 3)
N, W, H = shape
L = _capi.lib()
eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
eng.begin_tracking(1)
for _ in range(3):
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize()
    assert not eng.check_overflow()
eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
torch.cuda.synchronize()
R = int(eng.buf['status'][0])
ws = eng._workspace(False, False)
stream = torch.cuda.current_stream(dev).cuda_stream


def t(fn, iters=30):
    ms = C.c_float(0)
    for n in (5, iters):
        _capi.check(L.splat_iter_time_kernel(fn, n, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms)), "time")
    return ms.value * 1e3


print(f"## K7 account, workload {wl}: {N} Gaussians, {W}x{H}, {R} instances, longest list {eng.max_list_hint}, bucket stride {eng.tile_stride}\n")
k6 = t(2) if eng.max_list_hint * 5 // 4 <= 1024 else t(0)
rows = []
for bits, what in ((0, "product kernel"), (2, "stage + commit only (no visit)"), (8, "phase 1 only (pairs written, never reduced)"),
                   (4, "both phases, no accumulator atomics")):
    L.splat_debug_option(4, bits)
    row = (what, t(1), t(3) - k6, t(4))
    rows.append(row)
L.splat_debug_option(4, 0)
print("| build | mapping form, 30 in a row (us) | mapping form between forward composites (us) | tracking form, 30 in a row (us) |")
print("|---|---|---|---|")
for what, a, b, c in rows:
    print(f"| {what} | {a:.1f} | {b:.1f} | {c:.1f} |")
print(f"\n(forward composite alone in that alternation: {k6:.1f} us)\n")

# per-workgroup stamps of single launches (100 MHz wall clock)
T = ((W + 15) // 16) * ((H + 15) // 16)
blocks = 8 * ((T + 7) // 8)
stamps = torch.zeros(blocks, 2, dtype=torch.int64, device=dev)
L.splat_debug_stamps(stamps.data_ptr())
for form, fn in (("mapping", 1), ("tracking", 4)):
    L.splat_debug_option(4, 1)
    stamps.zero_()
    ms = C.c_float(0)
    _capi.check(L.splat_iter_time_kernel(fn, 1, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms)), "time")
    torch.cuda.synchronize()
    L.splat_debug_option(4, 0)
    s = stamps.cpu().numpy().astype(np.float64)
    s = s[s[:, 1] > 0]
    t0 = s[:, 0].min()
    b, e = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0           # us
    d = e - b
    span = e.max()
    busy = d.sum()
    # resident workgroups over time
    ev = np.concatenate([np.stack([b, np.ones_like(b)], 1), np.stack([e, -np.ones_like(e)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    occ = np.cumsum(ev[:, 1])
    peak = occ.max()
    # time-weighted mean occupancy, time below half of the peak at the end (tail) and at the start (ramp)
    tt = ev[:, 0]
    mean_occ = float((occ[:-1] * np.diff(tt)).sum() / span)
    half = 0.5 * peak
    tail_start = tt[np.where(occ >= half)[0][-1]]
    ramp_end = tt[np.where(occ >= half)[0][0]]
    q = np.quantile(d, [0.05, 0.5, 0.95, 1.0])
    print(f"**{form} form, one launch of {len(d)} workgroups ({ms.value * 1e3:.1f} us by events):** span {span:.1f} us; workgroup duration "
          f"5 % / median / 95 % / max = {q[0]:.1f} / {q[1]:.1f} / {q[2]:.1f} / {q[3]:.1f} us; peak resident workgroups {int(peak)}, "
          f"time-weighted mean {mean_occ:.0f} ({100 * mean_occ / peak:.0f} % of peak); ramp to half occupancy {ramp_end:.1f} us, "
          f"tail below half occupancy {span - tail_start:.1f} us; sum of workgroup durations / (span x peak) = {100 * busy / (span * peak):.0f} %; "
          f"last workgroup STARTED at {b.max():.1f} us.\n")
L.splat_debug_stamps(None)
