#!/usr/bin/env python3
"""Is the under-filled last round of the backward composite (profiles/r04_k7_account.md 2) a property of the tile granularity, or of
the STATIC assignment of tiles to XCDs?  The composites give XCD x the x-th band of tiles (neighbouring tiles share an L2) and start
each band's workgroups heaviest first; the hardware sends workgroup b to XCD b % 8, so the eight bands are eight independent queues.
From the per-workgroup wall-clock stamps of one launch (splat_debug_option(4, 1)): per XCD the number of workgroups, the sum of their
durations, when its last workgroup ended -- and what a launch with ONE queue over all 1 024 slots (any workgroup to any XCD, the same
durations, longest first / in the present order) would take.   usage: scripts/r05_xcd_balance.py [workload]"""
import ctypes as C
import heapq
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
ws = eng._workspace(False, False)
stream = torch.cuda.current_stream(dev).cuda_stream
T = ((W + 15) // 16) * ((H + 15) // 16)
blocks = 8 * ((T + 7) // 8)
stamps = torch.zeros(blocks, 2, dtype=torch.int64, device=dev)
L.splat_debug_stamps(stamps.data_ptr())


def packed(durations, slots):
    """Makespan of list scheduling: every job, in the given order, to the slot that frees first."""
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for d in durations:
        t = heapq.heappop(free) + d
        end = max(end, t)
        heapq.heappush(free, t)
    return end


print(f"## XCD balance of the backward composite, workload {wl}: {N} Gaussians, {W}x{H}, {T} tiles\n")
for form, fn, slots_per_xcd in (("mapping", 1, 128), ("tracking", 4, 160)):
    L.splat_debug_option(4, 1)
    stamps.zero_()
    ms = C.c_float(0)
    _capi.check(L.splat_iter_time_kernel(fn, 1, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms)), "time")
    torch.cuda.synchronize()
    L.splat_debug_option(4, 0)
    s = stamps.cpu().numpy().astype(np.float64)
    ok = s[:, 1] > 0
    t0 = s[ok, 0].min()
    b, e = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0
    d = np.where(ok, e - b, 0.0)
    xcd = np.arange(blocks) % 8
    print(f"**{form} form** ({ms.value * 1e3:.1f} us by events; span of the stamps {e[ok].max():.1f} us):\n")
    print("| XCD | workgroups | sum of durations (us) | per slot (us) | last workgroup ends (us) | its own queue packed (us) |")
    print("|---|---|---|---|---|---|")
    for x in range(8):
        m = ok & (xcd == x)
        order = np.argsort(np.nonzero(m)[0])            # launch order within the XCD = slot order
        dur = d[m][order]
        print(f"| {x} | {int(m.sum())} | {dur.sum():.0f} | {dur.sum() / slots_per_xcd:.1f} | {e[m].max():.1f} | {packed(dur, slots_per_xcd):.1f} |")
    total = d[ok].sum()
    print(f"\nall XCDs: {total:.0f} us of workgroup time = {total / (8 * slots_per_xcd):.1f} us per slot; one queue over {8 * slots_per_xcd} slots, "
          f"present order {packed(d[ok], 8 * slots_per_xcd):.1f} us, longest first {packed(np.sort(d[ok])[::-1], 8 * slots_per_xcd):.1f} us; "
          f"eight queues as launched {max(packed(d[ok & (xcd == x)], slots_per_xcd) for x in range(8)):.1f} us, each longest first "
          f"{max(packed(np.sort(d[ok & (xcd == x)])[::-1], slots_per_xcd) for x in range(8)):.1f} us.\n")
L.splat_debug_stamps(None)
