#!/usr/bin/env python3
"""Kernel-only times of the fused iteration's composites (splat_iter_time_kernel: K6 without its sort, K7 mapping form) and the
iteration rates at a workload.  Developer tool (run through gpurun)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import _capi, slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene(wl, dev, 3)
N = shape[0]
L = _capi.lib()
eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
eng.begin_tracking(1)
for _ in range(2):
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize()
    assert not eng.check_overflow()
eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
torch.cuda.synchronize()
ws = eng._workspace(False, False)
out = []
for fn in (0, 1):
    ms = C.c_float(0)
    for iters in (5, 50):
        _capi.check(L.splat_iter_time_kernel(fn, iters, C.byref(eng._cam), N, C.byref(ws), torch.cuda.current_stream(dev).cuda_stream, C.byref(ms)), "time")
    out.append(ms.value * 1e3)


host_us = []


def rate(fn, n=120):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    host_us.append(1e6 * (t1 - t0) / n)         # time the host needs to enqueue one iteration
    return n / (time.perf_counter() - t0)


tr = rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING))
mp = rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING))
assert not eng.check_overflow(grow=False)
print(f"K6 (no sort) {out[0]:.1f} us   K7 (mapping form) {out[1]:.1f} us   tracking {tr:.0f} it/s   mapping {mp:.0f} it/s   host enqueue {host_us[0]:.0f} / {host_us[1]:.0f} us per iteration", flush=True)
