#!/usr/bin/env python3
"""A/B of the backward composite's entries-per-loop-trip (splat_debug_option(2, 1 | 2)): K7 kernel time through
splat_iter_time_kernel + fused iteration rates at a workload.  Developer tool (run through gpurun)."""
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
N, W, H = shape
L = _capi.lib()
for ne in (1, 2, 1, 2):
    L.splat_debug_option(2, ne)
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    eng.begin_tracking(1)
    eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
    torch.cuda.synchronize()
    ws = eng._workspace(False, False)
    ms = C.c_float(0)
    for iters in (5, 40):
        _capi.check(L.splat_iter_time_kernel(1, iters, C.byref(eng._cam), N, C.byref(ws), torch.cuda.current_stream(dev).cuda_stream, C.byref(ms)), "time")
    eng.buf['accum'].zero_()
    assert not eng.check_overflow()

    def rate(fn, n=60):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)
    tr = rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING))
    mp = rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING))
    print(f"entries per trip {ne}: K7(map form) {ms.value * 1e3:.1f} us  tracking {tr:.0f} it/s  mapping {mp:.0f} it/s  loss {eng.loss():.6f}", flush=True)
