#!/usr/bin/env python3
"""A/B of the backward composite generations (splat_debug_option(3, 3 | 5), library built with EXPERIMENTS=1): K7 kernel time (mapping form) through
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
if L.splat_debug_option(3, 3) < 0:
    sys.exit("generation 3 is not in this library: rebuild with `make -C splatam_amd/csrc EXPERIMENTS=1`")
ref = None
for gen in (3, 5, 3, 5):
    L.splat_debug_option(3, gen)
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    eng.begin_tracking(1)
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize()
    g = eng.grads['means3D'].clone()
    if ref is None:
        ref = g
    else:
        print(f"  gen {gen} vs first: max |d means3D grad| {float((g - ref).abs().max()):.3e} of {float(ref.abs().max()):.3e}")
    ws = eng._workspace(False, False)
    ms = C.c_float(0)
    ms0 = C.c_float(0)
    for iters in (5, 40):
        _capi.check(L.splat_iter_time_kernel(0, iters, C.byref(eng._cam), N, C.byref(ws), torch.cuda.current_stream(dev).cuda_stream, C.byref(ms0)), "time")
        _capi.check(L.splat_iter_time_kernel(1, iters, C.byref(eng._cam), N, C.byref(ws), torch.cuda.current_stream(dev).cuda_stream, C.byref(ms)), "time")
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
    print(f"K7 generation {gen}: K6(no sort) {ms0.value * 1e3:.1f} us  K7(map form) {ms.value * 1e3:.1f} us  tracking {tr:.0f} it/s  mapping {mp:.0f} it/s  loss {eng.loss():.6f}", flush=True)
