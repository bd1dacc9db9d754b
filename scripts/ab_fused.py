#!/usr/bin/env python3
"""A/B timing of the composite-kernel generations on the FUSED path (6-channel K6 / K7 through splat_iter_time_kernel,
HIP events on the launch stream) + per-phase iteration rates.  usage: scripts/ab_fused.py [workload] [versions...]
Developer tool (run through gpurun)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import _capi, slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    versions = [int(v) for v in sys.argv[2:]] or [3, 4]
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene(wl, dev, 3)
    N, W, H = shape
    L = _capi.lib()
    for ver in versions:
        L.splat_debug_option(1, ver)
        eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
        eng.begin_tracking(1)
        eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
        torch.cuda.synchronize()
        ws = eng._workspace(False, False)          # exact lists (tile_base persists; bucketed counters are consumed per iteration)
        stream = torch.cuda.current_stream(dev).cuda_stream
        res = {}
        for fn, name in ((0, "K6"), (1, "K7")):
            ms = C.c_float(0)
            for iters in (5, 40):
                _capi.check(L.splat_iter_time_kernel(fn, iters, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms)), "time")
            res[name] = ms.value * 1e3
        eng.buf['accum'].zero_()
        assert not eng.check_overflow()

        def rate(fn, n=40):
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
        assert not eng.check_overflow(grow=False)
        print(f"version {ver}: K6 {res['K6']:.1f} us  K7(map form) {res['K7']:.1f} us  tracking {tr:.0f} it/s  mapping {mp:.0f} it/s  "
              f"mix {5.0 / (2.0 / tr + 3.0 / mp):.0f} it/s  loss {eng.loss():.6f}", flush=True)


if __name__ == "__main__":
    main()
