#!/usr/bin/env python3
"""Fused iteration rates at a workload in THIS process (environment switches of experimental builds are read once per process):
usage  VAR=value python scripts/ab_env.py [workload].  Developer tool (run through gpurun)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "B"
dev = torch.device("cuda", 0)
params, variables, frames, shape = bench.build_scene(wl, dev, 3)
eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
eng.begin_tracking(1)
for _ in range(2):
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize()
    assert not eng.check_overflow()


def rate(fn, n=150):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


res = []
for rep in range(2):
    if os.environ.get("AB_SEPARATE_ADAM"):
        def mp():
            eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
            eng.adam_map(slam.REPLICA_MAPPING['lrs'])
    else:
        def mp():
            eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING)
    res.append((rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)), rate(mp)))
assert not eng.check_overflow(grow=False)
sw = {k: v for k, v in os.environ.items() if k.startswith("SPLAT_") or k.startswith("AB_")}
print(f"{sw}: tracking {res[0][0]:.0f} / {res[1][0]:.0f} it/s  mapping {res[0][1]:.0f} / {res[1][1]:.0f} it/s  loss {eng.loss():.6f}", flush=True)
