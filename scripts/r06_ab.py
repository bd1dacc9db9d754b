"""Round 6: in-process A/B of engine-level switches at one workload, alternating rounds in ONE process on one box.

Per variant and round: tracking iterations/s, mapping iterations/s, the 2:3 mix, and the composites' launch times (K6 sorting form,
K6 + K7 alternating pair, K7 = pair - K6: splat_iter_time_kernel on the iteration's stream).
usage (GPU box): python scripts/r06_ab.py [workload] [rounds] variant[,variant...]
  variants: name=ATTR:VALUE[+ATTR:VALUE...]  (attributes of FusedEngine set before its lists are learnt), e.g.
            base= recs0=use_recs:0"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from splatam_amd import _capi, slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402


def make_engine(params, frames, attrs):
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    for k, v in attrs.items():
        setattr(eng, k, v)
    eng.begin_tracking(1)
    for _ in range(3):
        eng.loss_backward(frames[1], 1, slam.REPLICA_TRACKING, tracking=True)
        torch.cuda.synchronize()
        eng.check_overflow()
    return eng


def measure(eng, frames, shape, dev, n=150):
    N, W, H = shape
    L = _capi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    snap = {k: v.detach().clone() for k, v in eng.params.items()}

    def restore():
        with torch.no_grad():
            for k, v in eng.params.items():
                v.copy_(snap[k])
        eng.reset_map_optimizer()
        eng.begin_tracking(1)
    for _ in range(30):
        eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING)
    out["mapping"] = bench.phase_rate(lambda: eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING), n, dev)
    restore()
    for _ in range(30):
        eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
    out["tracking"] = bench.phase_rate(lambda: eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING), n, dev)
    restore()

    def track_full():
        eng.loss_backward(frames[1], eng.track_time_idx, slam.REPLICA_TRACKING, tracking=True, map_grads=True,
                          pose_adam=eng._pose_adam_args(slam.REPLICA_TRACKING))
    for _ in range(30):
        track_full()
    out["tracking_full"] = bench.phase_rate(track_full, n, dev)
    restore()
    out["mix"] = 5.0 / (2.0 / out["tracking"] + 3.0 / out["mapping"])
    out["mix_full"] = 5.0 / (2.0 / out["tracking_full"] + 3.0 / out["mapping"])
    eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
    torch.cuda.synchronize(dev)
    ws = eng._workspace(False, False)
    t = {}
    for fn in (2, 3):
        ms = C.c_float(0)
        for iters in (5, 30):
            rc = L.splat_iter_time_kernel(fn, iters, C.byref(eng._cam), N, C.byref(ws), stream, C.byref(ms))
        t[fn] = ms.value if rc == 0 else float("nan")
    out["K6_us"] = 1e3 * t[2]
    out["K7_us"] = 1e3 * (t[3] - t[2])
    assert not eng.check_overflow(grow=False)
    return {k: round(v, 1) for k, v in out.items()}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    variants = {}
    for spec in (sys.argv[3] if len(sys.argv) > 3 else "base=").split(","):
        name, _, rest = spec.partition("=")
        attrs = {}
        for kv in filter(None, rest.split("+")):
            k, _, v = kv.partition(":")
            attrs[k] = int(v) if v.lstrip("-").isdigit() else v
        variants[name] = attrs
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene(wl, dev, 8)
    engines = {name: make_engine(params, frames, attrs) for name, attrs in variants.items()}
    # clocks
    t0 = time.perf_counter()
    e0 = next(iter(engines.values()))
    while time.perf_counter() - t0 < 0.3:
        for _ in range(25):
            e0.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
        torch.cuda.synchronize()
    for r in range(rounds):
        for name, eng in engines.items():
            print(json.dumps({"workload": wl, "round": r, "variant": name, **measure(eng, frames, shape, dev)}), flush=True)


if __name__ == "__main__":
    main()
