"""Round 6: host time of a drop-in forward + backward by segment (perf_counter around the pieces, auto policy, steady state).
usage (GPU box): python scripts/r06_dropin_segments.py [workload] [N]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from splatam_amd import rasterizer as rz  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.rasterizer import GaussianRasterizer as Renderer  # noqa: E402

acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return w


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene(wl, dev, 8)
    N, W, H = shape
    with torch.no_grad():
        tg = slam.transform_to_frame(params, 1, False, False)
        rv = {k: v.detach() for k, v in slam.transformed_params2rendervar(params, tg).items()}
    inp = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    gout = torch.randn(3, H, W, device=dev)
    cam = frames[1]['cam']
    rz._build_pack = timed("fwd._build_pack", rz._build_pack)
    rz._alloc_state_fast = timed("fwd._alloc_state_fast", rz._alloc_state_fast)
    rz._rasterize_forward_fast = timed("fwd._rasterize_forward_fast (total)", rz._rasterize_forward_fast)
    rz.rasterize_forward = timed("fwd.rasterize_forward (total)", rz.rasterize_forward)
    rz.rasterize_backward = timed("bwd.rasterize_backward (total)", rz.rasterize_backward)
    rz._RasterizeGaussians.forward = staticmethod(timed("fwd.Function.forward (total)", rz._RasterizeGaussians.forward))
    rz._RasterizeGaussians.backward = staticmethod(timed("bwd.Function.backward (total)", rz._RasterizeGaussians.backward))
    L = rz._capi.lib()
    for name in ("splat_forward", "splat_backward"):
        f = getattr(L, name)
        setattr(L, name, timed("C." + name, f))

    def once():
        t0 = time.perf_counter()
        im, _, _ = Renderer(raster_settings=cam)(**inp)
        t1 = time.perf_counter()
        im.backward(gout)
        t2 = time.perf_counter()
        for v in inp.values():
            v.grad = None
        acc["once.forward call"] = acc.get("once.forward call", 0.0) + t1 - t0
        acc["once.backward call"] = acc.get("once.backward call", 0.0) + t2 - t1
    for _ in range(20):
        once()
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(n):
        once()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out = {k: round(1e6 * v / n, 1) for k, v in sorted(acc.items())}
    out["total per call (host)"] = round(1e6 * (t1 - t0) / n, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
