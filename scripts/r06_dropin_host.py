"""Round 6: is the drop-in forward + backward bound by the GPU or by the host?  N calls enqueued back to back: time until the host is
done enqueueing vs time until the GPU is done (auto policy, steady state).  usage: python scripts/r06_dropin_host.py [workload] [N]"""
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


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene(wl, dev, 8)
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
    res = {}
    for mode in ("auto", "exact"):
        rz.set_sync_mode(mode)
        rz.reset_scene_stats()
        for _ in range(10):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            once()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[mode] = {"host_enqueue_ms_per_call": round(1e3 * (t1 - t0) / n, 4), "total_ms_per_call": round(1e3 * (t2 - t0) / n, 4)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
