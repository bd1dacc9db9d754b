"""Round 6: where the HOST time of a drop-in forward + backward goes (cProfile over N steady-state calls, auto policy).
usage (GPU box): python scripts/r06_dropin_cprofile.py [workload] [N]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.rasterizer import GaussianRasterizer as Renderer  # noqa: E402


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

    def once():
        im, _, _ = Renderer(raster_settings=cam)(**inp)
        im.backward(gout)
        for v in inp.values():
            v.grad = None
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        once()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
