#!/usr/bin/env python3
"""Launches the rasterizer kernels of one call a few times on a valid state (workload B by
default) -- the target of `rocprofv3 --pmc ...` counter passes and of per-stage timing.
Developer tool (run through gpurun)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import _capi, slam  # noqa: E402
from splatam_amd import rasterizer as rz  # noqa: E402


def setup(workload, dev):
    params, variables, frames, shape = bench.build_scene(workload, dev, 1)
    N, W, H = shape
    with torch.no_grad():
        tg = slam.transform_to_frame(params, 1, False, False)
        rv = slam.transformed_params2rendervar(params, tg)
    empty = torch.empty(0, device=dev)
    args = (frames[1]['cam'], rv['means3D'].contiguous(), rv['colors_precomp'].detach().contiguous(),
            rv['opacities'].detach().reshape(-1).contiguous(), rv['scales'].detach().contiguous(),
            rv['rotations'].detach().contiguous(), empty, empty)
    color, radii, depth, pk = rz.rasterize_forward(*args)
    gcol = torch.randn_like(color)
    f32 = torch.float32
    bufs = dict(accum=torch.empty(N, _capi.SPLAT_GRAD_STRIDE, dtype=f32, device=dev), m3=torch.empty(N, 3, device=dev),
                m2=torch.empty(N, 3, device=dev), col=torch.empty(N, 3, device=dev), op=torch.empty(N, device=dev),
                sc=torch.empty(N, 3, device=dev), ro=torch.empty(N, 4, device=dev), gcol=gcol)
    gr = _capi.SplatGrads()
    gr.dL_dcolor, gr.accum = gcol.data_ptr(), bufs['accum'].data_ptr()
    gr.dL_dmeans3D, gr.dL_dmeans2D, gr.dL_dcolors = bufs['m3'].data_ptr(), bufs['m2'].data_ptr(), bufs['col'].data_ptr()
    gr.dL_dopacities, gr.dL_dscales, gr.dL_drotations = bufs['op'].data_ptr(), bufs['sc'].data_ptr(), bufs['ro'].data_ptr()
    return dict(pk=pk, gr=gr, color=color, depth=depth, bufs=bufs, shape=shape, args=args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="B")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--time", action="store_true", help="print per-stage times (torch events) instead of just launching")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    S = setup(a.workload, dev)
    L = _capi.lib()
    pk, gr, color, depth = S['pk'], S['gr'], S['color'], S['depth']
    s = torch.cuda.current_stream(dev).cuda_stream
    cam, g, st = C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st)

    def chain12():
        assert L.splat_preprocess_forward(cam, g, st, s) == 0

    def chain1234():
        assert L.splat_preprocess_forward(cam, g, st, s) == 0
        assert L.splat_bin_forward(cam, g, st, s) == 0
    stages = [("K1+K2 preprocess+scan", chain12), ("K1..K4 preprocess+scan+scatter+sort", chain1234),
              ("K6 render_forward", lambda: L.splat_render_forward(cam, g, st, color.data_ptr(), depth.data_ptr(), s)),
              ("memset+K7 render_backward", lambda: L.splat_render_backward(cam, g, st, C.byref(gr), s)),
              ("K8+K9 preprocess_backward", lambda: L.splat_preprocess_backward(cam, g, st, C.byref(gr), s))]
    N, W, H = S['shape']
    if a.time:
        stat = pk.tensors['status'].tolist()
        print(f"workload {a.workload}: P={N} {W}x{H} num_rendered={stat[0]} longest list={stat[2]}")
    for name, fn in stages:
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if a.time:
            print(f"{name:40s} {1e3 * e0.elapsed_time(e1) / a.reps:9.1f} us")


if __name__ == "__main__":
    main()
