#!/usr/bin/env python3
"""Per-launch timing of the rasterizer kernels in three regimes (developer tool, run through gpurun):
  (1) the same kernel launched back to back (caches warm),
  (2) the realistic sequence K1..K4 -> K6 -> K7 -> K8 repeated, each stage timed on its own,
  (3) the same sequence with an L2/MALL-polluting copy between iterations.
Prints per-launch microseconds so that warm-up / cache effects are visible."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from scripts import kernel_driver as kd  # noqa: E402
from splatam_amd import _capi  # noqa: E402


def clocks(tag):
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
        keep = [ln.strip() for ln in out.splitlines() if "sclk" in ln or "mclk" in ln or "fclk" in ln]
        print(f"[clocks {tag}] " + " | ".join(keep[:4]))
    except Exception as e:  # noqa: BLE001
        print(f"[clocks {tag}] unavailable: {e}")


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    return e0, e1


def fmt(xs):
    return " ".join(f"{x:6.1f}" for x in xs)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    dev = torch.device("cuda", 0)
    S = kd.setup(wl, dev)
    L = _capi.lib()
    pk, gr, color, depth = S['pk'], S['gr'], S['color'], S['depth']
    s = torch.cuda.current_stream(dev).cuda_stream
    cam, g, st = C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st)
    stages = {
        "K1K2": lambda: L.splat_preprocess_forward(cam, g, st, s),
        "K3K4": lambda: L.splat_bin_forward(cam, g, st, s),
        "K6": lambda: L.splat_render_forward(cam, g, st, color.data_ptr(), depth.data_ptr(), s),
        "K7": lambda: L.splat_render_backward(cam, g, st, C.byref(gr), s),
        "K8": lambda: L.splat_preprocess_backward(cam, g, st, C.byref(gr), s),
    }
    clocks("start")
    print("== (1) back-to-back, per launch [us] (K3K4 re-runs K1K2 untimed first: the scatter cursors are consumed)")
    for name, fn in stages.items():
        evs = []
        for _ in range(24):
            if name == "K3K4":
                stages["K1K2"]()
            evs.append(timed(fn))
        torch.cuda.synchronize()
        print(f"{name:5s} {fmt([1e3 * a.elapsed_time(b) for a, b in evs])}")
    clocks("after-1")
    print("== (2) realistic sequence, per iteration [us]")
    rows = {k: [] for k in stages}
    for _ in range(16):
        evs = {k: timed(fn) for k, fn in stages.items()}
        torch.cuda.synchronize()
        for k, (a, b) in evs.items():
            rows[k].append(1e3 * a.elapsed_time(b))
    for k, v in rows.items():
        print(f"{k:5s} {fmt(v)}")
    print("== (3) sequence with a 512 MiB copy between iterations [us]")
    junk_a = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
    junk_b = torch.empty_like(junk_a)
    rows = {k: [] for k in stages}
    for _ in range(8):
        junk_b.copy_(junk_a)
        evs = {k: timed(fn) for k, fn in stages.items()}
        torch.cuda.synchronize()
        for k, (a, b) in evs.items():
            rows[k].append(1e3 * a.elapsed_time(b))
    for k, v in rows.items():
        print(f"{k:5s} {fmt(v)}")
    print("== (4) sequence with host idle gaps (2 ms sleep) between iterations [us]")
    import time
    rows = {k: [] for k in stages}
    for _ in range(8):
        time.sleep(0.002)
        evs = {k: timed(fn) for k, fn in stages.items()}
        torch.cuda.synchronize()
        for k, (a, b) in evs.items():
            rows[k].append(1e3 * a.elapsed_time(b))
    for k, v in rows.items():
        print(f"{k:5s} {fmt(v)}")
    clocks("end")
    # splat_time_kernel for comparison
    for fn, name in ((0, "K6"), (1, "K7")):
        for iters in (3, 20, 200):
            ms = C.c_float(0)
            L.splat_time_kernel(fn, iters, cam, g, st, C.byref(gr), color.data_ptr(), depth.data_ptr(), s, C.byref(ms))
            print(f"splat_time_kernel {name} iters={iters}: {1e3 * ms.value:.1f} us")


if __name__ == "__main__":
    main()
