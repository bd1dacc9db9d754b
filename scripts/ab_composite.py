#!/usr/bin/env python3
"""A/B timing of the composite kernel generations / prefetch variants inside the realistic launch sequence
(K1..K4 -> K6 -> K7 -> K8), per stage, on workload B.  Developer tool (run through gpurun)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from scripts import kernel_driver as kd  # noqa: E402
from splatam_amd import _capi  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device("cuda", 0)
    S = kd.setup(wl, dev)
    L = _capi.lib()
    pk, gr, color, depth = S['pk'], S['gr'], S['color'], S['depth']
    s = torch.cuda.current_stream(dev).cuda_stream
    cam, g, st = C.byref(pk.cam), C.byref(pk.g), C.byref(pk.st)
    stages = [
        ("K1K2", lambda: L.splat_preprocess_forward(cam, g, st, s)),
        ("K3K4", lambda: L.splat_bin_forward(cam, g, st, s)),
        ("K6", lambda: L.splat_render_forward(cam, g, st, color.data_ptr(), depth.data_ptr(), s)),
        ("K7", lambda: L.splat_render_backward(cam, g, st, C.byref(gr), s)),
        ("K8", lambda: L.splat_preprocess_backward(cam, g, st, C.byref(gr), s)),
    ]
    ref = None
    for version, variant in ((2, 0), (3, 0), (2, 0), (3, 0)):
        L.splat_debug_option(1, version)
        acc = {k: 0.0 for k, _ in stages}
        for it in range(reps + 2):
            evs = []
            for k, fn in stages:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert fn() == 0
                e1.record()
                evs.append((k, e0, e1))
            torch.cuda.synchronize()
            if it >= 2:
                for k, e0, e1 in evs:
                    acc[k] += 1e3 * e0.elapsed_time(e1) / reps
        chk = (float(color.double().sum()), float(S['bufs']['m3'].double().abs().sum()))
        if ref is None:
            ref = chk
        print(f"gen {version} variant {variant}: " + "  ".join(f"{k} {v:7.1f}" for k, v in acc.items())
              + f"   | checksum color {chk[0]:.6e} (ref {ref[0]:.6e}) |dmeans3D| {chk[1]:.6e} (ref {ref[1]:.6e})")
    L.splat_debug_option(1, 3)


if __name__ == "__main__":
    main()
