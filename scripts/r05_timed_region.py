"""Where the fixed ~0.4 ms of bench.py's timed region (barrier, K steps, barrier) goes: the same K steps of the 2:3 mix at workload B, the host
clock read (a) when the last launch has been queued, (b) when a spin on an event recorded behind the last kernel sees it done, (c) when
torch.cuda.synchronize() returns -- for (c) with and without the spin in front of it.   usage: python scripts/r05_timed_region.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
params, variables, frames, shape = bench.build_scene("B", dev, 8)
eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'], track_max_radius=variables['max_2D_radius'])
eng.begin_tracking(1)
bench.run_steps_fused(eng, frames, 0, 1, 10, 0)
assert not eng.check_overflow()
bench.run_steps_fused(eng, frames, 0, 1, 10, 10)
start = 20
for mode in ("sync", "spin+sync", "sync", "spin+sync", "sync", "spin+sync"):
    rows = []
    for rep in range(5):
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        bench.run_steps_fused(eng, frames, 0, 1, K, start)
        ev1.record()
        t_queued = time.perf_counter()
        t_spin = None
        if mode == "spin+sync":
            while not ev1.query():
                pass
            t_spin = time.perf_counter()
        torch.cuda.synchronize(dev)
        t_sync = time.perf_counter()
        start += K
        rows.append((1e3 * (t_queued - t0), None if t_spin is None else 1e3 * (t_spin - t0), 1e3 * (t_sync - t0), ev0.elapsed_time(ev1)))
    best = min(rows, key=lambda r: r[2])
    med = sorted(rows, key=lambda r: r[2])[len(rows) // 2]
    print(f"{mode:10s} K={K}: median run: queued {med[0]:.3f} ms, spin saw it {med[1] if med[1] is None else round(med[1], 3)} ms, synchronize returned {med[2]:.3f} ms; "
          f"GPU events {med[3]:.3f} ms   (best: {best[2]:.3f} / events {best[3]:.3f})", flush=True)

# ---- the same with bench.py's own warm-up (--warmup 5: two steps, learn the lists, three steps), then K-step regions back to back
for trial in range(2):
    e2 = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'], track_max_radius=variables['max_2D_radius'])
    e2.begin_tracking(1)
    bench.run_steps_fused(e2, frames, 0, 1, 2, 0)
    assert not e2.check_overflow()
    bench.run_steps_fused(e2, frames, 0, 1, 3, 2)
    assert not e2.check_overflow(grow=False)
    st, out = 5, []
    for region in range(4):
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        bench.run_steps_fused(e2, frames, 0, 1, K, st)
        ev1.record()
        tq = time.perf_counter()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        st += K
        out.append(f"region {region}: queued {1e3 * (tq - t0):.3f}, host {1e3 * (t1 - t0):.3f}, events {ev0.elapsed_time(ev1):.3f} ms")
    print(f"bench-like warm-up, engine {trial}: " + "; ".join(out), flush=True)

# ---- does an IDLE GPU run the first milliseconds slower?  the engine above, warmed; sleep, 5 warm-up steps, one K-step region
for idle_ms in (0, 20, 200, 1000, 0, 200):
    torch.cuda.synchronize(dev)
    time.sleep(idle_ms * 1e-3)
    bench.run_steps_fused(e2, frames, 0, 1, 5, st)
    st += 5
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    bench.run_steps_fused(e2, frames, 0, 1, K, st)
    ev1.record()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    st += K
    print(f"idle {idle_ms:4d} ms, 5 steps, then {K} timed steps: host {1e3 * (t1 - t0):.3f} ms, events {ev0.elapsed_time(ev1):.3f} ms", flush=True)

# ---- how long does the ramp take?  idle 500 ms, then consecutive K-step regions without a gap
for trial in range(2):
    torch.cuda.synchronize(dev)
    time.sleep(0.5)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    evs[0].record()
    for r in range(40):
        bench.run_steps_fused(e2, frames, 0, 1, K, st)
        st += K
        evs[r + 1].record()
        if r % 10 == 9:                      # (the synthetic map moves under Adam: back to the start so that the lists keep their size)
            pass
    torch.cuda.synchronize(dev)
    ms = [evs[r].elapsed_time(evs[r + 1]) for r in range(40)]
    print("after 500 ms idle, consecutive regions (ms): " + " ".join(f"{m:.2f}" for m in ms), flush=True)
