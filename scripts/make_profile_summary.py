#!/usr/bin/env python3
"""profiles/<tag>_summary.md (+ the copies of the raw files it is built from) out of one scripts/r04_profile.sh call:
gpurun_out/bench_<tag>.log (the un-profiled bench line), gpurun_out/<tag>_bench_kernel_stats.csv and <tag>_Bloop_kernel_stats.csv
(rocprofv3 --kernel-trace --stats of the bench command at B / B-loop), gpurun_out/<tag>_k7_account_*.md.
usage: scripts/make_profile_summary.py <tag>"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def stats(name):
    src = os.path.join(G, name)
    if not os.path.exists(src):
        return []
    shutil.copy(src, os.path.join(P, name))
    return list(csv.DictReader(open(src)))


def avg(rows, *needles):
    for r in rows:
        if all(n in r["Name"] for n in needles):
            return float(r["AverageNs"]) / 1e3
    return 0.0


line = [l for l in open(os.path.join(G, f"bench_{tag}.log")) if l.startswith("{")][-1]
open(os.path.join(P, f"{tag}_bench.json.log"), "w").write(line)
d = json.loads(line)
out = [f"# `{tag}`", "",
       "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop --sustain-s 1` on one MI355X",
       f"(gfx950); the bench line of the un-profiled default run of the same call is `{tag}_bench.json.log`:",
       f"**{d['value']} iters/s** (sustained {d['sustained'].get('iters_per_s')}), tracking {d['tracking_iters_per_s']}/s, mapping {d['mapping_iters_per_s']}/s; the reference's loop",
       f"statements through `splatam_amd.plugin` {d.get('plugin_iters_per_s')} /s; on the drop-in rasterizer {d['dropin_iters_per_s']} /s (rasterizer",
       f"{d.get('dropin_mapping_rasterizer_ms_per_iter')} ms of a {d.get('dropin_mapping_ms_per_iter')} ms mapping iteration); CPU oracle {d['cpu_baseline']['value']} /s on {d['cpu_baseline']['cores']} threads.", ""]
for wl, fname in (("B (300 000 Gaussians, 1200x680)", f"{tag}_bench_kernel_stats.csv"), ("B-loop (816 000 Gaussians: the map the frame loop builds)", f"{tag}_Bloop_kernel_stats.csv")):
    rows = stats(fname)
    if not rows:
        continue
    f1 = avg(rows, "fused_preprocess_kernel<2") or avg(rows, "fused_preprocess_kernel<1") or avg(rows, "fused_preprocess")
    trk = avg(rows, "render_track_fused_kernel<false>")
    k6 = avg(rows, "render_forward_kernel<6, 8, false, true, false>")
    k7 = avg(rows, "render_backward_kernel5<6, 8, 15u, 15u")
    f4, f5 = avg(rows, "ssim_forward"), avg(rows, "map_loss_backward")
    f6t, f6m, f7 = avg(rows, "fused_backward_kernel<false, false"), avg(rows, "fused_backward_kernel<true"), avg(rows, "pose_finish")
    out += [f"## workload {wl}: `{fname}`", "",
            "| kernel | avg us |", "|---|---|",
            f"| F1 `fused_preprocess_kernel` (group binning) | {f1:.1f} |",
            f"| tracking: `render_track_fused_kernel` (forward composite + loss + backward composite) | {trk:.1f} |",
            f"| mapping: K6 `render_forward_kernel<6,8,sort>` | {k6:.1f} |",
            f"| mapping: F4 `ssim_forward_kernel` / F5 `map_loss_backward_kernel` | {f4:.1f} / {f5:.1f} |",
            f"| mapping: K7 `render_backward_kernel5<6,8,15,15>` | {k7:.1f} |",
            f"| F6 `fused_backward_kernel` tracking / mapping (+ Adam of the map) | {f6t:.1f} / {f6m:.1f} |",
            f"| F7 `pose_finish_kernel` (+ pose Adam, + tile launch order in its spare workgroups) | {f7:.1f} |",
            "",
            f"Per tracking iteration {f1 + trk + f6t + f7:.0f} us of kernels, per mapping iteration {f1 + k6 + f4 + f5 + k7 + f6m + f7:.0f} us.", ""]
src = os.path.join(G, f"bench_{tag}_Bloop.log")
if os.path.exists(src):
    lines = [l for l in open(src) if l.startswith("{")]
    if lines:
        open(os.path.join(P, f"{tag}_bench_Bloop.json.log"), "w").write(lines[-1])
        dl = json.loads(lines[-1])
        rl = dl.get("roofline", {})
        out += [f"Workload B-loop, un-profiled bench line `{tag}_bench_Bloop.json.log`: **{dl['value']} iters/s** (tracking {dl['tracking_iters_per_s']}, mapping "
                f"{dl['mapping_iters_per_s']}); K7 {rl.get('kernel_ms')} ms live = {rl.get('achieved')} GB/s algorithmic = {100 * (rl.get('frac') or 0):.2f} % of HBM peak, "
                f"PMC traffic {rl.get('traffic')} B, live pairs per launch {rl.get('other', {}).get('pairs_per_launch')}, lane-operations per live pair "
                f"{rl.get('other', {}).get('lane_ops_per_live_pair')}.", ""]
for wl in ("B", "B-loop"):
    src = os.path.join(G, f"{tag}_k7_account_{wl}.md")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_k7_account_{wl}.md"))
for key in ("slam_loop", "slam_loop_plugin"):
    sl = d.get(key)
    if sl:
        r0 = sl["runs"][0]
        out += [f"`{key}` ({sl['frames_counted']} counted frames of {sl['image']}, {sl['tracking_iters_per_frame']} + {sl['mapping_iters_per_frame']} iterations per frame, map "
                f"{r0['gaussians_first_last'][0]} -> {r0['gaussians_first_last'][1]} Gaussians): **{sl['frames_per_s']} frames/s** (two runs within {100 * sl['runs_agree_within']:.1f} %), "
                f"per frame {r0['phase_ms_per_frame']}, max translation error {1e3 * r0['max_translation_error_m']:.2f} mm.", ""]
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
