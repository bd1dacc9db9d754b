#!/usr/bin/env python3
"""profiles/<tag>_summary.md from a bench log (gpurun_out/bench_<tag>.log) and the rocprofv3 kernel stats of the same command
(gpurun_out/<tag>_bench_kernel_stats.csv).  usage: scripts/make_profile_summary.py r01_v7 [pmc-file-name]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
pmc = sys.argv[2] if len(sys.argv) > 2 else None
src_csv = os.path.join(ROOT, "gpurun_out", f"{tag}_bench_kernel_stats.csv")
src_log = os.path.join(ROOT, "gpurun_out", f"bench_{tag}.log")
shutil.copy(src_csv, os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv"))
line = [l for l in open(src_log) if l.startswith("{")][-1]
open(os.path.join(ROOT, "profiles", f"{tag}_bench.json.log"), "w").write(line)
d = json.loads(line)
rows = list(csv.DictReader(open(src_csv)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def avg(sub):
    for r in rows:
        if sub in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return 0.0


k7m = avg("render_backward_kernel5<6, 8, 15u, 15u") or avg("render_backward_kernel<6, 8, 15u, 15u")
k7t = avg("render_backward_kernel5_w5<6, 8, 15u, 8u") or avg("render_backward_kernel5<6, 8, 15u, 8u") or avg("render_backward_kernel<6, 8, 15u, 8u")
k6s = avg("render_forward_kernel<6, 8, false, true, false>") or avg("render_forward_kernel<6, 8, false, true>")
k6t = avg("render_forward_kernel<6, 8, false, true, true>")
f1, f4, f5 = avg("fused_preprocess_kernel<2") or avg("fused_preprocess"), avg("ssim_forward"), avg("map_loss_backward")
f6 = avg("fused_backward_kernel<false, false") or avg("fused_backward_kernel<false>") or avg("fused_backward_kernel")   # tracking form
f6m = avg("fused_backward_kernel<true") or f6                                     # single-view mapping step: Adam inside
f7, ap, am = avg("pose_finish"), avg("adam_pose"), avg("adam_map")
out = [f"# `{tag}`: composites on compact visit lists (one-byte entries, four per trip), forward composite with one 4x4-pixel block per 16-lane row",
       "(four Gaussians per trip), group binning, generation-5 backward composite, Adam of the map inside F6 for the single-view mapping step\n",
       "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-slam-loop` on MI355X (gfx950), workload B",
       "(300k Gaussians, 1200x680), engine = fused.  The run also times the drop-in path (reference-shaped PyTorch glue around the drop-in",
       f"rasterizer), hence the MIOpen / rocBLAS rows and the 3-channel kernels.  Full CSV: `{tag}_bench_kernel_stats.csv`; bench line of the",
       f"un-profiled run (with `cpu_baseline` and `slam_loop`): `{tag}_bench.json.log` (**{d['value']} iters/s**; tracking {d['tracking_iters_per_s']}/s, mapping",
       f"{d['mapping_iters_per_s']}/s; the reference's loop statements through splatam_amd.plugin {d.get('plugin_iters_per_s')} iters/s, on the drop-in path {d['dropin_iters_per_s']} iters/s; CPU oracle {d['cpu_baseline']['value']} iters/s on {d['cpu_baseline']['cores']} threads)."]
if pmc:
    out.append(f"PMC passes of the fused path: `{pmc}` (`scripts/pmc.sh`).")
out += ["", f"Total kernel time {tot / 1e6:.1f} ms.\n", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:28]:
    if "at::native" in r["Name"]:
        continue
    out.append(f"| `{r['Name'][:92]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
ro = d["roofline"]
out += ["",
        f"Reading: the dominant kernel of the fused iteration is `render_backward_kernel5<6,8,15,15>` (mapping form, {k7m:.1f} us;",
        f"tracking form `<6,8,15,8,false>` {k7t:.1f} us); bench.py's live HIP-event figure for it is {ro['kernel_ms'] * 1e3:.1f} us (`roofline.kernel_ms`, {ro['achieved']} GB/s",
        f"algorithmic = {100 * ro['frac']:.2f} % of HBM peak; K6 {ro['other']['render_forward_ms'] * 1e3:.1f} us = {ro['other']['render_forward_GBps']} GB/s).",
        f"Per fused tracking iteration (us): fused_preprocess {f1:.1f} + render_forward<6,8,sort,+tracking loss> {k6t:.1f} + render_backward<6,8,15,8> {k7t:.1f} +",
        f"fused_backward {f6:.1f} + pose_finish {f7:.1f} + adam_pose {ap:.1f} = {f1 + k6t + k7t + f6 + f7 + ap:.0f} -> {d['tracking_iters_per_s']:.0f} iterations/s measured; mapping: {f1:.1f} +",
        f"render_forward<6,8,sort> {k6s:.1f} + ssim {f4:.1f} + map_loss_backward {f5:.1f} + render_backward {k7m:.1f} + fused_backward<Adam> {f6m:.1f} + {f7:.1f} = "
        f"{f1 + k6s + f4 + f5 + k7m + f6m + f7:.0f} -> {d['mapping_iters_per_s']:.0f} iterations/s (adam_map as a kernel of its own, {am:.1f} us, only in the exchanged / batched step).",
        "GPU-bound, no host gaps, no memset launches.\n"]
sl = d.get("slam_loop")
if sl:
    out += [f"`slam_loop` (the whole frame loop of scripts/splatam.py:654-905 on a synthetic {sl['image']} RGB-D sequence of a smooth textured surface, Replica",
            f"iteration counts): {sl['frames']} frames, map {sl['gaussians_per_frame'][0]} -> {sl['gaussians_per_frame'][-1]} Gaussians, tracking {sl['tracking_iters_per_s']} it/s, mapping incl.",
            f"densification / keyframe selection / pruning / list re-learning {sl['mapping_iters_per_s_incl_densify_keyframes_prune']} it/s, {sl['frames_per_s']} frames/s, max translation error",
            f"{sl['max_translation_error_m'] * 1e3:.2f} mm."]
open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[-16:]))
