#!/usr/bin/env python3
"""After scripts/r06_profile.sh <tag> <head> (one gpurun call): copy the set from gpurun_out/ into profiles/, convert the counter passes
(scripts/pmc_to_json.py) with the call's git head, write profiles/<tag>_meta.json (head + the files of the set: bench.py pairs a trace
with a counter file only when both name the same head) and profiles/r06_workloads.md (it/s, K6 / K7 us, roofline fraction per workload).
   usage: scripts/r05_collect.py <tag>"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
head = open(os.path.join(G, f"{tag}_head.txt")).read().strip()
files = []


def keep(src_name, dst_name=None):
    src = os.path.join(G, src_name)
    if not os.path.exists(src) or os.path.getsize(src) == 0:
        return None
    dst_name = dst_name or src_name
    shutil.copy(src, os.path.join(P, dst_name))
    files.append(dst_name)
    return os.path.join(P, dst_name)


def last_json(path):
    if not path:
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except Exception:
                pass
    return None


lines = {"B": last_json(keep(f"bench_{tag}.log", f"{tag}_bench.json.log"))}
lines["B-loop"] = last_json(keep(f"bench_{tag}_Bloop.log", f"{tag}_bench_Bloop.json.log"))
for wl in ("A", "D", "E", "E-clustered"):
    lines[wl] = last_json(keep(f"bench_{tag}_{wl}.log", f"{tag}_bench_{wl}.json.log"))
for sfx in ("bench", "Bloop"):
    keep(f"{tag}_{sfx}_kernel_stats.csv")
for wl in ("B", "B-loop"):
    keep(f"{tag}_k7_account_{wl}.md")
for name in (f"{tag}_dropin_B.json", f"{tag}_dropin_host_B.json", f"{tag}_dropin_kernel_stats.csv"):
    keep(name)
for pmc_tag, out_tag, stats, wl in ((tag, tag, f"{tag}_bench_kernel_stats.csv", "B"), (f"{tag}_Bloop", f"{tag}_Bloop", f"{tag}_Bloop_kernel_stats.csv", "B-loop")):
    txt = os.path.join(G, f"pmc_{pmc_tag}.txt")
    if os.path.exists(txt) and os.path.getsize(txt) > 200:
        keep(f"pmc_{pmc_tag}.txt", f"{out_tag}_fused_pmc_kernels.txt")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "pmc_to_json.py"), pmc_tag, out_tag, os.path.join(G, stats), wl])
        path = os.path.join(P, f"{out_tag}_pmc.json")
        d = json.load(open(path))
        d["git_head"] = head                                  # the head of the CALL, not of the tree the conversion runs in
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
        files.append(f"{out_tag}_pmc.json")
json.dump({"tag": tag, "git_head": head, "files": sorted(files),
           "source": "scripts/r06_profile.sh: one gpurun call on one MI355X box"}, open(os.path.join(P, f"{tag}_meta.json"), "w"), indent=1)

# ---- the table
rows = []
for wl in ("A", "B", "B-loop", "D", "E", "E-clustered"):
    d = lines.get(wl)
    if not d:
        rows.append(f"| {wl} | (no line) | | | | | | |")
        continue
    r = d.get("roofline") or {}
    o = r.get("other") or {}
    cfgd = d["config"]
    sus = (d.get("sustained") or {}).get("iters_per_s") or float("nan")
    rows.append(f"| {wl} | {cfgd['gaussians']:,} | {cfgd['width']}x{cfgd['height']} | {d['value']:.0f} ({sus:.0f}) | {d['tracking_iters_per_s']:.0f} / "
                f"{d.get('tracking_full_gradients_iters_per_s') or float('nan'):.0f} | {d['mapping_iters_per_s']:.0f} | "
                f"{1e3 * (o.get('render_forward_ms') or float('nan')):.1f} / {1e3 * (o.get('render_backward_ms') or float('nan')):.1f} | "
                f"{r.get('frac', float('nan')):.4f} | {d.get('plugin_iters_per_s', float('nan')):.0f} | {d.get('dropin_iters_per_s', float('nan')):.0f} |")
with open(os.path.join(P, "r06_workloads.md"), "w") as f:
    f.write(f"# Every BASELINE configuration on the final code of round 6 (`{tag}` @ `{head}`, one `gpurun` call, one MI355X)\n\n"
            "`bench.py --workload <W>`: the 2:3 tracking:mapping mix (`value`), the phases alone (tracking: the headline form / with every "
            "gradient of the reference's backward formed), the two composites live (HIP events), the fraction of the 8 TB/s HBM peak the "
            "dominant kernel's algorithmic bytes reach, and the reference's own loop statements through the plug-in / on the drop-in package.\n\n"
            "| workload | Gaussians | frame | mix it/s: the K timed steps (sustained) | tracking it/s (headline / full gradients) | mapping it/s | K6 / K7 us | roofline frac | plug-in it/s | drop-in it/s |\n"
            "|---|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n")
    b = lines.get("B") or {}
    for key in ("slam_loop", "slam_loop_plugin", "slam_loop_plugin_map_edits"):
        if key in b:
            f.write(f"\n`{key}` at B: {b[key]['frames_per_s']} frames/s (runs agree within {b[key]['runs_agree_within']}).\n")
print("\n".join(rows))
print("files:", files)
