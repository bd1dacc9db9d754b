#!/usr/bin/env python3
"""profiles/<tag>_pmc.json from the counter passes of scripts/pmc.sh (gpurun_out/pmc_<tag>.txt) and, optionally, the rocprofv3
--stats kernel CSV of an un-profiled-counter run (average durations).  bench.py loads the newest profiles/*_pmc.json for its
`roofline.traffic` / `roofline.other` block instead of literals.
  usage: scripts/pmc_to_json.py <pmc tag> <out tag> [kernel_stats.csv] [workload]"""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out_tag = sys.argv[1], sys.argv[2]
stats = sys.argv[3] if len(sys.argv) > 3 else None
workload = sys.argv[4] if len(sys.argv) > 4 else "B"
kern = {}
for line in open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}.txt")):
    if line.startswith("##") or "=" not in line:
        continue
    m = re.match(r"^(.*?)\s((?:[A-Za-z_0-9]+=[-+.e0-9]+\s)+)\(n=(\d+)\)", line.strip() + " ")
    if not m:
        continue
    d = kern.setdefault(m.group(1).strip(), {})
    for kv in m.group(2).split():
        k, v = kv.split("=")
        d[k] = float(v)
if stats:
    for r in csv.DictReader(open(stats)):
        m = re.search(r"([A-Za-z_0-9]+_kernel[0-9]*(?:_w[0-9])?(?:<[^>]*>)?)", r["Name"])
        if m and m.group(1) in kern:
            kern[m.group(1)]["avg_us"] = float(r["AverageNs"]) / 1e3
            kern[m.group(1)]["calls"] = int(r["Calls"])
for k, d in kern.items():
    # HBM-side bytes per launch as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950
    # FETCH_SIZE reports 1/2 of a wide streaming read (x2); WRITE_SIZE is uncalibrated for 4-byte atomics (taken as is)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["traffic_bytes"] = int((2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
    if "SQ_INSTS_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0                     # GRBM_GUI_ACTIVE sums the 8 XCDs
        d["kernel_cycles"] = cycles
        # vector-pipe cycles of the kernel's instruction mix with the issue costs MEASURED on gfx950 (profiles/r03_valu_issue_bench.txt,
        # r03_visit_replay.txt): a plain wave64 VALU instruction 2 cycles, compares / selects / DPP 4 -- 2.5 on average over the
        # composites' non-transcendental mix (counted in their ISA) --, exp / rcp ~20 inside the visit's dependent mix (8 back to back).
        # Round 2 priced every instruction at 4 cycles ("valu_issue_frac").
        trans = d.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        d["valu_cycles_frac"] = round(((d["SQ_INSTS_VALU"] - trans) * 2.5 + trans * 20.0) / (1024.0 * cycles), 4)
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip()
out = {"source": f"scripts/pmc.sh {tag} {workload} fused (rocprofv3 --pmc, separate passes: inst / wait / fetch / write)", "workload": workload,
       "git_head": head, "kernels": kern}
path = os.path.join(ROOT, "profiles", f"{out_tag}_pmc.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(path, len(kern), "kernels")
