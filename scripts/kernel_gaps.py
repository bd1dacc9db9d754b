#!/usr/bin/env python3
"""Gaps between consecutive kernels of the bench loop from a rocprofv3 kernel trace (kernel_trace.csv): start[i+1] - end[i], per
(previous kernel -> next kernel) pair, median over the steady state.   usage: scripts/kernel_gaps.py <kernel_trace.csv>"""
import collections, csv, re, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "splat" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)", "").replace("void ", "")).replace("splat::", "").replace("::", "")[:48]
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g < 50:      # (host-side pauses between the bench's timed regions are not launch gaps)
        gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:24]:
    if len(v) >= 20:
        print(f"{a:48s} -> {b:48s} n {len(v):5d}  median gap {statistics.median(v):6.2f} us  p90 {sorted(v)[int(0.9 * len(v))]:6.2f}")
