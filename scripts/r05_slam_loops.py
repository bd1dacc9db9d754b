"""The frame-loop figure of bench.py for every engine that runs the loop, from one process (same box, same clocks):
python scripts/r05_slam_loops.py [frames] [engine ...] -> one JSON line per engine."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 13
dev = torch.device("cuda:0")
for engine in (sys.argv[2:] or ("fused", "plugin", "plugin_map_edits")):
    fig = bench.slam_loop_figure("B", dev, frames=frames, engine=engine)
    print(json.dumps({"engine": engine, "frames_per_s": fig["frames_per_s"], "runs_agree_within": fig["runs_agree_within"],
                      "runs": fig["runs"]}), flush=True)
