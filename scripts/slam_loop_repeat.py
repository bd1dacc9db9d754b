#!/usr/bin/env python3
"""The frame-loop figure of bench.py (slam_loop) three times in one process: its run-to-run spread (the 0.2 s mapping timer varies by +-15 %).  Developer tool."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
for i in range(3):
    r = bench.slam_loop_figure("B", dev)
    print({k: r[k] for k in ("tracking_iters_per_s", "mapping_iters_per_s", "mapping_iters_per_s_incl_densify_keyframes_prune", "frames_per_s", "redone_frames")})
