"""Round 6: the drop-in rasterizer's forward + backward at a workload, per capacity policy (exact: the reference's one host read per
forward; auto: the default -- group binning + sorting composite, nothing read back).  Same call, alternating rounds.
usage (GPU box): python scripts/r06_dropin.py [workload] [rounds]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from splatam_amd import rasterizer as rz  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene(wl, dev, 8)
    params_d = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    out = {"workload": wl, "rounds": []}
    for r in range(rounds):
        row = {}
        for mode in ("exact", "auto"):
            rz.set_sync_mode(mode)
            rz.reset_scene_stats()
            mpix, ms = bench.render_mpix(params_d, frames, shape, dev, reps=40)
            row[mode] = {"ms": round(ms, 4), "mpix_per_s": round(mpix, 1)}
        row["fast_path_stats"] = dict(rz.fast_path_stats)
        out["rounds"].append(row)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
