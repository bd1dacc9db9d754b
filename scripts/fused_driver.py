#!/usr/bin/env python3
"""Runs a few fused tracking + mapping iterations on workload B (or the one named) -- the target of
`rocprofv3 --pmc ...` counter passes over the fused path.  Developer tool (run through gpurun)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from splatam_amd import slam  # noqa: E402
from splatam_amd.fused import FusedEngine  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "B"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    params, variables, frames, shape = bench.build_scene(wl, dev, 2)
    eng = FusedEngine({k: v.detach().clone() for k, v in params.items()}, frames[1]['cam'])
    eng.begin_tracking(1)
    # the steady state of the frame loop: list statistics learnt (bucketed lists, group records, lists sorted inside the composite)
    for _ in range(2):
        eng.loss_backward(frames[2], 2, slam.REPLICA_MAPPING, tracking=False)
        torch.cuda.synchronize()
        assert not eng.check_overflow()
    assert eng.tile_stride > 0
    for _ in range(reps):
        eng.tracking_iteration(frames[1], slam.REPLICA_TRACKING)
        eng.mapping_iteration(frames[2], 2, slam.REPLICA_MAPPING)
    torch.cuda.synchronize()
    assert not eng.check_overflow(grow=False)
    print("fused driver ok: loss", eng.loss())


if __name__ == "__main__":
    main()
