#!/usr/bin/env python3
"""Does RCCL initialise on this box?  One rank: torch.distributed's nccl backend, then splatam_amd.dist.InStreamRccl, each with the
environment given on the command line (KEY=VALUE ...).  Developer tool (gpurun)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
what = sys.argv[1]
torch.cuda.set_device(0)
if what == "pg":
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize(); print("pg ok", t[0].item())
else:
    from splatam_amd.dist import InStreamRccl
    c = InStreamRccl(0, 1); t = torch.ones(8, device="cuda"); c.all_reduce(t); torch.cuda.synchronize(); print("instream ok", t[0].item())
''' % ROOT
for env_extra in ({}, {"NCCL_SOCKET_IFNAME": "lo"}):
    for what in ("pg", "instream"):
        env = dict(os.environ, NCCL_DEBUG="WARN", **env_extra)
        r = subprocess.run([sys.executable, "-c", CHILD, what], env=env, capture_output=True, text=True, timeout=180)
        tail = (r.stdout + r.stderr).strip().splitlines()[-6:]
        print(f"== {what} {env_extra}: rc {r.returncode}")
        for line in tail:
            print("   ", line[:220])
