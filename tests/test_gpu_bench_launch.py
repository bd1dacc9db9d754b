"""`python bench.py --gpus 2` started from a bare shell on a ONE-GPU box: bench.py launches its two ranks itself (torch.distributed.run,
127.0.0.1), the ranks share the GPU over gloo, rank 0 prints the JSON line with the N > 1 fields (sharded tracking, the exchanged mapping
step, both small-message forms).  The launch path the driver's scaling run takes; not a scaling figure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_from_a_bare_shell():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SPLAT_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "5", "--workload", "A",
                        "--no-roofline", "--sustain-s", "0.2"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    for k in ("allreduce_ms", "allreduce_small_ms", "allreduce_small_folded_ms", "tracking_replicated_iters_per_s",
              "mapping_with_exchange_iters_per_s"):
        assert d[k] is not None and d[k] > 0, k
    assert "gloo" in d["collectives"]
