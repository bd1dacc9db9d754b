"""scripts/scale_table.py: the one-command 1 -> 8 GPU run.  CPU only: the commands it would launch, and the table it
builds from bench lines (canned here: no 8-GPU node in reach), parsed back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "scale_table.py")


def _line(n, mode, value, allreduce=None, small=None, folded=None, track=5500.0, track_repl=None, map_=3800.0, map_x=None):
    return json.dumps({"metric": "x", "value": value, "unit": "iters/s", "n_gpus": n, "steps": 50, "warmup": 10, "ms_per_step": round(1e3 * 50 / value / 50, 3),
                       "scaling": "strong" if mode == "C" else "weak", "config": {"gaussians": 300000, "workload": mode},
                       "allreduce_ms": allreduce, "allreduce_small_ms": small, "allreduce_small_folded_ms": folded,
                       "tracking_iters_per_s": track, "tracking_replicated_iters_per_s": track_repl, "mapping_iters_per_s": map_,
                       "mapping_with_exchange_iters_per_s": map_x})


def test_dry_run_lists_every_launch():
    res = subprocess.run([sys.executable, SCRIPT, "--dry-run"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0
    cmds = [ln for ln in res.stdout.splitlines() if "bench.py" in ln]
    # mix and C, N = 1 once each (no collective), N = 2, 4, 8 over torch.distributed and in-stream RCCL
    assert len(cmds) == 2 * (1 + 3 * 2)
    assert sum("--instream-rccl" in c for c in cmds) == 6 and sum("--workload C" in c for c in cmds) == 7
    assert all("--gpus" in c and "--no-slam-loop" in c for c in cmds)


def test_table_from_logs(tmp_path):
    logs = {("mix", "torch", 1): _line(1, "mix", 4300.0), ("mix", "torch", 8): _line(8, "mix", 20000.0, 0.112, 0.03, 0.02, 9000.0, 5400.0, 3700.0, 3300.0),
            ("mix", "instream", 8): _line(8, "mix", 22000.0, 0.035, 0.02, 0.012, 9500.0, 5400.0, 3700.0, 3500.0),
            ("C", "torch", 1): _line(1, "C", 3800.0), ("C", "torch", 8): _line(8, "C", 24000.0, 0.11)}
    for (mode, rccl, n), text in logs.items():
        (tmp_path / f"scale_{mode}_{rccl}_N{n}.log").write_text("some stderr-ish noise\n" + text + "\n")
    res = subprocess.run([sys.executable, SCRIPT, "--from-logs", str(tmp_path), "--out", str(tmp_path / "out")], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    rows = [[c.strip() for c in ln.strip().strip("|").split("|")] for ln in res.stdout.splitlines() if ln.startswith("| mix") or ln.startswith("| C")]
    assert len(rows) == 5
    by = {(r[0], r[1], int(r[2])): r for r in rows}
    assert abs(float(by["mix", "torch", 8][5]) - 20000.0 / 4300.0 / 8) < 1e-3          # efficiency vs N = 1
    assert abs(float(by["C", "torch", 8][5]) - 24000.0 / 3800.0 / 8) < 1e-3
    assert float(by["mix", "--", 1][5]) == 1.0                                           # N = 1 is its own baseline
    # the measured all-reduce beside DESIGN.md 7's two predictions for 9.6 MB over 7 x 153 GB/s links
    assert by["mix", "torch", 8][6].startswith("0.1120 (0.110 / 0.031")
    text = res.stdout
    assert "closer to the ring prediction" in text and "closer to the one hop prediction" in text
    assert "sharded wins" in text
    assert os.path.exists(tmp_path / "out" / "scale_table.md")
