#!/usr/bin/env python3
"""The 1 -> 8 GPU run in one command: what DESIGN.md 7 predicts, measured, as one table.

On an 8-GPU MI355X node:
    python scripts/scale_table.py                       # runs bench.py --gpus {1,2,4,8} (the 2:3 mix) and --workload C, each over
                                                        # torch.distributed's RCCL and with --instream-rccl; ~10 minutes
    python scripts/scale_table.py --gpus 1 2            # a subset
    python scripts/scale_table.py --dry-run             # print the commands only
    python scripts/scale_table.py --from-logs DIR       # build the table from earlier runs' logs (scale_<mode>_<rccl>_N<n>.log)

Per (mode, collective, N): whole-job rate, scaling efficiency against N = 1 (the driver computes its own from the per-N values; this is
for the reader), `allreduce_ms` of the 9.6 MB gradient bucket (ring: 2 (N-1)/N S / 153 GB/s = 110 us at N = 8; one hop: ~31 us), the
small-message latency L of the 256-byte exchange of tile-row-sharded tracking (sharding pays while L < 114 us), sharded vs replicated
tracking and local vs exchanged mapping rates.  The last block answers DESIGN.md 7's three open questions from those numbers."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = (("mix", []), ("C", ["--workload", "C"]))
RCCL = (("torch", []), ("instream", ["--instream-rccl"]))
# 7 xGMI links x ~153 GB/s per GPU, fully connected
LINK_GBS = 153.0


def command(mode_args, rccl_args, n, steps, warmup):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup),
            "--no-cpu-baseline", "--no-slam-loop", "--sustain-s", "2"] + mode_args + (rccl_args if n > 1 else [])


def last_json(text):
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def predicted_allreduce_ms(bytes_, n):
    if n <= 1:
        return None, None
    ring = 2.0 * (n - 1) / n * bytes_ / (LINK_GBS * 1e9) * 1e3
    one_hop = 2.0 * (bytes_ / n) / (LINK_GBS * 1e9) * 1e3 + 0.015
    return ring, one_hop


def build_table(results):
    """results: {(mode, rccl, n): bench JSON}.  Returns the markdown text."""
    out = ["| mode | collectives | N | value (it/s) | ms/step | efficiency vs N=1 | allreduce 9.6 MB ms (ring / one-hop predicted) | small exchange ms (folded) | "
           "tracking sharded / replicated it/s | mapping local / with exchange it/s |", "|---|---|---|---|---|---|---|---|---|---|"]
    base = {}
    for (mode, rccl, n), d in sorted(results.items()):
        if n == 1:
            base[mode] = d["value"]
    for (mode, rccl, n), d in sorted(results.items()):
        b = base.get(mode)
        # whole-job rate against N x the one-GPU rate (mix: a mapping step renders one view PER RANK, tracking steps are counted once --
        # its ideal is below N; C: 8 views per step whatever N -- ideal N)
        eff = None if not b else d["value"] / b / n
        grad_bytes = 8 * 4 * d["config"]["gaussians"]
        ring, hop = predicted_allreduce_ms(grad_bytes, n)
        f = lambda v, p=4: "--" if v is None else f"{v:.{p}f}"       # noqa: E731
        out.append(f"| {mode} | {rccl if n > 1 else '--'} | {n} | {d['value']:.1f} | {d['ms_per_step']:.3f} | {f(eff, 3)} | "
                   f"{f(d.get('allreduce_ms'))} ({f(ring, 3)} / {f(hop, 3)}) | {f(d.get('allreduce_small_ms'))} ({f(d.get('allreduce_small_folded_ms'))}) | "
                   f"{f(d.get('tracking_iters_per_s'), 0)} / {f(d.get('tracking_replicated_iters_per_s'), 0)} | "
                   f"{f(d.get('mapping_iters_per_s'), 0)} / {f(d.get('mapping_with_exchange_iters_per_s'), 0)} |")
    # DESIGN.md 7's open questions
    ans = ["", "What the table settles (DESIGN.md 7):"]
    top = max((n for (_, _, n) in results), default=1)
    for rccl in ("torch", "instream"):
        d = results.get(("mix", rccl, top))
        if d and top > 1 and d.get("allreduce_ms") is not None:
            ring, hop = predicted_allreduce_ms(8 * 4 * d["config"]["gaussians"], top)
            which = "ring" if abs(d["allreduce_ms"] - ring) < abs(d["allreduce_ms"] - hop) else "one hop"
            ans.append(f"* gradient all-reduce at N = {top} ({rccl}): {d['allreduce_ms']:.3f} ms -- closer to the {which} prediction "
                       f"(ring {ring:.3f}, one hop {hop:.3f}).")
        if d and top > 1 and d.get("allreduce_small_folded_ms") is not None:
            L = d["allreduce_small_folded_ms"] * 1e3
            sh, rp = d.get("tracking_iters_per_s"), d.get("tracking_replicated_iters_per_s")
            ans.append(f"* small-message latency L ({rccl}) = {L:.0f} us (sharding was predicted to pay while L < 114 us): tracking sharded {sh} vs "
                       f"replicated {rp} it/s -> {'sharded' if (sh or 0) > (rp or 0) else 'replicated'} wins.")
    if len(ans) == 2:
        ans.append("* (single-GPU results only: nothing to settle)")
    return "\n".join(out + ans) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--from-logs", default=None, help="read scale_<mode>_<rccl>_N<n>.log from this directory instead of running")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale"))
    args = ap.parse_args()
    results = {}
    os.makedirs(args.out, exist_ok=True)
    for mode, margs in MODES:
        for rccl, rargs in RCCL:
            for n in args.gpus:
                if n == 1 and rccl != "torch":
                    continue                    # (one rank issues no collective)
                if mode == "C" and 8 % n != 0:
                    continue
                tag = f"scale_{mode}_{rccl}_N{n}"
                cmd = command(margs, rargs, n, args.steps, args.warmup)
                if args.dry_run:
                    print(" ".join(cmd))
                    continue
                if args.from_logs:
                    path = os.path.join(args.from_logs, tag + ".log")
                    if not os.path.exists(path):
                        continue
                    text = open(path).read()
                else:
                    res = subprocess.run(cmd, capture_output=True, text=True)
                    text = res.stdout
                    with open(os.path.join(args.out, tag + ".log"), "w") as f:
                        f.write(res.stdout)
                    if res.returncode != 0:
                        sys.stderr.write(f"{tag}: rc {res.returncode}\n{res.stderr[-1500:]}\n")
                        continue
                d = last_json(text)
                if d is not None and "value" in d:
                    results[(mode, rccl, n)] = d
    if args.dry_run:
        return 0
    table = build_table(results)
    with open(os.path.join(args.out, "scale_table.md"), "w") as f:
        f.write(table)
    print(table)
    return 0 if results else 1


if __name__ == "__main__":
    sys.exit(main())
