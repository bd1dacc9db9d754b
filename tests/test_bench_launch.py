"""`python bench.py --gpus N` from a bare shell (the way the driver's scaling run starts it): bench.py re-launches itself under
torch.distributed.run with N ranks; rank 0 prints ONE JSON line.  Here without a GPU: `--launch-check` stops after the rendezvous and
one all-reduce (gloo).  The real two-rank bench line on one GPU over gloo is tests/test_gpu_bench_launch.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_gpus_2_self_launches_and_rank0_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["launch_check"] is True and d["world"] == 2 and d["sum_of_ranks_plus_one"] == 3.0
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1          # rank 0 only
    # torchrun would have pinned OMP_NUM_THREADS=1: the ranks share the host cores instead
    assert int(d["omp_num_threads"]) == max(1, (os.cpu_count() or 1) // 2)


def test_single_rank_launch_check_needs_no_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["world"] == 1


def test_roofline_rows_find_their_kernels_in_the_committed_profiles():
    """bench.py fills `roofline.other.kernels` from the newest committed kernel-stats CSV of the bench command and the newest
    committed counter file: every kernel it prices must be present in the trace under the name it looks for (a renamed or
    re-templated kernel would silently drop its row), and counters are only paired with a kernel of the same name."""
    sys.path.insert(0, ROOT)
    import bench
    for workload in ("B", "B-loop"):
        trace = bench.load_kernel_trace(workload)
        assert trace is not None, workload
        for needle in ("render_track_fused_kernel", "fused_preprocess_kernel", "ssim_forward_kernel<", "map_loss_backward_kernel<",
                       "fused_backward_kernel", "render_backward_kernel5<6, 8, 15u, 15u", "render_forward_kernel<6, 8, false, true, false>"):
            us = bench.trace_avg_us(trace, needle)
            assert us is not None and 1.0 < us < 2000.0, (workload, needle, us)
        pmc = bench.load_pmc(workload)
        assert pmc is not None and pmc[1].get("workload") == workload
        k7 = bench.pmc_kernel(pmc, "render_backward_kernel", "<6, 8, 15u, 15u")
        assert k7 is not None and k7.get("traffic_bytes", 0) > 0
        for needle in ("ssim_forward_kernel<", "map_loss_backward_kernel<"):
            d = bench.pmc_kernel(pmc, needle)
            assert d is None or needle in d["name"]
