"""The frame loop on the HIP engines held to the recording of the REFERENCE'S OWN ``rgbd_slam`` (tests/golden/loop_reference.npz:
/root/reference/scripts/splatam.py:455-990 executed on the C oracle by tests/golden/make_golden_loop.py; the CPU half of this test
is tests/test_loop_golden.py).  Same frames, same configuration, same seeds through

  * ``engine="dropin"``: the reference-shaped statements on the drop-in rasterizer,
  * ``engine="plugin"``: the same statements with ``splatam_amd.plugin`` installed (the fused iteration under them),
  * ``engine="plugin_map_edits"``: ``plugin.install(map_edits=True)`` -- add_new_gaussians / prune_gaussians are adapters too (the engine
    owns the map); the call sequence recorded is still the statements' own,
  * ``engine="fused"``: ``FusedEngine``'s own loop (map edits in place on the device).

Drop-in and plug-in must make the reference loop's CALLS in its order with its arguments; all three must take its DECISIONS: the
keyframe list, the selected keyframes, the view of every mapping iteration, the tracking budget of every frame (the depth-loss
retry), the frames that map / densify, the prune schedule.  Row counts: the HIP rasterizer and the oracle are two float32
evaluations, so a pixel whose silhouette sits on the 0.5 threshold (add_new_gaussians) or a row whose scale / opacity sits on a
prune threshold after Adam's eps = 1e-15 steps (tests/test_loop_golden.py) may fall on the other side: counts may differ by a few
rows, bounded below.  Poses: within the loop tolerance of tests/test_gpu_pipeline.py."""
import os

import numpy as np
import pytest
import torch

import loop_trace as LT
from test_loop_golden import GOLD, seed_everything

pytestmark = pytest.mark.gpu

ROW_TOL = 0.002          # relative bound on a row-count difference (measured: 0.021 %, 1 row of 4 7xx; profiles/r05_loop_golden_gpu.log)


def run_engine(case, engine):
    from splatam_amd import pipeline, plugin, slam
    cfg = LT.load_config(GOLD, case)
    ds = LT.RecordedRGBDSequence(GOLD, case, device="cuda")
    rec = LT.LoopRecorder().wrap(slam).wrap(pipeline).wrap(plugin)
    try:
        seed_everything(cfg['seed'])
        params, variables, stats = pipeline.rgbd_slam(ds, cfg, engine=engine)
        torch.cuda.synchronize()
    finally:
        rec.restore()
    return cfg, rec, params, variables, stats


def gold_decisions(case, cfg):
    n = GOLD[f"{case}/frames/color"].shape[0]
    return LT.per_frame_decisions(GOLD[f"{case}/events"], GOLD[f"{case}/selected"], GOLD[f"{case}/final/keyframe_time_indices"], n,
                                  cfg['mapping']['pruning_dict'])


def close_rows(a, b):
    return a == b or (a is not None and b is not None and abs(a - b) <= max(3, int(ROW_TOL * b)))


def check_decisions(case, cfg, stats, what):
    want, got = gold_decisions(case, cfg), stats['decisions']
    assert len(want) == len(got)
    worst = 0.0
    for w, g in zip(want, got):
        for k in ('time_idx', 'tracking_iters', 'selected', 'views', 'keyframe'):
            assert w[k] == g[k], (what, w['time_idx'], k, w[k], g[k])
        assert [p[0] for p in w['prunes']] == [p[0] for p in g['prunes']], (what, w['time_idx'], w['prunes'], g['prunes'])
        pairs = [(w['rows_after_add'], g['rows_after_add']), (w['rows_end'], g['rows_end'])]
        pairs += [(x, y) for pw, pg in zip(w['prunes'], g['prunes']) for x, y in zip(pw[1:], pg[1:])]
        for x, y in pairs:
            assert close_rows(x, y), (what, w['time_idx'], w, g)
            if x:
                worst = max(worst, abs(x - y) / x)
    print(f"{what}: decisions equal the reference loop's on {len(want)} frames; largest row-count difference {100 * worst:.3f} %")
    assert stats['keyframe_time_indices'] == GOLD[f"{case}/final/keyframe_time_indices"].tolist()
    assert stats['redone_iterations'] == 0


def check_trajectory(case, params, what):
    for k, tol in (('cam_unnorm_rots', 2e-4), ('cam_trans', 2e-4)):
        d = np.abs(GOLD[f"{case}/final/{k}"] - params[k].detach().cpu().numpy())
        print(f"{what}: {k}: max |difference| to the reference loop {d.max():.1e}")
        assert d.max() < tol, (what, k, d.max())


@pytest.mark.parametrize("case", ["base", "variant", "gtposes"])
@pytest.mark.parametrize("engine", ["dropin", "plugin", "plugin_map_edits"])
def test_statement_engines_make_the_reference_loops_calls(case, engine):
    cfg, rec, params, variables, stats = run_engine(case, engine)
    events, values, selected = rec.arrays()
    gold = GOLD[f"{case}/events"]
    diff = LT.first_difference(gold, events, ignore_row_counts=True)
    assert diff is None, f"{case}/{engine}: reference vs pipeline: {diff[1]}"
    assert selected.tolist() == GOLD[f"{case}/selected"].tolist()
    check_decisions(case, cfg, stats, f"{case}/{engine}")
    check_trajectory(case, params, f"{case}/{engine}")
    is_loss = events[:, 0] == LT.LOSS
    rel = np.abs(values[is_loss] - GOLD[f"{case}/values"][is_loss]) / np.abs(GOLD[f"{case}/values"][is_loss])
    print(f"{case}/{engine}: {int(is_loss.sum())} losses, relative difference to the reference loop: first {rel[0]:.1e}, median "
          f"{np.median(rel):.1e}, max {rel.max():.1e}")
    assert rel[0] < 1e-4 and np.median(rel) < 2e-3 and rel.max() < 3e-2
    if engine.startswith("plugin"):
        assert stats['plugin']['skipped_iterations'] == 0, stats['plugin']
    if engine == "plugin_map_edits":
        assert stats['plugin']['engines_built'] == 1, stats['plugin']


@pytest.mark.parametrize("case", ["base", "variant", "gtposes"])
def test_fused_engine_takes_the_reference_loops_decisions(case):
    cfg, rec, params, variables, stats = run_engine(case, "fused")
    check_decisions(case, cfg, stats, f"{case}/fused")
    check_trajectory(case, params, f"{case}/fused")
    n = min(variables['timestep'].shape[0], GOLD[f"{case}/final/timestep"].shape[0])
    ts = variables['timestep'][:n].cpu().numpy()
    assert float((ts != GOLD[f"{case}/final/timestep"][:n]).mean()) < 2e-2
