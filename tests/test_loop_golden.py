"""The frame loop held to a recording of the REFERENCE'S OWN ``rgbd_slam`` (/root/reference/scripts/splatam.py:455-990, executed by
tests/golden/make_golden_loop.py on the C oracle).  CPU: ``pipeline.rgbd_slam(engine="dropin")`` with the same oracle behind the
``Renderer`` name must make the same calls in the same order with the same arguments -- keyframe cadence (:912-925), the
map_every / add_new_gaussians gating (:777-795), the selected-keyframe window and the random view of every mapping iteration
(:809-845), the prune schedule (:858), the doubled tracking budget (:713-738), initialize_camera_pose (:423-441), the number of
optimizer steps -- produce the same row counts, and the same losses / poses / parameters to float32 rounding."""
import os
import random

import numpy as np
import pytest
import torch

import loop_trace as LT

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_reference.npz"))
CASES = ("base", "variant", "gtposes")


def seed_everything(seed):
    """/root/reference/utils/common_utils.py:8-22"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def run_on_oracle(case):
    from oracle import c_ref
    from splatam_amd import pipeline, slam
    cfg = LT.load_config(GOLD, case)
    ds = LT.RecordedRGBDSequence(GOLD, case)
    saved = slam.Renderer
    slam.Renderer = c_ref.CRasterizer
    rec = LT.LoopRecorder().wrap(slam).wrap(pipeline)
    try:
        seed_everything(cfg['seed'])
        params, variables, stats = pipeline.rgbd_slam(ds, cfg, engine="dropin")
    finally:
        rec.restore()
        slam.Renderer = saved
    return cfg, rec, params, variables, stats


@pytest.fixture(scope="module", params=CASES)
def run(request):
    return (request.param,) + run_on_oracle(request.param)


def test_call_sequence_equals_the_reference_loop(run):
    case, cfg, rec, params, variables, stats = run
    events, values, selected = rec.arrays()
    gold = GOLD[f"{case}/events"]
    diff = LT.first_difference(gold, events)
    assert diff is None, f"{case}: reference vs pipeline: {diff[1]}"
    assert selected.tolist() == GOLD[f"{case}/selected"].tolist()
    assert stats['keyframe_time_indices'] == GOLD[f"{case}/final/keyframe_time_indices"].tolist()


def test_losses_follow_the_reference_loop(run):
    """Every iteration's loss.  Two float32 formulations of the same statements (the mirror computes `pts @ R.T + t` where the
    reference multiplies 4x4 matrices, sums `where(mask, x, 0)` where the reference indexes with the mask) agree to rounding on the
    first iterations; afterwards the difference is fed through Adam, whose step is lr * sign-like for gradients at rounding level
    (eps 1e-8 against summed gradients while tracking, 1e-15 while mapping), so the trajectories drift apart by parts in 1e4."""
    case, cfg, rec, params, variables, stats = run
    events, values, _ = rec.arrays()
    gold_v = GOLD[f"{case}/values"]
    is_loss = events[:, 0] == LT.LOSS
    rel = np.abs(values[is_loss] - gold_v[is_loss]) / np.abs(gold_v[is_loss])
    print(f"{case}: {int(is_loss.sum())} losses, relative difference: first three {rel[:3].max():.1e}, median {np.median(rel):.1e}, max {rel.max():.1e}")
    assert rel[:3].max() < 1e-6 and np.median(rel) < 2e-5 and rel.max() < 2e-3


def test_final_state_equals_the_reference_loop(run):
    """Final map and trajectory.  Row counts and `timestep` exactly.  Values: mapping's Adam runs with eps = 1e-15
    (scripts/splatam.py:166), i.e. a row whose gradient is rounding noise still moves by +-lr per step, so two correct float32
    evaluations differ by up to (steps x lr) on such rows (the rotations of still-isotropic Gaussians are the extreme: their true
    gradient is zero) while the typical row agrees to 1e-2 lr.  Bounds per tensor, in units of its mapping learning rate: median
    <= 0.01, 99 % quantile <= 2, maximum <= the number of mapping steps.  Poses (tracking lr 4e-4 / 2e-3, eps 1e-8): 5e-5."""
    case, cfg, rec, params, variables, stats = run
    steps = stats['mapping_iters']
    for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales'):
        want, got = GOLD[f"{case}/final/{k}"], params[k].detach().numpy()
        assert want.shape == got.shape, k
        d, lr = np.abs(want - got), cfg['mapping']['lrs'][k]
        q50, q99 = np.quantile(d, [0.5, 0.99])
        print(f"{case}: {k}: |difference| / lr: median {q50 / lr:.1e}, 99 % {q99 / lr:.2f}, max {d.max() / lr:.2f} ({steps} mapping steps)")
        assert q50 <= 0.01 * lr and q99 <= 2 * lr and d.max() <= steps * lr, k
    for k in ('cam_unnorm_rots', 'cam_trans'):
        d = np.abs(GOLD[f"{case}/final/{k}"] - params[k].detach().numpy())
        print(f"{case}: {k}: max |difference| {d.max():.1e}")
        assert d.max() < 5e-5, k
    assert np.array_equal(GOLD[f"{case}/final/timestep"], variables['timestep'].numpy())


def test_decisions_view(run):
    """The per-frame decision table the GPU tests compare (tests/test_gpu_loop_golden.py) says what the recording says."""
    case, cfg, rec, params, variables, stats = run
    n = len(LT.RecordedRGBDSequence(GOLD, case))
    want = LT.per_frame_decisions(GOLD[f"{case}/events"], GOLD[f"{case}/selected"], GOLD[f"{case}/final/keyframe_time_indices"], n,
                                  cfg['mapping']['pruning_dict'])
    events, _, selected = rec.arrays()
    got = LT.per_frame_decisions(events, selected, stats['keyframe_time_indices'], n, cfg['mapping']['pruning_dict'])
    assert want == got
    assert stats['decisions'] == want                     # ... and so does the table the loop itself keeps (every engine fills it)
    assert [f['rows_end'] for f in want] == stats['num_gaussians']
    if case == "gtposes":
        assert all(f['tracking_iters'] == 0 for f in want) and all(f['keyframe'] for f in want) and not any(f['prunes'] for f in want)
    if case == "variant":
        budgets = [f['tracking_iters'] for f in want]
        assert set(budgets[1:]) == {5, 10}, budgets               # the depth-loss retry doubled some frames' budget and not others'
        assert [f['selected'] is None for f in want] == [False, False, True, False, True, False]   # map_every = 2
        assert [f['keyframe'] for f in want] == [True, False, True, False, True, True]       # 0, every 3rd, num_frames - 2
