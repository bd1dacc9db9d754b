"""splatam_amd.plugin: the reference's OWN loop statements (/root/reference/scripts/splatam.py:690-711 tracking, :846-869 mapping:
get_loss -> loss.backward() -> [prune_gaussians] -> optimizer.step() -> optimizer.zero_grad()) running the fused iteration.  The loop
bodies below are those statements, restated; tests/test_plugin_cpu.py pins their order to the reference's source when it is present.
Compared with the same statements on the drop-in path (PyTorch glue + autograd + torch.optim.Adam around the HIP rasterizer)."""
import copy
import time

import pytest
import torch

from tests.test_gpu_fused import _scene
from tests.util import track_loop_loss_rtol

pytestmark = pytest.mark.gpu


def _tracking_loop(slam_mod, params, variables, frame, time_idx, cfg, iters):
    """scripts/splatam.py:680-744 (optimizer per frame, best-candidate bookkeeping)."""
    optimizer = slam_mod.initialize_optimizer(params, cfg['lrs'], tracking=True)
    candidate_rot = params['cam_unnorm_rots'][..., time_idx].detach().clone()
    candidate_tran = params['cam_trans'][..., time_idx].detach().clone()
    current_min_loss = float(1e20)
    losses = []
    for _ in range(iters):
        loss, variables, _ = slam_mod.get_loss(params, frame, variables, time_idx, cfg['loss_weights'], cfg['use_sil_for_loss'],
                                               cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'], tracking=True)
        loss.backward()
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        with torch.no_grad():
            if loss < current_min_loss:
                current_min_loss = loss
                candidate_rot = params['cam_unnorm_rots'][..., time_idx].detach().clone()
                candidate_tran = params['cam_trans'][..., time_idx].detach().clone()
        losses.append(float(loss))
    with torch.no_grad():
        params['cam_unnorm_rots'][..., time_idx] = candidate_rot
        params['cam_trans'][..., time_idx] = candidate_tran
    return losses


def _mapping_loop(slam_mod, params, variables, frame, time_idx, cfg, iters, prune_dict):
    """scripts/splatam.py:821-869: optimizer per frame; prune_gaussians between backward() and step()."""
    optimizer = slam_mod.initialize_optimizer(params, cfg['lrs'], tracking=False)
    for it in range(iters):
        loss, variables, _ = slam_mod.get_loss(params, frame, variables, time_idx, cfg['loss_weights'], cfg['use_sil_for_loss'],
                                               cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'], mapping=True)
        loss.backward()
        with torch.no_grad():
            if prune_dict is not None:
                params, variables = slam_mod.prune_gaussians(params, variables, optimizer, it, prune_dict)
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
    return params, variables, optimizer


def _variables(params):
    n = params['means3D'].shape[0]
    z = lambda: torch.zeros(n, device="cuda")       # noqa: E731
    return {'max_2D_radius': z(), 'means2D_gradient_accum': z(), 'denom': z(), 'timestep': z(), 'scene_radius': torch.tensor(2.0, device="cuda")}


def test_tracking_statements_run_fused_and_match_the_dropin_path():
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(12000, 256, 192, aniso=False, seed=11)
    cfg = slam.REPLICA_TRACKING
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    ref_losses = _tracking_loop(slam, ref, _variables(ref), frame, 1, cfg, 6)
    with plugin.install(slam):
        assert slam.get_loss is plugin.get_loss
        my_losses = _tracking_loop(slam, mine, _variables(mine), frame, 1, cfg, 6)
        stats = plugin.session_stats()
    assert slam.get_loss is not plugin.get_loss                     # uninstalled
    assert stats["iterations"] == 6 and stats["rebuilds"] == 1 and stats["engines_built"] == 1, stats
    assert stats["skipped_iterations"] == 0, stats
    for it, (a, b) in enumerate(zip(my_losses, ref_losses)):
        assert abs(a - b) <= track_loop_loss_rtol(it) * abs(b), (my_losses, ref_losses)
    assert (mine['cam_unnorm_rots'] - ref['cam_unnorm_rots']).abs().max() <= 1e-4
    assert (mine['cam_trans'] - ref['cam_trans']).abs().max() <= 4e-4
    assert torch.equal(mine['means3D'], ref['means3D'])


def _prune_dict():
    return dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
                final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500)


def _assert_adam_step_matches(name, p_mine, p_ref, p_before, m_mine, m_ref, lr):
    """ONE Adam step with eps = 1e-15 moves an element by lr * sign(g) (m / sqrt(v) = g / |g|): where the sign of the gradient is
    certain the two paths must land on the same value; the first moment (0.1 g) is compared at the north star's gradient tolerance."""
    scale = float(m_ref.abs().max())
    err = (m_mine - m_ref).abs()
    assert float(torch.quantile(err.reshape(-1)[:4_000_000].float(), 0.9999)) <= 1e-3 * scale + 1e-20, (name, float(err.max()), scale)
    certain = m_ref.abs() > 2e-3 * scale
    d = (p_mine - p_ref).abs()[certain]
    moved = (p_ref - p_before).abs()[certain]
    assert float(moved.min()) > 0.5 * lr, (name, float(moved.min()), lr)                # every certain element took its step
    assert float(d.max()) <= 0.02 * lr, (name, float(d.max()), lr)                      # ... to the same place (float32 rounding of p)
    flipped = float(((p_mine - p_ref).abs() > 0.5 * lr).float().mean())
    assert flipped <= 5e-3, (name, flipped)                                             # elements whose gradient is rounding noise


def test_mapping_statements_with_the_references_pruning_run_fused():
    """Mapping iterations in the reference's statements; the pruning schedule removes rows at iteration 0 (the reference's own
    remove_points slices the optimizer state and re-creates every parameter: that iteration takes no Adam step), iteration 1 steps
    the smaller map: parameters and moments after that ONE step element-wise against torch.optim.Adam on the drop-in path."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(8000, 208, 160, aniso=False, seed=7)
    with torch.no_grad():
        params['logit_opacities'][::5] = -6.0                       # a fifth of the map is transparent: pruned at iteration 0
    cfg = slam.REPLICA_MAPPING
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    ref, ref_vars, ref_opt = _mapping_loop(slam, ref, _variables(ref), frame, 1, cfg, 1, _prune_dict())
    before = {k: ref[k].detach().clone() for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales")}
    with plugin.install(slam):
        mine, my_vars, my_opt = _mapping_loop(slam, mine, _variables(mine), frame, 1, cfg, 1, _prune_dict())
    n = ref['means3D'].shape[0]
    assert n < 8000 and mine['means3D'].shape[0] == n
    for k in before:                                                # iteration 0 pruned and took no step: the rows are the reference's
        assert torch.equal(mine[k].detach(), before[k]), k

    def one_more(mod, p, v, opt):
        loss, v, losses = mod.get_loss(p, frame, v, 1, cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                       cfg['ignore_outlier_depth_loss'], mapping=True)
        loss.backward()
        with torch.no_grad():
            p, v = mod.prune_gaussians(p, v, opt, 1, _prune_dict())
            opt.step()
            opt.zero_grad(set_to_none=True)
        return p, v, losses
    ref, ref_vars, ref_losses = one_more(slam, ref, ref_vars, ref_opt)
    with plugin.install(slam):
        # (a fresh session: the optimizer of the first loop is the caller's object and keeps working across it)
        mine, my_vars, my_losses = one_more(slam, mine, my_vars, my_opt)
        stats = plugin.session_stats()
    assert stats["iterations"] == 1 and stats["skipped_iterations"] == 0, stats
    assert isinstance(my_opt, torch.optim.Adam)
    for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales"):
        _assert_adam_step_matches(k, mine[k].detach(), ref[k].detach(), before[k], my_opt.state[mine[k]]['exp_avg'],
                                  ref_opt.state[ref[k]]['exp_avg'], cfg['lrs'][k])
        assert float(my_opt.state[mine[k]]['step']) == 1.0 == float(ref_opt.state[ref[k]]['step'])
    assert torch.equal(my_vars['seen'], ref_vars['seen'])
    assert torch.equal(my_vars['max_2D_radius'], ref_vars['max_2D_radius'])
    # the whole losses dict of the reference (scripts/splatam.py:339-346: weighted terms + their sum)
    assert set(my_losses) == set(ref_losses) == {'depth', 'im', 'loss'}
    for k in ref_losses:
        assert abs(float(my_losses[k]) - float(ref_losses[k])) <= 2e-5 * abs(float(ref_losses[k])) + 1e-7, (k, float(my_losses[k]), float(ref_losses[k]))


@pytest.mark.parametrize("tracking", [True, False])
@pytest.mark.parametrize("use_l1", [True, False])
def test_losses_dict_is_the_references(tracking, use_l1):
    """get_loss returns (loss, variables, weighted_losses) with weighted_losses = {'depth' (only with use_l1), 'im', 'loss'}
    (scripts/splatam.py:275-346; report_loss reads losses['loss'], utils/eval_helpers.py:82): tracking sums, mapping means."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(9000, 224, 160, aniso=False, seed=5)
    cfg = dict(slam.REPLICA_TRACKING if tracking else slam.REPLICA_MAPPING)
    w = {'im': 0.5, 'depth': 1.7}
    kw = dict(tracking=True) if tracking else dict(mapping=True)
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    _, _, want = slam.get_loss(ref, frame, _variables(ref), 1, w, cfg['use_sil_for_loss'], cfg['sil_thres'], use_l1, False, **kw)
    with plugin.install(slam):
        loss, _, got = slam.get_loss(mine, frame, _variables(mine), 1, w, cfg['use_sil_for_loss'], cfg['sil_thres'], use_l1, False, **kw)
        assert set(got) == set(want) == ({'depth', 'im', 'loss'} if use_l1 else {'im', 'loss'})
        for k in want:
            assert abs(float(got[k]) - float(want[k])) <= 3e-5 * abs(float(want[k])) + 1e-7, (k, float(got[k]), float(want[k]))
        assert float(got['loss']) == float(loss) and got['loss'].item() == float(loss)
        assert f"{loss:.3f}" == f"{float(loss):.3f}"
        assert (loss < 1e20) is True and (loss > 1e20) is False and bool(loss < torch.tensor(1e20, device="cuda"))


@pytest.mark.parametrize("map_edits", [False, True])
def test_a_flagged_iteration_moves_nothing_and_is_repeated_or_reported(map_edits):
    """Per-tile lists that do not fit raise a flag on the device; the Adam kernels skip while it is up.  Tracking: the caller's own
    `loss < current_min_loss` read fetches the flag and the iteration is repeated on re-sized lists (same values as an undisturbed run).
    Mapping: the iteration is skipped, the next ones run on re-sized lists, session_stats() reports it."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(12000, 256, 192, aniso=False, seed=11)
    cfg = slam.REPLICA_TRACKING
    clean = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    with plugin.install(slam, map_edits=map_edits):
        clean_losses = _tracking_loop(slam, clean, _variables(clean), frame, 1, cfg, 5)
    with plugin.install(slam, map_edits=map_edits):
        v = _variables(mine)
        _tracking_loop(slam, mine, v, frame, 1, cfg, 2)          # learns bucketed lists
        with torch.no_grad():
            for k in mine:
                mine[k].copy_(params[k])
        eng = next(iter(plugin._session.engines.values()))
        plugin._session.drain()
        assert eng.tile_stride > 0
        eng.tile_stride = 64                                        # buckets far too small for this scene: the next iteration overflows
        eng.max_list_hint = 40
        losses = _tracking_loop(slam, mine, v, frame, 1, cfg, 5)
        stats = plugin.session_stats()
    assert stats["repeats"] >= 1, stats
    for a, b in zip(losses, clean_losses):
        assert abs(a - b) <= 1e-4 * abs(b), (losses, clean_losses)
    assert (mine['cam_trans'] - clean['cam_trans']).abs().max() <= 1e-5
    # mapping: the flagged iteration takes no step
    mcfg = slam.REPLICA_MAPPING
    mp = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    with plugin.install(slam, map_edits=map_edits):
        mv = _variables(mp)
        mp, mv, opt = _mapping_loop(slam, mp, mv, frame, 1, mcfg, 3, None)
        plugin._session.drain()
        eng = next(iter(plugin._session.engines.values()))
        snap = {k: mp[k].detach().clone() for k in mp}
        eng.tile_stride, eng.max_list_hint = 64, 40
        loss, mv, _ = slam.get_loss(mp, frame, mv, 1, mcfg['loss_weights'], mcfg['use_sil_for_loss'], mcfg['sil_thres'], mcfg['use_l1'],
                                    mcfg['ignore_outlier_depth_loss'], mapping=True)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        for k in snap:
            assert torch.equal(mp[k].detach(), snap[k]), k          # nothing moved
        plugin._session.drain()
        assert plugin.session_stats()["skipped_iterations"] >= 1
        if map_edits:
            assert eng.map_step == 3                                # the flagged step's count was taken back
        assert eng.tile_stride != 64                                # lists re-sized
        loss, mv, _ = slam.get_loss(mp, frame, mv, 1, mcfg['loss_weights'], mcfg['use_sil_for_loss'], mcfg['sil_thres'], mcfg['use_l1'],
                                    mcfg['ignore_outlier_depth_loss'], mapping=True)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        assert not torch.equal(mp['means3D'].detach(), snap['means3D'])      # ... and the loop goes on
        # several iterations launched back to back behind a flagged one (their reports are in flight when the first is digested): every
        # one the device gated is taken back from the host-side step counts -- the count equals the steps that MOVED the map
        plugin._session.drain()
        torch.cuda.synchronize()
        steps_before = eng.map_step if map_edits else int(opt.state[opt.param_groups[0]['params'][0]]['step'])
        eng.tile_stride, eng.max_list_hint = 64, 40
        snaps = [mp['means3D'].detach().clone()]
        for _ in range(4):
            loss, mv, _ = slam.get_loss(mp, frame, mv, 1, mcfg['loss_weights'], mcfg['use_sil_for_loss'], mcfg['sil_thres'], mcfg['use_l1'],
                                        mcfg['ignore_outlier_depth_loss'], mapping=True)
            loss.backward()
            opt.step()
            snaps.append(mp['means3D'].detach().clone())
        plugin._session.drain()
        torch.cuda.synchronize()
        moved = sum(0 if torch.equal(a, b) else 1 for a, b in zip(snaps[:-1], snaps[1:]))
        assert moved < 4                                            # (at least the first of them was gated)
        steps_after = eng.map_step if map_edits else int(opt.state[opt.param_groups[0]['params'][0]]['step'])
        assert steps_after - steps_before == moved, (steps_before, steps_after, moved, plugin.session_stats())


def test_plugin_speed_against_the_dropin_statements():
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(100000, 640, 480, aniso=False, seed=3)
    cfg = slam.REPLICA_TRACKING

    def rate(p, n):
        _tracking_loop(slam, p, _variables(p), frame, 1, cfg, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _tracking_loop(slam, p, _variables(p), frame, 1, cfg, n)
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    r_drop = rate(ref, 10)
    with plugin.install(slam):
        r_plug = rate(mine, 40)
    print(f"tracking statements, 100 k Gaussians 640x480: drop-in {r_drop:.0f} it/s, plug-in {r_plug:.0f} it/s")
    assert r_plug > 3.0 * r_drop


def test_learning_rates_the_shipped_configs_leave_at_zero():
    """Pose learning rates in the mapping optimizer (bundle adjustment: get_loss(..., do_ba=True), /root/reference/scripts/splatam.py:226-233)
    and Gaussian learning rates in the tracking optimizer: the reference's statements through the plug-in against the same statements
    on the drop-in path, after ONE step (Adam's first step is lr * sign(g) wherever the gradient's sign is certain)."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(9000, 224, 160, aniso=False, seed=13)

    def run(mod, p, cfg, lrs, tracking):
        v = _variables(p)
        opt = mod.initialize_optimizer(p, lrs, tracking=tracking)
        kw = dict(tracking=True) if tracking else dict(mapping=True, do_ba=True)
        loss, v, _ = mod.get_loss(p, frame, v, 1, cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                  cfg['ignore_outlier_depth_loss'], **kw)
        loss.backward()
        with torch.no_grad():
            opt.step()
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        return float(loss)

    def check(mine, ref, before, lrs, keys):
        for k in keys:
            lr = lrs[k]
            moved = (ref[k].detach() - before[k]).abs()
            d = (mine[k].detach() - ref[k].detach()).abs()
            certain = moved > 0.5 * lr                          # the reference stepped this element by ~lr: its gradient is not noise
            assert float(certain.float().mean()) > 0.02, (k, float(certain.float().mean()))
            flipped = float((d[certain] > 0.5 * lr).float().mean())
            assert flipped <= 5e-3, (k, flipped)
            assert float(d[certain & (d <= 0.5 * lr)].max()) <= 0.02 * lr, (k, float(d[certain & (d <= 0.5 * lr)].max()), lr)
    # (1) mapping with bundle adjustment
    lrs = dict(slam.REPLICA_MAPPING['lrs'], cam_unnorm_rots=0.0004, cam_trans=0.002)
    before = {k: v.detach().clone() for k, v in params.items()}
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    l_ref = run(slam, ref, slam.REPLICA_MAPPING, lrs, tracking=False)
    with plugin.install(slam):
        l_mine = run(slam, mine, slam.REPLICA_MAPPING, lrs, tracking=False)
    assert abs(l_mine - l_ref) <= 1e-4 * abs(l_ref)
    check(mine, ref, before, lrs, ("means3D", "rgb_colors", "logit_opacities", "log_scales"))
    for k in ("cam_unnorm_rots", "cam_trans"):                      # frame 1's pose stepped by lr * sign(g); every other column untouched
        d = (mine[k].detach() - ref[k].detach()).abs()
        assert float(d.max()) <= 0.02 * lrs[k], (k, float(d.max()))
        assert float((ref[k].detach() - before[k])[0, :, 1].abs().min()) > 0.5 * lrs[k]
        assert torch.equal(mine[k].detach()[0, :, 0], before[k][0, :, 0]) and torch.equal(mine[k].detach()[0, :, 2], before[k][0, :, 2])
    # (2) tracking with Gaussian learning rates: rgb / opacity / scale move (centres and rotations are detached while tracking)
    lrs = dict(slam.REPLICA_TRACKING['lrs'], rgb_colors=0.0025, logit_opacities=0.05, log_scales=0.001, means3D=0.0001)
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    l_ref = run(slam, ref, slam.REPLICA_TRACKING, lrs, tracking=True)
    with plugin.install(slam):
        l_mine = run(slam, mine, slam.REPLICA_TRACKING, lrs, tracking=True)
    assert abs(l_mine - l_ref) <= 1e-4 * abs(l_ref)
    check(mine, ref, before, lrs, ("rgb_colors", "logit_opacities", "log_scales"))
    assert torch.equal(mine['means3D'].detach(), before['means3D']) and torch.equal(ref['means3D'].detach(), before['means3D'])
    for k in ("cam_unnorm_rots", "cam_trans"):
        assert float((mine[k].detach() - ref[k].detach()).abs().max()) <= 0.02 * lrs[k], k


def test_means2d_gradient_for_the_references_densification():
    """use_gaussian_splatting_densification (/root/reference/scripts/splatam.py:863-866): the reference's own statement
    `accumulate_mean2d_gradient(variables)` (/root/reference/utils/slam_external.py:100-104, mirrored in slam.accumulate_mean2d_gradient)
    reads variables['means2D'].grad -- the COLOUR pass' screen-space gradient -- after backward().  Through the plug-in that gradient
    is formed on first access (one RGB-only backward composite over the iteration's lists): equal to the drop-in path's autograd result,
    and the statement accumulates the same statistic.  Tracking iterations keep no planes: reading it there raises."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(8000, 208, 160, aniso=False, seed=11)
    cfg = slam.REPLICA_MAPPING
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    args = (cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'])
    ref_vars = _variables(ref)
    loss, ref_vars, _ = slam.get_loss(ref, frame, ref_vars, 1, *args, mapping=True)
    loss.backward()
    want = ref_vars['means2D'].grad
    slam.accumulate_mean2d_gradient(ref_vars)
    with plugin.install(slam):
        my_vars = _variables(mine)
        loss, my_vars, _ = slam.get_loss(mine, frame, my_vars, 1, *args, mapping=True)
        loss.backward()
        got = my_vars['means2D'].grad
        slam.accumulate_mean2d_gradient(my_vars)                    # the reference's statement, unchanged
        scale = float(want[:, :2].abs().max())
        err = float((got[:, :2] - want[:, :2]).abs().max())
        print(f"means2D.grad: max |difference| {err:.3e} at scale {scale:.3e}")
        assert got.shape == want.shape and err <= 1e-3 * scale and float(got[:, 2].abs().max()) == 0.0
        assert torch.equal(my_vars['denom'], ref_vars['denom'])
        acc_err = float((my_vars['means2D_gradient_accum'] - ref_vars['means2D_gradient_accum']).abs().max())
        assert acc_err <= 1e-3 * float(ref_vars['means2D_gradient_accum'].max())
        # the gradient belongs to the iteration: after the next get_loss the old object refuses
        stale = my_vars['means2D']
        stale._grad = None
        loss, my_vars, _ = slam.get_loss(mine, frame, my_vars, 1, *args, mapping=True)
        with pytest.raises(RuntimeError, match="before the next get_loss"):
            stale.grad
        tcfg = slam.REPLICA_TRACKING
        slam.initialize_optimizer(mine, tcfg['lrs'], tracking=True)
        loss, my_vars, _ = slam.get_loss(mine, frame, my_vars, 1, tcfg['loss_weights'], tcfg['use_sil_for_loss'], tcfg['sil_thres'],
                                         tcfg['use_l1'], tcfg['ignore_outlier_depth_loss'], tracking=True)
        with pytest.raises(RuntimeError, match="TRACKING"):
            my_vars['means2D'].grad


def test_map_edits_mode_prunes_in_place_and_steps_like_torch():
    """plugin.install(map_edits=True): prune_gaussians is an adapter too and the engine owns the map.  The same two iterations as above
    in ONE session: iteration 0 prunes (no Adam step, as the reference's re-created parameters get none), iteration 1 steps the smaller
    map -- rows after the prune bit-equal to the reference's remove_points, parameters and moments after the step element-wise against
    torch.optim.Adam on the drop-in path; the caller's dicts are the same objects throughout, their entries views of the engine's rows."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(8000, 208, 160, aniso=False, seed=7)
    with torch.no_grad():
        params['logit_opacities'][::5] = -6.0
    cfg = slam.REPLICA_MAPPING
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    ref1, ref_vars, ref_opt = _mapping_loop(slam, {k: torch.nn.Parameter(v.detach().clone()) for k, v in ref.items()}, _variables(ref), frame, 1, cfg, 1,
                                            _prune_dict())
    before = {k: ref1[k].detach().clone() for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales")}
    ref2, ref_vars2, ref_opt2 = _mapping_loop(slam, ref, _variables(ref), frame, 1, cfg, 2, _prune_dict())
    my_vars = _variables(mine)
    with plugin.install(slam, map_edits=True):
        assert slam.prune_gaussians is plugin.prune_gaussians and slam.add_new_gaussians is plugin.add_new_gaussians
        out, out_vars, my_opt = _mapping_loop(slam, mine, my_vars, frame, 1, cfg, 1, _prune_dict())
        assert out is mine and out_vars is my_vars
        n = ref1['means3D'].shape[0]
        assert n < 8000 and mine['means3D'].shape[0] == n and my_vars['timestep'].shape[0] == n
        for k in before:
            assert torch.equal(mine[k].detach(), before[k]), k
        eng = next(iter(plugin._session.engines.values()))
        assert eng.managed and eng.P == n and mine['means3D'].data_ptr() == eng.store['means3D'].data_ptr()
        # iteration 1 (not on the pruning schedule): the first step of this optimizer
        loss, my_vars, _ = slam.get_loss(mine, frame, my_vars, 1, cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                         cfg['ignore_outlier_depth_loss'], mapping=True)
        loss.backward()
        with torch.no_grad():
            slam.prune_gaussians(mine, my_vars, my_opt, 1, _prune_dict())
            my_opt.step()
            my_opt.zero_grad(set_to_none=True)
        stats = plugin.session_stats()
        assert eng.map_step == 1
        for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales"):
            _assert_adam_step_matches(k, mine[k].detach(), ref2[k].detach(), before[k], eng.exp_avg[k], ref_opt2.state[ref2[k]]['exp_avg'], cfg['lrs'][k])
        assert torch.equal(my_vars['seen'], ref_vars2['seen'])
        assert torch.equal(my_vars['max_2D_radius'], ref_vars2['max_2D_radius'])
    assert stats["iterations"] == 2 and stats["engines_built"] == 1 and stats["skipped_iterations"] == 0, stats
    assert slam.prune_gaussians is not plugin.prune_gaussians                     # uninstalled


def test_map_edits_mode_adds_gaussians_in_place():
    """add_new_gaussians through the adapter (the engine's in-place append) against the reference-shaped one on the drop-in path, from
    the same map and frame: the same pixels selected (a count may differ by pixels whose silhouette sits on the threshold), the new
    rows appended after the old ones with the frame's time index, the old rows untouched, the dicts re-pointed."""
    from splatam_amd import plugin, slam
    W, H = 208, 160
    params, _, frame, cam = _scene(6000, W, H, aniso=False, seed=5)
    frame['intrinsics'] = torch.tensor([[0.5 * W, 0, W / 2 - 0.5], [0, 0.5 * W, H / 2 - 0.5], [0, 0, 1]], device="cuda")
    with torch.no_grad():
        keep = params['means3D'][:, 0] < params['means3D'][:, 0].median()          # half of the scene is missing from the map
    half = {k: (v.detach()[keep].clone() if v.shape[0] == keep.shape[0] and k not in ('cam_unnorm_rots', 'cam_trans') else v.detach().clone())
            for k, v in params.items()}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in half.items()}
    mine = {k: torch.nn.Parameter(v.clone()) for k, v in half.items()}
    ref_vars, my_vars = _variables(ref), _variables(mine)
    n0 = ref['means3D'].shape[0]
    ref, ref_vars = slam.add_new_gaussians(ref, ref_vars, frame, 0.5, 1, "projective", "isotropic")
    with plugin.install(slam, map_edits=True):
        out, out_vars = slam.add_new_gaussians(mine, my_vars, frame, 0.5, 1, "projective", "isotropic")
        assert out is mine and out_vars is my_vars
        n_ref, n = ref['means3D'].shape[0], mine['means3D'].shape[0]
        assert n_ref > n0 + 1000 and abs(n - n_ref) <= max(3, int(2e-3 * n_ref)), (n0, n_ref, n)
        for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales"):
            assert mine[k].shape[0] == n and torch.equal(mine[k].detach()[:n0], half[k]), k
        assert my_vars['timestep'].shape[0] == n and bool((my_vars['timestep'][n0:] == 1).all()) and bool((my_vars['timestep'][:n0] == 0).all())
        if n == n_ref:
            assert float((mine['means3D'].detach() - ref['means3D'].detach()).abs().max()) < 1e-4
            assert float((mine['log_scales'].detach() - ref['log_scales'].detach()).abs().max()) < 1e-4
        # and the statements go on with the grown map: one mapping iteration steps it
        cfg = slam.REPLICA_MAPPING
        _mapping_loop(slam, mine, my_vars, frame, 1, cfg, 2, None)
        stats = plugin.session_stats()
    assert stats["engines_built"] == 1 and stats["skipped_iterations"] == 0, stats


def test_map_edits_mode_refuses_what_it_does_not_own():
    """plugin.install(map_edits=True): a Gaussian tensor replaced behind the engine's back is an error (not a silent re-bind), and
    densify is refused by name."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(3000, 160, 112, aniso=False, seed=3)
    cfg = slam.REPLICA_MAPPING
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    variables = _variables(mine)
    args = (cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'], cfg['ignore_outlier_depth_loss'])
    with plugin.install(slam, map_edits=True):
        opt = slam.initialize_optimizer(mine, cfg['lrs'], tracking=False)
        loss, variables, _ = slam.get_loss(mine, frame, variables, 1, *args, mapping=True)
        loss.backward()
        opt.step()
        with pytest.raises(NotImplementedError, match="densify"):
            slam.densify(mine, variables, opt, 0, {})
        mine['rgb_colors'] = torch.nn.Parameter(mine['rgb_colors'].detach().clone())
        with pytest.raises(RuntimeError, match="replaced outside"):
            slam.get_loss(mine, frame, variables, 1, *args, mapping=True)
