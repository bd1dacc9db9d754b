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
    assert stats["iterations"] == 6 and stats["rebuilds"] == 1, stats
    for it, (a, b) in enumerate(zip(my_losses, ref_losses)):
        assert abs(a - b) <= track_loop_loss_rtol(it) * abs(b), (my_losses, ref_losses)
    assert (mine['cam_unnorm_rots'] - ref['cam_unnorm_rots']).abs().max() <= 1e-4
    assert (mine['cam_trans'] - ref['cam_trans']).abs().max() <= 4e-4
    assert torch.equal(mine['means3D'], ref['means3D'])


def test_mapping_statements_with_the_references_pruning_run_fused():
    """Three mapping iterations; the pruning schedule removes rows at iteration 0 (the reference's own remove_points slices the
    optimizer state and re-creates every parameter: that iteration takes no Adam step), iterations 1 and 2 step the smaller map."""
    from splatam_amd import plugin, slam
    params, _, frame, cam = _scene(8000, 208, 160, aniso=False, seed=7)
    with torch.no_grad():
        params['logit_opacities'][::5] = -6.0                       # a fifth of the map is transparent: pruned at iteration 0
    cfg = slam.REPLICA_MAPPING
    pd = dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
              final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500)
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    mine = {k: torch.nn.Parameter(v.detach().clone()) for k, v in params.items()}
    ref, ref_vars, ref_opt = _mapping_loop(slam, ref, _variables(ref), frame, 1, cfg, 3, copy.deepcopy(pd))
    with plugin.install(slam):
        mine, my_vars, my_opt = _mapping_loop(slam, mine, _variables(mine), frame, 1, cfg, 3, copy.deepcopy(pd))
        stats = plugin.session_stats()
    n = ref['means3D'].shape[0]
    assert n < 8000 and mine['means3D'].shape[0] == n
    assert stats["iterations"] == 3 and stats["rebuilds"] == 2, stats          # the pruned map is a new set of tensors
    assert isinstance(my_opt, torch.optim.Adam)
    for k in ("means3D", "rgb_colors", "logit_opacities", "log_scales"):
        lr = cfg['lrs'][k]
        diff = (mine[k].detach() - ref[k].detach()).abs()
        # two Adam steps with eps = 1e-15 move an element by ~2 lr; elements whose gradient is rounding noise may step the other way
        assert float((diff > 0.1 * lr).float().mean()) < 2e-2, (k, float(diff.max()), lr)
        m_ref = ref_opt.state[ref[k]]['exp_avg']
        m_my = my_opt.state[mine[k]]['exp_avg']
        assert float((m_my - m_ref).abs().max()) <= 2e-3 * float(m_ref.abs().max()) + 1e-12, k
    assert torch.equal(my_vars['seen'], ref_vars['seen'])
    assert torch.equal(my_vars['max_2D_radius'], ref_vars['max_2D_radius'])


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
