"""Map growth / maintenance on the device (FusedEngine.add_new_gaussians / add_valid_depth_points / prune_gaussians /
remove_points over the C ABI's splat_map_* entry points) against

 * the golden vectors produced by the REFERENCE's own code (tests/golden/make_golden_mapedit.py): same inputs,
   the stored depth/silhouette image in place of the render;
 * the torch formulation (splatam_amd/slam.py, itself pinned to the same vectors) on a real render, full size.

Row ORDER and counts are exact; float values to 2e-6 (the reference back-projects with torch.inverse + a GEMM, the
kernel with R^T (p - t) in registers)."""
import os

import numpy as np
import pytest
import torch

from tests.test_mapedit_mirror import GOLD, PARAM_KEYS, VAR_KEYS, check_add_outputs, load_add_case, load_prune_case

pytestmark = pytest.mark.gpu


def _dummy_cam(W, H, f=50.0):
    from splatam_amd import slam
    k = [[f, 0, W / 2 - 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]]
    return slam.setup_camera(W, H, k, np.eye(4, dtype=np.float32), device="cuda")


@pytest.mark.parametrize("name", ["add_iso", "add_aniso", "add_nan", "add_nothing"])
@pytest.mark.parametrize("room", ["exact", "grow"])
def test_add_new_gaussians_matches_reference_vectors(name, room):
    from splatam_amd.fused import FusedEngine
    params, variables, curr, depth_sil, time_idx, sil_thres, dist = load_add_case(name, device="cuda")
    params = {k: torch.nn.Parameter(v.detach().contiguous()) for k, v in params.items()}
    W, H = int(GOLD[f"{name}/in/meta"][0]), int(GOLD[f"{name}/in/meta"][1])
    n0 = params['means3D'].shape[0]
    n1 = GOLD[f"{name}/out/param/means3D"].shape[0]
    cap = n1 + 7 if room == "exact" else n0 + 3          # "grow": the first attempt cannot fit, the engine re-allocates
    eng = FusedEngine(params, _dummy_cam(W, H), gaussian_capacity=cap, variables=variables)
    added = eng.add_new_gaussians(curr, sil_thres, time_idx, "projective", dist, depth_sil=depth_sil)
    torch.cuda.synchronize()
    assert added == n1 - n0 and eng.P == n1
    if n1 > n0:
        assert eng.Pcap >= n1
    med = np.frombuffer(np.int32(eng.buf['counts'][4].item()).tobytes(), dtype=np.float32)[0]
    ref_med = np.float32(GOLD[f"{name}/median"])
    assert (np.isnan(med) and np.isnan(ref_med)) or med == ref_med, (med, ref_med)      # torch.median, bit for bit
    check_add_outputs(name, params, variables)
    for k in eng.exp_avg:                                   # moments of appended rows start from zero
        assert float(eng.exp_avg[k][n0:].abs().sum()) == 0.0 and float(eng.exp_avg_sq[k][n0:].abs().sum()) == 0.0


@pytest.mark.parametrize("name", ["init_iso", "init_aniso"])
def test_first_frame_points_match_reference_vectors(name):
    from splatam_amd.fused import FusedEngine
    im, depth = torch.tensor(GOLD[f"{name}/in/im"]).cuda(), torch.tensor(GOLD[f"{name}/in/depth"]).cuda()
    k, w2c = torch.tensor(GOLD[f"{name}/in/intrinsics"]), torch.tensor(GOLD[f"{name}/in/w2c"])
    H, W = im.shape[1], im.shape[2]
    cols = GOLD[f"{name}/out/param/log_scales"].shape[1]
    z = lambda *s: torch.nn.Parameter(torch.zeros(*s, device="cuda"))      # noqa: E731
    params = {'means3D': z(0, 3), 'rgb_colors': z(0, 3), 'unnorm_rotations': z(0, 4), 'logit_opacities': z(0, 1),
              'log_scales': z(0, cols), 'cam_unnorm_rots': torch.nn.Parameter(torch.tensor(GOLD[f"{name}/out/param/cam_unnorm_rots"]).cuda()),
              'cam_trans': z(1, 3, 5)}
    variables = {kk: torch.zeros(0, device="cuda") for kk in VAR_KEYS}
    eng = FusedEngine(params, _dummy_cam(W, H), gaussian_capacity=H * W, variables=variables)
    n = eng.add_valid_depth_points(im, depth, k, w2c)
    torch.cuda.synchronize()
    assert n == GOLD[f"{name}/out/param/means3D"].shape[0] == eng.P
    for kk in PARAM_KEYS:
        np.testing.assert_allclose(params[kk].detach().cpu().numpy(), GOLD[f"{name}/out/param/{kk}"], rtol=2e-6, atol=2e-6, err_msg=kk)
    for kk in VAR_KEYS:
        np.testing.assert_array_equal(variables[kk].cpu().numpy(), GOLD[f"{name}/out/var/{kk}"])


@pytest.mark.parametrize("name", ["prune_iso", "prune_aniso"])
def test_prune_gaussians_matches_reference_vectors(name):
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    params, variables, moments = load_prune_case(name, device="cuda")
    params = {k: torch.nn.Parameter(v.detach().contiguous()) for k, v in params.items()}
    scene_radius = variables.pop('scene_radius')
    eng = FusedEngine(params, _dummy_cam(64, 48), gaussian_capacity=params['means3D'].shape[0] + 100, variables=variables)
    for k, (m, v) in moments.items():
        eng.exp_avg[k].copy_(m)
        eng.exp_avg_sq[k].copy_(v)
    n0 = eng.P
    removed = eng.prune_gaussians(0, slam.REPLICA_PRUNE, scene_radius)
    torch.cuda.synchronize()
    n1 = GOLD[f"{name}/out/param/means3D"].shape[0]
    assert removed == n0 - n1 and eng.P == n1
    for k in PARAM_KEYS:                                    # rows move, values do not change: bit-exact
        np.testing.assert_array_equal(params[k].detach().cpu().numpy(), GOLD[f"{name}/out/param/{k}"], err_msg=k)
    for k in slam.GAUSSIAN_KEYS:
        np.testing.assert_array_equal(eng.exp_avg[k].cpu().numpy(), GOLD[f"{name}/out/exp_avg/{k}"])
        np.testing.assert_array_equal(eng.exp_avg_sq[k].cpu().numpy(), GOLD[f"{name}/out/exp_avg_sq/{k}"])
    for k in VAR_KEYS:
        np.testing.assert_array_equal(variables[k].cpu().numpy(), GOLD[f"{name}/out/var/{k}"], err_msg=k)
    assert eng.prune_gaussians(7, slam.REPLICA_PRUNE, scene_radius) == 0          # off the schedule
    # caller-supplied flags (remove_points as densify uses it)
    flags = torch.zeros(eng.P, dtype=torch.bool, device="cuda")
    flags[::3] = True
    before = params['means3D'].detach().clone()
    assert eng.remove_points(flags) == int(flags.sum())
    assert torch.equal(params['means3D'].detach(), before[~flags])


def test_render_only_pass_matches_full_iteration_and_densification_full_size():
    """splat_iter_render == the forward half of splat_iter_loss_backward; add_new_gaussians on a REAL render at workload
    size (1200x680, 300k) == the torch formulation on the same render; the grown map then optimises (bucketed lists are
    re-learnt) and prunes."""
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    W, H, n = 1200, 680, 300000
    f, cx, cy = 600.0, 599.5, 339.5
    params, variables = slam.synthetic_params(n, W, H, f, f, cx, cy, num_frames=3, seed=0, device="cuda")
    k = torch.tensor([[f, 0, cx], [0, f, cy], [0, 0, 1]])
    w2c = torch.eye(4, device="cuda")
    cam = slam.setup_camera(W, H, k.numpy(), np.eye(4, dtype=np.float32), device="cuda")
    im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.6, trans_m=0.02)
    depth = depth.clone()
    depth[:, 100:140, 200:300] *= 0.5                      # a new foreground object in front of the map
    frame = {'cam': cam, 'im': im.contiguous(), 'depth': depth.contiguous(), 'id': 1, 'w2c': w2c, 'intrinsics': k}
    with torch.no_grad():                                   # thin the map in one corner: low silhouette there
        params['logit_opacities'][params['means3D'][:, 0] / params['means3D'][:, 2] > 0.8] = -6.0
        params['cam_unnorm_rots'][0, :, 1] = torch.tensor([0.9999, 0.0, 0.004, 0.001], device="cuda") * 1.05
        params['cam_trans'][0, :, 1] = torch.tensor([0.015, -0.008, 0.01], device="cuda")
    ref_params = {kk: torch.nn.Parameter(v.detach().clone()) for kk, v in params.items()}
    ref_vars = {kk: v.clone() for kk, v in variables.items()}
    eng = FusedEngine(params, cam, gaussian_capacity=n + 200000, variables=variables)
    cfg = slam.REPLICA_MAPPING
    eng.loss_backward(frame, 1, cfg, tracking=False)
    full = eng.buf['out6'].clone()
    out = eng.render(frame, 1)
    torch.cuda.synchronize()
    assert torch.equal(eng.buf['out6'], full)
    assert not eng.check_overflow()
    eng.render(frame, 1)                                    # bucketed lists now
    assert eng.tile_stride > 0 and torch.allclose(eng.buf['out6'], full, atol=1e-6)
    assert int(eng.buf['tile_count'].abs().sum()) == 0      # counters folded and reset
    depth_sil = torch.stack([out[1][0], out[2]]).clone()
    added = eng.add_new_gaussians(frame, 0.5, 1, "projective", "isotropic")
    ref_params, ref_vars = slam._add_from_render(ref_params, ref_vars, frame, depth_sil, 0.5, 1, "projective", "isotropic")
    n1 = ref_params['means3D'].shape[0]
    assert added == n1 - n and added > 3000
    for kk in slam.GAUSSIAN_KEYS:
        np.testing.assert_allclose(params[kk].detach().cpu().numpy(), ref_params[kk].detach().cpu().numpy(), rtol=3e-6, atol=3e-6,
                                   err_msg=kk)
    for kk in VAR_KEYS:
        assert torch.equal(variables[kk], ref_vars[kk]), kk
    # the grown map optimises
    eng.relearn_lists(frame, 1)
    assert eng.tile_stride > 0
    eng.reset_map_optimizer()
    l0 = None
    for it in range(6):
        eng.mapping_iteration(frame, 1, cfg)
        if it == 0:
            l0 = eng.loss()
    assert not eng.check_overflow()
    assert np.isfinite(eng.loss()) and eng.loss() < l0
    with torch.no_grad():
        rem = (torch.sigmoid(params['logit_opacities']).squeeze(-1) < 0.005) | \
              (torch.exp(params['log_scales']).max(dim=1).values > 0.1 * 3.0)
        kept = {kk: params[kk].detach()[~rem].clone() for kk in slam.GAUSSIAN_KEYS}
        kept_m = eng.exp_avg['means3D'][~rem].clone()
    removed = eng.prune_gaussians(0, slam.REPLICA_PRUNE, 3.0)
    assert removed == int(rem.sum()) and removed > 100 and eng.P == n1 - removed
    for kk in slam.GAUSSIAN_KEYS:
        assert torch.equal(params[kk].detach(), kept[kk]), kk
    assert torch.equal(eng.exp_avg['means3D'], kept_m)
    eng.relearn_lists(frame, 1)
    eng.mapping_iteration(frame, 1, cfg)
    assert not eng.check_overflow() and np.isfinite(eng.loss())


# ---------------------------------------------------------------------------------------------------------------------
# gradient-based densification on the device (FusedEngine.accumulate_mean2d_gradient / densify)
# ---------------------------------------------------------------------------------------------------------------------

def _densify_scene(n=6000, W=160, H=112):
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    f, cx, cy = 150.0, W / 2 - 0.5, H / 2 - 0.5
    params, variables = slam.synthetic_params(n, W, H, f, f, cx, cy, num_frames=3, seed=4, device="cuda")
    k = [[f, 0, cx], [0, f, cy], [0, 0, 1]]
    w2c = torch.eye(4, device="cuda")
    cam = slam.setup_camera(W, H, k, np.eye(4, dtype=np.float32), device="cuda")
    im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.4, trans_m=0.01)
    g = torch.Generator().manual_seed(9)
    im = (im + 0.05 * torch.randn(im.shape, generator=g).cuda()).clamp(0, 1).contiguous()
    frame = {'cam': cam, 'im': im, 'depth': depth.contiguous(), 'id': 1, 'w2c': w2c}
    mirror = {k_: torch.nn.Parameter(v.detach().clone()) for k_, v in params.items()}
    eng = FusedEngine(params, cam, gaussian_capacity=4 * n, variables=variables)
    return eng, params, variables, mirror, frame


@pytest.mark.parametrize("lists", ["exact", "buckets", "groups"])
def test_colour_pass_means2d_gradient_and_its_accumulation(lists):
    """The colour pass' own dL/dmeans2D (what the reference reads from variables['means2D'].grad,
    /root/reference/utils/slam_external.py:100-104) from the fused iteration's extra RGB-only backward composite, against
    autograd through the drop-in rasterizer; and the accumulation into means2D_gradient_accum / denom for the seen Gaussians."""
    from splatam_amd import slam
    eng, params, variables, mirror, frame = _densify_scene()
    cfg = slam.REPLICA_MAPPING
    mv = {'max_2D_radius': torch.zeros(eng.P, device="cuda"), 'means2D_gradient_accum': torch.zeros(eng.P, device="cuda"),
          'denom': torch.zeros(eng.P, device="cuda")}
    eng.group_bins = lists == "groups"
    if lists != "exact":
        # bucketed lists (learnt from one iteration's statistics): their tile counters are consumed and reset by the iteration's
        # last kernel, the extra backward composite reads the counts it left behind
        eng.loss_backward(frame, 1, cfg, tracking=False)
        assert not eng.check_overflow() and eng.tile_stride > 0
    for it in range(2):
        loss, mv, _ = slam.get_loss(mirror, frame, mv, 1, cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                    cfg['ignore_outlier_depth_loss'], mapping=True)
        loss.backward()
        mv = slam.accumulate_mean2d_gradient(mv)
        eng.loss_backward(frame, 1, cfg, tracking=False)
        g = eng.accumulate_mean2d_gradient(want_grad=True)
        torch.cuda.synchronize()
        ref = mv['means2D'].grad[:, :2]
        scale = float(ref.abs().max())
        err = (g - ref).abs()
        assert float(torch.quantile(err.reshape(-1), 0.9995)) <= 1e-3 * scale and float(err.max()) <= 0.05 * scale, (float(err.max()), scale)
        for p in mirror.values():
            p.grad = None
    acc, ref_acc = variables['means2D_gradient_accum'], mv['means2D_gradient_accum']
    assert torch.equal(variables['denom'], mv['denom']) and float(mv['denom'].max()) == 2.0
    assert float((acc - ref_acc).abs().max()) <= 2e-3 * float(ref_acc.max())
    assert float(eng.buf['accum'].abs().max()) == 0.0           # the workspace invariant survives the extra pass
    g0 = eng.grads['means3D'].clone()
    eng.loss_backward(frame, 1, cfg, tracking=False)
    torch.cuda.synchronize()
    assert torch.allclose(eng.grads['means3D'], g0, rtol=1e-4, atol=1e-7 * float(g0.abs().max()))


def test_densify_on_the_device_matches_the_torch_formulation():
    """FusedEngine.densify against splatam_amd.slam.densify (pinned to the reference's own densify on the CPU by
    tests/test_mapedit_mirror.py) on the GPU under the same torch seed: clone + split (torch.normal draws in the reference's
    order) + removal of the split originals + opacity / size pruning; parameters, Adam moments and per-Gaussian variables of
    the surviving rows, in the same order."""
    from splatam_amd import slam
    eng, params, variables, mirror, frame = _densify_scene()
    cfg = slam.REPLICA_MAPPING
    n = eng.P
    gen = torch.Generator(device="cuda").manual_seed(3)
    # an accumulated gradient history and non-zero Adam moments on both sides
    hist = torch.rand(n, generator=gen, device="cuda") * 2e-3
    den = torch.randint(1, 5, (n,), generator=gen, device="cuda").float()
    den[::7] = 0.0
    hist[::7] = 0.0                                             # 0 / 0 -> NaN -> 0 rows
    opt = slam.initialize_optimizer(mirror, cfg['lrs'], tracking=False)
    for p in mirror.values():
        p.grad = torch.zeros_like(p)
    opt.step()                                                  # creates the state (zero gradient: no parameter moves)
    for key in slam.GAUSSIAN_KEYS:
        m = torch.randn(mirror[key].shape, generator=gen, device="cuda")
        v = torch.rand(mirror[key].shape, generator=gen, device="cuda")
        opt.state[mirror[key]]['exp_avg'].copy_(m)
        opt.state[mirror[key]]['exp_avg_sq'].copy_(v)
        eng.exp_avg[key].copy_(m)
        eng.exp_avg_sq[key].copy_(v)
    # this iteration's colour-pass gradient: the mirror accumulates autograd's, the engine its own; start the mirror so that
    # after ITS accumulation both hold the same sums
    mv = {'max_2D_radius': torch.zeros(n, device="cuda"), 'means2D_gradient_accum': torch.zeros(n, device="cuda"), 'denom': torch.zeros(n, device="cuda")}
    loss, mv, _ = slam.get_loss(mirror, frame, mv, 1, cfg['loss_weights'], cfg['use_sil_for_loss'], cfg['sil_thres'], cfg['use_l1'],
                                cfg['ignore_outlier_depth_loss'], mapping=True)
    loss.backward()
    eng.loss_backward(frame, 1, cfg, tracking=False)
    variables['means2D_gradient_accum'].copy_(hist)
    variables['denom'].copy_(den)
    eng.accumulate_mean2d_gradient()
    seen = mv['seen']
    own = torch.zeros(n, device="cuda")
    own[seen] = torch.norm(mv['means2D'].grad[seen, :2], dim=-1)
    mv['means2D_gradient_accum'] = variables['means2D_gradient_accum'].clone() - own
    mv['denom'] = variables['denom'].clone() - seen.float()
    scales = torch.exp(mirror['log_scales'].detach()).max(dim=1).values
    scene_radius = float(torch.quantile(scales, 0.6)) / 0.01     # ~60 % "small" (clone), ~40 % "large" (split)
    mv['scene_radius'] = torch.tensor(scene_radius, device="cuda")
    # threshold in a gap of the mean-gradient distribution (the two accumulations agree to rounding, not bit for bit)
    grads = variables['means2D_gradient_accum'] / variables['denom']
    grads[grads.isnan()] = 0.0
    sv = torch.sort(grads).values
    lo, hi = int(0.55 * n), int(0.75 * n)
    kgap = lo + int(torch.argmax(sv[lo + 1:hi] - sv[lo:hi - 1]))
    thr = float(0.5 * (sv[kgap] + sv[kgap + 1]))
    dd = dict(start_after=0, remove_big_after=0, stop_after=100, densify_every=1, grad_thresh=thr, num_to_split_into=2,
              removal_opacity_threshold=0.3, final_removal_opacity_threshold=0.3, reset_opacities=False, reset_opacities_every=3000)
    torch.manual_seed(17)
    mirror, mv = slam.densify(mirror, mv, opt, 1, dd)
    variables['means2D_gradient_accum'].copy_(hist)             # FusedEngine.densify accumulates this iteration's gradient itself
    variables['denom'].copy_(den)
    torch.manual_seed(17)
    assert eng.densify(1, dd, scene_radius)
    torch.cuda.synchronize()
    n1 = mirror['means3D'].shape[0]
    assert eng.P == n1 and n1 != n, (eng.P, n1, n)
    for key in slam.GAUSSIAN_KEYS:
        ref = mirror[key].detach()
        assert float((params[key].detach() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), key
        st = opt.state[mirror[key]]
        assert torch.equal(eng.exp_avg[key], st['exp_avg']) and torch.equal(eng.exp_avg_sq[key], st['exp_avg_sq']), key
    for key in ('means2D_gradient_accum', 'denom', 'max_2D_radius'):
        assert torch.equal(variables[key], mv[key]) and float(variables[key].abs().max()) == 0.0, key
    # and the engine keeps running on the edited map
    eng.relearn_lists(frame, 1)
    eng.loss_backward(frame, 1, cfg, tracking=False)
    eng.adam_map(cfg['lrs'])
    torch.cuda.synchronize()
    assert not eng.check_overflow() and np.isfinite(eng.loss())


def test_accumulate_before_a_prune_then_densify_without_accumulating():
    """with pruning AND densification in one iteration the screen-space gradient must be accumulated from the
    iteration's own workspace BEFORE rows are removed (pipeline._map_frame: accumulate_mean2d_gradient(), prune_gaussians(),
    densify(accumulate=False)); accumulating after the compaction would index the old lists with the new rows."""
    from splatam_amd import slam
    eng, params, variables, mirror, frame = _densify_scene()
    cfg = slam.REPLICA_MAPPING
    n = eng.P
    eng.loss_backward(frame, 1, cfg, tracking=False)
    eng.accumulate_mean2d_gradient()
    acc_full, den_full = variables['means2D_gradient_accum'][:n].clone(), variables['denom'][:n].clone()
    assert float(acc_full.max()) > 0 and float(den_full.max()) == 1.0
    keep = torch.arange(n, device="cuda") % 3 != 0
    assert eng.remove_points((~keep).to(torch.uint8)) == int((~keep).sum())
    dd = dict(start_after=10 ** 9, remove_big_after=0, stop_after=10 ** 9, densify_every=1, grad_thresh=1.0, num_to_split_into=2,
              removal_opacity_threshold=0.0, final_removal_opacity_threshold=0.0, reset_opacities=False, reset_opacities_every=3000)
    assert not eng.densify(1, dd, 1.0, accumulate=False)        # off its schedule: nothing but the (skipped) accumulation would happen
    torch.cuda.synchronize()
    assert eng.P == int(keep.sum())
    assert torch.equal(variables['means2D_gradient_accum'][:eng.P], acc_full[keep])
    assert torch.equal(variables['denom'][:eng.P], den_full[keep])


def test_opacity_reset_leaves_the_reset_value_through_the_adam_step():
    """the reference's reset re-creates logit_opacities without a .grad (/root/reference/utils/slam_external.py:186-190),
    so the optimizer.step() of that iteration does not move it; the other groups step as usual."""
    import math
    from splatam_amd import slam
    eng, params, variables, mirror, frame = _densify_scene()
    cfg = slam.REPLICA_MAPPING
    eng.reset_map_optimizer()
    eng.loss_backward(frame, 1, cfg, tracking=False)
    assert float(eng.grads['logit_opacities'].abs().max()) > 0
    before = params['means3D'].detach().clone()
    pd = dict(start_after=10 ** 9, remove_big_after=0, stop_after=10 ** 9, prune_every=1, removal_opacity_threshold=0.0,
              final_removal_opacity_threshold=0.0, reset_opacities=True, reset_opacities_every=1)
    assert eng.prune_gaussians(1, pd, 1.0) == 0
    eng.adam_map(cfg['lrs'])
    torch.cuda.synchronize()
    want = math.log(0.01 / (1 - 0.01))
    assert float((params['logit_opacities'].detach() - want).abs().max()) <= 1e-6
    assert float((params['means3D'].detach() - before).abs().max()) > 0
