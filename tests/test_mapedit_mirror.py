"""Map growth / maintenance, torch formulation (splatam_amd/slam.py) against golden vectors produced by the
REFERENCE's own code (tests/golden/make_golden_mapedit.py: add_new_gaussians, get_pointcloud, initialize_params,
initialize_camera_pose cut out of scripts/splatam.py; prune_gaussians / remove_points of utils/slam_external.py
driving a real torch.optim.Adam).  CPU only."""
import os

import numpy as np
import pytest
import torch

from splatam_amd import slam

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mapedit_reference.npz"))
PARAM_KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales', 'cam_unnorm_rots', 'cam_trans')
VAR_KEYS = ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'timestep')


def load_add_case(name, device="cpu"):
    t = lambda a: torch.tensor(a, device=device)        # noqa: E731
    params = {k: torch.nn.Parameter(t(GOLD[f"{name}/in/param/{k}"])) for k in PARAM_KEYS}
    variables = {k: t(GOLD[f"{name}/in/var/{k}"]) for k in VAR_KEYS}
    W, H, time_idx, sil_thres, iso = GOLD[f"{name}/in/meta"]
    curr = {'cam': None, 'im': t(GOLD[f"{name}/in/im"]), 'depth': t(GOLD[f"{name}/in/depth"]), 'id': int(time_idx),
            'intrinsics': t(GOLD[f"{name}/in/intrinsics"]), 'w2c': torch.eye(4, device=device)}
    return params, variables, curr, t(GOLD[f"{name}/in/depth_sil"]), int(time_idx), float(sil_thres), \
        ("isotropic" if iso else "anisotropic")


def check_add_outputs(name, params, variables):
    for k in PARAM_KEYS:
        got, ref = params[k].detach().cpu().numpy(), GOLD[f"{name}/out/param/{k}"]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6, err_msg=k)
    for k in VAR_KEYS:
        got, ref = variables[k].detach().cpu().numpy(), GOLD[f"{name}/out/var/{k}"]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        np.testing.assert_array_equal(got, ref, err_msg=k)


@pytest.mark.parametrize("name", ["add_iso", "add_aniso", "add_nan", "add_nothing"])
def test_add_new_gaussians(name):
    params, variables, curr, depth_sil, time_idx, sil_thres, dist = load_add_case(name)
    params, variables = slam._add_from_render(params, variables, curr, depth_sil, sil_thres, time_idx, "projective", dist)
    check_add_outputs(name, params, variables)


@pytest.mark.parametrize("name", ["init_iso", "init_aniso"])
def test_pointcloud_and_initialize_params(name):
    im, depth = torch.tensor(GOLD[f"{name}/in/im"]), torch.tensor(GOLD[f"{name}/in/depth"])
    k, w2c = torch.tensor(GOLD[f"{name}/in/intrinsics"]), torch.tensor(GOLD[f"{name}/in/w2c"])
    mask = (depth > 0).reshape(-1)
    cloud, msd = slam.get_pointcloud(im, depth, k, w2c, mask=mask, compute_mean_sq_dist=True)
    np.testing.assert_allclose(cloud.numpy(), GOLD[f"{name}/cloud"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(msd.numpy(), GOLD[f"{name}/msd"], rtol=1e-6)
    np.testing.assert_allclose(slam.get_pointcloud(im, depth, k, w2c, transform_pts=False).numpy(), GOLD[f"{name}/cloud_cam"],
                               rtol=1e-6, atol=1e-7)
    params, variables = slam.initialize_params(cloud, 5, msd, "isotropic" if name == "init_iso" else "anisotropic")
    for kk in PARAM_KEYS:
        np.testing.assert_allclose(params[kk].detach().numpy(), GOLD[f"{name}/out/param/{kk}"], rtol=2e-6, atol=2e-6, err_msg=kk)
        assert params[kk].requires_grad
    for kk in VAR_KEYS:
        np.testing.assert_array_equal(variables[kk].numpy(), GOLD[f"{name}/out/var/{kk}"])
    with pytest.raises(ValueError):
        slam.initialize_new_params(cloud, msd, "spherical")


def load_prune_case(name, device="cpu"):
    t = lambda a: torch.tensor(a, device=device)        # noqa: E731
    params = {k: torch.nn.Parameter(t(GOLD[f"{name}/in/param/{k}"])) for k in PARAM_KEYS}
    variables = {k: t(GOLD[f"{name}/in/var/{k}"]) for k in VAR_KEYS + ('scene_radius',)}
    moments = {k: (t(GOLD[f"{name}/in/exp_avg/{k}"]), t(GOLD[f"{name}/in/exp_avg_sq/{k}"])) for k in slam.GAUSSIAN_KEYS}
    return params, variables, moments


@pytest.mark.parametrize("name", ["prune_iso", "prune_aniso"])
def test_prune_gaussians(name):
    params, variables, moments = load_prune_case(name)
    opt = slam.initialize_optimizer(params, slam.REPLICA_MAPPING['lrs'], tracking=False)
    for k, (m, v) in moments.items():
        opt.state[params[k]] = {'step': torch.tensor(1.0), 'exp_avg': m, 'exp_avg_sq': v}
    n0 = params['means3D'].shape[0]
    params, variables = slam.prune_gaussians(params, variables, opt, 0, slam.REPLICA_PRUNE)
    assert 0 < params['means3D'].shape[0] < n0
    for k in PARAM_KEYS:
        np.testing.assert_array_equal(params[k].detach().numpy(), GOLD[f"{name}/out/param/{k}"], err_msg=k)
    for k in slam.GAUSSIAN_KEYS:
        st = opt.state[params[k]]
        np.testing.assert_array_equal(st['exp_avg'].numpy(), GOLD[f"{name}/out/exp_avg/{k}"])
        np.testing.assert_array_equal(st['exp_avg_sq'].numpy(), GOLD[f"{name}/out/exp_avg_sq/{k}"])
    for k in VAR_KEYS:
        np.testing.assert_array_equal(variables[k].numpy(), GOLD[f"{name}/out/var/{k}"], err_msg=k)
    n1 = params['means3D'].shape[0]
    params, variables = slam.prune_gaussians(params, variables, opt, 7, slam.REPLICA_PRUNE)      # off the schedule
    assert params['means3D'].shape[0] == n1


def test_initialize_camera_pose():
    params = {'cam_unnorm_rots': torch.nn.Parameter(torch.tensor(GOLD["pose/in/cam_unnorm_rots"])),
              'cam_trans': torch.nn.Parameter(torch.tensor(GOLD["pose/in/cam_trans"]))}
    slam.initialize_camera_pose(params, 3, True)
    slam.initialize_camera_pose(params, 1, True)
    slam.initialize_camera_pose(params, 5, False)
    np.testing.assert_allclose(params['cam_unnorm_rots'].detach().numpy(), GOLD["pose/out/cam_unnorm_rots"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(params['cam_trans'].detach().numpy(), GOLD["pose/out/cam_trans"], rtol=1e-6, atol=1e-7)


def test_densify_matches_reference():
    """Gradient-based densification (clone + split + prune + moments) against the reference's own densify under the same
    torch seed (the split draws torch.normal samples)."""
    name = "densify_iso"
    params = {k: torch.nn.Parameter(torch.tensor(GOLD[f"{name}/in/param/{k}"])) for k in PARAM_KEYS}
    variables = {k: torch.tensor(GOLD[f"{name}/in/var/{k}"]) for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'scene_radius', 'seen')}
    m2d = torch.zeros(params['means3D'].shape[0], 3, requires_grad=True)
    m2d.grad = torch.tensor(GOLD[f"{name}/in/means2D_grad"])
    variables['means2D'] = m2d
    opt = slam.initialize_optimizer(params, slam.REPLICA_MAPPING['lrs'], tracking=False)
    for k in slam.GAUSSIAN_KEYS:
        opt.state[params[k]] = {'step': torch.tensor(1.0), 'exp_avg': torch.tensor(GOLD[f"{name}/in/exp_avg/{k}"]),
                                'exp_avg_sq': torch.tensor(GOLD[f"{name}/in/exp_avg_sq/{k}"])}
    dd = dict(start_after=0, remove_big_after=0, stop_after=5000, densify_every=100, grad_thresh=0.0002, num_to_split_into=2,
              removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=True, reset_opacities_every=300)
    torch.manual_seed(1234)
    params, variables = slam.densify(params, variables, opt, 100, dd)
    n0, n1 = GOLD[f"{name}/counts"]
    assert params['means3D'].shape[0] == n1 != n0
    for k in PARAM_KEYS:
        np.testing.assert_allclose(params[k].detach().numpy(), GOLD[f"{name}/out/param/{k}"], rtol=1e-6, atol=1e-7, err_msg=k)
    for k in slam.GAUSSIAN_KEYS:
        st = opt.state[params[k]]
        np.testing.assert_array_equal(st['exp_avg'].numpy(), GOLD[f"{name}/out/exp_avg/{k}"])
        np.testing.assert_array_equal(st['exp_avg_sq'].numpy(), GOLD[f"{name}/out/exp_avg_sq/{k}"])
    for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom'):
        np.testing.assert_array_equal(variables[k].numpy(), GOLD[f"{name}/out/var/{k}"], err_msg=k)
    # past the schedule nothing happens
    n = params['means3D'].shape[0]
    params, variables = slam.densify(params, variables, opt, 6000, dd)
    assert params['means3D'].shape[0] == n
