"""CPU side of splatam_amd.plugin: the adapters keep the reference's signatures, and the loop statements that
tests/test_gpu_plugin.py restates appear in the reference's source in that order (checked when /root/reference is present: the GPU
box has no copy of it)."""
import inspect
import os
import re

import pytest

REF = "/root/reference/scripts/splatam.py"


def test_adapter_signatures_match_the_reference_call_sites():
    from splatam_amd import plugin
    sig = inspect.signature(plugin.get_loss)
    names = list(sig.parameters)
    assert names[:9] == ['params', 'curr_data', 'variables', 'iter_time_idx', 'loss_weights', 'use_sil_for_loss', 'sil_thres', 'use_l1',
                         'ignore_outlier_depth_loss']
    for kw in ('tracking', 'mapping', 'do_ba', 'plot_dir', 'visualize_tracking_loss', 'tracking_iteration'):
        assert kw in sig.parameters and sig.parameters[kw].default in (False, None)
    assert list(inspect.signature(plugin.initialize_optimizer).parameters) == ['params', 'lrs_dict', 'tracking']


def test_install_refuses_a_module_without_the_loop_names():
    import types
    from splatam_amd import plugin
    with pytest.raises(RuntimeError, match="get_loss"):
        plugin.install(types.ModuleType("empty"))


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is not on this machine")
def test_restated_loop_statements_follow_the_reference_source():
    src = open(REF).read().splitlines()

    def first(pattern, lo, hi):
        for i in range(lo - 1, hi):
            if re.search(pattern, src[i]):
                return i + 1
        raise AssertionError(f"{pattern!r} not found in lines {lo}-{hi}")
    # the reference's signatures the adapters mirror
    assert re.search(r"def get_loss\(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss,", src[213])
    assert "def initialize_optimizer(params, lrs_dict, tracking):" in src[159]
    # tracking loop: optimizer per frame, then get_loss -> backward -> step -> zero_grad -> best candidate
    t_opt = first(r"optimizer = initialize_optimizer\(params, config\['tracking'\]\['lrs'\], tracking=True\)", 670, 690)
    t = [first(p, t_opt, 745) for p in (r"loss, variables, losses = get_loss\(params, tracking_curr_data, variables, iter_time_idx",
                                         r"loss\.backward\(\)", r"optimizer\.step\(\)", r"optimizer\.zero_grad\(set_to_none=True\)",
                                         r"if loss < current_min_loss")]
    assert t == sorted(t), t
    # mapping loop: optimizer per frame, get_loss -> backward -> prune_gaussians -> step -> zero_grad
    m_opt = first(r"optimizer = initialize_optimizer\(params, config\['mapping'\]\['lrs'\], tracking=False\)", 800, 830)
    m = [first(p, m_opt, 895) for p in (r"loss, variables, losses = get_loss\(params, iter_data, variables, iter_time_idx",
                                         r"loss\.backward\(\)", r"params, variables = prune_gaussians\(params, variables, optimizer, iter",
                                         r"optimizer\.step\(\)", r"optimizer\.zero_grad\(set_to_none=True\)")]
    assert m == sorted(m), m


def test_loss_scalars_read_through_their_report():
    """The loss / losses['depth' | 'im'] values the adapters return are views of a copy of the iteration's report; comparisons with
    numbers and with each other, float(), item() and format() are decided from ONE host copy of that report (the read the reference's
    `if loss < current_min_loss` makes anyway), torch operations see plain tensors, backward() has nothing left to do."""
    import torch
    from splatam_amd import plugin
    rep = plugin._Report(torch.arange(32, dtype=torch.float32))
    loss, depth = plugin._scalar(rep, 7), plugin._scalar(rep, 14)
    assert rep.host is None
    assert (loss < 1e20) is True and (1e20 > loss) is True and (loss > 1e20) is False
    assert rep.host is not None and rep.host[7] == 7.0              # one read fetched the whole report
    assert float(loss) == 7.0 and loss.item() == 7.0 and f"{loss:.1f}" == "7.0"
    assert (loss < depth) is True and (depth <= loss) is False
    current_min = loss                                               # the reference keeps the tensor: `current_min_loss = loss`
    later = plugin._scalar(plugin._Report(torch.full((32,), 3.0)), 7)
    assert (later < current_min) is True
    assert loss.backward() is None
    assert type(loss + 1) is torch.Tensor and type(loss.detach()) is torch.Tensor
    other = torch.tensor(8.0)
    assert isinstance(loss < other, torch.Tensor) and bool(loss < other)


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is not on this machine")
def test_install_patches_what_the_reference_loop_resolves():
    """``plugin.install`` on the REAL module (/root/reference/scripts/splatam.py imported as tests/golden/make_golden_loop.py imports
    it; in a subprocess -- the import shims are process wide): the two names are replaced; the loop looks both of them up as module
    globals at call time (so the replacement is what runs); every ``get_loss`` / ``initialize_optimizer`` call of ``rgbd_slam`` binds to
    the adapters' signatures; ``uninstall`` puts the reference's functions back.  No machine available to this repository has both the
    reference tree and a GPU, so the loop itself runs with the plug-in on the restated statements (tests/test_gpu_loop_golden.py),
    which tests/test_loop_golden.py pins call by call to a recording of this very module."""
    import subprocess
    import sys
    code = r'''
import ast, inspect, sys, types
sys.path.insert(0, "tests/golden")
import make_golden_loop as G
G.install_device_shim()
dummy = types.ModuleType("diff_gaussian_rasterization")
dummy.GaussianRasterizer = object
from splatam_amd.rasterizer import GaussianRasterizationSettings
dummy.GaussianRasterizationSettings = GaussianRasterizationSettings
S = G.load_reference_module(dummy)
from splatam_amd import plugin
ref_get_loss, ref_init = S.get_loss, S.initialize_optimizer
for name in ("get_loss", "initialize_optimizer"):
    assert name in S.rgbd_slam.__code__.co_names and name not in S.rgbd_slam.__code__.co_varnames and name not in S.rgbd_slam.__code__.co_freevars, name
    assert S.rgbd_slam.__globals__ is vars(S)
handle = plugin.install(S)
assert S.get_loss is plugin.get_loss and S.initialize_optimizer is plugin.initialize_optimizer
tree = ast.parse(inspect.getsource(S.rgbd_slam))
calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in ("get_loss", "initialize_optimizer")]
seen = {"get_loss": 0, "initialize_optimizer": 0}
for c in calls:
    sig = inspect.signature(getattr(plugin, c.func.id))
    sig.bind(*[None] * len(c.args), **{k.arg: None for k in c.keywords})
    seen[c.func.id] += 1
assert seen == {"get_loss": 2, "initialize_optimizer": 2}, seen
handle.uninstall()
assert S.get_loss is ref_get_loss and S.initialize_optimizer is ref_init
# map_edits=True: the two map edits (looked up the same way by the loop) and densify as well, every call site binds, all restored
edits = ("add_new_gaussians", "prune_gaussians", "densify")
ref_edits = {n: getattr(S, n) for n in edits}
for name in edits:
    assert name in S.rgbd_slam.__code__.co_names and name not in S.rgbd_slam.__code__.co_varnames, name
handle = plugin.install(S, map_edits=True)
assert all(getattr(S, n) is getattr(plugin, n) for n in edits) and S.get_loss is plugin.get_loss
bound = {n: 0 for n in edits}
for c in (n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in edits):
    inspect.signature(getattr(plugin, c.func.id)).bind(*[None] * len(c.args), **{k.arg: None for k in c.keywords})
    bound[c.func.id] += 1
assert bound == {"add_new_gaussians": 1, "prune_gaussians": 1, "densify": 1}, bound
handle.uninstall()
assert all(getattr(S, n) is ref_edits[n] for n in edits) and S.get_loss is ref_get_loss
print("OK", seen, bound)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_map_edits_prune_adapter_follows_the_references_schedule():
    """plugin.prune_gaussians (map_edits mode) decides on the host WHEN the engine prunes / resets opacities and whether the
    ``optimizer.step()`` that follows moves the Gaussians.  Against the reference-shaped prune_gaussians run on CPU tensors with a real
    torch.optim.Adam: the engine is called exactly on the iterations where that function re-creates a parameter, and the step is
    skipped exactly when it re-creates ALL of them (remove_points does, whether or not a row goes: torch then finds no .grad)."""
    import torch
    from splatam_amd import plugin, slam

    class StubEngine:
        H, W = 4, 4

        def __init__(self, params):
            self.params, self.calls = params, []

        def prune_gaussians(self, it, pd, scene_radius):
            self.calls.append(it)
            return 0

    class StubReport:
        args = (None, {}, 0, {}, False, False)

    schedules = [
        dict(start_after=0, remove_big_after=0, stop_after=20, prune_every=20, removal_opacity_threshold=0.005,
             final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=500),
        dict(start_after=3, remove_big_after=5, stop_after=17, prune_every=4, removal_opacity_threshold=0.005,
             final_removal_opacity_threshold=0.01, reset_opacities=True, reset_opacities_every=6),
        dict(start_after=2, remove_big_after=0, stop_after=9, prune_every=3, removal_opacity_threshold=0.005,
             final_removal_opacity_threshold=0.005, reset_opacities=True, reset_opacities_every=3),
    ]
    for pd in schedules:
        n = 12
        ref = {k: torch.nn.Parameter(torch.zeros(n, w)) for k, w in (('means3D', 3), ('rgb_colors', 3), ('unnorm_rotations', 4), ('logit_opacities', 1), ('log_scales', 1))}
        ref['cam_unnorm_rots'] = torch.nn.Parameter(torch.zeros(1, 4, 2))
        ref['cam_trans'] = torch.nn.Parameter(torch.zeros(1, 3, 2))
        variables = {k: torch.zeros(n) for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius', 'timestep')}
        variables['scene_radius'] = torch.tensor(100.0)
        opt = slam.initialize_optimizer(ref, slam.REPLICA_MAPPING['lrs'], tracking=False)
        mine = {k: v.detach().clone() for k, v in ref.items()}
        eng = StubEngine(mine)
        s = plugin._session
        saved = (s.current, s.map_edits, s.skip_gaussian_step, dict(s.bound))
        try:
            s.map_edits, s.current = True, (eng, (), StubReport())
            for it in range(0, 25):
                before = {k: ref[k] for k in ('means3D', 'logit_opacities')}
                ref, variables = slam.prune_gaussians(ref, variables, opt, it, pd)
                all_recreated = ref['means3D'] is not before['means3D']
                any_recreated = all_recreated or ref['logit_opacities'] is not before['logit_opacities']
                eng.calls.clear()
                s.skip_gaussian_step = False
                plugin.prune_gaussians(mine, dict(variables), None, it, pd)
                assert bool(eng.calls) == any_recreated, (pd, it)
                assert s.skip_gaussian_step == all_recreated, (pd, it)
        finally:
            s.current, s.map_edits, s.skip_gaussian_step = saved[:3]
            s.bound.clear()
            s.bound.update(saved[3])
