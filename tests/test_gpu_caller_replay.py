"""The unmodified caller's contract on the HIP path itself.

tests/golden/caller_reference.npz (tests/golden/make_golden_caller.py) holds what the REFERENCE's own get_loss
(/root/reference/scripts/splatam.py:214-347, exec'd from its source) does at the rasterizer boundary: the settings tuple, the
exact kwargs of both ``Renderer(raster_settings=curr_data['cam'])(**rendervar)`` calls (:249, :253), the gradients autograd
hands to the two renders, what each call's inputs receive, the retained non-leaf ``means2D`` (:248) and the ``radius`` bookkeeping
(:341-345).  Here exactly those kwargs go through ``diff_gaussian_rasterization`` on the GPU -- two forwards, then the two
backwards in autograd's order -- and every output the caller reads is compared with the recording (oracle renders)."""
import os

import numpy as np
import pytest
import torch

from tests.util import assert_close_outliers, grad_scale

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caller_reference.npz"))
IN_KEYS = ('means3D', 'colors_precomp', 'rotations', 'opacities', 'scales', 'means2D')


def _camera():
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    n, W, H = (int(x) for x in GOLD["meta"][:3])
    t = lambda k: torch.tensor(GOLD[f"cam/{k}"]).cuda()      # noqa: E731
    return Camera(image_height=H, image_width=W, tanfovx=float(GOLD["cam/tanfovx"]), tanfovy=float(GOLD["cam/tanfovy"]), bg=t("bg"),
                  scale_modifier=float(GOLD["cam/scale_modifier"]), viewmatrix=t("viewmatrix"), projmatrix=t("projmatrix"),
                  sh_degree=0, campos=t("campos"), prefiltered=False)


@pytest.fixture(params=["geometry cache on", "geometry cache off", "auto policy: fast path"])
def geometry_cache(request):
    """The reference's capacity policy ("exact": one host read per forward, as the CUDA original) with the geometry shared between the
    two renders or not; and the default policy's steady state (no host read, every render its own two launches)."""
    from splatam_amd import rasterizer as rz
    on = request.param.endswith("on")
    auto = request.param.startswith("auto")
    rz.set_sync_mode("auto" if auto else "exact")
    rz.set_geometry_cache(on)
    before = dict(rz.geometry_cache_stats)
    yield on, before
    rz.set_geometry_cache(True)
    rz.set_sync_mode("auto")


@pytest.mark.parametrize("mode", ["tracking", "mapping"])
def test_reference_call_sequence_replayed_on_hip(mode, geometry_cache):
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from splatam_amd import rasterizer as rz
    cam = _camera()
    calls = []
    shared = {}
    for ci in (0, 1):
        kw = {}
        for k in IN_KEYS:
            if k == 'means3D' and k in shared:
                # both render-variable dicts of the caller hold the SAME transformed_gaussians['means3D'] tensor
                # (/root/reference/utils/slam_helpers.py:131,241): its recorded gradient is the sum over the two calls
                kw[k] = shared[k]
                continue
            src = GOLD[f"call{ci}/in/{k}"] if f"call{ci}/in/{k}" in GOLD.files else GOLD[f"call0/in/{k}"]
            leaf = torch.tensor(src).cuda()
            wants = f"{mode}/call{ci}/grad_in/{k}" in GOLD.files
            if k == 'means2D':
                # the caller's construction (/root/reference/utils/slam_helpers.py:137): a NON-LEAF zero tensor with retain_grad()
                kw[k] = torch.zeros_like(leaf, requires_grad=True) + 0
                kw[k].retain_grad()
            else:
                kw[k] = leaf.requires_grad_(wants)
            if k == 'means3D':
                shared[k] = kw[k]
        calls.append(kw)
    if rz.get_sync_mode() == "auto":
        with torch.no_grad():           # the scene's first call (exact lists; learns the longest list): the replay below is its steady state
            Renderer(raster_settings=cam)(**calls[0])
        fast_before = rz.fast_path_stats["fast"]
    # forward: two fresh modules, keyword arguments, 3-tuples (the second forward precedes the first backward)
    outs = []
    for kw in calls:
        res = Renderer(raster_settings=cam)(**kw)
        assert isinstance(res, tuple) and len(res) == 3
        outs.append(res)
    # the second render re-used the first one's geometry and sorted lists (proven equal on the device: its opacities / scales /
    # rotations are distinct tensors, as in the caller) -- or, with the cache off, ran the whole pass
    on, before = geometry_cache
    assert rz.geometry_cache_stats["shared"] - before["shared"] == (1 if on else 0), (rz.geometry_cache_stats, before)
    if rz.get_sync_mode() == "auto":
        assert rz.fast_path_stats["fast"] == fast_before + 2, rz.fast_path_stats
    for ci, (color, radii, depth) in enumerate(outs):
        assert radii.dtype == torch.int32 and tuple(depth.shape) == (1,) + tuple(color.shape[1:])
        assert (radii.cpu().numpy() != GOLD[f"call{ci}/out/radii"]).sum() <= 1
        ref = GOLD[f"call{ci}/out/color"]
        assert_close_outliers(color.detach().cpu().numpy(), ref, 1e-4, rtol=1e-4, max_outlier_frac=1e-4, outlier_atol=0.03 * max(1.0, np.abs(ref).max()),
                              what=f"call {ci} color")
        assert_close_outliers(depth.detach().cpu().numpy(), GOLD[f"call{ci}/out/depth"], 1e-4, rtol=1e-4, max_outlier_frac=1e-4, outlier_atol=0.1,
                              what=f"call {ci} depth")
    # backward: ONE backward over both renders with the gradients the reference's loss delivered (depth-silhouette pass first
    # in autograd's reverse order, then RGB -- per-call state, no globals)
    torch.autograd.backward([outs[0][0], outs[1][0]],
                            [torch.tensor(GOLD[f"{mode}/call0/grad_out/color"]).cuda(), torch.tensor(GOLD[f"{mode}/call1/grad_out/color"]).cuda()])
    torch.cuda.synchronize()
    for ci, kw in enumerate(calls):
        for k in IN_KEYS:
            key = f"{mode}/call{ci}/grad_in/{k}"
            if key not in GOLD.files:
                assert kw[k].grad is None, (ci, k)
                continue
            ref = GOLD[key]
            got = kw[k].grad
            assert got is not None, (ci, k)
            if k == 'means3D':
                # in the caller's graph the colours of the second render are a function of the centres
                # ([z, 1, z^2] with z = (w2c @ [X; 1])[2], /root/reference/utils/slam_helpers.py:196-213; w2c = identity in the
                # recording), so the recorded dL/dmeans3D also holds dL/dcolours chained through z; here colours are a leaf
                gc = calls[1]['colors_precomp'].grad
                z = kw[k].detach()[:, 2]
                got = got.clone()
                got[:, 2] += gc[:, 0] + 2.0 * z * gc[:, 2]
            if float(np.abs(ref).max()) == 0.0:
                assert float(got.abs().max()) == 0.0, (ci, k)
                continue
            assert_close_outliers(got.cpu().numpy().reshape(ref.shape), ref, 1e-3 * grad_scale(ref), max_outlier_frac=2e-4,
                                  outlier_atol=0.05 * grad_scale(ref), what=f"{mode} call {ci} grad {k}")
    # what the caller reads afterwards: variables['means2D'].grad (the colour pass') and the radius bookkeeping
    m2 = calls[0]['means2D'].grad.cpu().numpy()
    ref = GOLD[f"{mode}/means2D_grad"]
    assert float(np.abs(m2[:, 2]).max()) == 0.0
    assert_close_outliers(m2, ref, 1e-3 * grad_scale(ref), max_outlier_frac=2e-4, outlier_atol=0.05 * grad_scale(ref), what="means2D.grad")
    radius = outs[0][1]
    seen = radius > 0
    max_r = torch.zeros(radius.shape[0], device="cuda")
    max_r[seen] = torch.max(radius[seen], max_r[seen])          # /root/reference/scripts/splatam.py:341-343
    assert (seen.cpu().numpy() != GOLD[f"{mode}/seen"]).sum() <= 1
    assert (max_r.cpu().numpy() != GOLD[f"{mode}/max_2D_radius"]).sum() <= 1


def test_geometry_cache_shares_only_what_is_proven_equal():
    """K1-K5 of a call are re-used by the next one only when camera, means3D (same storage, same version), opacities, scales and
    rotations (same tensors, or bitwise equal on the device) are the same; anything else runs the full pass.  Shared or not, the
    render is the same function."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from splatam_amd import rasterizer as rz
    cam = _camera()
    g = lambda k: torch.tensor(GOLD[f"call0/in/{k}"]).cuda()      # noqa: E731
    base = {k: g(k) for k in ('means3D', 'colors_precomp', 'rotations', 'opacities', 'scales', 'means2D')}
    rz.set_sync_mode("exact")           # (the cache belongs to the reference's capacity policy: the one host read proves the equality)
    rz.set_geometry_cache(True)
    try:
        def render(**over):
            kw = dict(base)
            kw.update(over)
            n = rz.geometry_cache_stats["shared"]
            out = Renderer(raster_settings=cam)(**kw)
            return out, rz.geometry_cache_stats["shared"] - n
        (c0, r0, d0), s = render()
        assert s == 0                                               # nothing cached yet
        other_colors = base['colors_precomp'].flip(0).contiguous()
        (c1, r1, d1), s = render(colors_precomp=other_colors, scales=base['scales'].clone(), rotations=base['rotations'].clone(),
                                 opacities=base['opacities'].clone())
        assert s == 1                                               # equal values in distinct tensors: verified on the device
        rz.set_geometry_cache(False)
        (c1_ref, r1_ref, d1_ref), _ = render(colors_precomp=other_colors)
        rz.set_geometry_cache(True)
        assert torch.equal(c1, c1_ref) and torch.equal(d1, d1_ref) and torch.equal(r1, r1_ref)
        render()                                                    # (fills the cache again)
        sc = base['scales'].clone()
        sc[123, 1] *= 1.5
        (_, _, _), s = render(scales=sc)
        assert s == 0                                               # one scale differs: full pass
        render()
        (_, _, _), s = render(means3D=base['means3D'].clone())
        assert s == 0                                               # another means3D tensor: no proof, full pass
        render()
        base['means3D'][7, 0] += 0.01                               # modified in place: the version moved
        (_, _, _), s = render()
        assert s == 0
        (_, _, _), s = render()
        assert s == 1                                               # ... and the unchanged map shares again
        from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
        cam2 = Camera(**{**cam._asdict(), 'viewmatrix': cam.viewmatrix.clone()})
        n = rz.geometry_cache_stats["shared"]
        Renderer(raster_settings=cam2)(**base)
        assert rz.geometry_cache_stats["shared"] == n                # another view matrix tensor: full pass
    finally:
        rz.set_geometry_cache(True)


def test_geometry_cache_sees_writes_that_bump_no_version_counter():
    """`.data` arithmetic and raw-pointer kernels (this library's own Adam / map edits) change a tensor without moving its
    version counter.  The cache compares VALUES on the device -- centres included, also when the second call passes the very same
    tensors -- so such a write gives a full pass, not a stale projection; and the second call's radii are a tensor of their own."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from splatam_amd import rasterizer as rz
    cam = _camera()
    g = lambda k: torch.tensor(GOLD[f"call0/in/{k}"]).cuda()      # noqa: E731
    base = {k: g(k) for k in ('means3D', 'colors_precomp', 'rotations', 'opacities', 'scales', 'means2D')}
    rz.set_sync_mode("exact")           # (the cache belongs to the reference's capacity policy: the one host read proves the equality)
    rz.set_geometry_cache(True)
    try:
        def render():
            n = rz.geometry_cache_stats["shared"]
            out = Renderer(raster_settings=cam)(**base)
            return out, rz.geometry_cache_stats["shared"] - n
        (c0, r0, d0), s = render()
        (c1, r1, d1), s = render()
        assert s == 1 and torch.equal(c0, c1) and r0.data_ptr() != r1.data_ptr() and torch.equal(r0, r1)
        for key, idx in (('means3D', (7, 0)), ('scales', (11, 1)), ('opacities', (13, 0))):
            render()                                                    # (fills the cache)
            v = base[key]._version
            base[key].data[idx] += 0.05 if key != 'opacities' else -0.2
            assert base[key]._version == v                               # the write is invisible to autograd's bookkeeping
            (c2, _, _), s = render()
            assert s == 0, key                                          # ... but not to the cache
            rz.set_geometry_cache(False)
            (c2_ref, _, _), _ = render()
            rz.set_geometry_cache(True)
            assert torch.equal(c2, c2_ref), key
    finally:
        rz.set_geometry_cache(True)
