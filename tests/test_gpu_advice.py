"""Regression tests for the round-1 advisor findings: lazy-mode overflow under no_grad, densification render
on spilled buckets, per-call camera check, `seen` shape, the timing helper's workspace invariant."""
import numpy as np
import pytest
import torch

from tests.util import scene

pytestmark = pytest.mark.gpu


def _settings(cam):
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    return Camera(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                  bg=cam.bg.cuda(), scale_modifier=cam.scale_modifier, viewmatrix=cam.viewmatrix.cuda(),
                  projmatrix=cam.projmatrix.cuda(), sh_degree=cam.sh_degree, campos=cam.campos.cuda(), prefiltered=cam.prefiltered)


def test_lazy_mode_overflow_under_no_grad_rerenders():
    """Lazy sync mode sizes the lists from a per-(P, H, W) high-water mark.  A forward-only render (torch.no_grad) of a scene
    with the same P but many more (Gaussian, tile) instances used to return a background-only image silently."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from splatam_amd import rasterizer as rz
    W, H, n = 160, 112, 4000
    cam, rv = scene(n, W, H, 150.0, seed=4)
    cs = _settings(cam)
    small = {k: v.cuda() for k, v in rv.items()}
    big = dict(small)
    big['scales'] = small['scales'] * 6.0            # ~20x the instances for the same P
    rz.set_sync_mode("exact")
    with torch.no_grad():
        ref_big = Renderer(raster_settings=cs)(**big)[0].clone()
    rz._capacity_hint.clear()
    rz.set_sync_mode("lazy")
    try:
        with torch.no_grad():
            Renderer(raster_settings=cs)(**small)                  # learns a (too small) capacity for (P, H, W)
            hint = dict(rz._capacity_hint)
            got = Renderer(raster_settings=cs)(**big)[0]
        torch.cuda.synchronize()
        assert rz._capacity_hint != hint, "the overflow was not noticed"
        assert float((got - ref_big).abs().max()) < 1e-5
        # the same with inputs that REQUIRE grad (the params dict holds nn.Parameters) rendered under no_grad: needs_input_grad
        # is True there whatever the grad mode, so the wrapper passes the caller's grad mode along
        rz._capacity_hint.clear()
        with torch.no_grad():
            Renderer(raster_settings=cs)(**small)
            got_p = Renderer(raster_settings=cs)(**{k: torch.nn.Parameter(v.clone()) for k, v in big.items()})[0]
        torch.cuda.synchronize()
        assert float((got_p - ref_big).abs().max()) < 1e-5
        # with autograd the check happens in backward and raises (the forward that was handed out is invalid)
        rz._capacity_hint.clear()
        with torch.no_grad():
            Renderer(raster_settings=cs)(**small)
        inp = {k: v.clone().requires_grad_(True) for k, v in big.items()}
        col = Renderer(raster_settings=cs)(**inp)[0]
        with pytest.raises(RuntimeError, match="lazy sync mode"):
            col.sum().backward()
    finally:
        rz.set_sync_mode("exact")
        rz._capacity_hint.clear()


def _engine(n=6000, W=160, H=112, managed=False):
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    f, cx, cy = 150.0, W / 2 - 0.5, H / 2 - 0.5
    params, variables = slam.synthetic_params(n, W, H, f, f, cx, cy, num_frames=3, seed=1, device="cuda")
    k = [[f, 0, cx], [0, f, cy], [0, 0, 1]]
    w2c = torch.eye(4, device="cuda")
    cam = slam.setup_camera(W, H, k, np.eye(4, dtype=np.float32), device="cuda")
    im, depth = slam.synthetic_frame(params, cam, w2c, 1, rot_deg=0.4, trans_m=0.01)
    frame = {'cam': cam, 'im': im.contiguous(), 'depth': depth.contiguous(), 'id': 1, 'w2c': w2c,
             'intrinsics': torch.tensor(k, device="cuda")}
    eng = FusedEngine(params, cam, gaussian_capacity=4 * n if managed else None, variables=variables if managed else None)
    return eng, params, variables, frame, cam, k


def test_engine_rejects_a_different_camera():
    from splatam_amd import slam
    eng, params, variables, frame, cam, k = _engine()
    same = slam.setup_camera(160, 112, k, np.eye(4, dtype=np.float32), device="cuda")       # equal values, another tuple: fine
    eng.loss_backward(dict(frame, cam=same), 1, slam.REPLICA_MAPPING, tracking=False)
    k2 = [[140.0, 0, k[0][2]], [0, 140.0, k[1][2]], [0, 0, 1]]
    other = slam.setup_camera(160, 112, k2, np.eye(4, dtype=np.float32), device="cuda")
    with pytest.raises(RuntimeError, match="differs from the camera"):
        eng.loss_backward(dict(frame, cam=other), 1, slam.REPLICA_MAPPING, tracking=False)
    with pytest.raises(RuntimeError, match="differs from the camera"):
        eng.render(dict(frame, cam=other), 1)


def test_seen_has_the_maps_row_count():
    from splatam_amd import slam
    eng, params, variables, frame, cam, k = _engine(managed=True)
    eng.loss_backward(frame, 1, slam.REPLICA_MAPPING, tracking=False)
    assert eng.seen.shape == (eng.P,) and eng.seen.shape[0] == params['means3D'].shape[0]
    removed = eng.remove_points(torch.arange(eng.P, device="cuda") % 3 == 0)
    assert removed > 0
    eng.relearn_lists(frame, 1)
    eng.loss_backward(frame, 1, slam.REPLICA_MAPPING, tracking=False)
    assert eng.seen.shape == (eng.P,) == (params['means3D'].shape[0],)


def test_densification_render_notices_a_spilled_bucket():
    """add_new_gaussians renders with the current (bucketed) lists; a bucket that spills during THAT render must not feed
    the append.  Force it: learn buckets on the map, then shrink the bucket stride below the longest list."""
    from splatam_amd import slam
    eng, params, variables, frame, cam, k = _engine(managed=True)
    eng.relearn_lists(frame, 1)
    assert eng.tile_stride > 0
    ref_eng, *_ = _engine(managed=True)
    ref_eng.allow_buckets = False
    n_ref = ref_eng.add_new_gaussians(frame, 0.5, 1)
    eng.tile_stride = 64                     # stale: far below the longest list -> buckets spill in the densification render
    n_got = eng.add_new_gaussians(frame, 0.5, 1)
    assert n_got == n_ref
    assert torch.equal(eng.params['means3D'].detach(), ref_eng.params['means3D'].detach())


def test_time_kernel_leaves_the_accumulator_zeroed():
    import ctypes as C
    from splatam_amd import _capi, slam
    eng, params, variables, frame, cam, k = _engine()
    cfg = slam.REPLICA_MAPPING
    eng.loss_backward(frame, 1, cfg, tracking=False)
    torch.cuda.synchronize()
    g0 = eng.grads['means3D'].clone()
    ws = eng._workspace(True, with_ssim=True)
    ms = C.c_float(0)
    _capi.check(eng.L.splat_iter_time_kernel(1, 3, C.byref(eng._cam), eng.P, C.byref(ws), eng._stream(), C.byref(ms)), "time")
    torch.cuda.synchronize()
    assert float(eng.buf['accum'].abs().max()) == 0.0
    eng.loss_backward(frame, 1, cfg, tracking=False)          # the next real iteration is not polluted
    torch.cuda.synchronize()
    assert torch.allclose(eng.grads['means3D'], g0, rtol=1e-4, atol=1e-7 * float(g0.abs().max()))


# ---- round-2 advisor findings ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("stale_hint", [2000, 3000])
def test_stale_hint_with_lists_beyond_lds_is_flagged_and_memory_safe(stale_hint):
    """Exact-list path (tile_stride == 0) with the caller's scratch for the multi-workgroup sort: a stale list-length hint of
    ~683..2730 launches the workgroup-per-tile sort but NOT the multi-workgroup kernels (hint 2000), or launches them with too
    few merge passes (hint 3000 -> bound 4500 -> one pass for lists that need three).  Either way a list beyond 4 096 keys
    used to be left unsorted / half merged WITHOUT a flag.  Now: flagged (the host repeats) and published with valid ids."""
    from splatam_amd import slam
    from splatam_amd.fused import FusedEngine
    from tests.test_gpu_fused import _scene, _cmp
    params, variables, frame, cam = _scene(200000, 96, 64, seed=11)      # ~20 k instances per tile
    eng = FusedEngine(params, cam)
    eng.allow_buckets = False
    cfg = slam.REPLICA_MAPPING
    eng.loss_backward(frame, 1, cfg, tracking=False)
    torch.cuda.synchronize()
    assert not eng.check_overflow()
    assert eng.max_list_hint > 3 * 4096, eng.max_list_hint
    good = eng.grads['means3D'].clone()
    good_loss = eng.loss()
    eng.buf['point_list'].fill_(0x7f7f7f7f)
    if 'keys_alt' in eng.buf:
        eng.buf['keys_alt'].fill_(0x7f7f7f7f7f7f7f7f)
    eng.max_list_hint = stale_hint
    eng.loss_backward(frame, 1, cfg, tracking=False)
    torch.cuda.synchronize()                                            # no memory fault
    assert eng.check_overflow()                                          # the iteration is reported as invalid
    eng.loss_backward(frame, 1, cfg, tracking=False)
    torch.cuda.synchronize()
    assert not eng.check_overflow()
    assert abs(eng.loss() - good_loss) <= 1e-5 * abs(good_loss)
    _cmp(eng.grads['means3D'], good, "dL/dmeans3D after recovery", tol=1e-4)
