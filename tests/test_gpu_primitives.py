"""GPU self-tests of the wave64 / LDS building blocks (through the C ABI's test hook)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from splatam_amd import _capi
    return _capi.lib()


def test_wave_reduce4_packed_matches_sum():
    L = _lib()
    nw = 37
    x = torch.randn(nw, 4, 64, device="cuda")
    out = torch.empty(nw, 64, device="cuda")
    rc = L.splat_selftest(0, x.data_ptr(), out.data_ptr(), nw, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    sums = x.double().sum(dim=2).cpu().numpy()                 # [nw,4]
    got = out.cpu().numpy().reshape(nw, 4, 16)
    row_value = [0, 2, 1, 3]
    for r in range(4):
        want = sums[:, row_value[r]][:, None]
        np.testing.assert_allclose(got[:, r, :], np.broadcast_to(want, (nw, 16)), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 255, 256, 257, 1000, 2048, 4095, 4096])
def test_lds_bitonic_sort(n):
    L = _lib()
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 2 ** 62, (n,), generator=g, dtype=torch.int64).cuda()
    out = torch.empty_like(keys)
    rc = L.splat_selftest(1, keys.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.sort(keys.cpu())[0])


def _depth_id_keys(n, kind, seed):
    """Keys as the binning builds them: float32 bits of a positive depth << 32 | Gaussian id (unique ids)."""
    rng = np.random.default_rng(seed)
    ids = rng.permutation(5_000_000)[:n].astype(np.uint64)
    if kind == "wide":                                  # every depth byte varies
        depth = rng.uniform(0.01, 900.0, n).astype(np.float32)
    elif kind == "narrow":                              # a tile's depth range: the exponent byte never varies (pass skipped)
        depth = rng.uniform(2.0, 2.4, n).astype(np.float32)
    elif kind == "short_ties":                          # a few equal depths next to each other (neighbour pass)
        depth = np.round(rng.uniform(1.0, 3.0, n) * (max(n, 8) / 4)).astype(np.float32)
    elif kind == "plane":                               # long runs of equal depth (the all-bytes attempt)
        depth = rng.choice(np.array([1.5, 2.0, 2.0000002], np.float32), n)
    else:                                               # one depth
        depth = np.full(n, 3.25, np.float32)
    return (depth.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids


@pytest.mark.parametrize("kind", ["wide", "narrow", "short_ties", "plane", "constant"])
@pytest.mark.parametrize("which,n", [(3, 1), (3, 2), (3, 63), (3, 64), (3, 65), (3, 257), (3, 1000), (3, 2500), (3, 4095), (3, 4096),
                                     (4, 1), (4, 5), (4, 64), (4, 65), (4, 200), (4, 777), (4, 1023), (4, 1024),
                                     (5, 1), (5, 2), (5, 64), (5, 65), (5, 1000), (5, 1025), (5, 3000), (5, 4095), (5, 4096),
                                     (6, 3), (6, 4097), (6, 6000), (6, 8191), (6, 8192),
                                     (7, 1), (7, 2), (7, 7), (7, 8), (7, 9), (7, 513), (7, 4097), (7, 6000), (7, 8191), (7, 8192),
                                     (8, 1), (8, 3), (8, 4), (8, 5), (8, 255), (8, 1025), (8, 3000), (8, 4095), (8, 4096)])
def test_lds_radix_sort_on_depth_key(which, n, kind):
    """radix_sort_lds (splat_device.h): the per-tile sort of the list kernels -- 8-bit passes over the depth bits with wave-ballot
    ranking, equal depths ordered by id: exactly numpy's sort of the 64-bit keys.  which: 3 = 4 waves, 5 = 16 waves (the list
    kernels' workgroup), 6 = 16 waves on up to 8 192 keys, 4 = one wave (tile_sort_wave_kernel); 7 / 8 = radix_sort_lds_private (thread-private
    ranking, 4-bit digits) on up to 8 192 / 4 096 keys: the run sort and the block sort."""
    L = _lib()
    keys_np = _depth_id_keys(n, kind, seed=1000 * which + n)
    keys = torch.from_numpy(keys_np.view(np.int64)).cuda()
    out = torch.zeros_like(keys)
    rc = L.splat_selftest(which, keys.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64), np.sort(keys_np))


def test_lds_radix_sort_rejects_lists_beyond_lds():
    L = _lib()
    keys = torch.zeros(5000, dtype=torch.int64, device="cuda")
    assert L.splat_selftest(3, keys.data_ptr(), keys.data_ptr(), 4097, torch.cuda.current_stream().cuda_stream) != 0
    assert L.splat_selftest(4, keys.data_ptr(), keys.data_ptr(), 1025, torch.cuda.current_stream().cuda_stream) != 0
    big = torch.zeros(9000, dtype=torch.int64, device="cuda")
    assert L.splat_selftest(6, big.data_ptr(), big.data_ptr(), 8193, torch.cuda.current_stream().cuda_stream) != 0


@pytest.mark.parametrize("n", [4097, 10000, 40000])
def test_global_bitonic_sort(n):
    L = _lib()
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 2 ** 62, (n,), generator=g, dtype=torch.int64).cuda()
    out = torch.empty_like(keys)
    rc = L.splat_selftest(2, keys.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.sort(keys.cpu())[0])


@pytest.mark.parametrize("n,item_table", [(70_000, True), (250_000, True), (600_000, True), (250_000, False)])
def test_lists_beyond_lds_multi_workgroup_sort(n, item_table, monkeypatch):
    """Per-tile lists of ~7 k / ~25 k / ~60 k entries (0, 2 and 3 merge passes of the multi-workgroup sort on 8 192-key runs,
    binning.hip L1-L4): tile ranges and sorted ids must be the C oracle's, bit for bit; the render still matches.  item_table = False:
    a caller that does not pass SplatState.long_items (ABI <= 6 layouts): the kernels find an item's tile by binary search."""
    import numpy as np
    from oracle import c_ref
    from splatam_amd import rasterizer as rz
    monkeypatch.setattr(rz, "LONG_ITEM_TABLE", item_table)
    from tests.util import scene
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera
    W, H = 96, 64
    cam, rv = scene(n, W, H, 90.0, seed=n)
    cs = Camera(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=cam.bg.cuda(), scale_modifier=1.0,
                viewmatrix=cam.viewmatrix.cuda(), projmatrix=cam.projmatrix.cuda(), sh_degree=0, campos=cam.campos.cuda(), prefiltered=False)
    empty = torch.empty(0, device="cuda")
    rz._longest_seen.clear()
    args = (cs, rv['means3D'].cuda(), rv['colors_precomp'].cuda(), rv['opacities'].cuda().reshape(-1), rv['scales'].cuda(), rv['rotations'].cuda(),
            empty, empty)
    col1, radii1, dep1, pk1 = rz.rasterize_forward(*args)           # one counter per tile (nothing known about this shape yet)
    rz.clear_geometry_cache()                                       # (the same inputs again: a full pass is wanted here, not the first call's lists)
    col, radii, dep, pk = rz.rasterize_forward(*args)               # the first call saw very long lists: 16 counters per tile now
    torch.cuda.synchronize()
    assert pk1.st.sub_bins == 1 and pk.st.sub_bins == rz.SUB_BINS_LONG
    assert torch.equal(pk1.tensors['point_list'][:pk1.num_rendered], pk.tensors['point_list'][:pk.num_rendered]) and torch.equal(col1, col)
    cr = c_ref.CRef()
    oc, orad, od = cr.forward(rv['means3D'].numpy(), rv['colors_precomp'].numpy(), rv['opacities'].numpy(), rv['scales'].numpy(),
                              rv['rotations'].numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(), cam.tanfovx, cam.tanfovy, W, H,
                              cam.bg.numpy())
    base = cr.ranges()
    longest = int(np.diff(base).max())
    assert longest > 4096 and 'keys_alt' in pk.tensors, longest
    assert ('long_items' in pk.tensors) == item_table
    assert pk.num_rendered == cr.num_rendered()
    assert (pk.tensors['tile_base'].cpu().numpy() == base).all()
    got, ref = pk.tensors['point_list'].cpu().numpy()[:pk.num_rendered], cr.point_list()
    assert (got == ref).all(), f"{int((got != ref).sum())} of {ref.size} list entries differ (longest list {longest})"
    assert np.quantile(np.abs(col.cpu().numpy() - oc), 0.9999) < 1e-4
