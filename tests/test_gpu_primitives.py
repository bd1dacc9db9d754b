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


@pytest.mark.parametrize("n", [70_000, 250_000, 600_000])
def test_lists_beyond_lds_multi_workgroup_sort(n):
    """Per-tile lists of ~7 k / ~25 k / ~60 k entries (1, 3 and 4 merge passes of the multi-workgroup sort, binning.hip L1-L4):
    tile ranges and sorted ids must be the C oracle's, bit for bit; the render still matches."""
    import numpy as np
    from oracle import c_ref
    from splatam_amd import rasterizer as rz
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
    assert pk.num_rendered == cr.num_rendered()
    assert (pk.tensors['tile_base'].cpu().numpy() == base).all()
    got, ref = pk.tensors['point_list'].cpu().numpy()[:pk.num_rendered], cr.point_list()
    assert (got == ref).all(), f"{int((got != ref).sum())} of {ref.size} list entries differ (longest list {longest})"
    assert np.quantile(np.abs(col.cpu().numpy() - oc), 0.9999) < 1e-4
