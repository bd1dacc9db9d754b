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
