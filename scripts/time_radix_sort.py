#!/usr/bin/env python3
"""Time of ONE LDS sort of n keys by one workgroup (splat_selftest: 1 bitonic network on 256 threads, radix_sort_lds: 3 with 4 waves, 5 with 16 waves, 4 with one wave).
Developer tool (gpurun)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from splatam_amd import _capi  # noqa: E402
from test_gpu_primitives import _depth_id_keys  # noqa: E402

L = _capi.lib()
s = torch.cuda.current_stream().cuda_stream
for which, n, kind in ((1, 4096, "narrow"), (7, 8192, "narrow"), (7, 8192, "wide"), (8, 4096, "narrow"), (8, 2000, "narrow"), (6, 8192, "narrow"), (5, 4096, "narrow"), (5, 4096, "wide"), (5, 2000, "narrow"), (3, 4096, "narrow"), (3, 4096, "wide"), (3, 4096, "plane"), (3, 2000, "narrow"), (1, 1024, "narrow"),
                       (4, 1024, "narrow"), (4, 200, "narrow"), (1, 200, "narrow")):
    keys = torch.from_numpy(_depth_id_keys(n, kind, 1).view("int64")).cuda()
    out = torch.empty_like(keys)
    for _ in range(3):
        L.splat_selftest(which, keys.data_ptr(), out.data_ptr(), n, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.splat_selftest(which, keys.data_ptr(), out.data_ptr(), n, s)
    e1.record()
    torch.cuda.synchronize()
    print(f"selftest {which} ({'bitonic' if which == 1 else ('radix, private ranking' if which >= 7 else 'radix')}), n = {n}, {kind}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (one workgroup)", flush=True)
