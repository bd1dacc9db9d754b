// binning.hip -- builds the per-tile, depth-ordered Gaussian lists.
//
// The reference does this with one global 64-bit radix sort over all
// (tile | depth) keys (5-6 passes over R instances in HBM).  Here every
// instance is scattered ONCE into its tile's bucket (bucket offsets come from
// the tile scan), and each bucket is then sorted on its own inside LDS by one
// workgroup: one read + one write of the instance stream in HBM in total.
// Keys are (float bits of depth << 32) | Gaussian id -- unique, so the order
// is exactly "ascending depth, ties by ascending index" (the order a stable
// sort of index-ordered emission produces), independent of scatter order.
#include "splat_device.h"

namespace splat {

constexpr int kBlock = 256;
constexpr int kSortLds = 4096;                 // keys sorted in LDS per tile (32 KiB); longer lists spill to HBM

// K3: one lane per Gaussian, one bucket slot per touched tile.  The slot comes from a
// returning atomic on the tile cursor (a fabric round trip of a few microseconds under
// load), so a lane keeps up to four of them in flight instead of chaining them.
__global__ __launch_bounds__(kBlock) void scatter_kernel(SplatGaussians g, SplatState st, int gx) {
    if ((long long)st.status[0] > st.capacity) return;      // lists would not fit: host re-sizes and retries
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.P) return;
    if (st.radii[i] <= 0) return;
    const uint2 r = reinterpret_cast<const uint2 *>(st.rect)[i];
    const int x0 = r.x & 0xffff, y0 = r.x >> 16, x1 = r.y & 0xffff, y1 = r.y >> 16;
    const int w = x1 - x0, nt = w * (y1 - y0);
    const uint64_t key = ((uint64_t)__float_as_uint(st.depth[i]) << 32) | (uint32_t)i;
    for (int t0 = 0; t0 < nt; t0 += 4) {
        unsigned slot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            if (t < nt) {
                const int yy = t / w, xx = t - yy * w;
                slot[u] = atomicAdd(&st.tile_cursor[(size_t)((y0 + yy) * gx + x0 + xx) * SPLAT_COUNTER_STRIDE], 1u);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + u < nt) st.keys[slot[u]] = key;
    }
}

// K4(+K5).  tile_base already holds the ranges.  Two kernels:
//  * short lists (n <= kSortWave, the normal case: ~200 entries at config B): ONE wave per tile,
//    8 KiB of LDS, the network's barriers degenerate to wave-local waits; every tile of the frame is
//    resident at once;
//  * long lists: one 256-thread workgroup per tile, 32 KiB of LDS, or in place in HBM beyond that.
constexpr int kSortWave = 1024;

// The host may know the longest list (status[2] of an earlier iteration).  The long-list kernel is skipped only when that
// hint leaves a 1.5x margin (the margin the bucket stride uses): lists grow a little from iteration to iteration.
__host__ __device__ inline bool long_sort_skipped(int max_list_hint) {
    return max_list_hint > 0 && max_list_hint + max_list_hint / 2 <= kSortWave;
}

__global__ __launch_bounds__(64) void tile_sort_wave_kernel(SplatState st) {
    __shared__ uint64_t s_keys[kSortWave];
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (st.tile_stride == 0) {
        if ((long long)st.status[0] > st.capacity) {
            if (blockIdx.x == 0 && threadIdx.x == 0) st.status[1] = 1;
            return;
        }
    }
    unsigned lo;
    int n;
    tile_range(st, tile, lo, n);
    if (n > kSortWave) {
        // long list: the workgroup-per-tile kernel sorts it -- unless the host's (possibly stale) hint said that no
        // list is that long and skipped that launch.  Then flag it so that the host re-runs, and publish the ids
        // UNSORTED so that the composite kernels of this (invalid) iteration still read valid Gaussian indices.
        if (long_sort_skipped(st.max_list_hint)) {
            if (tid == 0) atomicOr((unsigned *)&st.status[3], 1u);
            for (int i = tid; i < n; i += 64) st.point_list[lo + i] = (uint32_t)st.keys[lo + i];
        }
        return;
    }
    if (n == 0) return;
    const uint64_t *gk = st.keys + lo;
    for (int i = tid; i < n; i += 64) s_keys[i] = gk[i];
    __syncthreads();
    if (n > 1) bitonic_sort(s_keys, n, tid, 64);
    for (int i = tid; i < n; i += 64) st.point_list[lo + i] = (uint32_t)s_keys[i];
}

__global__ __launch_bounds__(kBlock) void tile_sort_block_kernel(SplatState st) {
    __shared__ uint64_t s_keys[kSortLds];
    if (st.tile_stride == 0 && (long long)st.status[0] > st.capacity) return;
    const int tile = blockIdx.x, tid = threadIdx.x;
    unsigned lo;
    int n;
    tile_range(st, tile, lo, n);
    if (n <= kSortWave) return;
    uint64_t *gk = st.keys + lo;
    if (n <= kSortLds) {
        for (int i = tid; i < n; i += kBlock) s_keys[i] = gk[i];
        __syncthreads();
        bitonic_sort(s_keys, n, tid, kBlock);
        for (int i = tid; i < n; i += kBlock) st.point_list[lo + i] = (uint32_t)s_keys[i];
    } else {
        // spilled tile list: same network run in place on the HBM bucket (L2-resident)
        bitonic_sort(gk, n, tid, kBlock);
        for (int i = tid; i < n; i += kBlock) st.point_list[lo + i] = (uint32_t)gk[i];
    }
}

hipError_t launch_bin_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, hipStream_t s, bool sort) {
    const int gx = (cam.image_width + kTile - 1) / kTile;
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    // bucketed lists were filled by the per-Gaussian kernel: only the per-tile sort remains
    if (g.P > 0 && st.tile_stride == 0) hipLaunchKernelGGL(scatter_kernel, dim3((g.P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, g, st, gx);
    if (T > 0 && sort) {
        hipLaunchKernelGGL(tile_sort_wave_kernel, dim3(T), dim3(64), 0, s, st);
        // the host may know the longest list (status[2]); only then can the long-list kernel be skipped
        if (!long_sort_skipped(st.max_list_hint))
            hipLaunchKernelGGL(tile_sort_block_kernel, dim3(T), dim3(kBlock), 0, s, st);
    }
    return hipGetLastError();
}

// ---- self tests (driven from tests/test_gpu_primitives.py) ------------------
// which = 0: wave_reduce4_packed on 4*64 floats per wave -> 64 floats per wave
// which = 1: bitonic_sort of n uint64 keys in LDS (n <= kSortLds), one workgroup
// which = 2: bitonic_sort of n uint64 keys in global memory, one workgroup
__global__ __launch_bounds__(64) void selftest_reduce_kernel(const float *in, float *out) {
    const int l = threadIdx.x, w = blockIdx.x;
    const float *p = in + (size_t)w * 256;
    out[(size_t)w * 64 + l] = wave_reduce4_packed(p[l], p[64 + l], p[128 + l], p[192 + l]);
}
__global__ __launch_bounds__(kBlock) void selftest_sort_lds_kernel(const uint64_t *in, uint64_t *out, int n) {
    __shared__ uint64_t s_keys[kSortLds];
    for (int i = threadIdx.x; i < n; i += kBlock) s_keys[i] = in[i];
    __syncthreads();
    if (n > 1) bitonic_sort(s_keys, n, threadIdx.x, kBlock);
    for (int i = threadIdx.x; i < n; i += kBlock) out[i] = s_keys[i];
}
__global__ __launch_bounds__(kBlock) void selftest_sort_global_kernel(uint64_t *keys, int n) {
    if (n > 1) bitonic_sort(keys, n, threadIdx.x, kBlock);
}

hipError_t launch_selftest(int which, const void *in, void *out, int n, hipStream_t s) {
    if (which == 0) {
        hipLaunchKernelGGL(selftest_reduce_kernel, dim3(n), dim3(64), 0, s, (const float *)in, (float *)out);
    } else if (which == 1) {
        if (n > kSortLds) return hipErrorInvalidValue;
        hipLaunchKernelGGL(selftest_sort_lds_kernel, dim3(1), dim3(kBlock), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
    } else if (which == 2) {
        hipError_t e = hipMemcpyAsync(out, in, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(selftest_sort_global_kernel, dim3(1), dim3(kBlock), 0, s, (uint64_t *)out, n);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace splat
