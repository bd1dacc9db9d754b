// binning.hip -- builds the per-tile, depth-ordered Gaussian lists.
//
// The reference does this with one global 64-bit radix sort over all
// (tile | depth) keys (5-6 passes over R instances in HBM).  Here every
// instance is scattered ONCE into its tile's bucket (bucket offsets come from
// the tile scan), and each bucket is then sorted on its own inside LDS by one
// workgroup -- an LDS radix sort on the depth key (radix_sort_lds / radix_sort_lds_private,
// splat_device.h) -- : one read + one write of the instance stream in HBM in total
// (lists beyond 4096 keys: 8192-key runs + merge-path passes, below).
// Keys are (float bits of depth << 32) | Gaussian id -- unique, so the order
// is exactly "ascending depth, ties by ascending index" (the order a stable
// sort of index-ordered emission produces), independent of scatter order.
#include "splat_device.h"

namespace splat {

constexpr int kBlock = 256;
constexpr int kSortLds = 4096;                 // keys sorted in LDS per tile (32 KiB); longer lists spill to HBM
// workgroup of the LDS radix sort of up to kSortLds keys: 16 waves, one workgroup per CU (82 KB of LDS).  Measured against 4 waves x two
// workgroups per CU (long_run_sort at 1 M / 5 M clustered: 117 / 333 us) and against 16 waves x two per CU with 16-bit histogram words and
// 64 registers (108 / 304 us): 92 / 334 us
constexpr int kSortWaves = 16, kSortBlock = 64 * kSortWaves;

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
                slot[u] = atomicAdd(&st.tile_cursor[sub_counter(st, (y0 + yy) * gx + x0 + xx, i)], 1u);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + u < nt) st.keys[slot[u]] = key;
    }
}

// K3 for exact lists known to be very long (dense_exact_lists, splat_device.h; the partner of preprocess_forward_dense_kernel: same
// partition, same sub-bin): the workgroup counts its instances per tile in LDS, reserves each non-empty tile's slots with ONE returning
// atomic on the cursor of (tile, its sub-bin) and hands them out through the same table (5 M clustered Gaussians: 860 us with one
// returning atomic per instance).
__global__ __launch_bounds__(kDenseThreads) void scatter_dense_kernel(SplatGaussians g, SplatState st, int gx, int T) {
    extern __shared__ unsigned s_tile[];
    if ((long long)st.status[0] > st.capacity) return;      // lists would not fit: host re-sizes and retries
    const int tid = threadIdx.x;
    for (int t = tid; t < T; t += kDenseThreads) s_tile[t] = 0u;
    __syncthreads();
    constexpr int K = kDenseGaussians / kDenseThreads;
    unsigned r0[K], r1[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = blockIdx.x * kDenseGaussians + k * kDenseThreads + tid;
        r0[k] = r1[k] = 0u;
        if (i < g.P && st.radii[i] > 0) {
            const uint2 r = reinterpret_cast<const uint2 *>(st.rect)[i];
            r0[k] = r.x;
            r1[k] = r.y;
            const int x0 = r.x & 0xffff, y0 = r.x >> 16, x1 = r.y & 0xffff, y1 = r.y >> 16;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) atomicAdd(&s_tile[y * gx + x], 1u);
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += kDenseThreads) {
        const unsigned cnt = s_tile[t];
        if (cnt) s_tile[t] = atomicAdd(&st.tile_cursor[sub_counter(st, t, (int)blockIdx.x)], cnt);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = blockIdx.x * kDenseGaussians + k * kDenseThreads + tid;
        const int x0 = r0[k] & 0xffff, y0 = r0[k] >> 16, x1 = r1[k] & 0xffff, y1 = r1[k] >> 16;
        if (y1 > y0 && x1 > x0) {
            const uint64_t key = ((uint64_t)__float_as_uint(st.depth[i]) << 32) | (uint32_t)i;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) st.keys[atomicAdd(&s_tile[y * gx + x], 1u)] = key;
        }
    }
}

// K4(+K5).  tile_base already holds the ranges.  Two kernels:
//  * short lists (n <= kSortWave, the normal case: ~200 entries at config B): ONE wave per tile,
//    17 KiB of LDS, barriers degenerate to wave-local waits (bitonic network up to 256 keys, radix
//    sort above);
//  * long lists (up to kSortLds): one 1024-thread workgroup per list (a fixed grid walks the tiles), 82 KiB
//    of LDS; beyond that the multi-workgroup kernels further down, or in place in HBM without their scratch.
constexpr int kSortWave = 1024;

// The host may know the longest list (status[2] of an earlier iteration).  The long-list kernel is skipped only when that
// hint leaves a 1.5x margin (the margin the bucket stride uses): lists grow a little from iteration to iteration.
__host__ __device__ inline bool long_sort_skipped(int max_list_hint) {
    return max_list_hint > 0 && max_list_hint + max_list_hint / 2 <= kSortWave;
}

__global__ __launch_bounds__(64) void tile_sort_wave_kernel(SplatState st) {
    __shared__ uint64_t s_keys[kSortWave], s_alt[kSortWave];
    __shared__ __attribute__((aligned(8))) unsigned s_hist[256];
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
    // one wave: the radix passes are a latency chain (nothing else runs on its SIMD lane of the workgroup); below a few hundred keys
    // the bitonic network's log^2 n short stages are faster (measured: 200 keys 7 vs 12 us, 1 024 keys: the radix sort wins)
    const uint64_t *sorted = s_keys;
    if (n <= 256) {
        if (n > 1) bitonic_sort(s_keys, n, tid, 64);
    } else {
        sorted = radix_sort_lds<1>(s_keys, s_alt, s_hist, n, tid);
    }
    for (int i = tid; i < n; i += 64) st.point_list[lo + i] = (uint32_t)sorted[i];
}

// `long_launched`: the host launches the multi-workgroup kernels below in this call (it decides from its list-length hint, which may
// be stale: a list beyond LDS that nobody is going to sort is flagged -- status[3], the host repeats the iteration -- and published
// unsorted, so that the composites of the invalid iteration still read valid Gaussian indices).
__global__ __launch_bounds__(kSortBlock) void tile_sort_block_kernel(SplatState st, bool long_launched, int T) {
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[kSortLds + 1024], s_alt[kSortLds + 1024];
    __shared__ __attribute__((aligned(8))) unsigned s_scratch[400];
    if (st.tile_stride == 0 && (long long)st.status[0] > st.capacity) return;
    const int tid = threadIdx.x;
    // (2 x (kSortLds + 1024) keys + scratch = ~83 KB of LDS: one workgroup per CU -- a fixed grid walks the tiles, most of which have
    //  nothing for this kernel)
    for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
        unsigned lo;
        int n;
        tile_range(st, tile, lo, n);
        if (n <= kSortWave) continue;
        uint64_t *gk = st.keys + lo;
        if (n <= kSortLds) {
            __syncthreads();                            // (the previous tile's result is still being read out of LDS)
            for (int i = tid; i < n; i += kSortBlock) s_keys[i] = gk[i];
            __syncthreads();
            const uint64_t *sorted = radix_sort_lds_private<kSortLds>(s_keys, s_alt, s_scratch, n, tid);
            for (int i = tid; i < n; i += kSortBlock) st.point_list[lo + i] = (uint32_t)sorted[i];
        } else if (!st.keys_alt || !st.long_base) {
            // list beyond LDS and no scratch from the caller: the bitonic network run in place on the HBM bucket by this ONE workgroup
            // (correct, slow: O(n log^2 n) barrier-separated stages); with scratch the multi-workgroup kernels below take the tile
            bitonic_sort(gk, n, tid, kSortBlock);
            for (int i = tid; i < n; i += kSortBlock) st.point_list[lo + i] = (uint32_t)gk[i];
        } else if (!long_launched) {
            if (tid == 0) atomicOr((unsigned *)&st.status[3], 1u);
            for (int i = tid; i < n; i += kSortBlock) st.point_list[lo + i] = (uint32_t)gk[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lists beyond LDS (BASELINE config 5: "per-tile Gaussian list spilling HBM"): many workgroups per tile.
//   L1 long_scan_kernel       item space: tile t owns ceil(n_t / 1024) items when n_t > 4096 (exclusive scan -> long_base)
//   L2 long_run_sort_kernel   every 8th item of a tile: one 8192-key run sorted in LDS (radix_sort_lds), in place
//   L3 long_merge_kernel      pass p merges neighbouring runs of 8192 * 2^p keys, keys <-> keys_alt ping-pong: merge-path partitions
//                             of 1024 outputs, the two input pieces staged in LDS, rank-merged there (keys are unique): any
//                             number of workgroups per tile, one streaming read + write per pass
//   L4 long_publish_kernel    ids of the sorted keys (from whichever buffer the tile's last pass wrote) -> point_list
// A tile of n keys takes ceil(log2(n / 8192)) passes (none up to 8192 keys); tiles that are done sit out the later passes.
// ---------------------------------------------------------------------------------------------------------------------
// keys per LDS-sorted run of a list beyond kSortLds: 8192 (128 KB of keys + 16 KB of histogram rows: one workgroup per CU, as with 4096 --
// the sort's cost per key is the same and every list needs one merge pass less; a list of up to 8192 keys needs none)
constexpr int kRun = 2 * kSortLds;
constexpr int kRunItems = kRun / 1024;  // work items per run
constexpr int kItemKeys = 1024;         // keys per workgroup item in the merge / publish kernels (4 per thread)

__host__ __device__ inline int long_passes(long long n) {      // merge passes a list of n keys needs
    int p = 0;
    while (((long long)kRun << p) < n) ++p;
    return p;
}

__global__ __launch_bounds__(1024) void long_scan_kernel(SplatState st, int T) {
    __shared__ unsigned wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool dead = st.tile_stride == 0 && (long long)st.status[0] > st.capacity;      // lists were published empty
    const int per = (T + 1023) / 1024;
    const int lo = tid * per, hi = min(T, lo + per);
    auto items_of = [&](int t) -> unsigned {
        unsigned l;
        int n;
        tile_range(st, t, l, n);
        return (!dead && n > kSortLds) ? (unsigned)((n + kItemKeys - 1) / kItemKeys) : 0u;
    };
    // (the counts of a thread's tiles are read once, all loads in flight together: this single workgroup is a chain of round trips)
    constexpr int kCache = 16;
    unsigned cached[kCache];
#pragma unroll
    for (int k = 0; k < kCache; ++k) cached[k] = (lo + k < hi) ? items_of(lo + k) : 0u;
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < kCache; ++k) sum += cached[k];
    for (int t = lo + kCache; t < hi; ++t) sum += items_of(t);
    unsigned incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned wave_off = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wave_off += wave_tot[w];
        total += wave_tot[w];
    }
    unsigned run = wave_off + incl - sum;
    const long long table = st.capacity / kItemKeys + T + 1;       // words of SplatState.long_items
    auto emit = [&](int t, unsigned cnt) {
        st.long_base[t] = run;
        if (st.long_items)
            for (unsigned k = 0; k < cnt; ++k)
                if ((long long)(run + k) < table) st.long_items[run + k] = (unsigned)t;
        run += cnt;
    };
#pragma unroll
    for (int k = 0; k < kCache; ++k)
        if (lo + k < hi) emit(lo + k, cached[k]);
    for (int t = lo + kCache; t < hi; ++t) emit(t, items_of(t));
    if (tid == 0) st.long_base[T] = total;
}

// item -> (tile, chunk): the last tile whose first item is <= item (tiles without items share their successor's base)
__device__ __forceinline__ bool long_item(const SplatState &st, int T, unsigned item, int &tile, int &chunk, unsigned &lo, int &n) {
    if (item >= st.long_base[T]) return false;
    int a = 0, b = T;                   // invariant: long_base[a] <= item < long_base[b]
    if (st.long_items) {
        a = (int)st.long_items[item];   // the scan's table (sum of ceil(n_t / 1024) <= capacity / 1024 + T: every item has an entry)
    } else {
        while (b - a > 1) {
            const int m = (a + b) >> 1;
            if (st.long_base[m] <= item) a = m; else b = m;
        }
    }
    tile = a;
    chunk = (int)(item - st.long_base[a]);
    tile_range(st, tile, lo, n);
    return true;
}

// (the long-list kernels walk the item space with a grid stride: the number of items is only known on the device, long_base[T])
__global__ __launch_bounds__(kSortBlock) void long_run_sort_kernel(SplatState st, int T) {
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[kRun + 1024], s_alt[kRun + 1024];
    __shared__ __attribute__((aligned(8))) unsigned s_scratch[400];
    const unsigned total = st.long_base[T];
    const int tid = threadIdx.x;
    // a run starts at every item whose chunk index within its tile is a multiple of kRunItems (an item is 1024 keys): the workgroups
    // walk the items kRunItems at a time and sort the runs that START among theirs
    for (unsigned item = blockIdx.x * (unsigned)kRunItems; item < total; item += gridDim.x * (unsigned)kRunItems) {
        for (unsigned it = item; it < item + (unsigned)kRunItems && it < total; ++it) {
            int tile, chunk, n;
            unsigned lo;
            if (!long_item(st, T, it, tile, chunk, lo, n)) break;
            if (chunk % kRunItems) continue;
            const int off = (chunk / kRunItems) * kRun, m = min(kRun, n - off);
            uint64_t *gk = st.keys + lo + off;
            __syncthreads();
            for (int i = tid; i < m; i += kSortBlock) s_keys[i] = gk[i];
            __syncthreads();
            const uint64_t *sorted = radix_sort_lds_private<kRun>(s_keys, s_alt, s_scratch, m, tid);
            for (int i = tid; i < m; i += kSortBlock) gk[i] = sorted[i];
        }
    }
}

// (merge path: `a` = how many of the first k keys of merge(A[0, la), B[0, lb)) come from A; keys are unique)
// One item = 1024 consecutive OUTPUT keys of a tile's list.  Two merge-path searches (uniform over the workgroup: ~13 dependent
// reads each, of two 8-byte keys) cut the pair of runs at the item's first and last output; the two input pieces -- 1024 keys together,
// contiguous -- are staged in LDS, and every key finds its place among the other piece's keys by a binary search IN LDS.  One
// streaming read and one write of the keys per pass (round 2 searched the partner run in L2 for every key: 13 scattered reads per key).
__global__ __launch_bounds__(kBlock) void long_merge_kernel(SplatState st, int T, int pass) {
    __shared__ uint64_t s_in[kItemKeys];
    const long long L = (long long)kRun << pass;               // run length going into this pass
    if (st.tile_stride == 0 && L >= (long long)st.status[2]) return;      // (exact lists: the scan knows the longest list of this iteration)
    const unsigned total = st.long_base[T];
    const int Li = (int)L, tid = threadIdx.x;
    for (unsigned item = blockIdx.x; item < total; item += gridDim.x) {
        int tile, chunk, n;
        unsigned lo;
        if (!long_item(st, T, item, tile, chunk, lo, n)) break;
        if ((long long)n <= L) continue;                           // this tile was finished by an earlier pass
        const uint64_t *src = ((pass & 1) ? st.keys_alt : st.keys) + lo;
        uint64_t *dst = ((pass & 1) ? st.keys : st.keys_alt) + lo;
        const int o0 = chunk * kItemKeys;                           // first output of this item (2 Li is a multiple of the item size)
        if (o0 >= n) continue;
        const int base = (o0 / (2 * Li)) * (2 * Li);                // the pair of runs this item's outputs belong to
        const int la = min(Li, n - base), lb = max(0, min(Li, n - base - Li));
        const uint64_t *A = src + base, *B = src + base + Li;
        const int k0 = o0 - base, k1 = min(k0 + kItemKeys, la + lb);
        if (lb == 0) {                                              // a run without a partner: copied through
            for (int k = k0 + tid; k < k1; k += kBlock) dst[base + k] = A[k];
            continue;
        }
        // the two searches in lockstep: their reads are in flight together (each step is a dependent round trip to L2)
        int lo0 = max(0, k0 - lb), hi0 = min(k0, la), lo1 = max(0, k1 - lb), hi1 = min(k1, la);
        while (lo0 < hi0 || lo1 < hi1) {
            const int m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
            const bool on0 = lo0 < hi0, on1 = lo1 < hi1;
            const uint64_t xa0 = on0 ? A[m0] : 0ull, xb0 = on0 ? B[k0 - m0 - 1] : 0ull;
            const uint64_t xa1 = on1 ? A[m1] : 0ull, xb1 = on1 ? B[k1 - m1 - 1] : 0ull;
            if (on0) { if (xa0 < xb0) lo0 = m0 + 1; else hi0 = m0; }
            if (on1) { if (xa1 < xb1) lo1 = m1 + 1; else hi1 = m1; }
        }
        const int a0 = lo0, a1 = lo1;
        const int b0 = k0 - a0, b1 = k1 - a1;
        const int na = a1 - a0, nb = b1 - b0;                       // na + nb = k1 - k0 <= 1024
        __syncthreads();                                            // (the previous item's searches are done with s_in)
        for (int i = tid; i < na; i += kBlock) s_in[i] = A[a0 + i];
        for (int i = tid; i < nb; i += kBlock) s_in[na + i] = B[b0 + i];
        __syncthreads();
        for (int i = tid; i < na + nb; i += kBlock) {
            const uint64_t key = s_in[i];
            const bool from_a = i < na;
            const uint64_t *other = from_a ? s_in + na : s_in;
            int lo_ = 0, hi_ = from_a ? nb : na;                    // number of keys of the other piece below `key`
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_) >> 1;
                if (other[mid] < key) lo_ = mid + 1; else hi_ = mid;
            }
            dst[base + k0 + (from_a ? i : i - na) + lo_] = key;
        }
    }
}

// `passes`: merge passes the host launched (from its bound on the list length).  A list that needed more (stale hint) is flagged
// -- status[3], the host repeats the iteration -- and published from the buffer its LAST LAUNCHED pass wrote: half merged, but valid ids.
__global__ __launch_bounds__(kBlock) void long_publish_kernel(SplatState st, int T, int passes) {
    const unsigned total = st.long_base[T];
    for (unsigned item = blockIdx.x; item < total; item += gridDim.x) {
        int tile, chunk, n;
        unsigned lo;
        if (!long_item(st, T, item, tile, chunk, lo, n)) break;
        int done = long_passes(n);
        if (done > passes) {
            done = passes;
            if (chunk == 0 && threadIdx.x == 0) atomicOr((unsigned *)&st.status[3], 1u);
        }
        const uint64_t *buf = ((done & 1) ? st.keys_alt : st.keys) + lo;
#pragma unroll
        for (int k = 0; k < kItemKeys / kBlock; ++k) {
            const int j = chunk * kItemKeys + k * kBlock + threadIdx.x;
            if (j < n) st.point_list[lo + j] = (uint32_t)buf[j];
        }
    }
}

hipError_t launch_bin_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, hipStream_t s, bool sort, bool counted_per_workgroup) {
    const int gx = (cam.image_width + kTile - 1) / kTile;
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    // group binning behind the reference API (launch_preprocess_forward filed the records): the forward composite builds the lists
    if (group_binning(st, cam.image_width, cam.image_height)) return hipSuccess;
    // bucketed lists were filled by the per-Gaussian kernel: only the per-tile sort remains
    if (g.P > 0 && st.tile_stride == 0) {
        if (counted_per_workgroup && dense_exact_lists(st, g.P, T))
            hipLaunchKernelGGL(scatter_dense_kernel, dim3((g.P + kDenseGaussians - 1) / kDenseGaussians), dim3(kDenseThreads),
                               sizeof(unsigned) * (size_t)T, s, g, st, gx, T);
        else
            hipLaunchKernelGGL(scatter_kernel, dim3((g.P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, g, st, gx);
    }
    if (T > 0 && sort) {
        hipLaunchKernelGGL(tile_sort_wave_kernel, dim3(T), dim3(64), 0, s, st);
        // the host may know the longest list (status[2]); only then can the long-list kernels be skipped
        if (!long_sort_skipped(st.max_list_hint)) {
            const long long hint = st.max_list_hint > 0 ? (long long)st.max_list_hint + st.max_list_hint / 2 : (long long)1 << 40;
            const bool long_path = st.keys_alt && st.long_base && hint > kSortLds && st.capacity > kSortLds;
            hipLaunchKernelGGL(tile_sort_block_kernel, dim3(T < 512 ? T : 512), dim3(kSortBlock), 0, s, st, long_path, T);
            if (long_path) {
                // no list is longer than `bound`: the hint, the capacity, or -- bucketed lists -- the bucket
                long long bound = st.capacity < hint ? st.capacity : hint;
                if (st.tile_stride > 0 && st.tile_stride < bound) bound = st.tile_stride;
                // the item count (sum of ceil(n_t / 1024) over the long tiles) lives on the device: fixed grids, grid-stride loops
                long long items = st.capacity / kItemKeys + T + 1;
                if (items > 16384) items = 16384;
                hipLaunchKernelGGL(long_scan_kernel, dim3(1), dim3(1024), 0, s, st, T);
                // (fixed grids: a workgroup that finds nothing to do still costs its launch -- 4096 of them with 69 KB of LDS, two per
                //  CU at a time, cost more than the sort itself.  Two run-sort workgroups per CU, eight of the light kernels)
                const long long run_wgs = (items + kRunItems - 1) / kRunItems < 512 ? (items + kRunItems - 1) / kRunItems : 512, item_wgs = items < 2048 ? items : 2048;
                hipLaunchKernelGGL(long_run_sort_kernel, dim3((unsigned)run_wgs), dim3(kSortBlock), 0, s, st, T);
                const int passes = long_passes(bound);
                for (int p = 0; p < passes; ++p)
                    hipLaunchKernelGGL(long_merge_kernel, dim3((unsigned)item_wgs), dim3(kBlock), 0, s, st, T, p);
                hipLaunchKernelGGL(long_publish_kernel, dim3((unsigned)item_wgs), dim3(kBlock), 0, s, st, T, passes);
            }
        }
    }
    return hipGetLastError();
}

// ---- self tests (driven from tests/test_gpu_primitives.py) ------------------
// which = 0: wave_reduce4_packed on 4*64 floats per wave -> 64 floats per wave
// which = 1: bitonic_sort of n uint64 keys in LDS (n <= kSortLds), one workgroup
// which = 2: bitonic_sort of n uint64 keys in global memory, one workgroup
// (the multi-workgroup long-list path is tested through splat_bin_forward: tests/test_gpu_primitives.py)
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
template <int NW, int MAXN>
__global__ __launch_bounds__(64 * NW) void selftest_radix_lds_kernel(const uint64_t *in, uint64_t *out, int n) {
    __shared__ uint64_t s_keys[MAXN], s_alt[MAXN];
    __shared__ __attribute__((aligned(8))) unsigned s_hist[NW * 256];
    for (int i = threadIdx.x; i < n; i += 64 * NW) s_keys[i] = in[i];
    __syncthreads();
    const uint64_t *sorted = radix_sort_lds<NW, MAXN>(s_keys, s_alt, s_hist, n, threadIdx.x);
    for (int i = threadIdx.x; i < n; i += 64 * NW) out[i] = sorted[i];
}
template <int MAXN>
__global__ __launch_bounds__(1024) void selftest_radix_private_kernel(const uint64_t *in, uint64_t *out, int n) {
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[MAXN + 1024], s_alt[MAXN + 1024];
    __shared__ __attribute__((aligned(8))) unsigned s_scratch[400];
    for (int i = threadIdx.x; i < n; i += 1024) s_keys[i] = in[i];
    __syncthreads();
    const uint64_t *sorted = radix_sort_lds_private<MAXN>(s_keys, s_alt, s_scratch, n, threadIdx.x);
    for (int i = threadIdx.x; i < n; i += 1024) out[i] = sorted[i];
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
    } else if (which == 7 || which == 8) {              // radix_sort_lds_private: 7 = n <= 8192 (the run sort), 8 = n <= 4096 (the block sort)
        if (n > (which == 7 ? kRun : kSortLds)) return hipErrorInvalidValue;
        if (which == 7) hipLaunchKernelGGL(selftest_radix_private_kernel<kRun>, dim3(1), dim3(1024), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
        else hipLaunchKernelGGL(selftest_radix_private_kernel<kSortLds>, dim3(1), dim3(1024), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
    } else if (which == 6) {                            // radix_sort_lds as the run sort uses it: 1024 threads, n <= 8192
        if (n > kRun) return hipErrorInvalidValue;
        hipLaunchKernelGGL((selftest_radix_lds_kernel<16, kRun>), dim3(1), dim3(1024), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
    } else if (which >= 3 && which <= 5) {              // radix_sort_lds: 3 = 256 threads, 5 = 1024 threads (n <= 4096), 4 = one wave (n <= 1024)
        if (n > (which == 4 ? kSortWave : kSortLds)) return hipErrorInvalidValue;
        if (which == 3) hipLaunchKernelGGL((selftest_radix_lds_kernel<4, 4096>), dim3(1), dim3(256), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
        else if (which == 5) hipLaunchKernelGGL((selftest_radix_lds_kernel<16, 4096>), dim3(1), dim3(1024), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
        else hipLaunchKernelGGL((selftest_radix_lds_kernel<1, 1024>), dim3(1), dim3(64), 0, s, (const uint64_t *)in, (uint64_t *)out, n);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace splat
