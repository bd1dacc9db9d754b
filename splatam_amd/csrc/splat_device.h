// splat_device.h -- device-side helpers shared by the gfx950 kernels: camera
// constant loading, wave64 cross-lane primitives (DPP row reductions,
// v_permlane{16,32}_swap packed reductions), and the launch-side declarations.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/splat_hip.h"
#include "splat_math.h"

namespace splat {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// launchers (one per .hip file); every one returns hipGetLastError()
// ---------------------------------------------------------------------------
hipError_t launch_preprocess_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, hipStream_t s);
hipError_t launch_tile_count_reset(SplatState &st, int T, hipStream_t s);
hipError_t launch_tile_scan(SplatState &st, int T, hipStream_t s);
// counted_per_workgroup: the instance counts came from preprocess_forward_dense_kernel (launch_preprocess_forward under dense_exact_lists):
// the scatter must use the same (workgroup, sub-bin) partition.  The fused iteration counts per Gaussian in F1: false.
hipError_t launch_bin_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, hipStream_t s, bool sort = true,
                              bool counted_per_workgroup = false);

// Exact lists known to be very long (SplatState.sub_bins > 1: the host has seen lists beyond 2048 entries -- BASELINE config E clustered):
// K1's count and K3's scatter take one atomic per (WORKGROUP, tile) instead of one per instance -- a 1024-thread workgroup over
// kDenseGaussians Gaussians counts / ranks them per tile in an LDS table of the whole frame (4 bytes x tiles).  Both kernels must agree
// on the partition (workgroup b: Gaussians [b kDenseGaussians, (b + 1) kDenseGaussians), sub-bin b mod S): one predicate for both.
constexpr int kDenseThreads = 1024, kDenseGaussians = 4096, kDenseMaxTilesLds = 12 * 1024;
inline bool dense_exact_lists(const SplatState &st, int P, int T) {
    return st.tile_stride == 0 && st.sub_bins > 1 && P >= 8 * kDenseGaussians && T <= kDenseMaxTilesLds;
}
// lists known (host hint, possibly stale: then flagged) to be short are sorted by the composite kernel itself
inline bool lists_sorted_by_composite(const SplatState &st) { return st.max_list_hint > 0 && st.max_list_hint + st.max_list_hint / 4 <= 1024; }
// group binning (SplatState.group_count) needs bucketed lists that the composite sorts itself, and one LDS counter per group
inline bool group_binning(const SplatState &st, int W, int H) {
    const int gx = (W + SPLAT_TILE - 1) / SPLAT_TILE, gy = (H + SPLAT_TILE - 1) / SPLAT_TILE;
    const int groups = ((gx + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES) * ((gy + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES);
    return st.group_stride > 0 && st.group_count && st.group_recs && st.tile_stride > 0 && lists_sorted_by_composite(st) && groups <= 8192;
}
hipError_t launch_render_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st,
                                 float *out_color, float *out_depth, hipStream_t s);
hipError_t launch_render_backward(const SplatCamera &cam, const SplatGaussians &g, const SplatState &st,
                                  SplatGrads &gr, hipStream_t s);
hipError_t launch_preprocess_backward(const SplatCamera &cam, const SplatGaussians &g, const SplatState &st,
                                      SplatGrads &gr, hipStream_t s);
hipError_t launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s);
hipError_t launch_same_geometry(int P, const float *oa, const float *ob, const float *sa, const float *sb, const float *ra, const float *rb,
                                int *differ, hipStream_t s);
// Optional epilogue of the 6-channel forward composite for the fused TRACKING iteration: the masked L1 losses of get_loss
// (/root/reference/scripts/splatam.py:256-286, tracking=True, no outlier rejection) and their gradient planes, formed from
// the pixel's six channels while they are still in registers (saves the separate loss kernel and its re-read of six planes).
struct TrackLossEpilogue {
    const float *im;        // [3][H][W] curr_data['im']
    const float *depth;     // [1][H][W] curr_data['depth']
    float *dL_dout6;        // [6][H][W]: planes 0..3 are written
    double *sums;           // [SPLAT_ITER_SUM_COPIES][SPLAT_ITER_SUMS]: [0] += masked depth L1, [1] += (masked) image L1
    float sil_thres, w_im, w_depth;
    int use_sil_for_loss, use_l1;
};
// *ep_done tells the caller whether the epilogue ran (generation-3 kernels) or the separate loss kernel is still needed.
hipError_t launch_render_forward_feat8(const SplatCamera &cam, const float *feat8, SplatState &st, float *out6, bool sort_in_kernel,
                                       hipStream_t s, const TrackLossEpilogue *ep = nullptr, bool *ep_done = nullptr);
hipError_t launch_render_track_fused(const SplatCamera &cam, const float *feat8, SplatState &st, float *out6, float *accum,
                                     const TrackLossEpilogue &ep, bool keep_planes, hipStream_t s, bool full_sums = false);
hipError_t launch_render_backward_feat8(const SplatCamera &cam, const float *feat8, const SplatState &st, const float *dL_dout6,
                                        float *accum, int P, bool zero_accum, bool rgb_sums, hipStream_t s, bool opacity_sum = true);
hipError_t launch_iter_loss_backward(const SplatCamera &cam, const SplatMap &map, const SplatFrameData &frame,
                                     const SplatLossConfig &cfg, SplatIterWorkspace &ws, hipStream_t s, const SplatPoseAdam *pose_adam = nullptr,
                                     const SplatAdamMap *map_adam = nullptr);
hipError_t launch_iter_finish(const SplatCamera &cam, const SplatMap &map, const SplatFrameData &frame, const SplatLossConfig &cfg,
                              SplatIterWorkspace &ws, hipStream_t s, const SplatPoseAdam *pose_adam);
hipError_t launch_iter_adam_map(const SplatMap &map, const SplatAdamMap &opt, hipStream_t s);
hipError_t launch_iter_fold_sums(double *sums, hipStream_t s);
hipError_t launch_iter_adam_pose(const SplatMap &map, int time_idx, const float *d_cam, float *state, float beta1, float beta2,
                                 float eps, float bc2_sqrt, float ss_rot, float ss_trans, hipStream_t s);
hipError_t launch_iter_render(const SplatCamera &cam, const SplatMap &map, const SplatFrameData &frame, SplatIterWorkspace &ws,
                              hipStream_t s);
hipError_t launch_map_add(const SplatMapStore &st, const SplatAddArgs &a, hipStream_t s);
hipError_t launch_map_prune(const SplatMapStore &st, const SplatPruneArgs &a, hipStream_t s);
hipError_t launch_map_densify_select(const SplatMapStore &st, const SplatDensifyArgs &a, hipStream_t s);
hipError_t launch_map_duplicate(const SplatMapStore &st, const SplatDensifyArgs &a, hipStream_t s);
hipError_t launch_iter_means2d_accumulate(const SplatCamera &cam, const SplatMap &map, SplatIterWorkspace &ws, float *accum, float *denom,
                                          float *means2D_grad, hipStream_t s);
hipError_t launch_render_backward_rgb_only(const SplatCamera &cam, const float *feat8, const SplatState &st, const float *dL_dout6,
                                           float *accum, int P, hipStream_t s);
size_t map_scratch_words(long long n);
int map_row_floats(const SplatMapStore &st);
extern int g_debug_skip_count;
extern int g_debug_k7_bits;
extern long long *g_debug_stamps;
hipError_t launch_selftest(int which, const void *in, void *out, int n, hipStream_t s);

#if defined(__HIPCC__)

__device__ __forceinline__ void load_cam(CamConst &c, const SplatCamera &cam) {
    // uniform addresses -> scalar loads into SGPRs
    init_cam(c, cam.viewmatrix, cam.projmatrix, cam.image_width, cam.image_height, cam.tanfovx, cam.tanfovy,
             cam.scale_modifier);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// status word k (1: a bucket overflowed, 3: a list beyond the composite's sort) is raised, and with it the caller's pinned host word
__device__ __forceinline__ void raise_status(const SplatState &st, int k) {
    st.status[k] = 1;
    if (st.status_host) *st.status_host = 1;
}

// counter of (tile, sub-bin of Gaussian i): SplatState.sub_bins counters per tile, one 128-byte line each
__device__ __forceinline__ size_t sub_counter(const SplatState &st, int tile, int i) {
    const int S = st.sub_bins > 1 ? st.sub_bins : 1;
    return ((size_t)tile * S + (size_t)(i & (S - 1))) * SPLAT_COUNTER_STRIDE;
}

// [lo, lo + n) of tile `tile` in keys / point_list: compact (exact path) or bucketed (SplatState.tile_stride)
__device__ __forceinline__ void tile_range(const SplatState &st, int tile, unsigned &lo, int &n) {
    if (st.tile_stride > 0) {
        lo = (unsigned)tile * (unsigned)st.tile_stride;
        n = min((int)st.tile_count[(size_t)tile * SPLAT_COUNTER_STRIDE], st.tile_stride);
    } else {
        lo = st.tile_base[tile];
        n = (int)(st.tile_base[tile + 1] - lo);
    }
}

// ---- DPP helpers -----------------------------------------------------------
// dpp_ctrl encodings: quad_perm = 0x00..0xFF, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, false);
}

// After this every lane of a 16-lane row holds the sum over its row (4 v_add_f32_dpp).
__device__ __forceinline__ float row16_allreduce_add(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);   // row_half_mirror
    v += dpp_f32<0x140>(v);   // row_mirror
    return v;
}

// Packed wave64 reduction of four values: returns u such that every lane of
// row r (lanes 16r..16r+15) holds the wave-wide sum of value row_value(r),
// row_value = {0, 2, 1, 3}.  3 permlane swaps + 3 adds + 4 DPP adds for four
// values instead of 4 x 7 for four separate butterflies.
__device__ __forceinline__ float wave_reduce4_packed(float v0, float v1, float v2, float v3) {
#if defined(SPLAT_SAFE_REDUCE)
    // reference formulation on ds_bpermute butterflies (used by the self-test)
    float v[4] = {v0, v1, v2, v3};
    for (int i = 0; i < 4; ++i)
        for (int m = 32; m >= 1; m >>= 1) v[i] += __shfl_xor(v[i], m, 64);
    const int row = lane_id() >> 4;
    return row == 0 ? v[0] : (row == 1 ? v[2] : (row == 2 ? v[1] : v[3]));
#else
    // lanes 0-31 <- partial sums of v0, lanes 32-63 <- partial sums of v1
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
    const float s01 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v2), __float_as_uint(v3), false, false);
    const float s23 = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    // rows: s01 = [v0 v0 v1 v1], s23 = [v2 v2 v3 v3]  ->  [v0 v2 v1 v3]
    auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
    const float u = __uint_as_float(c[0]) + __uint_as_float(c[1]);
    return row16_allreduce_add(u);
#endif
}
// value index held by row r after wave_reduce4_packed: {0, 2, 1, 3}
__device__ __forceinline__ int row_value(int row) { return ((row & 1) << 1) | (row >> 1); }

// Bitonic network with ascending-only compare-exchanges ("flip" form), so that
// virtual +inf padding above n never has to be stored: an exchange whose upper
// index is >= n is skipped.  All block sizes are powers of two: index arithmetic is shifts and masks.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr keys, const int n, const int tid, const int nthreads) {
    int lN2 = 1;
    while ((1 << lN2) < n) ++lN2;
    const int half_n = 1 << (lN2 - 1);
    for (int lk = 1; lk <= lN2; ++lk) {                 // k = 2^lk
        const int lhk = lk - 1, hk = 1 << lhk, k = 1 << lk;
        for (int i = tid; i < half_n; i += nthreads) {
            const int blk = i >> lhk, off = i & (hk - 1);
            const int a = (blk << lk) + off, b = (blk << lk) + k - 1 - off;
            if (b < n) {
                const uint64_t ka = keys[a], kb = keys[b];
                if (ka > kb) { keys[a] = kb; keys[b] = ka; }
            }
        }
        __syncthreads();
        for (int lj = lhk - 1; lj >= 0; --lj) {         // j = 2^lj
            const int j = 1 << lj;
            for (int i = tid; i < half_n; i += nthreads) {
                const int blk = i >> lj, off = i & (j - 1);
                const int a = (blk << (lj + 1)) + off, b = a + j;
                if (b < n) {
                    const uint64_t ka = keys[a], kb = keys[b];
                    if (ka > kb) { keys[a] = kb; keys[b] = ka; }
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS radix sort on the depth key (the "per-tile radix sort on depth key" of the path's design): n <= NW * 1024 unique 64-bit keys
// (float bits of depth << 32 | Gaussian id) ascending, by NW waves (workgroup of 64 * NW threads; n <= MAXN).
//   * least-significant-digit passes over the 32 DEPTH bits, 8 bits per pass; a pass whose digit is the same in every key (the
//     exponent byte of a tile's depth range, usually) is skipped;
//   * wave-ballot ranking: wave w owns the contiguous segment w of the keys and walks it 64 keys per round.  match_any (8 ballots)
//     gives every lane the set of lanes of its round with the same digit: its rank among them, and ONE lane per distinct digit
//     counts / reserves slots for all of them in the wave's own histogram row -- no LDS atomics, no same-address serialisation;
//   * offsets: exclusive scan over (digit, wave) -- stable: segments in wave order, rounds in order, lanes in order;
//   * equal depths (rare) come out in scatter order; one last pass moves a key by (#smaller ids right of it - #larger ids left of
//     it) within its run of equal depth: "ascending depth, ties by ascending id", the order the bitonic network produces.
// buf_a holds the keys on entry; buf_b is scratch of the same size; hist: NW * 256 counters (8-byte aligned, >= 16 NW bytes).  Returns the buffer that holds the
// result.  Barriers inside: every thread of the workgroup must call it (n is uniform).
// Measured, one workgroup alone on the GPU, 4 096 keys of a tile's depth range (3 passes): 256 threads 28 us (the bitonic network 73 us),
// see scripts/time_radix_sort.py; the list kernels use 1 024 threads.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long match_any8(unsigned d, bool valid) {
    unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
        peers &= bit ? m : ~m;
    }
    return valid ? peers : 0ull;
}

template <int NW, int MAXN = NW * 1024, typename HistT = unsigned>
__device__ __forceinline__ uint64_t *radix_sort_lds(uint64_t *buf_a, uint64_t *buf_b, HistT *hist, const int n, const int tid) {
    static_assert(MAXN <= 65535 || sizeof(HistT) == 4, "a histogram word holds an offset into the keys");
    static_assert(NW == 1 || NW == 4 || NW == 16, "one wave per list, or a workgroup of 256 / 1024 threads");
    static_assert(MAXN % (NW * 64) == 0, "whole rounds");
    // 64-key rounds per wave.  The rounds of a wave are a chain of dependent steps (eight ballots -- vector compare to scalar mask and
    // back, ~30 cycles each -- then an LDS read-modify-write): measured 1 400 cycles per round with ONE wave per SIMD.  Sixteen waves
    // on 4 096 keys (4 rounds each, 4 waves per SIMD) hide that latency behind each other.
    constexpr int kRounds = MAXN / (NW * 64);
    constexpr int kTieRun = 32;                         // longest run of equal depths the neighbour pass orders
    __shared__ unsigned s_x[8];                         // wave totals of the digit scan / flags
    const int lane = tid & 63, wave = tid >> 6;
    const int seg = ((n + NW - 1) / NW + 63) & ~63;     // keys per wave segment (whole rounds)
    const int seg_lo = wave * seg, seg_hi = min(n, seg_lo + seg);
    // which bit positions differ at all
    uint64_t all_or = 0ull, all_and = ~0ull;
    for (int i = tid; i < n; i += 64 * NW) {
        const uint64_t k = buf_a[i];
        all_or |= k;
        all_and &= k;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        all_or |= (uint64_t)__shfl_xor((long long)all_or, m, 64);
        all_and &= (uint64_t)__shfl_xor((long long)all_and, m, 64);
    }
    uint64_t *xch = reinterpret_cast<uint64_t *>(hist);
    if (lane == 0) { xch[2 * wave] = all_or; xch[2 * wave + 1] = all_and; }
    __syncthreads();
    uint64_t diff;
    {
        uint64_t o = 0ull, a = ~0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) { o |= xch[2 * w]; a &= xch[2 * w + 1]; }
        diff = o ^ a;
    }
    if (tid == 0) s_x[4] = 0u;
    __syncthreads();
    uint64_t *src = buf_a, *dst = buf_b;
    HistT *row = hist + wave * 256;
    // attempt 0: the four depth bytes, then the neighbour pass.  Only when that finds a run of > kTieRun equal depths (a fronto-parallel
    // plane of Gaussians): attempt 1, all eight bytes, least significant first -- the id bytes, then the depth bytes again.
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (int byte = attempt == 0 ? 4 : 0; byte < 8; ++byte) {
            if (((diff >> (8 * byte)) & 0xFFull) == 0ull) continue;      // (uniform) every key holds the same digit here
            const int shift = 8 * byte;
            for (int k = tid; k < NW * 256; k += 64 * NW) hist[k] = (HistT)0;
            __syncthreads();
            unsigned info[kRounds];                     // rank | leader lane << 8 | group size << 16 | is-leader << 24
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                info[r] = 0u;
                if (seg_lo + r * 64 < seg_hi) {          // (uniform per wave)
                    const int i = seg_lo + r * 64 + lane;
                    const bool valid = i < seg_hi;
                    const unsigned d = valid ? (unsigned)(src[i] >> shift) & 0xFFu : 0u;
                    const unsigned long long peers = match_any8(d, valid);
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
                    const unsigned cnt = (unsigned)__builtin_popcountll(peers);
                    const unsigned leader = valid ? (unsigned)__builtin_ctzll(peers) : 0u;
                    const bool is_leader = valid && leader == (unsigned)lane;
                    if (is_leader) row[d] = (HistT)(row[d] + cnt);       // leaders of one round hold distinct digits; rounds of a wave are in program order
                    info[r] = rank | (leader << 8) | (cnt << 16) | ((is_leader ? 1u : 0u) << 24);
                }
            }
            __syncthreads();
            // exclusive scan over (digit, wave)
            if constexpr (NW >= 4) {                    // thread t < 256 owns digit t
                unsigned tot = 0u, incl = 0u;
                if (tid < 256) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) tot += hist[w * 256 + tid];
                    incl = tot;
                    for (int dd = 1; dd < 64; dd <<= 1) {
                        const unsigned o = (unsigned)__shfl_up((int)incl, dd, 64);
                        if (lane >= dd) incl += o;
                    }
                    if (lane == 63) s_x[wave] = incl;
                }
                __syncthreads();
                if (tid < 256) {
                    unsigned base = incl - tot;
                    for (int w = 0; w < wave; ++w) base += s_x[w];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const unsigned c = hist[w * 256 + tid];
                        hist[w * 256 + tid] = (HistT)base;
                        base += c;
                    }
                }
            } else {                                    // one wave: the four 64-digit quarters in turn
                unsigned run = 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned cq = hist[q * 64 + lane];
                    unsigned incl = cq;
                    for (int dd = 1; dd < 64; dd <<= 1) {
                        const unsigned o = (unsigned)__shfl_up((int)incl, dd, 64);
                        if (lane >= dd) incl += o;
                    }
                    hist[q * 64 + lane] = (HistT)(run + incl - cq);
                    run += (unsigned)__shfl((int)incl, 63, 64);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                if (seg_lo + r * 64 < seg_hi) {
                    const int i = seg_lo + r * 64 + lane;
                    const bool valid = i < seg_hi;
                    const uint64_t key = valid ? src[i] : 0ull;
                    const unsigned d = (unsigned)(key >> shift) & 0xFFu;
                    unsigned base = 0u;
                    if ((info[r] >> 24) & 1u) {
                        base = row[d];
                        row[d] = (HistT)(base + ((info[r] >> 16) & 0xFFu));
                    }
                    base = (unsigned)__shfl((int)base, (int)((info[r] >> 8) & 0xFFu), 64);
                    if (valid) dst[base + (info[r] & 0xFFu)] = key;
                }
            }
            __syncthreads();
            uint64_t *t = src; src = dst; dst = t;
        }
        if (attempt == 1) return src;                    // sorted on all 64 bits
        // a run of more than kTieRun equal depths?
        bool long_run = false;
        for (int i = tid; i < n; i += 64 * NW)
            if (i >= kTieRun && (unsigned)(src[i] >> 32) == (unsigned)(src[i - kTieRun] >> 32)) long_run = true;
        if (long_run) s_x[4] = 1u;
        __syncthreads();
        if (s_x[4] == 0u) break;
    }
    // ties of equal depth (runs of at most kTieRun) -> ascending id
    for (int i = tid; i < n; i += 64 * NW) {
        const uint64_t key = src[i];
        const unsigned depth = (unsigned)(key >> 32);
        int pos = i;
        for (int j = i - 1; j >= 0; --j) {
            const uint64_t o = src[j];
            if ((unsigned)(o >> 32) != depth) break;
            if (o > key) --pos;
        }
        for (int j = i + 1; j < n; ++j) {
            const uint64_t o = src[j];
            if ((unsigned)(o >> 32) != depth) break;
            if (o < key) ++pos;
        }
        dst[pos] = key;
    }
    __syncthreads();
    return dst;
}

// wave64 inclusive prefix sum in 6 DPP adds (row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 -- the sequence LLVM's atomic optimizer emits for gfx9)
__device__ __forceinline__ unsigned wave_scan_incl_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same sort with THREAD-PRIVATE ranking (the run sort of lists beyond LDS, the block sort): 1 024 threads, thread t owns the
// KPT = MAXN / 1024 consecutive keys [KPT t, KPT t + KPT).  4-bit digits; a thread counts its keys per digit in two registers (16 x
// 4-bit counters, which also give a key its rank among the thread's earlier keys of that digit), the counters are widened to eight
// registers of two 16-bit fields and prefix-summed over the wave with DPP (48 adds for 16 digits x 64 lanes), wave totals are scanned in
// (digit, wave) order by one wave, and a key's place is base(digit, wave) + the lanes below + its rank in the thread: no ballots, no
// vector -> scalar -> vector round trips (the wave-ballot ranking above costs ~430 vector-pipe cycles per 64 keys and pass: its 8-bit
// passes are fewer, but the run sort was bound by exactly that).  The lane prefixes live in eight registers and a key's digit picks one of
// them at run time: they are parked in the destination buffer (own column [j][tid], read back by the same thread with a computed
// address) until the scatter overwrites it.  Stable; nibbles that are the same in every key are skipped; ties of equal depth and the
// all-bits second attempt as in radix_sort_lds.  buf_a / buf_b: MAXN + 1024 keys each (padded layout, below); scratch: 400 words; the
// result is compact.  Every thread of the workgroup must call it.
// ---------------------------------------------------------------------------------------------------------------------
template <int MAXN>
__device__ __forceinline__ uint64_t *radix_sort_lds_private(uint64_t *buf_a, uint64_t *buf_b, unsigned *scratch, const int n, const int tid) {
    constexpr int NT = 1024, KPT = MAXN / NT, kTieRun = 32;
    static_assert(MAXN % NT == 0 && KPT >= 1 && KPT <= 15, "a 4-bit counter holds a thread's keys of one digit");
    static_assert((KPT & (KPT - 1)) == 0, "the padded index is i + i / KPT");
    // A thread reads its KPT consecutive keys: lane stride 2 KPT words, i.e. the 64 lanes of a read would share 64 / (2 KPT) banks (the
    // first version: half of the sort's LDS cycles were bank conflicts).  The keys of every pass but the first live PADDED, one empty slot
    // after every KPT keys (lane stride 2 KPT + 2 words: all 64 banks); the buffers hold MAXN + NT keys.
    auto phys = [](int i) { return i + i / KPT; };
    static_assert(MAXN * 8 >= 8 * NT * 4, "the destination buffer parks eight prefix words per thread");
    const int lane = tid & 63, wave = tid >> 6;
    unsigned *wave_tot = scratch;               // [16 waves][8 registers of two 16-bit digit totals]
    unsigned *gbase = scratch + 128;            // [16 waves][16 digits]: first output slot of (digit, wave)
    unsigned *s_flag = scratch + 384;
    uint64_t all_or = 0ull, all_and = ~0ull;
    for (int i = tid; i < n; i += NT) {
        const uint64_t k = buf_a[i];
        all_or |= k;
        all_and &= k;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        all_or |= (uint64_t)__shfl_xor((long long)all_or, m, 64);
        all_and &= (uint64_t)__shfl_xor((long long)all_and, m, 64);
    }
    uint64_t *xch = reinterpret_cast<uint64_t *>(scratch);
    if (lane == 0) { xch[2 * wave] = all_or; xch[2 * wave + 1] = all_and; }
    if (tid == 0) *s_flag = 0u;
    __syncthreads();
    uint64_t diff;
    {
        uint64_t o = 0ull, a = ~0ull;
#pragma unroll
        for (int w = 0; w < 16; ++w) { o |= xch[2 * w]; a &= xch[2 * w + 1]; }
        diff = o ^ a;
    }
    __syncthreads();
    uint64_t *src = buf_a, *dst = buf_b;
    bool padded = false;                        // layout of src (the caller's keys are compact)
    const int first = tid * KPT;
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (int nib = attempt == 0 ? 8 : 0; nib < 16; ++nib) {
            if (((diff >> (4 * nib)) & 0xFull) == 0ull) continue;        // (uniform)
            const int shift = 4 * nib;
            uint64_t key[KPT];
            unsigned dig[KPT], lrank[KPT];
            unsigned c_lo = 0u, c_hi = 0u;              // 4-bit counters of digits 0..7 / 8..15
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const bool valid = first + k < n;
                key[k] = valid ? src[padded ? first + tid + k : first + k] : 0ull;        // phys(first + k) = first + tid + k
                const unsigned d = (unsigned)(key[k] >> shift) & 15u, sh = (d & 7u) * 4u;
                const bool hi = d >= 8u;
                dig[k] = d;
                lrank[k] = ((hi ? c_hi : c_lo) >> sh) & 15u;
                const unsigned inc = valid ? (1u << sh) : 0u;
                c_lo += hi ? 0u : inc;
                c_hi += hi ? inc : 0u;
            }
            unsigned excl[8];
            unsigned *park = reinterpret_cast<unsigned *>(dst);          // [8][NT]: this thread's lane prefixes
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned c = j < 4 ? c_lo : c_hi;
                const unsigned e = ((c >> (8 * (j & 3))) & 15u) | (((c >> (8 * (j & 3) + 4)) & 15u) << 16);
                const unsigned incl = wave_scan_incl_u32(e);             // (<= 64 * 15 per field: no carry into the upper field)
                excl[j] = incl - e;
                park[j * NT + tid] = excl[j];
                if (lane == 63) wave_tot[wave * 8 + j] = incl;
            }
            __syncthreads();
            if (wave == 0) {
                // 256 totals in (digit, wave) order, four per lane: lane l holds t = 4 l .. 4 l + 3, digit t >> 4, wave t & 15
                unsigned v[4], sum = 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = 4 * lane + q, d = t >> 4, w = t & 15;
                    const unsigned word = wave_tot[w * 8 + (d >> 1)];
                    v[q] = (d & 1) ? word >> 16 : word & 0xFFFFu;
                    sum += v[q];
                }
                unsigned run = wave_scan_incl_u32(sum) - sum;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = 4 * lane + q, d = t >> 4, w = t & 15;
                    gbase[w * 16 + d] = run;
                    run += v[q];
                }
            }
            __syncthreads();
            unsigned pos[KPT];
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const unsigned d = dig[k];
                const unsigned word = park[(d >> 1) * NT + tid];
                pos[k] = gbase[wave * 16 + d] + ((d & 1u) ? word >> 16 : word & 0xFFFFu) + lrank[k];
            }
            __syncthreads();                            // every thread has read its parked prefixes: the scatter may overwrite them
#pragma unroll
            for (int k = 0; k < KPT; ++k)
                if (first + k < n) dst[phys((int)pos[k])] = key[k];
            __syncthreads();
            uint64_t *t = src; src = dst; dst = t;
            padded = true;
        }
        auto at = [&](int i) { return src[padded ? phys(i) : i]; };
        if (attempt == 1) {                              // sorted on all 64 bits: compact it
            for (int i = tid; i < n; i += NT) dst[i] = at(i);
            __syncthreads();
            return dst;
        }
        bool long_run = false;
        for (int i = tid; i < n; i += NT)
            if (i >= kTieRun && (unsigned)(at(i) >> 32) == (unsigned)(at(i - kTieRun) >> 32)) long_run = true;
        if (long_run) *s_flag = 1u;
        __syncthreads();
        if (*s_flag == 0u) break;
    }
    auto at = [&](int i) { return src[padded ? phys(i) : i]; };
    for (int i = tid; i < n; i += NT) {                 // ties of equal depth (runs of at most kTieRun) -> ascending id; compact output
        const uint64_t key = at(i);
        const unsigned depth = (unsigned)(key >> 32);
        int pos = i;
        for (int j = i - 1; j >= 0; --j) {
            const uint64_t o = at(j);
            if ((unsigned)(o >> 32) != depth) break;
            if (o > key) --pos;
        }
        for (int j = i + 1; j < n; ++j) {
            const uint64_t o = at(j);
            if ((unsigned)(o >> 32) != depth) break;
            if (o < key) ++pos;
        }
        dst[pos] = key;
    }
    __syncthreads();
    return dst;
}

// ---------------------------------------------------------------------------------------------------------------------
// GROUP BINNING (SplatState.group_count / group_recs; the per-Gaussian kernels of both paths: fused.hip F1, preprocess.hip K1): the
// kGroupBlock Gaussians of a workgroup count their records per GROUP of 2 x 2 tiles in an LDS histogram (one counter per group of the
// frame: dynamic LDS, 4 bytes x groups), the workgroup takes ONE returning global atomic per non-empty group, and a record's slot is the
// group's base + its LDS rank: ~1.0 global atomics per Gaussian in ANY row order (1.56 records per Gaussian over 836 groups at workload B)
// instead of 2.36 per-tile ones, far fewer for a map in creation order.
// Measured at B (iterations/s, tracking / mapping): per-tile buckets 3 700 / 2 945; groups with 256-Gaussian workgroups 3 900 / 3 025,
// 512: 4 120 / 3 165, 1 024: 3 990 / 3 090 (fewer atomics, but one workgroup per CU leaves its phases unoverlapped).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kGroupBlock = 512;
constexpr int kGroupPerLane = 4;            // groups a lane files through the histogram; a Gaussian's further groups take own atomics
constexpr int kGT = SPLAT_GROUP_TILES;
static_assert(kGT == 2, "the group index is tile >> 1");

__host__ __device__ inline int tile_groups_x(int W) { return (((W + SPLAT_TILE - 1) / SPLAT_TILE) + kGT - 1) / kGT; }
__host__ __device__ inline int tile_groups(int W, int H) { return tile_groups_x(W) * ((((H + SPLAT_TILE - 1) / SPLAT_TILE) + kGT - 1) / kGT); }

// step 0 (every thread of the workgroup, BEFORE the projection work, which hides the barrier): the histogram starts at zero
template <int BLOCK>
__device__ __forceinline__ void group_hist_reset(unsigned *s_grp, int num_groups) {
    for (int g = (int)threadIdx.x; g < num_groups; g += BLOCK) s_grp[g] = 0u;
    __syncthreads();
}

// step 1 (every thread of the workgroup): one 16-byte record (Gaussian id, depth bits, tile rectangle) per touched group of Gaussian i,
// whose tile rectangle is [x0, x1) x [y0, y1) (`filed`: it has one).  A record that finds its group's bucket full raises status[1].
template <int BLOCK>
__device__ __forceinline__ void file_group_records(const SplatState &st, unsigned *s_grp, int i, bool filed, int x0, int y0, int x1, int y1, float depth,
                                                   int ggx, int num_groups) {
    const unsigned gstride = (unsigned)st.group_stride;
    const int gx0 = x0 >> 1, gy0 = y0 >> 1;
    const int gw = filed ? ((x1 - 1) >> 1) - gx0 + 1 : 0, ng = filed ? gw * (((y1 - 1) >> 1) - gy0 + 1) : 0;
    unsigned rank[kGroupPerLane];
    const int nh = min(ng, kGroupPerLane);
#pragma unroll
    for (int t = 0; t < kGroupPerLane; ++t)
        if (t < nh) {
            const int yy = t / gw, xx = t - yy * gw;
            rank[t] = atomicAdd(&s_grp[(gy0 + yy) * ggx + gx0 + xx], 1u);
        }
    __syncthreads();
    // (one returning atomic per non-empty group; two in flight per lane -- with a zero added where one of a lane's two groups is
    //  empty -- was measured: 27.4 -> 34.1 us, the kernel is bound by the NUMBER of L2 atomics, not by their latency)
    for (int g = (int)threadIdx.x; g < num_groups; g += BLOCK) {
        const unsigned cnt = s_grp[g];
        if (cnt) s_grp[g] = atomicAdd(&st.group_count[(size_t)g * SPLAT_COUNTER_STRIDE], cnt);
    }
    __syncthreads();
    const uint4 rec = make_uint4((unsigned)i, __float_as_uint(depth), (unsigned)x0 | ((unsigned)y0 << 16), (unsigned)x1 | ((unsigned)y1 << 16));
    uint4 *recs = reinterpret_cast<uint4 *>(st.group_recs);
    bool spilled = false;
    for (int t = 0; t < ng; ++t) {
        const int yy = t / gw, xx = t - yy * gw;
        const unsigned g = (unsigned)((gy0 + yy) * ggx + gx0 + xx);
        unsigned slot;
        if (t < kGroupPerLane) {
            // (compile-time indices only: a dynamically indexed rank[] would live in scratch memory)
            const unsigned r = t == 0 ? rank[0] : (t == 1 ? rank[1] : (t == 2 ? rank[2] : rank[3]));
            slot = s_grp[g] + r;
        } else {
            slot = atomicAdd(&st.group_count[(size_t)g * SPLAT_COUNTER_STRIDE], 1u);
        }
        if (slot < gstride) recs[(size_t)g * gstride + slot] = rec;
        else spilled = true;
    }
    if (spilled) raise_status(st, 1);
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, m, 64);
        v = v > o ? v : o;
    }
    return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// SplatState.tile_order from SplatState.tile_work for ONE XCD band (a 256-thread workgroup): the band's tiles by descending work
// through a 256-bin counting sort (the order inside a bin is whatever the atomics give: it is a schedule, not a result).
// `scratch`: 513 words of LDS.
__device__ __forceinline__ void tile_order_band(const uint32_t *work, uint32_t *order, int T, int per_xcd, int band, unsigned *scratch) {
    unsigned *s_hist = scratch, *s_off = scratch + 256, *s_max = scratch + 512;
    const int tid = threadIdx.x;
    const int t0 = band * per_xcd, t1 = min(T, t0 + per_xcd);
    if (tid == 0) *s_max = 1u;
    s_hist[tid] = 0u;
    __syncthreads();
    unsigned mx = 0u;
    for (int t = t0 + tid; t < t1; t += 256) mx = max(mx, work[t]);
    mx = wave_max_u32(mx);
    if ((tid & 63) == 0) atomicMax(s_max, mx);
    __syncthreads();
    const float scale = 255.0f / (float)*s_max;
    for (int t = t0 + tid; t < t1; t += 256) atomicAdd(&s_hist[255 - (int)((float)work[t] * scale)], 1u);
    __syncthreads();
    if (tid < 64) {                         // exclusive scan of the 256 bins by one wave (four bins per lane)
        unsigned c[4], sum = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = s_hist[4 * tid + k]; sum += c[k]; }
        unsigned incl = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = (unsigned)__shfl_up((int)incl, d, 64);
            if (tid >= d) incl += o;
        }
        unsigned run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_off[4 * tid + k] = run; run += c[k]; }
    }
    __syncthreads();
    for (int t = t0 + tid; t < t1; t += 256) {
        const unsigned pos = atomicAdd(&s_off[255 - (int)((float)work[t] * scale)], 1u);
        order[t0 + pos] = (uint32_t)t + 1u;              // (tile + 1: zero means "not written yet", block_tile)
    }
    for (int t = max(t1, t0) + tid; t < t0 + per_xcd; t += 256) order[t] = 0xFFFFFFFFu;       // (the last band may be short)
}

#endif  // __HIPCC__

}  // namespace splat
