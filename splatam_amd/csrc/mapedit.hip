// mapedit.hip -- map growth and maintenance on the device (include/splat_hip.h, "Map growth and maintenance").
//
// The reference edits the map with torch.cat / boolean indexing, re-allocating every parameter tensor, the Adam
// moments and the per-Gaussian variables (/root/reference/scripts/splatam.py:378-420,
// /root/reference/utils/slam_external.py:139-188).  Here the map is a capacity-managed struct of arrays edited in
// place, with the reference's row order preserved exactly:
//
//   add_new_gaussians   E1 depth error + 11-bit histogram          E2/E3 two refining histogram passes
//                       (select kernels between them: exact lower median = torch.median, by radix selection on the
//                        float bit patterns, which are monotone for the non-negative values |gt - d| * (gt > 0))
//                       E4 per-block counts of selected pixels     E5 block scan, capacity check, result counts
//                       E6 stable append: ballot + popcount rank inside the block, block offset from the scan
//   prune / remove      R1 keep flags + per-block counts           E5 (same scan)
//                       R2 stable gather of every array into the staging buffer   R3 copy back
//
// Everything is HBM-streaming work of a few MB per frame: one lane per pixel / per row, coalesced 4-byte streams.
#include "splat_device.h"

#include "fused_math.h"

namespace splat {

namespace {

constexpr int kBlock = 256;
constexpr int kRounds = 4;                          // rows / pixels per lane in the counting + compaction kernels
constexpr int kPerBlock = kBlock * kRounds;         // 1024 consecutive rows per workgroup
constexpr int kHistBins = 2048;

// scratch words
constexpr int kWPrefix = kHistBins;                 // bit prefix of the median found so far
constexpr int kWRank = kHistBins + 1;               // rank still to resolve inside the prefix bucket
constexpr int kWNan = kHistBins + 2;                // NaN depth errors seen (torch.median then returns NaN)
constexpr int kWNonPresence = kHistBins + 3;        // sum(non_presence_mask) before the valid-depth mask
constexpr int kWBlocks = kHistBins + 16;            // per-block counts, then (same slots) exclusive offsets

// exclusive prefix of v over the 256 threads of the block; *total = block sum.  s_tmp: 4 + 1 words.
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *s_tmp, unsigned *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < kBlock / 64; ++w) {
        const unsigned c = s_tmp[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------------------------------------------------
// add_new_gaussians
// ---------------------------------------------------------------------------------------------------------
// depth_error = torch.abs(gt_depth - render_depth) * (gt_depth > 0)   (/root/reference/scripts/splatam.py:390)
__device__ __forceinline__ float depth_error(float gt, float rd) { return fabsf(gt - rd) * (gt > 0.f ? 1.f : 0.f); }

__global__ __launch_bounds__(kBlock) void depth_error_hist_kernel(SplatAddArgs a, int HW) {
    __shared__ unsigned s_hist[kHistBins];
    for (int k = threadIdx.x; k < kHistBins; k += kBlock) s_hist[k] = 0;
    __syncthreads();
    const float *rd = a.out6 + 3 * (size_t)HW;
    unsigned nan = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) {
        const float e = depth_error(a.depth[i], rd[i]);
        a.err[i] = e;
        if (e != e) ++nan;
        else atomicAdd(&s_hist[__float_as_uint(e) >> 21], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kHistBins; k += kBlock)
        if (s_hist[k]) atomicAdd(&a.scratch[k], s_hist[k]);
    if (nan) atomicAdd(&a.scratch[kWNan], nan);
}

// pixels whose error shares the prefix found so far: histogram of the next `bins` bits
__global__ __launch_bounds__(kBlock) void refine_hist_kernel(SplatAddArgs a, int HW, int prefix_shift, int shift, int bins) {
    __shared__ unsigned s_hist[kHistBins];
    for (int k = threadIdx.x; k < bins; k += kBlock) s_hist[k] = 0;
    __syncthreads();
    const unsigned prefix = a.scratch[kWPrefix];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < HW; i += gridDim.x * kBlock) {
        const float e = a.err[i];
        const unsigned b = __float_as_uint(e);
        if (e == e && (b >> prefix_shift) == prefix) atomicAdd(&s_hist[(b >> shift) & (unsigned)(bins - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < bins; k += kBlock)
        if (s_hist[k]) atomicAdd(&a.scratch[k], s_hist[k]);
}

// one workgroup: the bin that holds rank `k` (0-based) of the current bucket; extends the prefix, re-bases the rank,
// clears the histogram for the next pass.  first: rank = (HW - 1) / 2, torch.median's lower median.
__global__ __launch_bounds__(kBlock) void select_bin_kernel(uint32_t *scratch, int bins, int bits, int first, int HW, int32_t *counts) {
    __shared__ unsigned s_tmp[8];
    __shared__ unsigned s_bin, s_before;
    const int per = bins / kBlock;
    if (threadIdx.x == 0) { s_bin = 0; s_before = 0; }      // all-NaN input: no bin holds the rank (the median is NaN anyway)
    const unsigned rank = first ? (unsigned)((HW - 1) / 2) : scratch[kWRank];
    unsigned mine[kHistBins / kBlock], sum = 0;
    for (int j = 0; j < per; ++j) {
        mine[j] = scratch[threadIdx.x * per + j];
        sum += mine[j];
    }
    unsigned total;
    unsigned before = block_exclusive_scan(sum, s_tmp, &total);
    if (rank >= before && rank < before + sum) {            // exactly one thread (bins are non-empty where the rank falls)
        unsigned b = before;
        for (int j = 0; j < per; ++j) {
            if (rank < b + mine[j]) { s_bin = (unsigned)(threadIdx.x * per + j); s_before = b; break; }
            b += mine[j];
        }
    }
    __syncthreads();
    for (int j = 0; j < per; ++j) scratch[threadIdx.x * per + j] = 0;
    if (threadIdx.x == 0) {
        const unsigned prefix = first ? 0u : scratch[kWPrefix];
        const unsigned np = (prefix << bits) | s_bin;
        scratch[kWPrefix] = np;
        scratch[kWRank] = rank - s_before;
        // last pass leaves the full bit pattern; NaN anywhere -> the median is NaN, as torch.median returns it
        counts[4] = scratch[kWNan] ? 0x7fc00000 : (int32_t)np;
    }
}

struct PixelPick {
    bool pick;          // becomes a Gaussian
    bool non_presence;  // non_presence_mask before the valid-depth mask (decides whether the variables are reset)
};

__device__ __forceinline__ PixelPick pick_pixel(const SplatAddArgs &a, int i, int HW, float median) {
    PixelPick r;
    const float gt = a.depth[i];
    if (a.mode == SPLAT_ADD_VALID_DEPTH) {
        r.pick = r.non_presence = gt > 0.f;
        return r;
    }
    const float rd = a.out6[3 * (size_t)HW + i], sil = a.out6[4 * (size_t)HW + i];
    const float e = a.err[i];
    // /root/reference/scripts/splatam.py:386-393
    const bool np = (sil < a.sil_thres) || ((rd > gt) && (e > 50.f * median));
    r.non_presence = np;
    r.pick = np && gt > 0.f;                    // :402-403
    return r;
}

__device__ __forceinline__ float add_median(const SplatAddArgs &a, const int32_t *counts) {
    return a.mode == SPLAT_ADD_NON_PRESENCE ? __int_as_float(counts[4]) : 0.f;
}

__global__ __launch_bounds__(kBlock) void add_count_kernel(SplatAddArgs a, int HW, const int32_t *counts) {
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const float median = add_median(a, counts);
    unsigned c = 0, n = 0;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int i = blockIdx.x * kPerBlock + r * kBlock + threadIdx.x;
        if (i < HW) {
            const PixelPick p = pick_pixel(a, i, HW, median);
            c += p.pick ? 1u : 0u;
            n += p.non_presence ? 1u : 0u;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { c += (unsigned)__shfl_xor((int)c, m, 64); n += (unsigned)__shfl_xor((int)n, m, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], c); atomicAdd(&s_cnt[1], n); }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.scratch[kWBlocks + blockIdx.x] = s_cnt[0];
        if (s_cnt[1]) atomicAdd(&a.scratch[kWNonPresence], s_cnt[1]);
    }
}

// one workgroup: exclusive scan of the per-block counts (in place) + the result words.
// append: rows0 + total must fit capacity, else counts[2] = 1 and counts[0] = rows0.  removal: counts[0] = total kept.
__global__ __launch_bounds__(kBlock) void scan_blocks_kernel(uint32_t *scratch, int nblocks, int rows0, int capacity, int append,
                                                             int32_t *counts) {
    __shared__ unsigned s_tmp[8];
    unsigned carry = 0;
    for (int base = 0; base < nblocks; base += kBlock) {
        const int b = base + threadIdx.x;
        const unsigned v = b < nblocks ? scratch[kWBlocks + b] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan(v, s_tmp, &total);
        if (b < nblocks) scratch[kWBlocks + b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        if (append) {
            const bool fits = (long long)rows0 + (long long)carry <= (long long)capacity;
            counts[0] = fits ? rows0 + (int)carry : rows0;
            counts[1] = (int)carry;
            counts[2] = fits ? 0 : 1;
            counts[3] = (int)scratch[kWNonPresence];
        } else {
            counts[0] = (int)carry;
            counts[1] = rows0 - (int)carry;
            counts[2] = 0;
        }
    }
}

struct Rigid { float R[9], t[3]; };      // row-major world-to-camera

__global__ __launch_bounds__(kBlock) void append_kernel(SplatMapStore st, SplatAddArgs a, int HW, const int32_t *counts) {
    __shared__ unsigned s_wave[kBlock / 64];
    const int rows0 = st.map.P;
    const bool fits = counts[2] == 0;
    const float median = add_median(a, counts);
    // world-to-camera of the frame: the map's pose (normalised quaternion -> build_rotation, :396-400) or the given one
    Rigid w;
    if (a.mode == SPLAT_ADD_NON_PRESENCE) {
        Pose P;
        pose_from_params(st.map.cam_unnorm_rots + a.time_idx, st.map.cam_trans + a.time_idx, st.map.num_frames, P);
#pragma unroll
        for (int k = 0; k < 9; ++k) w.R[k] = P.R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) w.t[k] = P.t[k];
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w.R[3 * r + c] = a.w2c[4 * r + c];
            w.t[r] = a.w2c[4 * r + 3];
        }
    }
    const bool iso = st.map.isotropic != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned base = fits ? (unsigned)rows0 + a.scratch[kWBlocks + blockIdx.x] : 0u;
    if (fits) {
#pragma unroll 1
        for (int r = 0; r < kRounds; ++r) {
            const int i = blockIdx.x * kPerBlock + r * kBlock + threadIdx.x;
            const bool pick = i < HW && pick_pixel(a, i, HW, median).pick;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pick);
            if (lane == 0) s_wave[wave] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned off = 0, tot = 0;
            for (int k = 0; k < kBlock / 64; ++k) {
                if (k < wave) off += s_wave[k];
                tot += s_wave[k];
            }
            __syncthreads();
            if (pick) {
                const unsigned row = base + off + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                // get_pointcloud (/root/reference/scripts/splatam.py:67-116)
                const int v = i / a.width, u = i - v * a.width;
                const float z = a.depth[i];
                const float xx = ((float)u - a.cx) / a.fx, yy = ((float)v - a.cy) / a.fy;
                const float pc[3] = {xx * z, yy * z, z};
                // c2w of a rigid w2c: R^T (p - t)
                const float d[3] = {pc[0] - w.t[0], pc[1] - w.t[1], pc[2] - w.t[2]};
                float *m3 = st.map.means3D + 3 * (size_t)row;
#pragma unroll
                for (int c = 0; c < 3; ++c) m3[c] = w.R[c] * d[0] + w.R[3 + c] * d[1] + w.R[6 + c] * d[2];
                float *col = st.map.rgb_colors + 3 * (size_t)row;
#pragma unroll
                for (int c = 0; c < 3; ++c) col[c] = a.im[(size_t)c * HW + i];
                reinterpret_cast<float4 *>(st.map.unnorm_rotations)[row] = make_float4(1.f, 0.f, 0.f, 0.f);
                st.map.logit_opacities[row] = 0.f;
                // mean3_sq_dist = (z / ((FX + FY) / 2))^2; log_scales = log(sqrt(mean3_sq_dist))   (:94-99, :126-129)
                const float sg = z / ((a.fx + a.fy) / 2.f);
                const float ls = logf(sqrtf(sg * sg));
                if (iso) st.map.log_scales[row] = ls;
                else { st.map.log_scales[3 * (size_t)row] = ls; st.map.log_scales[3 * (size_t)row + 1] = ls; st.map.log_scales[3 * (size_t)row + 2] = ls; }
                if (st.timestep) st.timestep[row] = (float)a.time_idx;
                const int width[5] = {3, 3, 4, 1, iso ? 1 : 3};
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    if (st.exp_avg[g])
                        for (int c = 0; c < width[g]; ++c) st.exp_avg[g][(size_t)row * width[g] + c] = 0.f;
                    if (st.exp_avg_sq[g])
                        for (int c = 0; c < width[g]; ++c) st.exp_avg_sq[g][(size_t)row * width[g] + c] = 0.f;
                }
            }
            base += tot;
        }
    }
    // the reference re-creates these three for ALL rows whenever non_presence_mask had a set pixel (:413-416)
    if (counts[3] > 0 && fits) {
        const int rows = counts[0];
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < rows; i += gridDim.x * kBlock) {
            if (st.max_2D_radius) st.max_2D_radius[i] = 0.f;
            if (st.means2D_gradient_accum) st.means2D_gradient_accum[i] = 0.f;
            if (st.denom) st.denom[i] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// prune_gaussians / remove_points
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxArrays = 19;
struct MapArrays {
    float *ptr[kMaxArrays];
    int width[kMaxArrays];
    long long stage_off[kMaxArrays];     // float offset of the array's region in the staging buffer
    int n;
};

__global__ __launch_bounds__(kBlock) void prune_flag_kernel(SplatMapStore st, SplatPruneArgs a) {
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int P = st.map.P;
    const bool iso = st.map.isotropic != 0;
    unsigned c = 0;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int i = blockIdx.x * kPerBlock + r * kBlock + threadIdx.x;
        if (i < P) {
            bool rem;
            if (a.to_remove) {
                rem = a.to_remove[i] != 0;
            } else {
                // /root/reference/utils/slam_external.py:176-181
                const float op = 1.0f / (1.0f + expf(-st.map.logit_opacities[i]));
                rem = op < a.removal_opacity_threshold;
                if (a.remove_big) {
                    float mx = expf(st.map.log_scales[iso ? i : 3 * i]);
                    if (!iso) mx = fmaxf(mx, fmaxf(expf(st.map.log_scales[3 * i + 1]), expf(st.map.log_scales[3 * i + 2])));
                    rem = rem || (mx > a.big_scale);
                }
            }
            a.flags[i] = rem ? 0 : 1;
            c += rem ? 0u : 1u;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) c += (unsigned)__shfl_xor((int)c, m, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) a.scratch[kWBlocks + blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(kBlock) void gather_rows_kernel(MapArrays arr, const uint8_t *flags, const uint32_t *scratch, float *stage, int P) {
    __shared__ unsigned s_wave[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned base = scratch[kWBlocks + blockIdx.x];
#pragma unroll 1
    for (int r = 0; r < kRounds; ++r) {
        const int i = blockIdx.x * kPerBlock + r * kBlock + threadIdx.x;
        const bool keep = i < P && flags[i] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned off = 0, tot = 0;
        for (int k = 0; k < kBlock / 64; ++k) {
            if (k < wave) off += s_wave[k];
            tot += s_wave[k];
        }
        __syncthreads();
        if (keep) {
            const size_t row = base + off + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < arr.n; ++k) {
                const int w = arr.width[k];
                const float *src = arr.ptr[k] + (size_t)i * w;
                float *dst = stage + arr.stage_off[k] + row * w;
                if (w == 4) {
                    *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
                } else {
                    for (int c = 0; c < w; ++c) dst[c] = src[c];
                }
            }
        }
        base += tot;
    }
}

__global__ __launch_bounds__(kBlock) void copy_back_kernel(MapArrays arr, const float *stage, const int32_t *counts) {
    const long long rows = counts[0];
    const int k = blockIdx.y;
    const long long n = rows * arr.width[k];
    const float *src = stage + arr.stage_off[k];
    float *dst = arr.ptr[k];
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long long)gridDim.x * kBlock) dst[e] = src[e];
}

// ---------------------------------------------------------------------------------------------------------
// densify (/root/reference/utils/slam_external.py:191-240): clone / split selection and the duplication of rows
// ---------------------------------------------------------------------------------------------------------
//   D1 densify_flag_kernel     flags[i] = grads[i] >= grad_thresh  &&  max exp(log_scales[i]) <= / > 0.01 scene_radius
//                              with grads = means2D_gradient_accum / denom, NaN (0 / 0) -> 0 and rows >= rows_with_grad -> 0
//                              (the reference's padded_grad), + per-block counts;  E5 scan (append form)
//   D2 duplicate_rows_kernel   the selected rows, in order, `repeat` times block after block (torch's v[mask].repeat(n, 1)),
//                              appended at the tail; split form: means3D += build_rotation(unnorm_rotations) @ sample,
//                              log_scales = log(exp(log_scales) / (0.8 n)); Adam moments of the new rows zero
// The normal samples of the split are drawn by torch (the reference's own generator stream) and handed in.
__device__ __forceinline__ float row_max_scale(const SplatMap &m, int i) {
    if (m.isotropic) return expf(m.log_scales[i]);
    return fmaxf(expf(m.log_scales[3 * (size_t)i]), fmaxf(expf(m.log_scales[3 * (size_t)i + 1]), expf(m.log_scales[3 * (size_t)i + 2])));
}

__global__ __launch_bounds__(kBlock) void densify_flag_kernel(SplatMapStore st, SplatDensifyArgs a) {
    __shared__ unsigned s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int P = st.map.P;
    unsigned c = 0;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int i = blockIdx.x * kPerBlock + r * kBlock + threadIdx.x;
        if (i < P) {
            float g = 0.f;
            if (i < a.rows_with_grad) {
                g = st.means2D_gradient_accum[i] / st.denom[i];
                if (g != g) g = 0.f;                                   // grads[grads.isnan()] = 0.0
            }
            const float mx = row_max_scale(st.map, i);
            const bool big = mx > a.small_scale;
            const bool sel = g >= a.grad_thresh && (a.mode == SPLAT_DENSIFY_SPLIT ? big : !big);
            a.flags[i] = sel ? 1 : 0;
            c += sel ? 1u : 0u;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) c += (unsigned)__shfl_xor((int)c, m, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) a.scratch[kWBlocks + blockIdx.x] = s_cnt;
}

// one workgroup: scan of the per-block counts; counts[1] = selected rows S, counts[0] = rows after appending S * repeat,
// counts[2] = does not fit
__global__ __launch_bounds__(kBlock) void densify_scan_kernel(uint32_t *scratch, int nblocks, int rows0, int capacity, int repeat, int32_t *counts) {
    __shared__ unsigned s_tmp[8];
    unsigned carry = 0;
    for (int base = 0; base < nblocks; base += kBlock) {
        const int b = base + threadIdx.x;
        const unsigned v = b < nblocks ? scratch[kWBlocks + b] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan(v, s_tmp, &total);
        if (b < nblocks) scratch[kWBlocks + b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        const long long after = (long long)rows0 + (long long)carry * repeat;
        const bool fits = after <= (long long)capacity;
        counts[0] = fits ? (int)after : rows0;
        counts[1] = (int)carry;
        counts[2] = fits ? 0 : 1;
    }
}

__global__ __launch_bounds__(kBlock) void duplicate_rows_kernel(SplatMapStore st, SplatDensifyArgs a, const int32_t *counts) {
    __shared__ unsigned s_wave[kBlock / 64];
    if (counts[2] != 0) return;                                        // does not fit: nothing is written
    const int P = st.map.P, S = counts[1], n = a.mode == SPLAT_DENSIFY_SPLIT ? a.num_to_split_into : 1;
    const bool iso = st.map.isotropic != 0;
    const bool split = a.mode == SPLAT_DENSIFY_SPLIT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int width[5] = {3, 3, 4, 1, iso ? 1 : 3};
    float *params[5] = {st.map.means3D, st.map.rgb_colors, st.map.unnorm_rotations, st.map.logit_opacities, st.map.log_scales};
    unsigned base = a.scratch[kWBlocks + blockIdx.x];
#pragma unroll 1
    for (int r = 0; r < kRounds; ++r) {
        const int i = blockIdx.x * kPerBlock + r * kBlock + threadIdx.x;
        const bool sel = i < P && a.flags[i] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(sel);
        if (lane == 0) s_wave[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned off = 0, tot = 0;
        for (int k = 0; k < kBlock / 64; ++k) {
            if (k < wave) off += s_wave[k];
            tot += s_wave[k];
        }
        __syncthreads();
        if (sel) {
            const unsigned q = base + off + (unsigned)__popcll(m & ((1ull << lane) - 1ull));      // rank among the selected rows
            float R[9];
            if (split) {
                const float4 u = reinterpret_cast<const float4 *>(st.map.unnorm_rotations)[i];
                // build_rotation normalises its argument (/root/reference/utils/slam_external.py:25-42)
                const float inv = 1.0f / sqrtf(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w);
                const float qn[4] = {u.x * inv, u.y * inv, u.z * inv, u.w * inv};
                quat_to_rot(qn, R);
            }
            for (int rep = 0; rep < n; ++rep) {
                const size_t row = (size_t)P + (size_t)rep * S + q;
                for (int g = 0; g < 5; ++g) {
                    const int w = width[g];
                    for (int c = 0; c < w; ++c) {
                        float v = params[g][(size_t)i * w + c];
                        if (split && g == 0) {
                            const float *sm = a.samples + 3 * ((size_t)rep * S + q);
                            v += R[3 * c] * sm[0] + R[3 * c + 1] * sm[1] + R[3 * c + 2] * sm[2];
                        }
                        if (split && g == 4) v = logf(expf(v) / (0.8f * (float)n));
                        params[g][row * w + c] = v;
                        if (st.exp_avg[g]) st.exp_avg[g][row * w + c] = 0.f;
                        if (st.exp_avg_sq[g]) st.exp_avg_sq[g][row * w + c] = 0.f;
                    }
                }
                // per-Gaussian variables of the new rows: the three the reference re-creates are zero; `timestep` is copied (the
                // reference's densify does not extend it at all: /root/reference/utils/slam_external.py:206-227)
                if (st.max_2D_radius) st.max_2D_radius[row] = 0.f;
                if (st.means2D_gradient_accum) st.means2D_gradient_accum[row] = 0.f;
                if (st.denom) st.denom[row] = 0.f;
                if (st.timestep) st.timestep[row] = st.timestep[i];
            }
        }
        base += tot;
    }
}

// tile counters of the bucketed lists after a forward-only pass (the full iteration folds them in its last per-Gaussian kernel)
__global__ __launch_bounds__(kBlock) void fold_tile_counters_kernel(SplatState st, int T, int G) {
    unsigned sum = 0, mx = 0;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < T; t += gridDim.x * kBlock) {
        const unsigned cnt = st.tile_count[(size_t)t * SPLAT_COUNTER_STRIDE];
        st.tile_count[(size_t)t * SPLAT_COUNTER_STRIDE] = 0;
        st.tile_cursor[(size_t)t * SPLAT_COUNTER_STRIDE] = cnt;
        if (st.group_count && t < G) {                                                          // (G <= T)
            st.group_count[(size_t)t * SPLAT_COUNTER_STRIDE + 1] = st.group_count[(size_t)t * SPLAT_COUNTER_STRIDE];    // kept, see fused_backward_kernel
            st.group_count[(size_t)t * SPLAT_COUNTER_STRIDE] = 0;
        }
        sum += cnt;
        mx = max(mx, cnt);
    }
    for (int m = 32; m >= 1; m >>= 1) { sum += (unsigned)__shfl_xor((int)sum, m, 64); mx = max(mx, (unsigned)__shfl_xor((int)mx, m, 64)); }
    if ((threadIdx.x & 63) == 0 && sum) { atomicAdd((unsigned *)&st.status[0], sum); atomicMax((unsigned *)&st.status[2], mx); }
}

MapArrays map_arrays(const SplatMapStore &st) {
    MapArrays a{};
    const int width[5] = {3, 3, 4, 1, st.map.isotropic ? 1 : 3};
    float *params[5] = {st.map.means3D, st.map.rgb_colors, st.map.unnorm_rotations, st.map.logit_opacities, st.map.log_scales};
    long long off = 0;
    auto push = [&](float *p, int w) {
        if (!p) return;
        a.ptr[a.n] = p;
        a.width[a.n] = w;
        a.stage_off[a.n] = off;
        off += (long long)((st.capacity + 3) & ~3) * w;      // regions start 16-byte aligned
        ++a.n;
    };
    for (int g = 0; g < 5; ++g) push(params[g], width[g]);
    for (int g = 0; g < 5; ++g) push(st.exp_avg[g], width[g]);
    for (int g = 0; g < 5; ++g) push(st.exp_avg_sq[g], width[g]);
    push(st.max_2D_radius, 1);
    push(st.means2D_gradient_accum, 1);
    push(st.denom, 1);
    push(st.timestep, 1);
    return a;
}

}  // namespace

size_t map_scratch_words(long long n) { return (size_t)kWBlocks + (size_t)((n + kPerBlock - 1) / kPerBlock) + 16; }

int map_row_floats(const SplatMapStore &st) {
    const MapArrays a = map_arrays(st);
    int w = 0;
    for (int k = 0; k < a.n; ++k) w += a.width[k];
    return w;
}

// torch.median(|gt - d| * (gt > 0)) of one render (exact lower median, NaN if any NaN): err[HW] and the bits of the median
// in counts[4].  Used by add_new_gaussians and by get_loss(ignore_outlier_depth_loss=True).
hipError_t launch_depth_error_median(const float *out6, const float *depth, float *err, uint32_t *scratch, int HW, int32_t *counts,
                                     hipStream_t s) {
    SplatAddArgs a{};
    a.out6 = out6;
    a.depth = depth;
    a.err = err;
    a.scratch = scratch;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint32_t) * kWBlocks, s);
    if (e != hipSuccess || HW <= 0) return e;
    const int sblocks = min((HW + kBlock - 1) / kBlock, 1024);
    hipLaunchKernelGGL(depth_error_hist_kernel, dim3(sblocks), dim3(kBlock), 0, s, a, HW);
    hipLaunchKernelGGL(select_bin_kernel, dim3(1), dim3(kBlock), 0, s, scratch, 2048, 11, 1, HW, counts);
    hipLaunchKernelGGL(refine_hist_kernel, dim3(sblocks), dim3(kBlock), 0, s, a, HW, 21, 10, 2048);
    hipLaunchKernelGGL(select_bin_kernel, dim3(1), dim3(kBlock), 0, s, scratch, 2048, 11, 0, HW, counts);
    hipLaunchKernelGGL(refine_hist_kernel, dim3(sblocks), dim3(kBlock), 0, s, a, HW, 10, 0, 1024);
    hipLaunchKernelGGL(select_bin_kernel, dim3(1), dim3(kBlock), 0, s, scratch, 1024, 10, 0, HW, counts);
    return hipGetLastError();
}

hipError_t launch_map_add(const SplatMapStore &st, const SplatAddArgs &a, hipStream_t s) {
    const int HW = a.width * a.height;
    const int nblocks = (HW + kPerBlock - 1) / kPerBlock;
    hipError_t e = hipMemsetAsync(a.scratch, 0, sizeof(uint32_t) * kWBlocks, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(st.counts, 0, sizeof(int32_t) * 8, s);
    if (e != hipSuccess) return e;
    if (HW > 0) {
        if (a.mode == SPLAT_ADD_NON_PRESENCE) {
            e = launch_depth_error_median(a.out6, a.depth, a.err, a.scratch, HW, st.counts, s);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(add_count_kernel, dim3(nblocks), dim3(kBlock), 0, s, a, HW, st.counts);
    }
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(kBlock), 0, s, a.scratch, nblocks, st.map.P, st.capacity, 1, st.counts);
    if (HW > 0) hipLaunchKernelGGL(append_kernel, dim3(nblocks), dim3(kBlock), 0, s, st, a, HW, st.counts);
    return hipGetLastError();
}

hipError_t launch_map_prune(const SplatMapStore &st, const SplatPruneArgs &a, hipStream_t s) {
    const int P = st.map.P;
    const int nblocks = (P + kPerBlock - 1) / kPerBlock;
    if (P > 0) hipLaunchKernelGGL(prune_flag_kernel, dim3(nblocks), dim3(kBlock), 0, s, st, a);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(kBlock), 0, s, a.scratch, nblocks, P, st.capacity, 0, st.counts);
    if (P > 0) {
        const MapArrays arr = map_arrays(st);
        hipLaunchKernelGGL(gather_rows_kernel, dim3(nblocks), dim3(kBlock), 0, s, arr, a.flags, a.scratch, a.stage, P);
        const int cblocks = min((int)(((long long)P * 4 + kBlock - 1) / kBlock), 2048);
        hipLaunchKernelGGL(copy_back_kernel, dim3(cblocks, arr.n), dim3(kBlock), 0, s, arr, a.stage, st.counts);
    }
    return hipGetLastError();
}

hipError_t launch_map_densify_select(const SplatMapStore &st, const SplatDensifyArgs &a, hipStream_t s) {
    const int P = st.map.P;
    const int nblocks = (P + kPerBlock - 1) / kPerBlock;
    if (P > 0) hipLaunchKernelGGL(densify_flag_kernel, dim3(nblocks), dim3(kBlock), 0, s, st, a);
    const int rep = a.mode == SPLAT_DENSIFY_SPLIT ? a.num_to_split_into : 1;
    hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(kBlock), 0, s, a.scratch, nblocks, P, st.capacity, rep, st.counts);
    return hipGetLastError();
}

hipError_t launch_map_duplicate(const SplatMapStore &st, const SplatDensifyArgs &a, hipStream_t s) {
    const int P = st.map.P;
    const int nblocks = (P + kPerBlock - 1) / kPerBlock;
    if (P > 0) hipLaunchKernelGGL(duplicate_rows_kernel, dim3(nblocks), dim3(kBlock), 0, s, st, a, st.counts);
    return hipGetLastError();
}

hipError_t launch_fold_tile_counters(SplatState &st, int T, int G, hipStream_t s) {
    if (T > 0) hipLaunchKernelGGL(fold_tile_counters_kernel, dim3(min((T + kBlock - 1) / kBlock, 64)), dim3(kBlock), 0, s, st, T, G);
    return hipGetLastError();
}

}  // namespace splat
