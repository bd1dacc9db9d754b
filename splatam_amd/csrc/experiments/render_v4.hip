// render_v4.hip -- tile-wise alpha compositing, generation 4: ONE 4x4-PIXEL BLOCK PER 16-LANE ROW.
//
// EXPERIMENT, not the default (splat_debug_option(1, 4) selects it; parity-tested like generation 3).  Measured on
// MI355X at workload B (profiles/r01_v4_experiment.md): K6 86 us (generation 3: 79 us), K7 350 us (generation 3: 204 us).
// The VALU instruction count of K7 does drop (1.03e8 -> 7.7e7 per launch) but the per-(Gaussian, block) partial sums
// double the accumulator atomics, and those are the wall: 125 us without any accumulation, 350 us with ds_add_f32 into
// LDS accumulators, 455 us with global float atomics (19.8 M per launch, ~45 atomics/ns -- the same rate generation 3
// already sustains with 9.7 M per launch in 204 us).
//
// Why: SplaTAM's splats are small (sigma ~1.2 px, {alpha >= 1/255} radius ~3.9 px at workload B).  Generation 3 walks
// one Gaussian per wave step over an 8x8 quadrant: 0.97 M (Gaussian, quadrant) visits per pass with ~21 % of the lanes
// live, and the kernels are VALU-issue bound.  The same scene has 1.98 M (Gaussian, 4x4 block) pairs;
// giving every 16-lane DPP row of a wave its OWN block and its OWN Gaussian list processes four pairs per wave step:
// 0.49 M steps (2x fewer), 36 % of the lanes live, and the cross-lane reduction of the backward pass shrinks to a
// 16-lane row (DPP only, no permlane swaps).
//
//   * workgroup = one 16x16 tile (256 threads), wave = one 8x8 quadrant, row r of the wave = block (r & 1, r >> 1) of
//     the quadrant, lane l of the row = pixel (l & 3, l >> 2) of the block;
//   * staging per batch of 256 list entries: every thread gathers one Gaussian's record into LDS (as before) and
//     tests it EXACTLY against the 16 blocks of the tile (axis-aligned extents of {alpha >= 1/255} + the radial
//     bound); 16 ballots per gathering wave -> per-block 256-bit masks -> per-block compacted u8 lists in LDS
//     (order preserved: the lists stay depth sorted);
//   * compute: each row walks its own list (ds_read_u8 -> record at a row-uniform LDS address); the wave runs
//     max(list length) steps; rows that are finished idle;
//   * backward: the 6 + |SMASK| partial sums are reduced inside the row with a packed DPP network (four values in 11
//     instructions: row_ror:8 / row_half_mirror selects + two quad_perm adds) and added to the Gaussian's accumulator
//     IN LDS (ds_add_f32 from four lanes per row); after the batch each thread flushes one Gaussian's accumulator with
//     global float atomics -- one per (Gaussian, tile, component) instead of one per (Gaussian, quadrant, component).
//
// Arithmetic and its order per pixel are those of generation 3 (render.hip): results are identical up to the
// summation order of the per-Gaussian partial sums.
#include "../splat_device.h"

namespace splat {

namespace v4 {

constexpr int kBatch = 256;
constexpr int kFusedSortMax = 1024;

template <int FP>
struct Staged {
    float4 ga;          // A = -0.5*cxx*log2e, B = -cxy*log2e, Cq = -0.5*cyy*log2e, opacity
    float2 mu;
    float feat[FP];
    unsigned mask;      // 16-bit block mask, bit = block_row * 4 + block_col
    unsigned id;
};

template <int C, int CS>
__device__ __forceinline__ void load_colors(const float *colors, unsigned id, float *out) {
    if constexpr (CS % 4 == 0) {
        const float4 *p = reinterpret_cast<const float4 *>(colors + (size_t)id * CS);
#pragma unroll
        for (int v = 0; v < (C + 3) / 4; ++v) {
            const float4 t = p[v];
            if (4 * v < C) out[4 * v] = t.x;
            if (4 * v + 1 < C) out[4 * v + 1] = t.y;
            if (4 * v + 2 < C) out[4 * v + 2] = t.z;
            if (4 * v + 3 < C) out[4 * v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out[ch] = colors[(size_t)id * CS + ch];
    }
}

template <int C, int CS, bool WITH_DEPTH, int FP>
__device__ __forceinline__ void gather(Staged<FP> &s, const SplatState &st, const float *colors, unsigned idx, bool valid,
                                       float tile_x0, float tile_y0, const uint64_t *lds_keys = nullptr, int lds_idx = 0) {
    s.ga = make_float4(0.f, 0.f, 0.f, 0.f);
    s.mu = make_float2(0.f, 0.f);
#pragma unroll
    for (int f = 0; f < FP; ++f) s.feat[f] = 0.f;
    s.mask = 0;
    s.id = 0;
    if (valid) {
        const unsigned id = lds_keys ? (unsigned)lds_keys[lds_idx] : st.point_list[idx];
        const float4 co = reinterpret_cast<const float4 *>(st.conic_opacity)[id];
        const float2 mu = reinterpret_cast<const float2 *>(st.xy)[id];
        load_colors<C, CS>(colors, id, s.feat);
        if constexpr (WITH_DEPTH) s.feat[C] = st.depth[id];
        // live region {alpha >= 1/255}: (p-mu)^T Q (p-mu) <= 2 tau, tau = ln(255 o); half extents sqrt(2 tau Q^-1_ii);
        // radial bound d^T Q d >= lambda_min |d|^2 (exact for round splats, conservative otherwise)
        unsigned mask = 0;
        const float tau2 = 2.0f * __logf(255.0f * co.w);
        if (tau2 >= 0.f) {
            const float det = co.x * co.z - co.y * co.y;
            const float hx = sqrtf(tau2 * co.z / det) * 1.00001f + 0.01f;
            const float hy = sqrtf(tau2 * co.x / det) * 1.00001f + 0.01f;
            const float mid = 0.5f * (co.x + co.z);
            const float lam_min = mid - sqrtf(fmaxf(0.f, mid * mid - det));
            const float thr = tau2 * 1.001f + 1e-3f;
            unsigned xm = 0, ym = 0;
            float dx2[4], dy2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float bx0 = tile_x0 + 4.f * k, by0 = tile_y0 + 4.f * k;
                if ((mu.x - hx <= bx0 + 3.f) && (mu.x + hx >= bx0)) xm |= 1u << k;
                if ((mu.y - hy <= by0 + 3.f) && (mu.y + hy >= by0)) ym |= 1u << k;
                const float ddx = fmaxf(fmaxf(bx0 - mu.x, mu.x - (bx0 + 3.f)), 0.f);
                const float ddy = fmaxf(fmaxf(by0 - mu.y, mu.y - (by0 + 3.f)), 0.f);
                dx2[k] = lam_min * ddx * ddx;
                dy2[k] = lam_min * ddy * ddy;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (((xm >> c) & (ym >> r) & 1u) && !(dx2[c] + dy2[r] > thr)) mask |= 1u << (4 * r + c);
            if (!(hx == hx) || !(hy == hy) || !(lam_min == lam_min)) mask = 0xffffu;     // NaN geometry: no culling
        }
        s.ga = make_float4(-0.5f * kLog2e * co.x, -kLog2e * co.y, -0.5f * kLog2e * co.z, co.w);
        s.mu = mu;
        s.mask = mask;
        s.id = id;
    }
}

// LDS image of one batch
template <int FP>
struct Batch {
    static constexpr int R4 = FP / 4 + 2;
    float4 rec[kBatch * R4];                    // [0] ga  [1 .. FP/4] features  [R4-1] mu_x, mu_y, id, 0
    unsigned long long bmask[16][4];            // [block][gathering wave]: ballot of "can touch the block"
    unsigned short pre[16][4];                  // entries of the block's list that come from earlier gathering waves
    unsigned short total[16];                   // list lengths
    unsigned char list[16][kBatch];             // per-block compacted entry indices, ascending
    unsigned flag[4];
};

// Three-phase commit.  Callers: barrier before (previous batch fully consumed); on return the batch is visible.
template <int FP>
__device__ __forceinline__ void commit(Batch<FP> &b, const Staged<FP> &s, int tid, unsigned flag) {
    constexpr int R4 = Batch<FP>::R4;
    b.rec[tid * R4] = s.ga;
#pragma unroll
    for (int v = 0; v < FP / 4; ++v)
        b.rec[tid * R4 + 1 + v] = make_float4(s.feat[4 * v], s.feat[4 * v + 1], s.feat[4 * v + 2], s.feat[4 * v + 3]);
    b.rec[tid * R4 + R4 - 1] = make_float4(s.mu.x, s.mu.y, __uint_as_float(s.id), 0.f);
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64((s.mask >> k) & 1u);
        if (lane == 0) b.bmask[k][wave] = m;
    }
    if (lane == 0) b.flag[wave] = flag;
    __syncthreads();
    if (tid < 16) {
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            b.pre[tid][w] = (unsigned short)run;
            run += (unsigned)__popcll(b.bmask[tid][w]);
        }
        b.total[tid] = (unsigned short)run;
    }
    __syncthreads();
    unsigned mk = s.mask;
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (mk) {
        const int k = __builtin_ctz(mk);
        mk &= mk - 1;
        const unsigned pos = (unsigned)b.pre[k][wave] + (unsigned)__popcll(b.bmask[k][wave] & lt);
        b.list[k][pos] = (unsigned char)tid;
    }
    __syncthreads();
}

template <int FP>
struct Entry {
    float4 ga;
    float mux, muy;
    unsigned id;
    float feat[FP];
};

template <int FP>
__device__ __forceinline__ void read_entry(const Batch<FP> &b, int e, Entry<FP> &o) {
    constexpr int R4 = Batch<FP>::R4;
    const float4 *r = b.rec + e * R4;
    o.ga = r[0];
#pragma unroll
    for (int v = 0; v < FP / 4; ++v) {
        const float4 t = r[1 + v];
        o.feat[4 * v] = t.x; o.feat[4 * v + 1] = t.y; o.feat[4 * v + 2] = t.z; o.feat[4 * v + 3] = t.w;
    }
    const float4 m = r[R4 - 1];
    o.mux = m.x; o.muy = m.y; o.id = __float_as_uint(m.z);
}

__device__ __forceinline__ int block_tile(int per_xcd, int T) {
    const int b = blockIdx.x, slot = b >> 3;
    const int tile = (b & 7) * per_xcd + slot;
    return (slot < per_xcd && tile < T) ? tile : -1;
}

// max of an int over the four rows of the wave (value is row-uniform): wave-uniform result
__device__ __forceinline__ int wave_max_of_rows(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x141>(v));
    v = max(v, dpp_u32<0x140>(v));
    return v;
}

// Packed reduction of four values over a 16-lane row: afterwards every lane of quad q (lanes 4q .. 4q+3 of the row)
// holds the ROW sum of value row_value(q) = {0, 2, 1, 3}[q].
//   level 1 (lanes 8 apart, row_ror:8): quads 0,1 collect v0 (v2), quads 2,3 collect v1 (v3)
//   level 2 (neighbouring quads, row_half_mirror): quad 0 <- v0, quad 2 <- v1 of r01; quad 1 <- v2, quad 3 <- v3 of r23
//   levels 3, 4: inside the quad
__device__ __forceinline__ float row_reduce4_packed(float v0, float v1, float v2, float v3, bool hi, bool odd) {
    const float s01 = hi ? v1 : v0, t01 = hi ? v0 : v1;
    const float r01 = s01 + dpp_f32<0x128>(t01);
    const float s23 = hi ? v3 : v2, t23 = hi ? v2 : v3;
    const float r23 = s23 + dpp_f32<0x128>(t23);
    const float s = odd ? r23 : r01, t = odd ? r01 : r23;
    float r = s + dpp_f32<0x141>(t);
    r += dpp_f32<0xB1>(r);
    r += dpp_f32<0x4E>(r);
    return r;
}

constexpr int popcount_c(unsigned m) { return m == 0 ? 0 : (int)(m & 1u) + popcount_c(m >> 1); }
constexpr int nth_set_bit(unsigned m, int n) {
    int idx = 0;
    while (true) {
        if (m & 1u) { if (n == 0) return idx; --n; }
        m >>= 1; ++idx;
        if (idx > 31) return -1;
    }
}

// ---------------------------------------------------------------------------
// K6 forward composite
// ---------------------------------------------------------------------------
template <int C, int CS, bool WITH_DEPTH, bool SORT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void render_forward_kernel(SplatCamera cam, const float *colors, SplatState st,
                                                             float *out_color, float *out_depth, int T, int per_xcd) {
    constexpr int F = C + (WITH_DEPTH ? 1 : 0);
    constexpr int FP = (F + 3) / 4 * 4;
    __shared__ Batch<FP> B;
    __shared__ uint64_t s_keys[SORT ? kFusedSortMax : 1];
    const int tile = block_tile(per_xcd, T);
    if (tile < 0) return;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane >> 4, l16 = lane & 15;
    const int tx = tile % gx, ty = tile / gx;
    const int bcol = (wave & 1) * 2 + (row & 1), brow = (wave >> 1) * 2 + (row >> 1);      // block of this row in the tile
    const int blk = brow * 4 + bcol;
    const int px = tx * kTile + bcol * 4 + (l16 & 3), py = ty * kTile + brow * 4 + (l16 >> 2);
    const float fpx = (float)px, fpy = (float)py;
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    const bool inside = px < W && py < H;
    float Tr = 1.f, D = 0.f, Cc[C];
    unsigned last = 0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) Cc[ch] = 0.f;
    bool done = !inside;
    bool wdone = __builtin_amdgcn_ballot_w64(done) == ~0ull;

    unsigned lo;
    int n;
    tile_range(st, tile, lo, n);
    if constexpr (SORT) {
        if (n > kFusedSortMax) {
            if (tid == 0) st.status[3] = 1;
            n = kFusedSortMax;
        }
        for (int i = tid; i < n; i += 256) s_keys[i] = st.keys[lo + i];
        __syncthreads();
        if (n > 1) bitonic_sort(s_keys, n, tid, 256);
        for (int i = tid; i < n; i += 256) st.point_list[lo + i] = (uint32_t)s_keys[i];
    }
    const uint64_t *lk = SORT ? s_keys : nullptr;
    const int nb = (n + kBatch - 1) / kBatch;

    if (nb > 0) {
        Staged<FP> pre;
        gather<C, CS, WITH_DEPTH, FP>(pre, st, colors, lo + tid, tid < n, tile_x0, tile_y0, lk, tid);
        for (int bi = 0; bi < nb; ++bi) {
            if (bi > 0) __syncthreads();
            commit(B, pre, tid, wdone ? 1u : 0u);
            const unsigned alldone = B.flag[0] & B.flag[1] & B.flag[2] & B.flag[3];
            if (__builtin_amdgcn_readfirstlane((int)alldone)) break;
            if (bi + 1 < nb) {
                const int e = (bi + 1) * kBatch + tid;
                gather<C, CS, WITH_DEPTH, FP>(pre, st, colors, lo + e, e < n, tile_x0, tile_y0, lk, e);
            }
            const unsigned base1 = (unsigned)(bi * kBatch + 1);
            const int n_r = (int)B.total[blk];
            const int nmax = wave_max_of_rows(n_r);
            const unsigned char *mylist = B.list[blk];
            // software pipeline: the list index is read two steps ahead and the record one step ahead, so that a step
            // never waits for the (dependent) pair of LDS round trips
            int e = (int)mylist[0], e_nxt = (int)mylist[1];
            Entry<FP> cur;
            read_entry(B, e, cur);
#pragma unroll 1
            for (int i = 0; i < nmax && !wdone; ++i) {
                const bool act = i < n_r;
                Entry<FP> nxt;
                read_entry(B, e_nxt, nxt);
                const int e_n2 = (int)mylist[min(i + 2, kBatch - 1)];
                const float dx = cur.mux - fpx, dy = cur.muy - fpy;
                const float p2 = dx * (cur.ga.x * dx + cur.ga.y * dy) + cur.ga.z * dy * dy;     // power * log2(e)
                const float alpha = fminf(kAlphaMax, cur.ga.w * fast_exp2(p2));
                const bool live = act && !done && p2 <= 0.f && alpha >= kAlphaMin;
                if (__builtin_amdgcn_ballot_w64(live) != 0) {
                    const float test_T = Tr * (1.f - alpha);
                    const bool stop = live && test_T < kTStop;
                    const bool upd = live && !stop;
                    const float wgt = upd ? alpha * Tr : 0.f;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) Cc[ch] += cur.feat[ch] * wgt;
                    if constexpr (WITH_DEPTH) D += cur.feat[C] * wgt;
                    Tr = upd ? test_T : Tr;
                    last = upd ? base1 + (unsigned)e : last;
                    done = done || stop;
                    if (__builtin_amdgcn_ballot_w64(done) == ~0ull) wdone = true;
                }
                cur = nxt;
                e = e_nxt;
                e_nxt = e_n2;
            }
        }
    }
    if (inside) {
        const size_t HW = (size_t)H * W;
        const size_t pix = (size_t)py * W + px;
        st.final_T[pix] = Tr;
        st.n_contrib[pix] = (int)last;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) out_color[ch * HW + pix] = Cc[ch] + Tr * cam.bg[ch];
        if constexpr (WITH_DEPTH) out_depth[pix] = D;
    }
}

// ---------------------------------------------------------------------------
// K7 backward composite
// ---------------------------------------------------------------------------
template <int C, int CS, unsigned DMASK, unsigned SMASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void render_backward_kernel(SplatCamera cam, const float *colors, SplatState st,
                                                              const float *dL_dcolor, float *accum, int T, int per_xcd) {
    constexpr int FP = (C + 3) / 4 * 4;
    constexpr int NS = popcount_c(SMASK);
    constexpr int NV = 6 + NS;                // partial sums per Gaussian
    constexpr int NG = (NV + 3) / 4;          // packed reduction groups
    __shared__ Batch<FP> B;
    __shared__ unsigned s_wmax[4];
    const int tile = block_tile(per_xcd, T);
    if (tile < 0) return;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane >> 4, l16 = lane & 15, quad = l16 >> 2;
    const int tx = tile % gx, ty = tile / gx;
    const int bcol = (wave & 1) * 2 + (row & 1), brow = (wave >> 1) * 2 + (row >> 1);
    const int blk = brow * 4 + bcol;
    const int px = tx * kTile + bcol * 4 + (l16 & 3), py = ty * kTile + brow * 4 + (l16 >> 2);
    const float fpx = (float)px, fpy = (float)py;
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;

    const float Tfin = inside ? st.final_T[pix] : 0.f;
    float Tr = Tfin;
    const unsigned last = inside ? (unsigned)st.n_contrib[pix] : 0u;
    float dpix[C], bgdot = 0.f, behind = 0.f, lcdot = 0.f, lalpha = 0.f;
    bool has_bg = false;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        dpix[ch] = 0.f;
        if ((DMASK >> ch) & 1u) {
            has_bg |= cam.bg[ch] != 0.f;
            dpix[ch] = inside ? dL_dcolor[ch * HW + pix] : 0.f;
            bgdot += cam.bg[ch] * dpix[ch];
        }
    }
    const unsigned rmax = row_max_u32(last);                      // deepest contributor of this block (row-uniform)
    const unsigned wmax = (unsigned)wave_max_of_rows((int)rmax);
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const unsigned tmax = (unsigned)__builtin_amdgcn_readfirstlane((int)max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));
    if (tmax == 0) return;                                     // uniform over the workgroup
    const unsigned lo = st.tile_stride > 0 ? (unsigned)tile * (unsigned)st.tile_stride : st.tile_base[tile];
    const int nb = (int)((tmax + kBatch - 1) / kBatch);
    const bool hi = quad >= 2, odd = (quad & 1) != 0;
    // Publish: after the packed row reductions every lane of quad q holds value row_value(q) of each group; lane 4 q + g
    // of the row takes group g's value, so that ONE global_atomic_add_f32 per step carries all sums of the four rows'
    // Gaussians to their 64-byte accumulator lines (<= 4 line-requests per step; the L2 atomic units retire ~21
    // line-requests per ns however many floats of a line a request carries: scripts/micro/atomic_bench.hip).
    const int pub_g = l16 & 3;
    int doff = -1;
#pragma unroll
    for (int grp = 0; grp < NG; ++grp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * grp + row_value(q);
            const int slot = k < 6 ? k : (k < NV ? 6 + nth_set_bit(SMASK, k - 6 < 0 ? 0 : k - 6) : -1);
            if (pub_g == grp && quad == q) doff = slot;
        }
    const bool pub = doff >= 0;

    Staged<FP> pre;
    {
        const unsigned e = (unsigned)((nb - 1) * kBatch + tid);
        gather<C, CS, false, FP>(pre, st, colors, lo + e, e < tmax, tile_x0, tile_y0);
    }
    for (int bi = nb - 1; bi >= 0; --bi) {
        // the previous iteration ended with a barrier
        commit(B, pre, tid, 0u);
        if (bi > 0) gather<C, CS, false, FP>(pre, st, colors, lo + (unsigned)((bi - 1) * kBatch + tid), true, tile_x0, tile_y0);
        const int base = bi * kBatch;
        // entries [0, lim) of this batch can matter to this block: count them in the block's list
        const int lim = (int)rmax - base;
        int n_eff = 0;
        if (lim >= kBatch) {
            n_eff = (int)B.total[blk];
        } else if (lim > 0) {
            const int gw = lim >> 6;
            const unsigned long long m = B.bmask[blk][gw] & ((1ull << (lim & 63)) - 1ull);
            n_eff = (int)B.pre[blk][gw] + (int)__popcll(m);
        }
        const int nmax = wave_max_of_rows(n_eff);
        const unsigned char *mylist = B.list[blk];
        // software pipeline (see the forward kernel): walk the list from its tail, index two steps / record one step ahead
        int e = (int)mylist[max(n_eff - 1, 0)], e_nxt = (int)mylist[max(n_eff - 2, 0)];
        Entry<FP> cur;
        read_entry(B, e, cur);
#pragma unroll 1
        for (int i = 0; i < nmax; ++i) {
            const bool act = i < n_eff;
            Entry<FP> nxt;
            read_entry(B, e_nxt, nxt);
            const int e_n2 = (int)mylist[max(n_eff - 3 - i, 0)];
            const unsigned pos = (unsigned)(base + e + 1);
            const float dx = cur.mux - fpx, dy = cur.muy - fpy;
            const float p2 = dx * (cur.ga.x * dx + cur.ga.y * dy) + cur.ga.z * dy * dy;
            const float G = fast_exp2(p2);
            const float alpha = fminf(kAlphaMax, cur.ga.w * G);
            const bool live = act && pos <= last && p2 <= 0.f && alpha >= kAlphaMin;
            const unsigned long long live_m = __builtin_amdgcn_ballot_w64(live);
            if (live_m != 0) {
                const float rcp = __builtin_amdgcn_rcpf(1.f - alpha);
                const float Tn = Tr * rcp;                     // transmittance in front of this Gaussian
                float cdot = 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                    if ((DMASK >> ch) & 1u) cdot += cur.feat[ch] * dpix[ch];
                const float bh = lalpha * lcdot + (1.f - lalpha) * behind;
                float dL_dalpha = (cdot - bh) * Tn;
                if (has_bg) dL_dalpha += (-Tfin * rcp) * bgdot;
                // selects (not multiplies by 0) so that a non-live lane can never inject inf * 0
                const float Gl = live ? G : 0.f;
                const float wgt = live ? alpha * Tn : 0.f;
                const float q = live ? cur.ga.w * dL_dalpha : 0.f;   // dL/dG
                const float gdx = Gl * dx, gdy = Gl * dy;
                const float qgx = q * gdx, qgy = q * gdy;
                float s[NG * 4];
#pragma unroll
                for (int v = NV; v < NG * 4; ++v) s[v] = 0.f;
                s[0] = qgx;
                s[1] = qgy;
                s[2] = qgx * dx;
                s[3] = qgx * dy;
                s[4] = qgy * dy;
                s[5] = Gl * dL_dalpha;
#pragma unroll
                for (int n = 0; n < NS; ++n) s[6 + n] = wgt * dpix[nth_set_bit(SMASK, n)];
                Tr = live ? Tn : Tr;
                behind = live ? bh : behind;
                lcdot = live ? cdot : lcdot;
                lalpha = live ? alpha : lalpha;
                float r[NG];
#pragma unroll
                for (int grp = 0; grp < NG; ++grp)
                    r[grp] = row_reduce4_packed(s[4 * grp], s[4 * grp + 1], s[4 * grp + 2], s[4 * grp + 3], hi, odd);
                // rows with a live lane add their sums to their Gaussian's accumulator line
                const bool row_live = ((unsigned)(live_m >> (16 * row)) & 0xffffu) != 0u;
                float pv = r[0];
#pragma unroll
                for (int grp = 1; grp < NG; ++grp) pv = pub_g == grp ? r[grp] : pv;
                if (pub && row_live) atomicAdd(accum + (size_t)cur.id * SPLAT_GRAD_STRIDE + doff, pv);
            }
            cur = nxt;
            e = e_nxt;
            e_nxt = e_n2;
        }
        __syncthreads();            // every wave has finished reading this batch
    }
}

template <int C, int CS, bool WITH_DEPTH, bool SORT = false>
static void launch_fwd(const SplatCamera &cam, const float *colors, SplatState &st, float *oc, float *od, int T, hipStream_t s) {
    const int per = (T + 7) / 8;
    hipLaunchKernelGGL((render_forward_kernel<C, CS, WITH_DEPTH, SORT>), dim3(8 * per), dim3(256), 0, s, cam, colors, st, oc, od, T, per);
}
template <int C, int CS, unsigned DMASK = (1u << C) - 1u, unsigned SMASK = (1u << C) - 1u>
static void launch_bwd(const SplatCamera &cam, const float *colors, const SplatState &st, const float *dl, float *acc, int T,
                       hipStream_t s) {
    const int per = (T + 7) / 8;
    hipLaunchKernelGGL((render_backward_kernel<C, CS, DMASK, SMASK>), dim3(8 * per), dim3(256), 0, s, cam, colors, st, dl, acc, T, per);
}

}  // namespace v4

hipError_t launch_render_forward_v4(const SplatCamera &cam, const float *col, int channels, SplatState &st, float *out_color,
                                    float *out_depth, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    switch (channels) {
        case 1: v4::launch_fwd<1, 1, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 2: v4::launch_fwd<2, 2, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 3: v4::launch_fwd<3, 3, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 4: v4::launch_fwd<4, 4, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 5: v4::launch_fwd<5, 5, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 6: v4::launch_fwd<6, 6, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 7: v4::launch_fwd<7, 7, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 8: v4::launch_fwd<8, 8, true>(cam, col, st, out_color, out_depth, T, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_render_backward_v4(const SplatCamera &cam, const float *col, int channels, const SplatState &st,
                                     const float *dL_dcolor, float *accum, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    switch (channels) {
        case 1: v4::launch_bwd<1, 1>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 2: v4::launch_bwd<2, 2>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 3: v4::launch_bwd<3, 3>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 4: v4::launch_bwd<4, 4>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 5: v4::launch_bwd<5, 5>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 6: v4::launch_bwd<6, 6>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 7: v4::launch_bwd<7, 7>(cam, col, st, dL_dcolor, accum, T, s); break;
        case 8: v4::launch_bwd<8, 8>(cam, col, st, dL_dcolor, accum, T, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_render_forward_feat8_v4(const SplatCamera &cam, const float *feat8, SplatState &st, float *out6, bool sort_in_kernel,
                                          hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    if (sort_in_kernel) v4::launch_fwd<6, 8, false, true>(cam, feat8, st, out6, nullptr, T, s);
    else v4::launch_fwd<6, 8, false, false>(cam, feat8, st, out6, nullptr, T, s);
    return hipGetLastError();
}

hipError_t launch_render_backward_feat8_v4(const SplatCamera &cam, const float *feat8, const SplatState &st, const float *dL_dout6,
                                           float *accum, bool rgb_sums, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    if (rgb_sums) v4::launch_bwd<6, 8, 0xFu, 0xFu>(cam, feat8, st, dL_dout6, accum, T, s);
    else v4::launch_bwd<6, 8, 0xFu, 0x8u>(cam, feat8, st, dL_dout6, accum, T, s);
    return hipGetLastError();
}

}  // namespace splat
