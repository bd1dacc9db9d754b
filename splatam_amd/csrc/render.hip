// render.hip -- tile-wise alpha compositing, forward (K6) and backward (K7).
//
// MI355X mapping (forward: third generation, backward: fifth; the earlier / rejected ones are in the git history, their measurements in
// profiles/r0*_experiments.md):
//   * one 256-thread workgroup per 16x16 tile, ONE WAVE64 PER 8x8 QUADRANT, one pixel per lane.  A frame of
//     config B is 3 225 tiles = 12 900 waves = 12.6 per SIMD, so every SIMD always has several waves to
//     interleave (the one-wave-per-tile kernels were latency bound at ~3 waves per SIMD);
//   * the tile's depth-sorted list is staged 256 Gaussians at a time into double-buffered LDS (each thread
//     gathers one Gaussian's record from the L2-resident SoA arrays; the next batch is in flight while the
//     current one is composited; one barrier per batch);
//   * exact quadrant culling, resolved at staging time: the gathering thread computes which of the four
//     quadrants can hold a pixel with alpha >= 1/255 (axis-aligned box of {power >= -ln(255 o)}, inflated by a
//     rounding margin), the four 64-bit ballots per gathering wave go to LDS, and each compositing wave walks
//     ONLY the set bits of its own quadrant's 256-bit mask (s_ff1 / s_flbit on SGPRs).  A Gaussian that cannot
//     touch a quadrant costs that wave nothing; results are unchanged (a culled pixel fails the alpha test);
//   * inner loop: the record of the NEXT visit is fetched (uniform-address, broadcast ds_read_b128 / b64) while the current one is
//     composited (two register sets, ping-pong), so that a visit's critical path is instruction issue alone; loop control and
//     the per-lane predicates (done / live / stop) live in scalar registers as 64-bit masks;
//   * backward (generation 5): no cross-lane reduction per visit -- phase 1 (lane = pixel) leaves two numbers per live pixel
//     in a per-wave LDS pair buffer, phase 2 (every 8 visits; lane = (visit, pixel row)) turns them into the 6 + C sums with
//     dense FMAs and three DPP adds, and every accumulator line of a visit leaves the wave in ONE float atomic instruction;
//   * with SORT the forward kernel builds its tile's list itself: it filters its 2x2-tile group's records (or reads its bucket),
//     sorts the <= 1 024 keys in LDS in two levels (rank sort of chunks, then interleaved binary searches) and publishes ids and
//     count for the backward pass;
//   * blockIdx -> tile map gives each XCD (block b runs on XCD b % 8) a contiguous band of tiles, so that the
//     Gaussians shared by neighbouring tiles are served by one XCD's L2.
//
// Arithmetic: SURVEY.md Appendix A "Forward composite (K6)" / "Backward composite (K7)" -- the callee of
// /root/reference/scripts/splatam.py:249,253 and of the autograd backward reached from :702,854.
// exp(power) is evaluated as v_exp_f32(power * log2 e) with log2 e folded into the staged conic.
#include <type_traits>

#include "splat_device.h"

#ifndef SPLAT_BLOCK_RADIAL
#define SPLAT_BLOCK_RADIAL 1       // the forward composite's 4x4-block lists: radial test per block on top of the box test (r06_experiments.md 9)
#endif

namespace splat {

constexpr int kBatch = 256;         // records per LDS buffer (one per thread)
constexpr int kBatchEntries = 255;  // list entries staged per buffer: record 255 is the INERT record (zero opacity) that the padding of the
                                    // visit lists points to
constexpr int kSegBytes = 68;       // one visit list: <= 64 one-byte entries, padded with 0xFF to a multiple of 4, + the next group's look-ahead
constexpr int kBlockListBytes = 264; // a block's visit list over a whole batch: <= 255 entries + padding + look-ahead


// One staged Gaussian as the gathering thread holds it in registers.
template <int FP>
struct Staged {
    float4 ga;          // A = -0.5*cxx*log2e, B = -cxy*log2e, Cq = -0.5*cyy*log2e, opacity
    float2 mu;          // pixel centre
    float feat[FP];     // colours (then depth in the forward of the reference API), zero padded
    unsigned mask;      // 4-bit quadrant mask
    unsigned id;        // Gaussian index
    float aux;          // travels in the spare word of the record's last float4 (COMPACT6 forward records: depth^2)
};

// colours of Gaussian `id`: C floats at colors + id*CS (CS = record stride in floats; CS % 4 == 0 records are
// 16-byte aligned and read as float4)
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

// NL: visit lists per tile the staged mask addresses -- 4: one per 8x8 quadrant (bit q);  16: one per 4x4-PIXEL BLOCK, bit
// 4 * quadrant + block-in-quadrant (block r of quadrant w: columns 4 (2 (w & 1) + (r & 1)) .., rows 4 (2 (w >> 1) + (r >> 1)) ..)
// AUXZ2: s.aux = feat[3]^2 (the six channels of the fused iteration -- r, g, b, z, 1, z^2 -- from the FIRST float4 of the 8-float record:
// the silhouette colour is the constant 1, and z^2 is the product the per-Gaussian kernel stored, formed again from the same float)
template <int C, int CS, bool WITH_DEPTH, int FP, int NL = 4, bool AUXZ2 = false>
__device__ __forceinline__ void gather(Staged<FP> &s, const SplatState &st, const float *colors, unsigned idx, bool valid,
                                       float tile_x0, float tile_y0, const uint64_t *lds_keys = nullptr, int lds_idx = 0) {
    s.ga = make_float4(0.f, 0.f, 0.f, 0.f);
    s.mu = make_float2(0.f, 0.f);
#pragma unroll
    for (int f = 0; f < FP; ++f) s.feat[f] = 0.f;
    s.mask = 0;
    s.id = 0;
    s.aux = 0.f;
    if (valid) {
        // sorted id: from the tile's keys sorted in LDS by this workgroup (low 32 bits), or from the global list
        const unsigned id = lds_keys ? (unsigned)lds_keys[lds_idx] : st.point_list[idx];
        const float4 co = reinterpret_cast<const float4 *>(st.conic_opacity)[id];
        const float2 mu = reinterpret_cast<const float2 *>(st.xy)[id];
        load_colors<C, CS>(colors, id, s.feat);
        if constexpr (WITH_DEPTH) s.feat[C] = st.depth[id];
        // live region {alpha >= 1/255}: (p-mu)^T Q (p-mu) <= 2 tau, tau = ln(255 o); half extents sqrt(2 tau Q^-1_ii)
        unsigned mask = 0;
        // (the threshold itself is decided on the opacity, as the visit decides it -- alpha <= opacity --: 255 o may round below 1 for
        //  o == 1/255, and __logf is approximate)
        const float tau2 = fmaxf(2.0f * __logf(255.0f * co.w), 0.f);
        if (co.w >= kAlphaMin) {
            const float det = co.x * co.z - co.y * co.y;
            const float hx = sqrtf(tau2 * co.z / det) * 1.00001f + 0.01f;
            const float hy = sqrtf(tau2 * co.x / det) * 1.00001f + 0.01f;
            const bool x0 = (mu.x - hx <= tile_x0 + 7.f) && (mu.x + hx >= tile_x0);
            const bool x1 = (mu.x - hx <= tile_x0 + 15.f) && (mu.x + hx >= tile_x0 + 8.f);
            const bool y0 = (mu.y - hy <= tile_y0 + 7.f) && (mu.y + hy >= tile_y0);
            const bool y1 = (mu.y - hy <= tile_y0 + 15.f) && (mu.y + hy >= tile_y0 + 8.f);
            mask = (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
            // radial test on top of the box test: d^T Q d >= lambda_min |d|^2, with |d| the distance from the centre to the
            // quadrant's box of pixel centres; exact for round splats (cuts the box's corners), conservative otherwise
            const float mid = 0.5f * (co.x + co.z);
            const float lam_min = mid - sqrtf(fmaxf(0.f, mid * mid - det));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float bx0 = tile_x0 + 8.f * (q & 1), by0 = tile_y0 + 8.f * (q >> 1);
                const float ddx = fmaxf(fmaxf(bx0 - mu.x, mu.x - (bx0 + 7.f)), 0.f);
                const float ddy = fmaxf(fmaxf(by0 - mu.y, mu.y - (by0 + 7.f)), 0.f);
                if (lam_min * (ddx * ddx + ddy * ddy) > tau2 * 1.001f + 1e-3f) mask &= ~(1u << q);
            }
            if (!(hx == hx) || !(hy == hy) || !(lam_min == lam_min)) mask = 15u;     // NaN geometry: no culling, let it propagate as the reference would
            if constexpr (NL == 16) {
                // per 4x4 block: the box test on the block's columns / rows, the radial test of the block's quadrant
                const bool nan_geo = !(hx == hx) || !(hy == hy) || !(lam_min == lam_min);
                unsigned xh = 0, yh = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    xh |= ((mu.x - hx <= tile_x0 + 4.f * b + 3.f) && (mu.x + hx >= tile_x0 + 4.f * b)) ? (1u << b) : 0u;
                    yh |= ((mu.y - hy <= tile_y0 + 4.f * b + 3.f) && (mu.y + hy >= tile_y0 + 4.f * b)) ? (1u << b) : 0u;
                }
#if SPLAT_BLOCK_RADIAL
                // the radial test per BLOCK as well (exact for round splats -- every SplaTAM map): the box test alone visits the blocks in
                // the corners of the splat's bounding box, 6 % of the block visits at workload B, 8 % on the frame loop's map
                // (scripts/fwd_balance_stats.py).  lambda_min * distance^2 to the block's box of pixel centres, separable in x and y
                unsigned rad16 = 0;         // bit 4 * by + bx: block column bx, block row by of the tile passes
                {
                    float ex[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float x0 = tile_x0 + 4.f * b;
                        const float ddx = fmaxf(fmaxf(x0 - mu.x, mu.x - (x0 + 3.f)), 0.f);
                        ex[b] = lam_min * ddx * ddx;
                    }
                    const float thr = tau2 * 1.001f + 1e-3f;
#pragma unroll
                    for (int by = 0; by < 4; ++by) {
                        const float y0 = tile_y0 + 4.f * by;
                        const float ddy = fmaxf(fmaxf(y0 - mu.y, mu.y - (y0 + 3.f)), 0.f);
                        const float left = thr - lam_min * ddy * ddy;         // what the block row leaves for the x term
#pragma unroll
                        for (int bx = 0; bx < 4; ++bx) rad16 |= (ex[bx] > left) ? 0u : (1u << (4 * by + bx));
                    }
                    if (nan_geo) rad16 = 0xFFFFu;
                }
#endif
                // list L = 4 w + r: set when quadrant w passed the tests above and block r of it ((r & 1, r >> 1) within the quadrant)
                // is hit in x and in y
                unsigned m16 = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned xq = nan_geo ? 3u : (xh >> (2 * (w & 1))) & 3u, yq = nan_geo ? 3u : (yh >> (2 * (w >> 1))) & 3u;
                    unsigned sub = ((yq & 1u) ? xq : 0u) | ((yq & 2u) ? (xq << 2) : 0u);
#if SPLAT_BLOCK_RADIAL
                    // quadrant w's blocks: tile block columns 2 (w & 1) .., rows 2 (w >> 1) ..
                    const unsigned r0 = (rad16 >> (4 * (2 * (w >> 1)) + 2 * (w & 1))) & 3u, r1 = (rad16 >> (4 * (2 * (w >> 1) + 1) + 2 * (w & 1))) & 3u;
                    sub &= r0 | (r1 << 2);
#endif
                    m16 |= (((mask >> w) & 1u) ? sub : 0u) << (4 * w);
                }
                mask = m16;
            }
        }
        s.ga = make_float4(-0.5f * kLog2e * co.x, -kLog2e * co.y, -0.5f * kLog2e * co.z, co.w);
        s.mu = mu;
        s.mask = mask;
        s.id = id;
        if constexpr (AUXZ2) s.aux = s.feat[3] * s.feat[3];
    }
}

// LDS image of one batch: one record of R4 = FP/4 + 2 float4 per staged Gaussian
//   [0] A, B, Cq, opacity   [1 .. FP/4] features   [FP/4 + 1] mu_x, mu_y, id (bits), 0
// so that the compositing wave needs ONE address register and reads the record with FP/4 + 2 broadcast
// ds_read_b128 (the 48 / 64-byte record stride keeps the staging ds_write_b128 conflict free).
template <int FP, int NL = 4, bool BOTH = false>
struct Batch {
    static constexpr int R4 = FP / 4 + 2;
    float4 rec[kBatch * R4];
    unsigned qmask[4][4][2];        // [quadrant][gathering wave][lo, hi]: ballot of "this Gaussian can touch the quadrant"
    // VISIT LISTS [quadrant][gathering wave]: the set bits of qmask as one-byte batch entry indices, ascending, padded with 0xFF
    // (the inert record) to a multiple of four.  The composites walk these four entries at a time (one ds_read_b32, v_bfe_u32 per
    // entry, no per-visit scalar bookkeeping): the bit walk over qmask cost ~13 SALU instructions per visit, and on gfx950 every
    // instruction of whatever type takes a 2-cycle issue slot of its SIMD, scalar ones at most every 4th cycle
    // (profiles/r03_valu_issue_bench.txt) -- the composites are bound by the TOTAL instruction count.
    // One word of slack on either side: the look-ahead of the first / last group reads it.
    // NL = 16 (forward composite): one list per 4x4-pixel block, see gather().  The four 16-lane rows of a compositing wave walk
    // their blocks' lists IN STEP, so a trip count is the longest of four lists: those lists run over the whole batch (contiguous
    // over the gathering waves, kBlockListBytes each) -- per gathering wave the maximum of four ~15-entry pieces sat 30 % above
    // their mean.  The gathering waves exchange their per-list counts (vcnt, double buffered over the batches) one barrier
    // before they write: commit_counts() / commit()
    unsigned char vcnt[NL == 4 ? 1 : 2][NL][4];     // entries per list and gathering wave
    unsigned char vtot[NL];         // NL == 16: entries per list
    unsigned flag[4];               // forward: wave w had no pixel left when this batch was committed
    // BOTH (the fused forward + backward composite, NL == 16): the quadrant visit lists of the NL == 4 layout BESIDE the block lists --
    // the backward pass walks the batch the forward pass has just composited
    unsigned qlist[BOTH ? 1 + 16 * (kSegBytes / 4) + 1 : 1];
    // (last member: the fused composite lets its backward pass' pair buffer start HERE -- the block lists are dead by then)
    unsigned vlist[NL == 4 ? 1 + 16 * (kSegBytes / 4) + 1 : NL * (kBlockListBytes / 4)];
};

template <int FP, int NL, bool BOTH>
__device__ __forceinline__ const unsigned char *visit_list(const Batch<FP, NL, BOTH> &b, int list, int gwave) {
    static_assert(NL == 4 || BOTH, "quadrant lists: the NL == 4 layout, or the second set of a BOTH batch");
    return reinterpret_cast<const unsigned char *>((BOTH ? b.qlist : b.vlist) + 1) + (list * 4 + gwave) * kSegBytes;
}

template <int FP, bool BOTH>
__device__ __forceinline__ const unsigned char *block_list(const Batch<FP, 16, BOTH> &b, int list) {
    return reinterpret_cast<const unsigned char *>(b.vlist) + list * kBlockListBytes;
}

// NL == 16, step 1 (before the barrier that ends the previous batch): this gathering wave's entry count per block list
template <int FP, bool BOTH>
__device__ __forceinline__ void commit_counts(Batch<FP, 16, BOTH> &b, const Staged<FP> &s, int tid, int parity) {
    const int wave = tid >> 6, lane = tid & 63;
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64((s.mask >> q) & 1u));
        mine = lane == q ? (unsigned)cnt : mine;
    }
    if (lane < 16) b.vcnt[parity][lane][wave] = (unsigned char)mine;
}

template <int FP, int NL, bool BOTH>
__device__ __forceinline__ void commit(Batch<FP, NL, BOTH> &b, const Staged<FP> &s, int tid, unsigned flag, int parity = 0) {
    constexpr int R4 = Batch<FP, NL, BOTH>::R4;
    b.rec[tid * R4] = s.ga;
#pragma unroll
    for (int v = 0; v < FP / 4; ++v)
        b.rec[tid * R4 + 1 + v] = make_float4(s.feat[4 * v], s.feat[4 * v + 1], s.feat[4 * v + 2], s.feat[4 * v + 3]);
    b.rec[tid * R4 + R4 - 1] = make_float4(s.mu.x, s.mu.y, __uint_as_float(s.id), s.aux);
    const int wave = tid >> 6, lane = tid & 63;
    if constexpr (NL == 16) {
        // step 2 (after that barrier): entries at the offset the lower gathering waves leave; the lists of quadrant `wave` get
        // their totals and their 0xFF padding (the inert record) up to the quadrant's longest list + the look-ahead from this wave
        unsigned char *lists = reinterpret_cast<unsigned char *>(b.vlist);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool hit = (s.mask >> q) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned c4 = *reinterpret_cast<const unsigned *>(b.vcnt[parity][q]);           // counts of gathering waves 0..3
            const unsigned below = c4 & ((1u << (8 * wave)) - 1u);                                 // (wave 0: none)
            const int off = (int)((below & 0xFFu) + ((below >> 8) & 0xFFu) + ((below >> 16) & 0xFFu));
            if (hit) lists[q * kBlockListBytes + off + rank] = (unsigned char)tid;
        }
        const unsigned *c4p = reinterpret_cast<const unsigned *>(b.vcnt[parity][wave * 4]);
        int tot[4], tmax = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned c4 = c4p[r];
            tot[r] = (int)((c4 & 0xFFu) + ((c4 >> 8) & 0xFFu) + ((c4 >> 16) & 0xFFu) + (c4 >> 24));
            tmax = max(tmax, tot[r]);
        }
        const int pad_end = ((tmax + 3) & ~3) + 4;                  // the trips run to the longest list, the look-ahead 4 further
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            for (int j = tot[r] + lane; j < pad_end; j += 64) lists[(wave * 4 + r) * kBlockListBytes + j] = 0xFFu;
            if (lane == 0) b.vtot[wave * 4 + r] = (unsigned char)tot[r];
        }
        if constexpr (BOTH) {
            // the quadrant lists of the backward pass: a Gaussian is on quadrant q's list when it is on the list of any of q's four blocks
            // (a subset of the NL == 4 quadrant test: a pixel with alpha >= 1/255 lies in a block whose box test passed)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool hit = ((s.mask >> (4 * q)) & 0xFu) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const int cnt = __builtin_popcountll(m);
                unsigned char *seg = const_cast<unsigned char *>(visit_list(b, q, wave));
                if (hit) seg[rank] = (unsigned char)tid;
                if (lane < 4) seg[cnt + lane] = 0xFFu;
                if (lane == 0) {
                    b.qmask[q][wave][0] = (unsigned)m;
                    b.qmask[q][wave][1] = (unsigned)(m >> 32);
                }
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const bool hit = (s.mask >> q) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int cnt = __builtin_popcountll(m);
            unsigned char *seg = const_cast<unsigned char *>(visit_list(b, q, wave));
            if (hit) seg[rank] = (unsigned char)tid;
            if (lane < 4) seg[cnt + lane] = 0xFFu;              // padding (cnt + 3 <= 67)
            if (lane == 0) {
                b.qmask[q][wave][0] = (unsigned)m;
                b.qmask[q][wave][1] = (unsigned)(m >> 32);
                b.vcnt[0][q][wave] = (unsigned char)cnt;
            }
        }
    }
    if (lane == 0) b.flag[wave] = flag;
}

// What the backward pass needs of a batch: the records and the QUADRANT lists.  An NL == 4 batch: commit(); a BOTH batch re-staged by
// the backward pass of the fused composite (s.mask holds the 4-bit quadrant mask of gather<.., NL = 4>)
template <int FP, int NL, bool BOTH>
__device__ __forceinline__ void commit_quadrants(Batch<FP, NL, BOTH> &b, const Staged<FP> &s, int tid) {
    if constexpr (!BOTH) {
        commit(b, s, tid, 0u);
    } else {
        constexpr int R4 = Batch<FP, NL, BOTH>::R4;
        b.rec[tid * R4] = s.ga;
#pragma unroll
        for (int v = 0; v < FP / 4; ++v)
            b.rec[tid * R4 + 1 + v] = make_float4(s.feat[4 * v], s.feat[4 * v + 1], s.feat[4 * v + 2], s.feat[4 * v + 3]);
        b.rec[tid * R4 + R4 - 1] = make_float4(s.mu.x, s.mu.y, __uint_as_float(s.id), s.aux);
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool hit = (s.mask >> q) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int cnt = __builtin_popcountll(m);
            unsigned char *seg = const_cast<unsigned char *>(visit_list(b, q, wave));
            if (hit) seg[rank] = (unsigned char)tid;
            if (lane < 4) seg[cnt + lane] = 0xFFu;
            if (lane == 0) {
                b.qmask[q][wave][0] = (unsigned)m;
                b.qmask[q][wave][1] = (unsigned)(m >> 32);
            }
        }
    }
}

// One staged Gaussian as a compositing wave holds it: the whole LDS record, fetched ONE VISIT AHEAD (the wave's critical path
// per visit is then instruction issue alone; with the record read inside the visit the dependent LDS round trips were
// exposed and the SIMDs sat ~40 % idle at the 4-5 waves each that the register / LDS budget allows).
template <int FP>
struct Rec {
    float4 a;               // A, B, Cq (conic * log2 e), opacity
    float4 f[FP / 4];       // colours
    float4 m;               // mu_x, mu_y, id (bits), -
};

// wave-uniform 64-bit word of this wave's quadrant mask
template <int FP, int NL, bool BOTH>
__device__ __forceinline__ unsigned long long mask_word(const Batch<FP, NL, BOTH> &b, int quadrant, int w) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)b.qmask[quadrant][w][0]);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)b.qmask[quadrant][w][1]);
    return ((unsigned long long)hi << 32) | lo;
}

// blockIdx -> tile: XCD x (blocks with b % 8 == x) owns tiles [x*per, (x+1)*per)
// `order` (SplatState.tile_order): the band's tiles in the order they should START (heaviest first), or NULL for the natural order.
// An entry holds tile + 1; 0 = "not written yet: the natural tile of this slot" (a zero-initialised buffer IS the natural order),
// 0xFFFFFFFF = no tile
__device__ __forceinline__ int block_tile(int per_xcd, int T, const uint32_t *order = nullptr) {
    const int b = blockIdx.x, slot = b >> 3;
    const int pos = (b & 7) * per_xcd + slot;
    if (slot >= per_xcd) return -1;
    if (order) {
        const unsigned t = order[pos];
        if (t != 0u) return t - 1u < (unsigned)T ? (int)(t - 1u) : -1;
    }
    return pos < T ? pos : -1;
}
// the launch order applies to whole frames (a band of tile rows is composited in the natural order)
__device__ __forceinline__ const uint32_t *launch_order(const SplatState &st) { return st.tile_row_end > st.tile_row_begin ? nullptr : st.tile_order; }

// background colour of channel ch; SplatCamera.bg == NULL: black
__device__ __forceinline__ float bg_of(const SplatCamera &cam, int ch) { return cam.bg ? cam.bg[ch] : 0.f; }

// Per-lane predicates of the inner loops are kept as wave-uniform 64-bit masks in SGPRs (ballot in, inverse
// ballot out): the kernels are instruction-issue bound, and masks in scalar registers cost one s_and / s_andn2
// where a per-lane bool in divergent control flow costs the compiler a chain of exec-mask bookkeeping.
__device__ __forceinline__ bool lane_of(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// ---------------------------------------------------------------------------
// K6 forward composite
// ---------------------------------------------------------------------------
constexpr int kFusedSortMax = 1024;     // longest list the composite sorts itself (8 KiB of LDS)

// Two-level sort of n <= kFusedSortMax UNIQUE keys by one 256-thread workgroup: A holds the keys (in: any order, A[n] = ~0 when n
// is odd; out: ascending), S is scratch for n keys (the batch buffer, idle before compositing starts).
//   level 1: the list is cut into C = 4 (n <= 512) or 8 chunks of <= 128 keys; wave w RANK-sorts chunk w (and w + 4): keys are
//            unique, so a key's position in its sorted chunk is the number of smaller keys in the chunk, counted against broadcast
//            ds_read_b128 of the chunk -- no barrier inside;
//   level 2: a key's final position is the sum over ALL chunks of the number of smaller keys there (its own chunk yields its
//            local rank): C independent 8-step binary searches per key, interleaved.
// ~250 instructions per wave at the ~220 entries of workload B (a full rank sort of the list: ~770; the bitonic network: ~36
// barrier-separated stages), two barriers.
#ifndef SPLAT_SORT_PAD
#define SPLAT_SORT_PAD 0       // 1: one pad key per 32 keys of the scratch array S (A/B of VERDICT r4 item 7: measured, no gain -- r05_experiments.md 4)
#endif
__device__ __forceinline__ int sort_pad(int i) { return SPLAT_SORT_PAD ? i + (i >> 5) : i; }
__device__ __forceinline__ void sort_keys_two_level(uint64_t *A, uint64_t *S, const int n, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int C = n <= 512 ? 4 : 8;
    const int cs = (((n + C - 1) / C) + 1) & ~1;            // chunk size: even (16-byte aligned pair reads), <= 128
    // ---- level 1
    for (int c = wave; c < C; c += 4) {
        const int start = c * cs, len = min(n, start + cs) - start;
        if (len <= 0) break;
        const uint64_t k0 = lane < len ? A[start + lane] : ~0ull;
        const uint64_t k1 = lane + 64 < len ? A[start + lane + 64] : ~0ull;
        unsigned r0 = 0, r1 = 0;
        const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(A + start);
        const int np = (len + 1) >> 1;
        if (len <= 64) {
            for (int j = 0; j < np; ++j) {
                const ulonglong2 ab = pairs[j];
                r0 += (ab.x < k0 ? 1u : 0u) + (ab.y < k0 ? 1u : 0u);
            }
        } else {
            for (int j = 0; j < np; ++j) {
                const ulonglong2 ab = pairs[j];
                r0 += (ab.x < k0 ? 1u : 0u) + (ab.y < k0 ? 1u : 0u);
                r1 += (ab.x < k1 ? 1u : 0u) + (ab.y < k1 ? 1u : 0u);
            }
        }
        // (a pair read of an odd-length chunk sees one key of the NEXT chunk or the ~0 pad: a key of the next chunk may be
        //  smaller than an own key, so the partner of the last pair is masked below)
        if (len & 1) {
            const uint64_t extra = A[start + len];
            r0 -= (lane < len && extra < k0) ? 1u : 0u;
            r1 -= (lane + 64 < len && extra < k1) ? 1u : 0u;
        }
        if (lane < len) S[sort_pad(start + (int)r0)] = k0;
        if (lane + 64 < len) S[sort_pad(start + (int)r1)] = k1;
    }
    __syncthreads();
    // ---- level 2: a key's rank = its index in its own chunk + its insertion points in the OTHER chunks.  The probes of one
    //      step are independent LDS reads (issued together, no branch around them: a probe past the chunk's end reads some
    //      other word of S, inside the LDS allocation, and is discarded by `probe <= len`); the step count follows the chunk size
    const int top = 1 << (31 - __builtin_clz((unsigned)cs));      // largest power of two <= cs (wave-uniform)
    for (int i = tid; i < n; i += 256) {
        const uint64_t key = S[sort_pad(i)];
        int oc = 0;
#pragma unroll
        for (int c = 1; c < 8; ++c) oc += (i >= c * cs) ? 1 : 0;
        unsigned r = (unsigned)(i - oc * cs);
        auto search = [&](auto CC) {
            constexpr int NC = decltype(CC)::value;
            unsigned pos[NC - 1];
            int base[NC - 1], len[NC - 1];
#pragma unroll
            for (int d = 1; d < NC; ++d) {
                const int start = ((oc + d) & (NC - 1)) * cs;
                pos[d - 1] = 0;
                base[d - 1] = start - 1;
                len[d - 1] = min(n, start + cs) - start;          // <= 0 for a chunk beyond the list
            }
#pragma unroll 1
            for (int step = top; step >= 1; step >>= 1) {
#pragma unroll
                for (int d = 0; d < NC - 1; ++d) {
                    const unsigned probe = pos[d] + (unsigned)step;
                    const uint64_t other = S[sort_pad(base[d] + (int)probe)];
                    pos[d] = (((int)probe <= len[d]) & (other < key)) ? probe : pos[d];
                }
            }
#pragma unroll
            for (int d = 0; d < NC - 1; ++d) r += pos[d];
        };
        if (C == 4) search(std::integral_constant<int, 4>{});
        else search(std::integral_constant<int, 8>{});
        A[r] = key;
    }
    __syncthreads();
}

// SORT: the workgroup first collects its tile's (depth, id) keys in LDS (from its bucket, or by filtering its group's records:
// SplatState.group_count), sorts them there (sort_keys_two_level) and publishes ids and count for the backward pass: no scan,
// scatter or sort launch, and the gathers take their ids from LDS instead of a dependent global load.
// The fused iteration's six channels (r, g, b, z, 1, z^2 read from 8-float records) travel as 48-byte LDS records -- r, g, b, z in the
// colour word, z^2 in the spare word of the centre word, the silhouette colour as the constant it is -- instead of 64-byte ones:
// three ds_read_b128 per visit instead of four, a quarter less LDS per workgroup, and half the colour gather.  Same sums, same order.
template <int C, int CS, bool WITH_DEPTH>
constexpr bool forward_compact6() { return C == 6 && CS == 8 && !WITH_DEPTH; }
template <int C, int CS, bool WITH_DEPTH>
constexpr int forward_fp() { return forward_compact6<C, CS, WITH_DEPTH>() ? 4 : (C + (WITH_DEPTH ? 1 : 0) + 3) / 4 * 4; }

// The forward composite of ONE tile up to its per-pixel results (the kernels below add their epilogues): builds / reads the tile's list,
// stages it batch by batch into B and composites front to back.  Returns the index of the LAST batch it staged (the one B still holds;
// -1: empty list).  Lane l of wave w = pixel (l & 3, (l >> 2) & 3) of block (l >> 4) of quadrant w (one 4x4 block per 16-lane row).
// WRITE_RECS: SplatState.tile_recs (when the caller gave one) receives the staged record of every list entry, for a SEPARATE backward
// composite (the fused forward + backward composite keeps gathering what it re-stages: measured -1 % with records at B-loop,
// profiles/r06_experiments.md 2)
template <int C, int CS, bool WITH_DEPTH, bool SORT, bool WRITE_RECS = true, class BatchT>
__device__ __forceinline__ int forward_tile(const float *colors, SplatState &st, BatchT &B, uint64_t *s_keys, const int tile, const int tx, const int ty,
                                            const int gx, const int tid, const float fpx, const float fpy, const bool inside, float &Tr, float &D,
                                            float (&Cc)[C], unsigned &last) {
    constexpr int F = C + (WITH_DEPTH ? 1 : 0);
    constexpr bool COMPACT6 = forward_compact6<C, CS, WITH_DEPTH>();
    constexpr int FP = forward_fp<C, CS, WITH_DEPTH>();
    static_assert(BatchT::R4 == FP / 4 + 2, "the batch holds this pass' records");
    constexpr int CG = COMPACT6 ? 4 : C;                        // channels the gather loads
    constexpr int NL = 16;
    const int lane = tid & 63, wave = tid >> 6;
    const int row = lane >> 4;                                  // this lane's block of the quadrant
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);      // wave-uniform: pixels with nothing left to composite
    bool wdone = done_m == ~0ull;
    int staged = -1;
    unsigned lo;
    int n;
    tile_range(st, tile, lo, n);
    const uint64_t *lk = SORT ? s_keys : nullptr;
    if constexpr (SORT) {
        if (st.group_stride > 0) {
            // group binning (SplatState.group_count): the tile's instances are the records of its 2 x 2-tile group whose tile
            // rectangle holds this tile; compacted into LDS in any order (the sort below fixes it), and the count is published
            // where the per-tile buckets would have left it (the backward composite and the counter fold read it there)
            __shared__ int s_n;
            if (tid == 0) s_n = 0;
            __syncthreads();
            const int ggx = (gx + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES;
            const int grp = (ty / SPLAT_GROUP_TILES) * ggx + tx / SPLAT_GROUP_TILES;
            const uint4 *recs = reinterpret_cast<const uint4 *>(st.group_recs) + (size_t)grp * st.group_stride;
            // The prologue is a chain of dependent round trips that every workgroup of the first round pays at once, with nothing to
            // hide it behind: the first two chunks of records are requested TOGETHER with the group's count (slots below the stride are
            // allocated, a group of workload B holds ~580 records), the rest as soon as the count is known -- two round trips for
            // what was one per chunk after the count's.
            constexpr int kChunks = 4;
            uint4 r[kChunks];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = k * 256 + tid;
                r[k] = i < st.group_stride ? recs[i] : make_uint4(0u, 0u, 0u, 0u);
            }
            const int cnt = min((int)st.group_count[(size_t)grp * SPLAT_COUNTER_STRIDE], st.group_stride);
#pragma unroll
            for (int k = 2; k < kChunks; ++k) {
                const int i = k * 256 + tid;
                r[k] = i < cnt ? recs[i] : make_uint4(0u, 0u, 0u, 0u);
            }
            const unsigned utx = (unsigned)tx, uty = (unsigned)ty;
            auto file = [&](const uint4 &rec, int i) {
                const bool hit = i < cnt && utx >= (rec.z & 0xFFFFu) && utx < (rec.w & 0xFFFFu) && uty >= (rec.z >> 16) && uty < (rec.w >> 16);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                int base = 0;
                if (lane == 0 && m) base = atomicAdd(&s_n, __builtin_popcountll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (hit && pos < kFusedSortMax) s_keys[pos] = ((uint64_t)rec.y << 32) | rec.x;
            };
#pragma unroll
            for (int k = 0; k < kChunks; ++k)
                if (k * 256 < cnt) file(r[k], k * 256 + tid);
            for (int i0 = kChunks * 256; i0 < cnt; i0 += 256) {
                const int i = i0 + tid;
                file(i < cnt ? recs[i] : make_uint4(0u, 0u, 0u, 0u), i);
            }
            __syncthreads();
            n = s_n;
            if (tid == 0) {
                // (a count that is flagged below is published CLAMPED: the backward composite must not walk past the published ids)
                st.tile_count[(size_t)tile * SPLAT_COUNTER_STRIDE] = (unsigned)min(n, min(st.tile_stride, kFusedSortMax));
                if (n > st.tile_stride) raise_status(st, 1);      // the published list would not fit its bucket
            }
            n = min(n, st.tile_stride);
        }
        if (n > kFusedSortMax) {        // the host's list-length hint was stale: flag it (the host repeats the iteration)
            if (tid == 0) {
                raise_status(st, 3);
                if (st.tile_stride > 0) st.tile_count[(size_t)tile * SPLAT_COUNTER_STRIDE] = (unsigned)kFusedSortMax;    // (see above)
            }
            n = kFusedSortMax;
        }
        if (st.group_stride == 0)
            for (int i = tid; i < n; i += 256) s_keys[i] = st.keys[lo + i];
        if (tid == 0 && (n & 1)) s_keys[n] = ~0ull;                  // pad to an even count (pair reads)
        __syncthreads();
        static_assert(sizeof(B.rec) >= (kFusedSortMax + 256 + (kFusedSortMax + 256) / 32 + 1) * sizeof(uint64_t), "level-2 probes of the sort may read up to 255 words past the last chunk");
        sort_keys_two_level(s_keys, reinterpret_cast<uint64_t *>(B.rec), n, tid);
        for (int i = tid; i < n; i += 256) st.point_list[lo + i] = (uint32_t)lk[i];
    }
    const int nb = (n + kBatchEntries - 1) / kBatchEntries;

    if (nb > 0) {
        // One LDS buffer (a tile's list is usually ONE batch; a second buffer would halve the resident workgroups):
        // the next batch's gather is in flight in registers while this one is composited, then barrier - commit - barrier.
        Staged<FP> pre;
        gather<CG, CS, WITH_DEPTH, FP, NL, COMPACT6>(pre, st, colors, lo + tid, tid < kBatchEntries && tid < n, tile_x0, tile_y0, lk, tid);
        commit_counts(B, pre, tid, 0);
        for (int bi = 0; bi < nb; ++bi) {
            __syncthreads();                        // every wave has finished reading the previous batch; this batch's list counts are in
            commit(B, pre, tid, wdone ? 1u : 0u, bi & 1);
            staged = bi;
            if constexpr (FP == 4 && WRITE_RECS) {
                // the staged record goes to memory as well (SplatState.tile_recs): the backward composite re-stages the list from ONE
                // coalesced 48-byte read per entry instead of the id -> conic / centre / colour gathers and the culling tests.  The
                // spare word carries the quadrant mask (a quadrant is visited when one of its four blocks is)
                if (st.tile_recs) {
                    const int e = bi * kBatchEntries + tid;
                    if (tid < kBatchEntries && e < n) {
                        unsigned qm = 0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) qm |= ((pre.mask >> (4 * q)) & 0xFu) ? (1u << q) : 0u;
                        float4 *dst = reinterpret_cast<float4 *>(st.tile_recs) + (size_t)(lo + (unsigned)e) * 3;
                        dst[0] = pre.ga;
                        dst[1] = make_float4(pre.feat[0], pre.feat[1], pre.feat[2], pre.feat[3]);
                        dst[2] = make_float4(pre.mu.x, pre.mu.y, __uint_as_float(pre.id), __uint_as_float(qm));
                    }
                }
            }
            __syncthreads();
            // every wave was finished when this batch was committed: the rest of the list cannot contribute
            const unsigned alldone = B.flag[0] & B.flag[1] & B.flag[2] & B.flag[3];
            if (__builtin_amdgcn_readfirstlane((int)alldone)) break;
            const bool more = bi + 1 < nb;
            if (more) {                         // next batch's gather stays in flight while this one is composited
                const int e = (bi + 1) * kBatchEntries + tid;
                gather<CG, CS, WITH_DEPTH, FP, NL, COMPACT6>(pre, st, colors, lo + e, tid < kBatchEntries && e < n, tile_x0, tile_y0, lk, e);
                commit_counts(B, pre, tid, (bi + 1) & 1);
            }
            const unsigned base1 = (unsigned)(bi * kBatchEntries + 1);
            unsigned last_loc = ~0u;            // record (byte offset in B.rec) of the pixel's last contributor in this batch (none yet)
            // a list entry -> its record's byte offset in B.rec: a wave-uniform value in a vector register (one byte-select shift per
            // visit, no scalar bookkeeping); the visit identifies its entry by that offset
            constexpr int R4 = BatchT::R4;
            auto rec_of = [&](unsigned e) { return e * (unsigned)(R4 * sizeof(float4)); };
            // one visit of this quadrant: `cur` was fetched from LDS during the previous visit (see Rec)
            auto visit = [&](unsigned rec, const Rec<FP> &cur) {
                const float dx = cur.m.x - fpx, dy = cur.m.y - fpy;
                const float p2 = dx * (cur.a.x * dx + cur.a.y * dy) + cur.a.z * dy * dy;     // power * log2(e)
                const float alpha = fminf(kAlphaMax, cur.a.w * fast_exp2(p2));
                const unsigned long long live_m = __builtin_amdgcn_ballot_w64(p2 <= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= kAlphaMin) & ~done_m;
                if (live_m == 0) return;
                const float test_T = Tr * (1.f - alpha);
                const unsigned long long stop_m = __builtin_amdgcn_ballot_w64(test_T < kTStop) & live_m;
                const bool upd = lane_of(live_m & ~stop_m);
                const float wgt = upd ? alpha * Tr : 0.f;
                if constexpr (COMPACT6) {
                    // r, g, b, z from the record's colour word; the silhouette colour is 1; z^2 rides in the centre word
                    Cc[0] += cur.f[0].x * wgt; Cc[1] += cur.f[0].y * wgt; Cc[2] += cur.f[0].z * wgt; Cc[3] += cur.f[0].w * wgt;
                    Cc[4] += wgt;
                    Cc[5] += cur.m.w * wgt;
                } else {
#pragma unroll
                    for (int ch = 0; ch < F; ++ch) {
                        const float4 &fv = cur.f[ch >> 2];
                        const float c = (ch & 3) == 0 ? fv.x : ((ch & 3) == 1 ? fv.y : ((ch & 3) == 2 ? fv.z : fv.w));
                        if (ch < C) Cc[ch] += c * wgt;
                        else D += c * wgt;
                    }
                }
                Tr = upd ? test_T : Tr;
                last_loc = upd ? rec : last_loc;
                done_m |= stop_m;                   // (rare: a pixel saturates)
            };
            auto load_rec = [&](unsigned rec, Rec<FP> &r) {
                const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(B.rec) + rec);
                r.a = p[0];
#pragma unroll
                for (int v = 0; v < FP / 4; ++v) r.f[v] = p[1 + v];
                r.m = p[R4 - 1];
            };
            // front to back over every row's block list, four entries per trip; the rows walk their own lists in step (each lane reads
            // its row's list and records: up to four addresses per instruction), a row whose list is shorter visits the inert record
            // (the 0xFF padding).  The NEXT record is in flight while the current one is composited (two register sets, ping-pong)
            const unsigned t4 = (unsigned)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<const unsigned *>(B.vtot + wave * 4));
            const int ng = (int)max(max(t4 & 0xFFu, (t4 >> 8) & 0xFFu), max((t4 >> 16) & 0xFFu, t4 >> 24));       // the longest of the four rows' lists
            if (ng > 0 && done_m != ~0ull) {
                const unsigned char *seg = block_list(B, wave * 4 + row);
                unsigned cur4 = *reinterpret_cast<const unsigned *>(seg);
                Rec<FP> ra, rb;
                unsigned r0 = rec_of(cur4 & 0xFFu);
                load_rec(r0, ra);
#pragma unroll 1
                for (int k = 0; k < ng; k += 4) {
                    const unsigned nxt4 = *reinterpret_cast<const unsigned *>(seg + k + 4);
                    const unsigned r1 = rec_of((cur4 >> 8) & 0xFFu);
                    load_rec(r1, rb);
                    visit(r0, ra);
                    const unsigned r2 = rec_of((cur4 >> 16) & 0xFFu);
                    load_rec(r2, ra);
                    visit(r1, rb);
                    const unsigned r3 = rec_of(cur4 >> 24);
                    load_rec(r3, rb);
                    visit(r2, ra);
                    r0 = rec_of(nxt4 & 0xFFu);
                    load_rec(r0, ra);
                    visit(r3, rb);
                    cur4 = nxt4;
                    if (done_m == ~0ull) break;     // every pixel of the quadrant is done: nothing left to visit
                }
            }
            last = last_loc != ~0u ? base1 + last_loc / (unsigned)(R4 * sizeof(float4)) : last;
            wdone = done_m == ~0ull;
        }
    }
    return staged;
}

// (the six fused channels on 48-byte records: 25 KB of LDS and 80 VGPRs -> six workgroups per CU; the reference API's forms stay at five)
#ifndef SPLAT_K6_WAVES
#define SPLAT_K6_WAVES 6
#endif
#ifndef SPLAT_K6_FP4_SIX
#define SPLAT_K6_FP4_SIX 1        // the reference API's forms with 48-byte records (<= 3 channels + depth) at six workgroups per CU too: 80 VGPRs, K6 63.8 -> 60.4 us at B (r06_experiments.md 8)
#endif
struct FwdArgs {
    SplatCamera cam;
    const float *colors;
    SplatState st;
    float *out_color, *out_depth;
    int T, per_xcd;
    TrackLossEpilogue ep;
};
template <int C, int CS, bool WITH_DEPTH, bool SORT, bool TRACK = false>
__global__ __launch_bounds__(256, ((forward_compact6<C, CS, WITH_DEPTH>() || (SPLAT_K6_FP4_SIX && forward_fp<C, CS, WITH_DEPTH>() == 4)) ? SPLAT_K6_WAVES : 5)) void render_forward_kernel(FwdArgs args) {
    static_assert(!TRACK || (C == 6 && !WITH_DEPTH), "the tracking-loss epilogue reads the six fused channels");
    constexpr int FP = forward_fp<C, CS, WITH_DEPTH>();
    // ONE 4x4-PIXEL BLOCK PER 16-LANE ROW: row r of wave w composites block r of quadrant w (gather(): NL = 16) from the block's OWN
    // visit lists -- the four rows of a wave process four different Gaussians per trip (each lane reads the record of its row's
    // entry; the vector instructions are shared).  SplaTAM's splats are small ({alpha >= 1/255} radius ~3.8 px at workload B): a
    // Gaussian that touches an 8x8 quadrant touches ~2 of its four blocks, so a wave takes ~0.6 trips per quadrant visit and ~36 %
    // of its lanes hold a live pixel instead of ~21 % -- on a kernel bound by the vector pipe.
    constexpr int NL = 16;
    __shared__ Batch<FP, NL> B;
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[SORT ? kFusedSortMax + 2 : 2];
    const int tile_local = block_tile(args.per_xcd, args.T, launch_order(args.st));     // T: tiles of this launch (SplatState.tile_row_begin: a band of tile rows)
    if (tile_local < 0) return;
    const FwdArgs &a = args;
    const SplatCamera &cam = a.cam;
    SplatState &st = const_cast<SplatState &>(a.st);            // (never written: forward_tile's signature is historical)
    const float *const colors = a.colors;
    float *const out_color = a.out_color, *const out_depth = a.out_depth;
    const TrackLossEpilogue &ep = a.ep;
    (void)out_depth; (void)ep;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tile = tile_local + st.tile_row_begin * gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int row = lane >> 4;                                  // this lane's block of the quadrant
    const int px = tx * kTile + (wave & 1) * 8 + (row & 1) * 4 + (lane & 3), py = ty * kTile + (wave >> 1) * 8 + (row >> 1) * 4 + ((lane >> 2) & 3);
    const float fpx = (float)px, fpy = (float)py;
    const bool inside = px < W && py < H;
    float Tr = 1.f, D = 0.f, Cc[C];
    unsigned last = 0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) Cc[ch] = 0.f;
    forward_tile<C, CS, WITH_DEPTH, SORT>(colors, st, B, s_keys, tile, tx, ty, gx, tid, fpx, fpy, inside, Tr, D, Cc, last);
    if (st.tile_work) {
        // what the backward composite will walk in this tile: per quadrant, the deepest list entry a pixel blended (SplatState.tile_work)
        __shared__ unsigned s_work;
        const unsigned wmax = wave_max_u32(inside ? last : 0u);
        if (tid == 0) s_work = 0u;
        __syncthreads();
        if (lane == 0) atomicAdd(&s_work, wmax);
        __syncthreads();
        if (tid == 0) st.tile_work[tile] = s_work;
    }
    float acc_depth = 0.f, acc_im = 0.f;
    if (inside) {
        const size_t HW = (size_t)H * W;
        const size_t pix = (size_t)py * W + px;
        st.final_T[pix] = Tr;
        st.n_contrib[pix] = (int)last;
        float o[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            o[ch] = Cc[ch] + Tr * bg_of(cam, ch);
            out_color[ch * HW + pix] = o[ch];
        }
        if constexpr (WITH_DEPTH) out_depth[pix] = D;
        if constexpr (TRACK) {
            // channels: r, g, b, depth, silhouette, depth^2; mask = (gt > 0) & ~isnan(depth) & ~isnan(uncertainty) [& sil > thres]
            // (the four inputs of the loss requested together, ahead of the gradient stores: the compiler cannot move a load over a store
            //  it cannot prove disjoint, and the epilogue is a tail nothing hides)
            const float gt = ep.depth[pix];
            const float im3[3] = {ep.im[pix], ep.im[HW + pix], ep.im[2 * HW + pix]};
            const float unc = o[5] - o[3] * o[3];
            bool m = gt > 0.f && !(o[3] != o[3]) && !(unc != unc);
            if (ep.use_sil_for_loss) m = m && (o[4] > ep.sil_thres);
            const float dd = gt - o[3];
            acc_depth = m ? fabsf(dd) : 0.f;
            const float dsign = m ? ((dd > 0.f) ? -1.f : ((dd < 0.f) ? 1.f : 0.f)) : 0.f;
            ep.dL_dout6[3 * HW + pix] = ep.use_l1 ? ep.w_depth * dsign : 0.f;
            const bool cm = ep.use_sil_for_loss ? m : true;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float di = im3[ch] - o[ch];
                acc_im += cm ? fabsf(di) : 0.f;
                ep.dL_dout6[ch * HW + pix] = cm ? -ep.w_im * ((di > 0.f) ? 1.f : ((di < 0.f) ? -1.f : 0.f)) : 0.f;
            }
        }
    }
    if constexpr (TRACK) {
        // workgroup sums -> one of the SPLAT_ITER_SUM_COPIES copies of the partial sums (double atomics)
        __shared__ double s_loss[2][4];
        float a0 = acc_depth, a1 = acc_im;
        for (int msk = 32; msk >= 1; msk >>= 1) { a0 += __shfl_xor(a0, msk, 64); a1 += __shfl_xor(a1, msk, 64); }
        if (lane == 0) { s_loss[0][wave] = (double)a0; s_loss[1][wave] = (double)a1; }
        __syncthreads();
        if (tid < 2) {
            const double t = s_loss[tid][0] + s_loss[tid][1] + s_loss[tid][2] + s_loss[tid][3];
            if (t != 0.0) atomicAdd(ep.sums + (size_t)(blockIdx.x % SPLAT_ITER_SUM_COPIES) * SPLAT_ITER_SUMS + tid, t);
        }
    }
}

// ---------------------------------------------------------------------------
// K7 backward composite
// ---------------------------------------------------------------------------
// DMASK: channels whose incoming gradient can be non-zero (the others are not even loaded);
// SMASK: channels whose colour sums (dL/dcolour) the caller needs.  The reference API uses all C for both; the fused
// iteration knows that the silhouette and depth^2 planes carry no gradient, and that tracking never reads dL/drgb.
constexpr int popcount_c(unsigned m) { return m == 0 ? 0 : (int)(m & 1u) + popcount_c(m >> 1); }
constexpr int highest_set_bit(unsigned m) { int h = -1; for (int i = 0; i < 32; ++i) if ((m >> i) & 1u) h = i; return h; }
constexpr int nth_set_bit(unsigned m, int n) {      // index of the n-th (0-based) set bit
    int idx = 0;
    while (true) {
        if (m & 1u) { if (n == 0) return idx; --n; }
        m >>= 1; ++idx;
        if (idx > 31) return -1;
    }
}


// ---------------------------------------------------------------------------
// K7 backward composite, generation 5: no cross-lane reduction per Gaussian visit
// ---------------------------------------------------------------------------
// Generation 3 (round 1) spent ~95 VALU wave-instructions per (Gaussian, 8x8 quadrant) visit, ~40 of them on reducing the
// 6 + |SMASK| partial sums across the 64 pixel lanes (packed permlane / DPP trees, zero fills, publish selects) -- while only
// ~21 % of the lanes of a visited quadrant hold a live pixel.  This generation splits the visit into two phases and turns
// the reduction over PIXELS into a loop, not a lane tree:
//   phase 1 (lane = pixel): the back-to-front recursion in its running-sum form
//         T_i = T_{i+1} / (1 - a_i),   dL/da_i = T_i c_i.dL/dC - R_i / (1 - a_i),   R_{i-1} = R_i + (c_i.dL/dC) a_i T_i,
//     (R_i = sum_{j>i} (c_j.dL/dC) a_j T_j, started at T_final (bg.dL/dC): the reference's recursion on accum_rec /
//     last_alpha, /root/reference callee of scripts/splatam.py:702,854, summed instead of blended) gives per live pixel just
//     TWO numbers, v = G dL/da and w = a T, which go to a per-wave LDS pair buffer [visit slot][pixel]; ~28 VALU;
//   phase 2 (every kChunk = 8 visits; lane = (visit, pixel ROW of the quadrant)): each lane walks the 8 pixels of its row and
//     accumulates the moments of v about the row's centre (sum v, sum v k, sum v k^2, k = column - 3.5: compile-time
//     constants) and the colour sums sum w dL/dC[ch] in registers -- dense FMAs, 3 + |SMASK| per pixel -- converts them to
//     the centred sums S1..S6 with the visit's (mu - quadrant centre) and opacity, reduces over its visit's 8 lanes with three
//     DPP adds per value, and publishes.  ~14 VALU per visit.
// All 6 + |SMASK| sums of one visit still leave the wave in ONE atomic instruction per accumulator line: a visit's values
// 0..7 sit in its own 8 lanes, values 8.. are rotated (DPP row_ror:8) into the lanes of its row partner, so that even visits
// publish in one instruction and odd visits in the next (4 lines each).
constexpr int kChunk = 8;               // visits per phase-2 pass (8 lanes each)
constexpr int kPairRow = 69;            // float2 per visit slot: 64 pixels + 4 (rows 4..7 shifted), padded to an odd stride

// (v, w) of (slot, pixel): conflict-free for the phase-1 ds_write_b64 (lane = pixel) and the phase-2 ds_read_b64
// (lane = (slot, row), same column): with the odd row stride 69 = 5 (mod 32) the 8-byte unit index mod 32 is a bijection
// of (slot & 3, row) for every column
__device__ __forceinline__ int pair_index(int slot, int pixel) { return slot * kPairRow + pixel + 4 * (pixel >> 5); }

typedef float v2f __attribute__((ext_vector_type(2)));

struct PairBuf {
    float2 vw[4][kChunk * kPairRow];    // [wave][slot][pixel] -> (v, w)
};

__device__ __forceinline__ float group8_allreduce_add(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);   // row_half_mirror: the other quad of the 8-lane group
    return v;
}

// The backward composite of ONE tile from the pixel state on (generation 5, see above): everything after the loads of final_T / n_contrib /
// dL/dC.  `staged`: index of the batch that is ALREADY in B -- records, quadrant masks and quadrant visit lists committed by the
// caller (the fused forward + backward composite hands over the batch its forward pass ended on) -- or -1.  B: a Batch with the
// quadrant lists (NL == 4, or a BOTH batch); PB, s_wmax: the caller's LDS.
template <int C, int CS, unsigned DMASK, unsigned SMASK, bool OPAC, int DBG, class BatchT>
__device__ __forceinline__ void backward_core(const float *colors, const SplatState &st, float *accum, BatchT &B, PairBuf &PB, unsigned *s_wmax,
                                              const int tile, const int tx, const int ty, const int tid, const bool inside, const float Tfin,
                                              const unsigned last, const float (&dpix)[C], float R, const int staged) {
    constexpr int CL = highest_set_bit(DMASK) + 1;        // colour channels staged per Gaussian: only those that carry gradient
    static_assert(CL >= 1 && CL <= C, "DMASK names channels of the call");
    constexpr int R4 = BatchT::R4;
    constexpr int FP = (R4 - 2) * 4;
    static_assert(FP >= CL, "the batch records hold the channels that carry gradient");
    constexpr int NS = popcount_c(SMASK);
    constexpr int NB = OPAC ? 6 : 5;          // published geometric sums: S1..S5 (+ S6)
    constexpr int NV = NB + NS;               // published values per visit
    static_assert(NV <= 16, "two publish instructions carry at most 16 values");
    constexpr int NX = NV > 8 ? NV - 8 : 0;   // values 8.. travel in the row partner's lanes
    const int lane = tid & 63, wave = tid >> 6;
    const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;      // this wave's quadrant
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const float fpx = (float)px, fpy = (float)py;
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    float Tr = Tfin;
    (void)inside;
    // ---- phase-2 constants: the incoming gradient of the 8 pixels of row s2 = lane & 7 (channels in SMASK), from a per-wave
    // LDS table [row][column] filled by the pixel lanes (row stride 8 records + 1: eight rows -> eight distinct bank groups).
    // Few channels (tracking: the depth plane only): held in registers; more: one broadcast ds_read_b128 per pixel in phase 2.
    const int v2 = lane >> 3, s2 = lane & 7;
    constexpr int NS4 = NS > 0 ? (NS + 3) / 4 : 1;              // float4 per pixel in the table
    constexpr bool kRowRegs = NS <= 2;
    // (with the rows in registers the table is only a transposition buffer at start-up: it then borrows the pair buffer, which is
    //  first written by the first visit -- 4.6 KB less LDS, a fifth workgroup per CU)
    constexpr int kTabWave = 8 * (8 * NS4 + 1);
    static_assert(!kRowRegs || sizeof(float4) * kTabWave <= sizeof(PB.vw[0]), "the start-up table fits one wave's pair buffer");
    __shared__ float4 s_dtab[kRowRegs ? 1 : 4 * kTabWave];
    float4 *const my_tab = kRowRegs ? reinterpret_cast<float4 *>(PB.vw[wave]) : s_dtab + wave * kTabWave;
    {
        float t[NS4 * 4];
#pragma unroll
        for (int n = 0; n < NS4 * 4; ++n) {
            t[n] = 0.f;
            if constexpr (NS > 0)
                if (n < NS) t[n] = dpix[nth_set_bit(SMASK, n < NS ? n : 0)];
        }
#pragma unroll
        for (int q = 0; q < NS4; ++q)
            my_tab[(lane >> 3) * (8 * NS4 + 1) + (lane & 7) * NS4 + q] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
    }
    const float4 *const my_drow = my_tab + s2 * (8 * NS4 + 1);
    float drow[kRowRegs && NS > 0 ? NS : 1][8];
    if constexpr (kRowRegs && NS > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 t = my_drow[k];
            drow[0][k] = t.x;
            if constexpr (NS > 1) drow[1][k] = t.y;
        }
    }
    const unsigned wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(last));   // deepest contributor of this quadrant
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const unsigned tmax = (unsigned)__builtin_amdgcn_readfirstlane((int)max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));
    if (tmax == 0) return;                                     // uniform over the workgroup
    const unsigned lo = st.tile_stride > 0 ? (unsigned)tile * (unsigned)st.tile_stride : st.tile_base[tile];
    const int nb = (int)((tmax + kBatchEntries - 1) / kBatchEntries);

    // accumulator slot of published value k: S1..S5 (S6) -> 0..5, colour sums -> 6 + channel.  The camera-tracking form (no opacity sum,
    // the depth channel's colour sum only) puts that one sum in the free slot 5: everything tracking publishes then lies in the first
    // half of the accumulator line, and the per-Gaussian kernel behind it reads and clears 32 bytes per Gaussian instead of 64
    // (fused.hip: fused_backward_kernel<.., MAPGRADS = false>)
    constexpr bool kTrackSlots = !OPAC && SMASK == 0x8u;
    auto slot_of = [](int k) { return k < NB ? k : (kTrackSlots ? 5 : 6 + nth_set_bit(SMASK, k - NB)); };
    int doff_own = 0, doff_x = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < NV && s2 == k) doff_own = slot_of(k);
#pragma unroll
    for (int k = 0; k < NX; ++k)
        if (s2 == k) doff_x = slot_of(8 + k);
    float2 *const my_vw = PB.vw[wave];
    const int wr_lane = pair_index(0, lane);
    // LDS byte address of column 0 of this lane's phase-2 row (generic shared pointer -> 32-bit LDS offset)
    const unsigned rd_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)(my_vw + pair_index(v2, s2 * 8));
    const float qcx = (float)qx0 + 3.5f, qcy = (float)(qy0 + s2);       // centre column of the quadrant, this lane's phase-2 row
    unsigned slot_rec = 0;              // lanes 8 n .. 8 n + 7 (the phase-2 lanes of slot n): record (byte offset in B.rec) of the visit waiting in slot n
    unsigned long long slot_m = 0xFFull;        // lanes of the slot the next live visit takes (wave-uniform)
    float2 *wr_ptr = my_vw + wr_lane;   // where this lane's pixel writes the next visit's pair

    // ---- phase 2: nvis (wave-uniform) visits are waiting in the pair buffer; their records are still in B
    auto phase2 = [&](int nvis) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 *rp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(B.rec) + slot_rec);
        const float op = rp[0].w;
        const float4 mm = rp[R4 - 1];
        float R0 = 0.f, RX = 0.f, RXX = 0.f, Cs[NS > 0 ? NS : 1];
#pragma unroll
        for (int n = 0; n < NS; ++n) Cs[n] = 0.f;
        // the row's 8 (v, w) pairs as EIGHT ds_read_b64 (2 LDS cycles each, conflict-free in this layout); left to itself the
        // compiler pairs them into ds_read2_b64, which the LDS serves in 16-lane groups at half the rate and with 2-way
        // bank conflicts here (SQ_LDS_BANK_CONFLICT was 83 % of the LDS-active cycles)
        v2f pr[8];
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                     "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(pr[0]), "=&v"(pr[1]), "=&v"(pr[2]), "=&v"(pr[3]), "=&v"(pr[4]), "=&v"(pr[5]), "=&v"(pr[6]), "=&v"(pr[7])
                     : "v"(rd_addr)
                     : "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float kc = (float)k - 3.5f;
            R0 += pr[k].x;
            RX = fmaf(pr[k].x, kc, RX);
            RXX = fmaf(pr[k].x, kc * kc, RXX);
            if constexpr (kRowRegs) {
#pragma unroll
                for (int n = 0; n < NS; ++n) Cs[n] = fmaf(pr[k].y, drow[n][k], Cs[n]);
            } else {
#pragma unroll
                for (int q = 0; q < NS4; ++q) {
                    const float4 t = my_drow[k * NS4 + q];
                    if (4 * q < NS) Cs[4 * q] = fmaf(pr[k].y, t.x, Cs[4 * q]);
                    if (4 * q + 1 < NS) Cs[4 * q + 1] = fmaf(pr[k].y, t.y, Cs[4 * q + 1]);
                    if (4 * q + 2 < NS) Cs[4 * q + 2] = fmaf(pr[k].y, t.z, Cs[4 * q + 2]);
                    if (4 * q + 3 < NS) Cs[4 * q + 3] = fmaf(pr[k].y, t.w, Cs[4 * q + 3]);
                }
            }
        }
        const float ax = mm.x - qcx, ay = mm.y - qcy;
        const float Sdx = fmaf(ax, R0, -RX);                   // sum v dx      (this row)
        const float Sdxx = fmaf(ax, Sdx - RX, RXX);            // sum v dx^2
        float val[NV > 8 ? NV : 8];
        val[0] = Sdx;
        val[1] = ay * R0;                                      // sum v dy
        val[2] = Sdxx;
        val[3] = ay * Sdx;                                     // sum v dx dy
        val[4] = ay * val[1];                                  // sum v dy^2
        if constexpr (OPAC) val[5] = R0;
#pragma unroll
        for (int n = 0; n < NS; ++n) val[NB + n] = Cs[n];
#pragma unroll
        for (int k = 0; k < NV; ++k) val[k] = group8_allreduce_add(val[k]);
        const unsigned id = __float_as_uint(mm.z);
        float pv = val[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < NV) pv = s2 == k ? val[k] : pv;
        if (s2 < 5) pv *= op;                                  // S1..S5 carry q = opacity * dL/dalpha
        const bool valid = v2 < nvis;
        if constexpr ((DBG & 4) != 0) {
            // (measurement: the values are formed, the atomics are not issued -- a dependent, never-true store keeps them alive)
            if (pv == 1.2345e-33f && valid) accum[0] = pv + val[NV > 8 ? 8 : 0];
        } else if constexpr (NX == 0) {
            if (valid && s2 < NV) atomicAdd(accum + (size_t)id * SPLAT_GRAD_STRIDE + doff_own, pv);
        } else {
            float px_ = val[8];
#pragma unroll
            for (int k = 1; k < NX; ++k) px_ = s2 == k ? val[8 + k] : px_;
            // the row partner's (visit v2 ^ 1) extra values, id and validity
            const float xr = dpp_f32<0x128>(px_);                                 // row_ror:8
            const unsigned idr = dpp_u32<0x128>(id);
            const bool pvalid = (v2 ^ 1) < nvis;
            const bool odd = (v2 & 1) != 0;
            // instruction 1: even visits (own 8 lanes + the first NX lanes of the odd partner); instruction 2: odd visits
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const bool own = odd == (pass == 1);
                const bool act = own ? valid : (pvalid && s2 < NX);
                const unsigned tid_ = own ? id : idr;
                const int off = own ? doff_own : doff_x;
                const float x = own ? pv : xr;
                if (act) atomicAdd(accum + (size_t)tid_ * SPLAT_GRAD_STRIDE + off, x);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    int nslot = 0;                                              // wave-uniform
    Staged<FP> pre;
    const bool handed_over = staged == nb - 1;                  // (uniform) the first batch of this pass is the one already in B
    // one list entry as the backward pass stages it: the record the forward composite left in SplatState.tile_recs (one coalesced
    // 48-byte read: parameters, colours, centre, id, quadrant mask), or gathered and culled again from the per-Gaussian arrays
    const float4 *const recs = FP == 4 ? reinterpret_cast<const float4 *>(st.tile_recs) : nullptr;
    auto stage = [&](unsigned e, bool valid) {
        if (recs) {
            pre.ga = make_float4(0.f, 0.f, 0.f, 0.f);
            pre.mu = make_float2(0.f, 0.f);
#pragma unroll
            for (int f = 0; f < FP; ++f) pre.feat[f] = 0.f;
            pre.mask = 0;
            pre.id = 0;
            pre.aux = 0.f;
            if (valid) {
                const float4 *r = recs + (size_t)(lo + e) * 3;
                const float4 a = r[0], f = r[1], m = r[2];
                pre.ga = a;
                pre.feat[0] = f.x;
                if constexpr (FP > 1) pre.feat[1] = f.y;
                if constexpr (FP > 2) pre.feat[2] = f.z;
                if constexpr (FP > 3) pre.feat[3] = f.w;
                pre.mu = make_float2(m.x, m.y);
                pre.id = __float_as_uint(m.z);
                pre.mask = __float_as_uint(m.w) & 15u;
            }
        } else {
            gather<CL, CS, false, FP>(pre, st, colors, lo + e, valid, tile_x0, tile_y0);
        }
    };
    if (!handed_over) {
        const unsigned e = (unsigned)((nb - 1) * kBatchEntries + tid);
        stage(e, tid < kBatchEntries && e < tmax);
    }
    for (int bi = nb - 1; bi >= 0; --bi) {
        if (!(handed_over && bi == nb - 1)) {
            if (bi < nb - 1 || staged >= 0) __syncthreads();   // every wave has finished reading the previous batch
            commit_quadrants(B, pre, tid);
            __syncthreads();
        }
        const bool more = bi > 0;
        if (more) stage((unsigned)((bi - 1) * kBatchEntries + tid), tid < kBatchEntries);
        const int base = bi * kBatchEntries;
        const int lim = (int)wmax - base;                      // entries [0, lim) of this batch can matter to this wave
        const int last_rec = ((int)last - base - 1) * (int)(R4 * sizeof(float4));      // this pixel blended the batch's records at byte offsets [0, last_rec]

        // a list entry -> its record's byte offset in B.rec: a wave-uniform value in a vector register (one v_mul_u32_u24 with a byte
        // select per visit, no scalar bookkeeping); the visit identifies its entry by that offset
        auto rec_of = [&](unsigned e) { return e * (unsigned)(R4 * sizeof(float4)); };
        auto load_rec = [&](unsigned rec, Rec<FP> &r) {
            const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(B.rec) + rec);
            r.a = p[0];
#pragma unroll
            for (int v = 0; v < FP / 4; ++v) r.f[v] = p[1 + v];
            r.m = p[R4 - 1];
        };
        // one visit: `cur` was fetched during the previous visit
        auto visit = [&](unsigned rec, const Rec<FP> &cur) {
            const float dx = cur.m.x - fpx, dy = cur.m.y - fpy;
            const float p2 = dx * (cur.a.x * dx + cur.a.y * dy) + cur.a.z * dy * dy;
            const float G = fast_exp2(p2);
            const float alpha = fminf(kAlphaMax, cur.a.w * G);
            const unsigned long long live_m = __builtin_amdgcn_ballot_w64((int)rec <= last_rec) & __builtin_amdgcn_ballot_w64(p2 <= 0.f) &
                                              __builtin_amdgcn_ballot_w64(alpha >= kAlphaMin);
            if (live_m == 0) return;
            // lanes without a live pixel run the same arithmetic on G = alpha = 0 (two selects; 1 / (1 - 0) and T * 1 are exact,
            // so their T and R do not move and their pair is (0, 0); a select, not a multiply by 0, so that no inf * 0 can appear)
            const bool live = lane_of(live_m);
            const float Gl = live ? G : 0.f;
            const float al = live ? alpha : 0.f;
            const float rcp = __builtin_amdgcn_rcpf(1.f - al);
            const float Tn = Tr * rcp;                     // transmittance in front of this Gaussian
            // c . dL/dC as ONE chain of fused multiply-adds (|DMASK| instructions; on gfx950 the composites are bound by the vector
            // pipe -- plain operations 2 cycles, compares / selects / DPP 4, exp / rcp 8: profiles/r03_valu_issue_bench.txt)
            float cdot = 0.f;
            bool first = true;
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                if ((DMASK >> ch) & 1u) {
                    const float4 &fv = cur.f[ch >> 2];
                    const float c = (ch & 3) == 0 ? fv.x : ((ch & 3) == 1 ? fv.y : ((ch & 3) == 2 ? fv.z : fv.w));
                    cdot = first ? c * dpix[ch] : fmaf(c, dpix[ch], cdot);
                    first = false;
                }
            const float dL_dalpha = fmaf(cdot, Tn, -(R * rcp));
            const float vv = Gl * dL_dalpha;
            const float ww = al * Tn;
            R = fmaf(cdot, ww, R);
            Tr = Tn;
            *wr_ptr = make_float2(vv, ww);
            wr_ptr += kPairRow;
            slot_rec = lane_of(slot_m) ? rec : slot_rec;
            slot_m <<= 8;
            if (++nslot == kChunk) {
                if constexpr ((DBG & 8) == 0) phase2(kChunk);
                nslot = 0;
                slot_m = 0xFFull;
                wr_ptr = my_vw + wr_lane;
            }
        };
        // back to front over this quadrant's four visit lists (one per gathering wave), four entries per trip; the record of the NEXT
        // visit is fetched before the current one is processed (two register sets, ping-pong).  Entries past lim (this wave's deepest
        // contributor) are cut off through the quadrant masks; what the top group holds beyond them -- cut-off entries, the 0xFF
        // padding (the inert record) -- is visited without effect, as is whatever the look-ahead below the list's start names
#pragma unroll 1
        for (int g = 3; g >= 0; --g) {
            if constexpr ((DBG & 2) != 0) break;
            const int kk = lim - 64 * g;
            if (kk <= 0) continue;
            unsigned long long bits = mask_word(B, wave, g);
            if (kk < 64) bits &= (1ull << kk) - 1ull;
            if (bits == 0) continue;
            const unsigned char *seg = visit_list(B, wave, g);
            int kb = (__builtin_popcountll(bits) - 1) & ~3;
            unsigned cur4 = *reinterpret_cast<const unsigned *>(seg + kb);
            Rec<FP> ra, rb;
            unsigned r3 = rec_of(cur4 >> 24);
            load_rec(r3, ra);
#pragma unroll 1
            for (; kb >= 0; kb -= 4) {
                const unsigned nxt4 = *reinterpret_cast<const unsigned *>(seg + kb - 4);
                const unsigned r2 = rec_of((cur4 >> 16) & 0xFFu);
                load_rec(r2, rb);
                visit(r3, ra);
                const unsigned r1 = rec_of((cur4 >> 8) & 0xFFu);
                load_rec(r1, ra);
                visit(r2, rb);
                const unsigned r0 = rec_of(cur4 & 0xFFu);
                load_rec(r0, rb);
                visit(r1, ra);
                r3 = rec_of(nxt4 >> 24);
                load_rec(r3, ra);
                visit(r0, rb);
                cur4 = nxt4;
            }
        }
        // the records of the waiting visits live in this batch: publish them before it is replaced
        if (nslot > 0) {
            if constexpr ((DBG & 8) == 0) phase2(nslot);
            nslot = 0;
            slot_m = 0xFFull;
            wr_ptr = my_vw + wr_lane;
        }
    }
}

// DBG (measurement builds of the SAME body, selected by splat_debug_option(4, bits); 0 in the product launches): 1 = every workgroup
// leaves (start, end) wall-clock stamps in `stamps` [gridDim.x][2]; 2 = stage and commit the batches but visit nothing; 4 = everything
// but the accumulator atomics; 8 = phase 1 only (the pairs are written, never read).  profiles/r04_k7_account.md is built from them.
struct BwdArgs {
    SplatCamera cam;
    const float *colors;
    SplatState st;
    const float *dL_dcolor;
    float *accum;
    int T, per_xcd;
    long long *stamps;
};
template <int C, int CS, unsigned DMASK, unsigned SMASK, bool OPAC, bool BG, int DBG = 0>
__device__ __forceinline__ void render_backward_body5(const BwdArgs &args) {
    long long *const stamps = args.stamps;
    long long t_begin = 0;
    if constexpr ((DBG & 1) != 0) t_begin = (long long)wall_clock64();
    auto stamp = [&]() {
        if constexpr ((DBG & 1) != 0)
            if (threadIdx.x == 0 && stamps) { stamps[2 * (size_t)blockIdx.x] = t_begin; stamps[2 * (size_t)blockIdx.x + 1] = (long long)wall_clock64(); }
    };
    constexpr int CL = highest_set_bit(DMASK) + 1;        // colour channels staged per Gaussian: only those that carry gradient
    static_assert(CL >= 1 && CL <= C, "DMASK names channels of the call");
    constexpr int FP = (CL + 3) / 4 * 4;
    __shared__ Batch<FP> B;
    __shared__ PairBuf PB;
    __shared__ unsigned s_wmax[4];
    const int tile_local = block_tile(args.per_xcd, args.T, launch_order(args.st));
    if (tile_local < 0) { stamp(); return; }
    const BwdArgs &a = args;
    const SplatCamera &cam = a.cam;
    const SplatState &st = a.st;
    const float *const colors = a.colors, *const dL_dcolor = a.dL_dcolor;
    float *const accum = a.accum;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tile = tile_local + st.tile_row_begin * gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7), py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;

    // ---- phase-1 state of this lane's pixel
    const float Tfin = inside ? st.final_T[pix] : 0.f;
    const unsigned last = inside ? (unsigned)st.n_contrib[pix] : 0u;
    float dpix[C], R = 0.f;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        dpix[ch] = 0.f;
        if ((DMASK >> ch) & 1u) {
            dpix[ch] = inside ? dL_dcolor[ch * HW + pix] : 0.f;
            if constexpr (BG) R += Tfin * bg_of(cam, ch) * dpix[ch];      // the background term of dL/dalpha: -T_final bg.dL/dC / (1 - alpha)
        }
    }
    backward_core<C, CS, DMASK, SMASK, OPAC, DBG>(colors, st, accum, B, PB, s_wmax, tile, tx, ty, tid, inside, Tfin, last, dpix, R, -1);
    stamp();
}

// Two launch shapes of the same body.  With at most two colour sums (the tracking form) the phase-2 gradient rows live in registers: the
// kernel fits 96 VGPRs and 31 KB of LDS -> FIVE workgroups per CU (tracking +2.3 % at B); the mapping form (120 VGPRs, 35.7 KB) stays
// at four with the compiler's own register budget (an explicit minimum of four waves made it 2 % slower).
#ifndef SPLAT_K7_MIN_WAVES
#define SPLAT_K7_MIN_WAVES 1
#endif
template <int C, int CS, unsigned DMASK, unsigned SMASK, bool OPAC = true, bool BG = true>
__global__ __launch_bounds__(256, (popcount_c(SMASK) <= 4 ? SPLAT_K7_MIN_WAVES : 1)) void render_backward_kernel5(BwdArgs args) {
    render_backward_body5<C, CS, DMASK, SMASK, OPAC, BG>(args);
}
template <int C, int CS, unsigned DMASK, unsigned SMASK, bool OPAC = true, bool BG = true>
__global__ __launch_bounds__(256, 5) void render_backward_kernel5_w5(BwdArgs args) {
    static_assert(popcount_c(SMASK) <= 2, "the five-wave shape is for the forms whose gradient rows live in registers");
    render_backward_body5<C, CS, DMASK, SMASK, OPAC, BG>(args);
}

// ---------------------------------------------------------------------------
// K6 + K7 in ONE kernel: the tracking iteration's forward composite, loss and backward composite
// ---------------------------------------------------------------------------
// The tracking loss is pixel-local (/root/reference/scripts/splatam.py:256-288 with ignore_outlier_depth_loss = False, as every shipped
// configuration has it): a tile's gradient planes exist the moment its forward composite ends.  This kernel keeps going: the loss terms
// and dL/d(render) of a pixel are formed in its registers (TrackLossEpilogue, above), moved from the forward pass' lane layout (one 4x4
// block per 16-lane row) to the backward pass' (8x8 quadrant, row-major) by one ds_bpermute per value, and the backward composite
// (backward_core) walks the batch the forward pass has just composited -- its records are still in LDS, and commit() left the quadrant
// visit lists beside the block lists (Batch<.., BOTH>).  Against the two-kernel form a tile with a one-batch list (workload B: ~220
// entries) saves the whole second staging (point list -> conic / centre / feature gathers, culling tests, ballots: 16-20 % of the
// backward composite, profiles/r04_k7_account.md), the round trip of ten per-pixel planes (six rendered, four gradient; final_T and
// n_contrib) and a launch; a tile with more batches re-stages all but its last one.  KEEP: the planes are written as well (what
// FusedEngine.rendered() / the tests read); the tracking step of the product loop has no reader for them.
// DBG (measurement builds, splat_debug_option(4, bits); 0 in the product launches): 2 / 4 / 8 as in backward_core (stage only / no accumulator
// atomics / phase 1 only), 16 = forward composite and loss only (the backward pass is not entered): the account of the kernel's memory
// traffic in profiles/r05_track_fused_traffic.md is the differences of their FETCH_SIZE / WRITE_SIZE.
struct TrackFusedArgs {
    SplatCamera cam;
    const float *feat8;
    SplatState st;
    float *out6, *accum;
    int T, per_xcd;
    TrackLossEpilogue ep;
};
// FULL: every sum the reference's backward() forms for a tracking iteration -- the colour sums of r, g, b and the opacity sum too
// (dL/d rgb, opacity, scale: the reference computes them and steps them with learning rate 0, /root/reference/utils/slam_helpers.py:133-136,
// configs/replica/splatam.py:71-79) -- i.e. the backward composite's MAPPING form behind the forward pass (120 registers, the gradient
// rows of phase 2 in LDS: four workgroups per CU instead of five)
template <bool KEEP, int DBG, bool FULL>
__device__ __forceinline__ void track_fused_body(const TrackFusedArgs &args) {
    constexpr int C = 6, CS = 8, FP = forward_fp<C, CS, false>();
    using BatchT = Batch<FP, 16, true>;
    // LDS, 31.4 KB (five workgroups per CU): [records | quadrant masks + lists | counts] live through both passes; behind them ONE region
    // that holds the forward pass' block lists and the sort's key array, then the backward pass' pair buffer
    constexpr size_t kHead = offsetof(BatchT, vlist);
    constexpr size_t kKeysAt = (kHead + sizeof(BatchT::vlist) + 15) / 16 * 16;
    constexpr size_t kFwd = kKeysAt + sizeof(uint64_t) * (kFusedSortMax + 2);
    constexpr size_t kBwd = (kHead + 15) / 16 * 16 + sizeof(PairBuf);
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kFwd > kBwd ? kFwd : kBwd];
    __shared__ unsigned s_wmax[4];
    __shared__ double s_loss[2][4];
    BatchT &B = *reinterpret_cast<BatchT *>(s_raw);
    uint64_t *const s_keys = reinterpret_cast<uint64_t *>(s_raw + kKeysAt);
    PairBuf &PB = *reinterpret_cast<PairBuf *>(s_raw + (kHead + 15) / 16 * 16);
    const int tile_local = block_tile(args.per_xcd, args.T, launch_order(args.st));
    if (tile_local < 0) return;
    const TrackFusedArgs &a = args;
    const SplatCamera &cam = a.cam;
    SplatState &st = const_cast<SplatState &>(a.st);
    const float *const feat8 = a.feat8;
    float *const out6 = a.out6, *const accum = a.accum;
    const TrackLossEpilogue &ep = a.ep;
    (void)out6;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tile = tile_local + st.tile_row_begin * gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const size_t HW = (size_t)H * W;
    // ---- forward pass: lane = pixel of block (lane >> 4) of quadrant `wave`
    const int row = lane >> 4;
    const int fx_ = tx * kTile + (wave & 1) * 8 + (row & 1) * 4 + (lane & 3), fy_ = ty * kTile + (wave >> 1) * 8 + (row >> 1) * 4 + ((lane >> 2) & 3);
    const bool finside = fx_ < W && fy_ < H;
    float Tr = 1.f, D = 0.f, Cc[C];
    unsigned flast = 0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) Cc[ch] = 0.f;
    const int staged = forward_tile<C, CS, false, true, false>(feat8, st, B, s_keys, tile, tx, ty, gx, tid, (float)fx_, (float)fy_, finside, Tr, D, Cc, flast);
    (void)D;
    // ---- the loss of this pixel and its gradient planes, in registers (the arithmetic of render_forward_kernel's TRACK epilogue)
    float acc_depth = 0.f, acc_im = 0.f, g[4] = {0.f, 0.f, 0.f, 0.f};
    if (finside) {
        const size_t pix = (size_t)fy_ * W + fx_;
        const float gt = ep.depth[pix];
        const float im3[3] = {ep.im[pix], ep.im[HW + pix], ep.im[2 * HW + pix]};
        float o[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) o[ch] = Cc[ch] + Tr * bg_of(cam, ch);
        const float unc = o[5] - o[3] * o[3];
        bool m = gt > 0.f && !(o[3] != o[3]) && !(unc != unc);
        if (ep.use_sil_for_loss) m = m && (o[4] > ep.sil_thres);
        const float dd = gt - o[3];
        acc_depth = m ? fabsf(dd) : 0.f;
        const float dsign = m ? ((dd > 0.f) ? -1.f : ((dd < 0.f) ? 1.f : 0.f)) : 0.f;
        g[3] = ep.use_l1 ? ep.w_depth * dsign : 0.f;
        const bool cm = ep.use_sil_for_loss ? m : true;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float di = im3[ch] - o[ch];
            acc_im += cm ? fabsf(di) : 0.f;
            g[ch] = cm ? -ep.w_im * ((di > 0.f) ? 1.f : ((di < 0.f) ? -1.f : 0.f)) : 0.f;
        }
        if constexpr (KEEP) {
            st.final_T[pix] = Tr;
            st.n_contrib[pix] = (int)flast;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out6[ch * HW + pix] = o[ch];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) ep.dL_dout6[ch * HW + pix] = g[ch];
        }
    }
    {
        float a0 = acc_depth, a1 = acc_im;
        for (int msk = 32; msk >= 1; msk >>= 1) { a0 += __shfl_xor(a0, msk, 64); a1 += __shfl_xor(a1, msk, 64); }
        if (lane == 0) { s_loss[0][wave] = (double)a0; s_loss[1][wave] = (double)a1; }
        __syncthreads();                    // (also: every wave has left the forward pass -- the key array may become the pair buffer)
        if (tid < 2) {
            const double t = s_loss[tid][0] + s_loss[tid][1] + s_loss[tid][2] + s_loss[tid][3];
            if (t != 0.0) atomicAdd(ep.sums + (size_t)(blockIdx.x % SPLAT_ITER_SUM_COPIES) * SPLAT_ITER_SUMS + tid, t);
        }
    }
    // ---- to the backward pass' layout: lane = pixel (lane & 7, lane >> 3) of the quadrant; its values sit in lane `src` of the forward layout
    const int qx = lane & 7, qy = lane >> 3;
    const int src = 16 * ((qx >> 2) + 2 * (qy >> 2)) + 4 * (qy & 3) + (qx & 3);
    const bool inside = tx * kTile + (wave & 1) * 8 + qx < W && ty * kTile + (wave >> 1) * 8 + qy < H;
    const float Tfin = inside ? __shfl(Tr, src, 64) : 0.f;
    const unsigned last = inside ? (unsigned)__shfl((int)flast, src, 64) : 0u;
    float dpix[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) dpix[ch] = ch < 4 ? __shfl(g[ch < 4 ? ch : 0], src, 64) : 0.f;      // (outside the image g is zero)
    // ---- backward pass over the same lists, starting on the batch that is still staged
    if constexpr ((DBG & 16) != 0) {
        // (measurement: forward + loss only; the per-pixel state is kept alive by a never-true store)
        if (Tfin == 1.2345e-33f && last == 0xFFFFFFFFu) accum[0] = dpix[0] + dpix[3];
        return;
    }
    if constexpr (FULL) backward_core<C, CS, 0xFu, 0xFu, true, (DBG & 14)>(feat8, st, accum, B, PB, s_wmax, tile, tx, ty, tid, inside, Tfin, last, dpix, 0.f, staged);
    else backward_core<C, CS, 0xFu, 0x8u, false, (DBG & 14)>(feat8, st, accum, B, PB, s_wmax, tile, tx, ty, tid, inside, Tfin, last, dpix, 0.f, staged);
    // what the next launch's order is built from (SplatState.tile_work): the quadrants' deepest contributors, as the core left them
    if (st.tile_work && tid == 0) st.tile_work[tile] = s_wmax[0] + s_wmax[1] + s_wmax[2] + s_wmax[3];
}
template <bool KEEP, int DBG = 0>
__global__ __launch_bounds__(256, 5) void render_track_fused_kernel(TrackFusedArgs args) { track_fused_body<KEEP, DBG, false>(args); }
template <bool KEEP>
__global__ __launch_bounds__(256, 4) void render_track_fused_full_kernel(TrackFusedArgs args) { track_fused_body<KEEP, 0, true>(args); }

// measurement builds (splat_debug_option(4, bits)): the fused iteration's two forms only
template <int C, int CS, unsigned DMASK, unsigned SMASK, bool OPAC, bool BG, int DBG>
__global__ __launch_bounds__(256) void render_backward_kernel5_dbg(BwdArgs args) {
    render_backward_body5<C, CS, DMASK, SMASK, OPAC, BG, DBG>(args);
}
template <int C, int CS, unsigned DMASK, unsigned SMASK, bool OPAC, bool BG, int DBG>
__global__ __launch_bounds__(256, 5) void render_backward_kernel5_w5_dbg(BwdArgs args) {
    render_backward_body5<C, CS, DMASK, SMASK, OPAC, BG, DBG>(args);
}

int g_debug_k7_bits = 0;                // splat_debug_option(4, bits)
long long *g_debug_stamps = nullptr;    // splat_debug_stamps(buffer)

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int C, int CS, bool WITH_DEPTH, bool SORT = false>
static void launch_fwd(const SplatCamera &cam, const float *colors, SplatState &st, float *oc, float *od, int T, hipStream_t s) {
    const int per = (T + 7) / 8;
    auto k = render_forward_kernel<C, CS, WITH_DEPTH, SORT, false>;
    hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s,
                       FwdArgs{cam, colors, st, oc, od, T, per, TrackLossEpilogue{}});
}
template <int C, int CS, unsigned DMASK = (1u << C) - 1u, unsigned SMASK = (1u << C) - 1u, bool OPAC = true, bool BG = true>
static void launch_bwd(const SplatCamera &cam, const float *colors, const SplatState &st, const float *dl, float *acc, int T,
                       hipStream_t s) {
    const int per = (T + 7) / 8;
    if constexpr (C == 6 && CS == 8 && DMASK == 0xFu && !BG && (SMASK == 0xFu || (SMASK == 0x8u && !OPAC))) {
        if (g_debug_k7_bits != 0) {
            constexpr bool W5 = popcount_c(SMASK) <= 2;
            auto go = [&](auto D) {
                constexpr int DBG = decltype(D)::value;
                if constexpr (W5)
                    hipLaunchKernelGGL((render_backward_kernel5_w5_dbg<C, CS, DMASK, SMASK, OPAC, BG, DBG>), dim3(8 * per), dim3(256), 0, s,
                                       BwdArgs{cam, colors, st, dl, acc, T, per, g_debug_stamps});
                else
                    hipLaunchKernelGGL((render_backward_kernel5_dbg<C, CS, DMASK, SMASK, OPAC, BG, DBG>), dim3(8 * per), dim3(256), 0, s,
                                       BwdArgs{cam, colors, st, dl, acc, T, per, g_debug_stamps});
            };
            switch (g_debug_k7_bits) {
                case 1: go(std::integral_constant<int, 1>{}); return;
                case 2: go(std::integral_constant<int, 2>{}); return;
                case 3: go(std::integral_constant<int, 3>{}); return;
                case 4: go(std::integral_constant<int, 4>{}); return;
                case 5: go(std::integral_constant<int, 5>{}); return;
                case 8: go(std::integral_constant<int, 8>{}); return;
                case 9: go(std::integral_constant<int, 9>{}); return;
                default: break;
            }
        }
    }
    if constexpr (popcount_c(SMASK) <= 2) {
        auto k = render_backward_kernel5_w5<C, CS, DMASK, SMASK, OPAC, BG>;
        hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s, BwdArgs{cam, colors, st, dl, acc, T, per, nullptr});
    } else {
        auto k = render_backward_kernel5<C, CS, DMASK, SMASK, OPAC, BG>;
        hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s, BwdArgs{cam, colors, st, dl, acc, T, per, nullptr});
    }
}

static const float *colour_source(const SplatGaussians &g, const SplatState &st) {
    return g.shs ? st.rgb : g.colors_precomp;
}

hipError_t launch_render_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, float *out_color,
                                 float *out_depth, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    const float *col = colour_source(g, st);
    if (T == 0) return hipSuccess;
    if (group_binning(st, cam.image_width, cam.image_height)) {
        // the front end of the fused iteration behind the reference API (splat_preprocess_forward filed one record per touched group
        // of 2 x 2 tiles): the composite filters, sorts and publishes its tile's list itself -- no scan, scatter or sort launch
        if (g.channels != 3) return hipErrorInvalidValue;
        launch_fwd<3, 3, true, true>(cam, col, st, out_color, out_depth, T, s);
        return hipGetLastError();
    }
    switch (g.channels) {
        case 1: launch_fwd<1, 1, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 2: launch_fwd<2, 2, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 3: launch_fwd<3, 3, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 4: launch_fwd<4, 4, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 5: launch_fwd<5, 5, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 6: launch_fwd<6, 6, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 7: launch_fwd<7, 7, true>(cam, col, st, out_color, out_depth, T, s); break;
        case 8: launch_fwd<8, 8, true>(cam, col, st, out_color, out_depth, T, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_render_backward(const SplatCamera &cam, const SplatGaussians &g, const SplatState &st, SplatGrads &gr,
                                  hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    const float *col = colour_source(g, st);
    if (!(gr.flags & SPLAT_GRADS_ACCUM_ZEROED)) {
        hipError_t e = hipMemsetAsync(gr.accum, 0, sizeof(float) * SPLAT_GRAD_STRIDE * (size_t)g.P, s);
        if (e != hipSuccess) return e;
    }
    if (T == 0 || g.P == 0) return hipSuccess;
    if (!cam.bg && g.channels == 3) {           // black background (SplatCamera.bg == NULL): the background term of dL/dalpha falls away at compile time
        launch_bwd<3, 3, 7u, 7u, true, false>(cam, col, st, gr.dL_dcolor, gr.accum, T, s);
        return hipGetLastError();
    }
    switch (g.channels) {
        case 1: launch_bwd<1, 1>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 2: launch_bwd<2, 2>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 3: launch_bwd<3, 3>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 4: launch_bwd<4, 4>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 5: launch_bwd<5, 5>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 6: launch_bwd<6, 6>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 7: launch_bwd<7, 7>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 8: launch_bwd<8, 8>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Fused-iteration path (fused.hip): 6 channels (r, g, b, z, 1, z^2) read from 8-float records, no separate depth plane.
// tiles a fused composite launch covers: the whole frame, or the band of tile rows SplatState.tile_row_begin / _end name
static int launch_tiles(const SplatCamera &cam, const SplatState &st) {
    const int gx = (cam.image_width + kTile - 1) / kTile;
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    if (st.tile_row_end > st.tile_row_begin) return min(T, st.tile_row_end * gx) - st.tile_row_begin * gx;
    return T;
}

hipError_t launch_render_forward_feat8(const SplatCamera &cam, const float *feat8, SplatState &st, float *out6, bool sort_in_kernel,
                                       hipStream_t s, const TrackLossEpilogue *ep, bool *ep_done) {
    const int T = launch_tiles(cam, st);
    if (ep_done) *ep_done = false;
    if (T == 0) return hipSuccess;
    const int per = (T + 7) / 8;
    if (ep && ep_done) {
        if (sort_in_kernel)
        {
            auto k = render_forward_kernel<6, 8, false, true, true>;
            hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s, FwdArgs{cam, feat8, st, out6, nullptr, T, per, *ep});
        }
        else
        {
            auto k = render_forward_kernel<6, 8, false, false, true>;
            hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s, FwdArgs{cam, feat8, st, out6, nullptr, T, per, *ep});
        }
        *ep_done = true;
        return hipGetLastError();
    }
    if (sort_in_kernel) launch_fwd<6, 8, false, true>(cam, feat8, st, out6, nullptr, T, s);
    else launch_fwd<6, 8, false, false>(cam, feat8, st, out6, nullptr, T, s);
    return hipGetLastError();
}

// The tracking iteration's composites as ONE launch (render_track_fused_kernel): lists short enough for the composite's own sort, a
// pixel-local loss (no outlier rejection).  keep_planes: also write out6 / final_T / n_contrib / dL_dout6.
hipError_t launch_render_track_fused(const SplatCamera &cam, const float *feat8, SplatState &st_in, float *out6, float *accum,
                                     const TrackLossEpilogue &ep, bool keep_planes, hipStream_t s, bool full_sums) {
    const int T = launch_tiles(cam, st_in);
    if (T == 0) return hipSuccess;
    const int per = (T + 7) / 8;
    SplatState st = st_in;
    st.tile_recs = nullptr;                               // (forward_tile<.., WRITE_RECS = false>: the backward pass gathers what it re-stages)
    if (!keep_planes && g_debug_k7_bits != 0) {           // measurement builds (see render_track_fused_kernel)
        switch (g_debug_k7_bits) {
            case 2: hipLaunchKernelGGL((render_track_fused_kernel<false, 2>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep}); break;
            case 4: hipLaunchKernelGGL((render_track_fused_kernel<false, 4>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep}); break;
            case 8: hipLaunchKernelGGL((render_track_fused_kernel<false, 8>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep}); break;
            case 16: hipLaunchKernelGGL((render_track_fused_kernel<false, 16>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep}); break;
            default: hipLaunchKernelGGL((render_track_fused_kernel<false>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep}); break;
        }
        return hipGetLastError();
    }
    if (full_sums) {
        if (keep_planes) hipLaunchKernelGGL((render_track_fused_full_kernel<true>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep});
        else hipLaunchKernelGGL((render_track_fused_full_kernel<false>), dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep});
        return hipGetLastError();
    }
    if (keep_planes) {
        auto k = render_track_fused_kernel<true>;
        hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep});
    } else {
        auto k = render_track_fused_kernel<false>;
        hipLaunchKernelGGL(k, dim3(8 * per), dim3(256), 0, s, TrackFusedArgs{cam, feat8, st, out6, accum, T, per, ep});
    }
    return hipGetLastError();
}

hipError_t launch_render_backward_feat8(const SplatCamera &cam, const float *feat8, const SplatState &st, const float *dL_dout6,
                                        float *accum, int P, bool zero_accum, bool rgb_sums, hipStream_t s, bool opacity_sum) {
    const int T = launch_tiles(cam, st);
    if (zero_accum) {
        hipError_t e = hipMemsetAsync(accum, 0, sizeof(float) * SPLAT_GRAD_STRIDE * (size_t)P, s);
        if (e != hipSuccess) return e;
    }
    if (T == 0 || P == 0) return hipSuccess;
    // channels r, g, b, z carry gradient; the silhouette and depth^2 planes never do.  dL/drgb is only summed on request
    // (tracking does not read it: LR 0 in /root/reference/configs/*/splatam.py, optimizer discarded after the frame).
    // (zero background: FusedEngine refuses anything else, as setup_camera builds it)
    if (rgb_sums) launch_bwd<6, 8, 0xFu, 0xFu, true, false>(cam, feat8, st, dL_dout6, accum, T, s);
    else if (opacity_sum) launch_bwd<6, 8, 0xFu, 0x8u, true, false>(cam, feat8, st, dL_dout6, accum, T, s);
    else launch_bwd<6, 8, 0xFu, 0x8u, false, false>(cam, feat8, st, dL_dout6, accum, T, s);      // camera tracking: no dL/dopacity wanted
    return hipGetLastError();
}

// The colour pass' own geometric sums S1..S5 (gradient planes r, g, b only; no colour sums, no opacity sum): the extra pass of
// splat_iter_means2d_accumulate.  `accum` must be zero on entry (the fused iteration leaves it so).
hipError_t launch_render_backward_rgb_only(const SplatCamera &cam, const float *feat8, const SplatState &st, const float *dL_dout6,
                                           float *accum, int P, hipStream_t s) {
    const int T = launch_tiles(cam, st);
    if (T == 0 || P == 0) return hipSuccess;
    const int per = (T + 7) / 8;
    hipLaunchKernelGGL((render_backward_kernel5_w5<6, 8, 0x7u, 0x0u, false, false>), dim3(8 * per), dim3(256), 0, s, BwdArgs{cam, feat8, st, dL_dout6, accum, T, per, nullptr});
    return hipGetLastError();
}

}  // namespace splat
