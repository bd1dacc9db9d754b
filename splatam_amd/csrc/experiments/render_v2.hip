// render_v2.hip -- PREVIOUS generation of the composite kernels (one wave64 per 16x16 tile, four pixels per
// lane), kept only for A/B timing against render.hip: splat_debug_option(1, 2) routes 3-channel calls here.
//
// MI355X mapping: ONE wave64 per 16x16 tile, FOUR pixels per lane (lane l owns
// pixel (l&7, l>>3) of each 8x8 quadrant).  Consequences:
//   * no workgroup barriers and no cross-wave reductions: "is every pixel of
//     the tile done" and "did anybody in the tile touch this Gaussian" are
//     single wave-level ballots;
//   * the tile's depth-sorted list is staged 64 Gaussians at a time through a
//     wave-private LDS slab (each lane gathers one Gaussian's record from the
//     L2-resident SoA arrays, next batch prefetched into registers while the
//     current one is composited); the inner loop reads the slab with
//     uniform-address (broadcast) ds_read_b128, one Gaussian ahead of use,
//     amortised over 4 pixels per lane;
//   * exact quadrant culling: the staging lane computes, per Gaussian, which of
//     the tile's four 8x8 quadrants can contain a pixel with alpha >= 1/255
//     (axis-aligned box of the ellipse power >= -ln(255*opacity), inflated by a
//     rounding margin).  The 4-bit mask is wave-uniform, so a dead quadrant
//     costs one scalar branch instead of 64 lanes of exp/compare work, and
//     results are unchanged (a culled pixel would have failed the alpha test);
//   * backward: a Gaussian's 6+C partial sums are first added over the lane's
//     4 pixels in registers, then reduced across the 64 lanes four values at a
//     time with v_permlane32_swap / v_permlane16_swap + 4 DPP adds, and leave
//     the wave as ONE float atomic per (Gaussian, tile, component).
//
// Arithmetic: SURVEY.md Appendix A "Forward composite (K6)" / "Backward
// composite (K7)" -- the callee of /root/reference/scripts/splatam.py:249,253
// and of the autograd backward reached from :702,854.  exp(power) is evaluated
// as v_exp_f32(power * log2 e) with log2 e folded into the staged conic.
#include "../splat_device.h"

namespace splat {
namespace v2 {

// One staged Gaussian, as the gathering lane holds it in registers.
template <int C>
struct Staged {
    float4 ga;          // A = -0.5*cxx*log2e, B = -cxy*log2e, Cq = -0.5*cyy*log2e, opacity
    float4 gb;          // mu_x, mu_y, quadrant mask (bits), Gaussian id (bits)
    float feat[C + 1];  // colours, then depth (forward only)
};

// LDS slab of one wave: 64 staged Gaussians.
template <int C>
struct Slab {
    float4 ga[64];
    float4 gb[64];
    float feat[(C + 1) * 64];                // C == 3: one float4 {r,g,b,depth} per Gaussian; otherwise [c][64]
};

template <int C, bool WITH_DEPTH>
__device__ __forceinline__ void gather(Staged<C> &s, const SplatState &st, const float *colors, unsigned idx, bool valid,
                                       float tile_x0, float tile_y0) {
    s.ga = make_float4(0.f, 0.f, 0.f, 0.f);
    s.gb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int f = 0; f <= C; ++f) s.feat[f] = 0.f;
    if (valid) {
        const unsigned id = st.point_list[idx];
        const float4 co = reinterpret_cast<const float4 *>(st.conic_opacity)[id];
        const float2 mu = reinterpret_cast<const float2 *>(st.xy)[id];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) s.feat[ch] = colors[(size_t)id * C + ch];
        if (WITH_DEPTH) s.feat[C] = st.depth[id];
        // live region {alpha >= 1/255}: (p-mu)^T Q (p-mu) <= 2 tau, tau = ln(255 o); half extents sqrt(2 tau Q^-1_ii)
        unsigned mask = 0;
        const float tau2 = 2.0f * __logf(255.0f * co.w);
        if (tau2 >= 0.f) {
            const float det = co.x * co.z - co.y * co.y;
            const float hx = sqrtf(tau2 * co.z / det) * 1.00001f + 0.01f;
            const float hy = sqrtf(tau2 * co.x / det) * 1.00001f + 0.01f;
            const bool x0 = (mu.x - hx <= tile_x0 + 7.f) && (mu.x + hx >= tile_x0);
            const bool x1 = (mu.x - hx <= tile_x0 + 15.f) && (mu.x + hx >= tile_x0 + 8.f);
            const bool y0 = (mu.y - hy <= tile_y0 + 7.f) && (mu.y + hy >= tile_y0);
            const bool y1 = (mu.y - hy <= tile_y0 + 15.f) && (mu.y + hy >= tile_y0 + 8.f);
            mask = (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
            if (!(hx == hx) || !(hy == hy)) mask = 15u;     // NaN geometry: no culling, let it propagate as the reference would
        }
        s.ga = make_float4(-0.5f * kLog2e * co.x, -kLog2e * co.y, -0.5f * kLog2e * co.z, co.w);
        s.gb = make_float4(mu.x, mu.y, __uint_as_float(mask), __uint_as_float(id));
    }
}

template <int C>
__device__ __forceinline__ void commit(Slab<C> &sl, const Staged<C> &s, int lane) {
    sl.ga[lane] = s.ga;
    sl.gb[lane] = s.gb;
    if constexpr (C == 3) {
        reinterpret_cast<float4 *>(sl.feat)[lane] = make_float4(s.feat[0], s.feat[1], s.feat[2], s.feat[3]);
    } else {
#pragma unroll
        for (int f = 0; f <= C; ++f) sl.feat[f * 64 + lane] = s.feat[f];
    }
}

template <int C>
struct Entry {          // one slab entry as the inner loop holds it (wave-uniform values)
    float4 ga, gb;
    float feat[C + 1];
};

template <int C>
__device__ __forceinline__ void read_entry(const Slab<C> &sl, int j, Entry<C> &e) {
    e.ga = sl.ga[j];
    e.gb = sl.gb[j];
    if constexpr (C == 3) {
        const float4 v = reinterpret_cast<const float4 *>(sl.feat)[j];
        e.feat[0] = v.x; e.feat[1] = v.y; e.feat[2] = v.z; e.feat[3] = v.w;
    } else {
#pragma unroll
        for (int f = 0; f <= C; ++f) e.feat[f] = sl.feat[f * 64 + j];
    }
}

// ---------------------------------------------------------------------------
// K6 forward composite
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(64) void render_forward_kernel(SplatCamera cam, const float *colors, SplatState st,
                                                            float *out_color, float *out_depth) {
    __shared__ Slab<C> sl;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px0 = tx * kTile + (lane & 7), py0 = ty * kTile + (lane >> 3);
    const float fpx[2] = {(float)px0, (float)(px0 + 8)}, fpy[2] = {(float)py0, (float)(py0 + 8)};
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    bool inside[4], done[4];
    float T[4], D[4], Cc[4][C];
    unsigned last[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        inside[k] = (px0 + 8 * (k & 1) < W) && (py0 + 8 * (k >> 1) < H);
        done[k] = !inside[k];
        T[k] = 1.f; D[k] = 0.f; last[k] = 0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) Cc[k][ch] = 0.f;
    }
    // wave-uniform: quadrants that still have a pixel to composite
    unsigned qalive = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) qalive |= __any(!done[k]) ? (1u << k) : 0u;

    const unsigned lo = st.tile_base[tile], hi = st.tile_base[tile + 1];
    const int n = (int)(hi - lo);

    Staged<C> pre;
    gather<C, true>(pre, st, colors, lo + lane, lane < n, tile_x0, tile_y0);
    for (int base = 0; base < n && qalive != 0; base += 64) {
        commit<C>(sl, pre, lane);
        __syncthreads();
        if (base + 64 < n) gather<C, true>(pre, st, colors, lo + base + 64 + lane, base + 64 + lane < n, tile_x0, tile_y0);
        const int cnt = min(64, n - base);
        Entry<C> cur, nxt;
        read_entry<C>(sl, 0, cur);
        for (int j = 0; j < cnt; ++j) {
            read_entry<C>(sl, (j + 1) & 63, nxt);          // one Gaussian ahead: LDS latency hidden behind the blend
            const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cur.gb.z)) & qalive;
            if (mask != 0) {
                const float dx[2] = {cur.gb.x - fpx[0], cur.gb.x - fpx[1]}, dy[2] = {cur.gb.y - fpy[0], cur.gb.y - fpy[1]};
                const unsigned pos = (unsigned)(base + j + 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (mask & (1u << k)) {                // scalar branch
                        const float ddx = dx[k & 1], ddy = dy[k >> 1];
                        const float p2 = ddx * (cur.ga.x * ddx + cur.ga.y * ddy) + cur.ga.z * ddy * ddy;   // power * log2(e)
                        const float alpha = fminf(kAlphaMax, cur.ga.w * fast_exp2(p2));
                        const bool live = !done[k] && p2 <= 0.f && alpha >= kAlphaMin;
                        if (__any(live)) {
                            const float test_T = T[k] * (1.f - alpha);
                            const bool stop = live && test_T < kTStop;
                            const bool upd = live && !stop;
                            const float w = upd ? alpha * T[k] : 0.f;
#pragma unroll
                            for (int ch = 0; ch < C; ++ch) Cc[k][ch] += cur.feat[ch] * w;
                            D[k] += cur.feat[C] * w;
                            T[k] = upd ? test_T : T[k];
                            last[k] = upd ? pos : last[k];
                            if (__any(stop)) {
                                done[k] = done[k] || stop;
                                if (!__any(!done[k])) qalive &= ~(1u << k);
                            }
                        }
                    }
                }
            }
            cur = nxt;
        }
        __syncthreads();
    }
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (inside[k]) {
            const size_t pix = (size_t)(py0 + 8 * (k >> 1)) * W + (px0 + 8 * (k & 1));
            st.final_T[pix] = T[k];
            st.n_contrib[pix] = (int)last[k];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out_color[ch * HW + pix] = Cc[k][ch] + T[k] * cam.bg[ch];
            out_depth[pix] = D[k];
        }
    }
}

// ---------------------------------------------------------------------------
// K7 backward composite
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(64) void render_backward_kernel(SplatCamera cam, const float *colors, SplatState st,
                                                             const float *dL_dcolor, float *accum) {
    constexpr int NV = 6 + C;                 // partial sums per Gaussian
    constexpr int NG = (NV + 3) / 4;          // packed reduction groups
    __shared__ Slab<C> sl;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int px0 = tx * kTile + (lane & 7), py0 = ty * kTile + (lane >> 3);
    const float fpx[2] = {(float)px0, (float)(px0 + 8)}, fpy[2] = {(float)py0, (float)(py0 + 8)};
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    const size_t HW = (size_t)H * W;

    // Per pixel: running transmittance, and the scalar form of the "colour behind" recursion:
    // with cdot_i = sum_ch c_i[ch] dL/dC[ch], the reference's accum_rec[ch] only ever enters through
    // behind = sum_ch accum_rec[ch] dL/dC[ch], which obeys behind <- a_prev cdot_prev + (1-a_prev) behind.
    float T[4], Tfin[4], dpix[4][C], bgdot[4], behind[4], lcdot[4], lalpha[4];
    unsigned last[4];
    unsigned max_last = 0;
    bool has_bg = false;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) has_bg |= cam.bg[ch] != 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool inside = (px0 + 8 * (k & 1) < W) && (py0 + 8 * (k >> 1) < H);
        const size_t pix = (size_t)(py0 + 8 * (k >> 1)) * W + (px0 + 8 * (k & 1));
        Tfin[k] = inside ? st.final_T[pix] : 0.f;
        T[k] = Tfin[k];
        last[k] = inside ? (unsigned)st.n_contrib[pix] : 0u;
        max_last = max(max_last, last[k]);
        bgdot[k] = 0.f; lalpha[k] = 0.f; behind[k] = 0.f; lcdot[k] = 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            dpix[k][ch] = inside ? dL_dcolor[ch * HW + pix] : 0.f;
            bgdot[k] += cam.bg[ch] * dpix[k][ch];
        }
    }
    max_last = wave_max_u32(max_last);
    if (max_last == 0) return;
    const unsigned lo = st.tile_base[tile];
    const int nb = (int)((max_last + 63) / 64);

    Staged<C> pre;
    gather<C, false>(pre, st, colors, lo + (nb - 1) * 64 + lane, (unsigned)((nb - 1) * 64 + lane) < max_last, tile_x0, tile_y0);
    for (int b = nb - 1; b >= 0; --b) {
        commit<C>(sl, pre, lane);
        __syncthreads();
        if (b > 0) gather<C, false>(pre, st, colors, lo + (b - 1) * 64 + lane, true, tile_x0, tile_y0);
        const int jhi = min(64, (int)max_last - b * 64);
        Entry<C> cur, nxt;
        read_entry<C>(sl, jhi - 1, cur);
        for (int j = jhi - 1; j >= 0; --j) {
            read_entry<C>(sl, (j - 1) & 63, nxt);
            const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cur.gb.z));
            if (mask != 0) {
                const unsigned pos = (unsigned)(b * 64 + j + 1);
                const float dx[2] = {cur.gb.x - fpx[0], cur.gb.x - fpx[1]}, dy[2] = {cur.gb.y - fpy[0], cur.gb.y - fpy[1]};
                float s[NG * 4];
#pragma unroll
                for (int v = 0; v < NG * 4; ++v) s[v] = 0.f;
                bool touched = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (mask & (1u << k)) {
                        const float ddx = dx[k & 1], ddy = dy[k >> 1];
                        const float p2 = ddx * (cur.ga.x * ddx + cur.ga.y * ddy) + cur.ga.z * ddy * ddy;
                        const float G = fast_exp2(p2);
                        const float alpha = fminf(kAlphaMax, cur.ga.w * G);
                        const bool live = pos <= last[k] && p2 <= 0.f && alpha >= kAlphaMin;
                        if (__any(live)) {
                            touched = true;
                            const float rcp = __builtin_amdgcn_rcpf(1.f - alpha);
                            const float Tn = T[k] * rcp;                     // transmittance in front of this Gaussian
                            float cdot = 0.f;
#pragma unroll
                            for (int ch = 0; ch < C; ++ch) cdot += cur.feat[ch] * dpix[k][ch];
                            const float bh = lalpha[k] * lcdot[k] + (1.f - lalpha[k]) * behind[k];
                            float dL_dalpha = (cdot - bh) * Tn;
                            if (has_bg) dL_dalpha += (-Tfin[k] * rcp) * bgdot[k];
                            // selects (not multiplies by 0) so that a non-live lane can never inject inf * 0
                            const float Gl = live ? G : 0.f;
                            const float w = live ? alpha * Tn : 0.f;
                            const float q = live ? cur.ga.w * dL_dalpha : 0.f;   // dL/dG
                            const float gdx = Gl * ddx, gdy = Gl * ddy;
                            const float qgx = q * gdx, qgy = q * gdy;
                            s[0] += qgx;
                            s[1] += qgy;
                            s[2] += qgx * ddx;
                            s[3] += qgx * ddy;
                            s[4] += qgy * ddy;
                            s[5] += live ? Gl * dL_dalpha : 0.f;
#pragma unroll
                            for (int ch = 0; ch < C; ++ch) s[6 + ch] += w * dpix[k][ch];
                            T[k] = live ? Tn : T[k];
                            behind[k] = live ? bh : behind[k];
                            lcdot[k] = live ? cdot : lcdot[k];
                            lalpha[k] = live ? alpha : lalpha[k];
                        }
                    }
                }
                if (touched) {
                    float *dst = accum + (size_t)__float_as_uint(cur.gb.w) * SPLAT_GRAD_STRIDE;
                    const int rv = row_value(lane >> 4);
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp) {
                        const float r = wave_reduce4_packed(s[4 * grp], s[4 * grp + 1], s[4 * grp + 2], s[4 * grp + 3]);
                        if ((lane & 15) == 0 && 4 * grp + rv < NV) atomicAdd(dst + 4 * grp + rv, r);
                    }
                }
            }
            cur = nxt;
        }
        __syncthreads();
    }
}

}  // namespace v2
using namespace v2;

// ---------------------------------------------------------------------------
// launchers (3-channel A/B path only)
// ---------------------------------------------------------------------------
hipError_t launch_render_forward_v2(const SplatCamera &cam, const float *col, SplatState &st, float *out_color,
                                    float *out_depth, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    if (T == 0) return hipSuccess;
    hipLaunchKernelGGL((render_forward_kernel<3>), dim3(T), dim3(64), 0, s, cam, col, st, out_color, out_depth);
    return hipGetLastError();
}

hipError_t launch_render_backward_v2(const SplatCamera &cam, const float *col, const SplatState &st, const float *dL_dcolor,
                                     float *accum, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    if (T == 0) return hipSuccess;
    hipLaunchKernelGGL((render_backward_kernel<3>), dim3(T), dim3(64), 0, s, cam, col, st, dL_dcolor, accum);
    return hipGetLastError();
}

}  // namespace splat
