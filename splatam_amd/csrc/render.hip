// render.hip -- tile-wise alpha compositing, forward (K6) and backward (K7).
//
// MI355X mapping: ONE wave64 per 16x16 tile, FOUR pixels per lane (lane l owns
// pixel (l&7, l>>3) of each 8x8 quadrant).  Consequences:
//   * no workgroup barriers and no cross-wave reductions: "is every pixel of
//     the tile done" and "did anybody in the tile touch this Gaussian" are
//     single wave-level ballots;
//   * the tile's depth-sorted list is staged 64 Gaussians at a time through a
//     wave-private LDS slab (each lane gathers one Gaussian's record from the
//     L2-resident SoA arrays, next batch prefetched into registers while the
//     current one is composited); the inner loop reads it with uniform-address
//     (broadcast) ds_read_b128/b64, amortised over 4 pixels per lane;
//   * backward: a Gaussian's 6+C partial sums are first added over the lane's
//     4 pixels in registers, then reduced across the 64 lanes four values at a
//     time with v_permlane32_swap / v_permlane16_swap + 4 DPP adds, and leave
//     the wave as ONE float atomic per (Gaussian, tile, component).
//
// Arithmetic: SURVEY.md Appendix A "Forward composite (K6)" / "Backward
// composite (K7)" -- the callee of /root/reference/scripts/splatam.py:249,253
// and of the autograd backward reached from :702,854.
#include "splat_device.h"

namespace splat {

template <int C>
struct Staged {
    unsigned id;
    float4 co;      // conic xx, xy, yy + opacity
    float2 xy;
    float feat[C + 1];   // colours, then depth
};

template <int C, bool WITH_DEPTH>
__device__ __forceinline__ void gather(Staged<C> &s, const SplatState &st, const float *colors, unsigned idx, bool valid) {
    s.id = 0; s.co = make_float4(0.f, 0.f, 0.f, 0.f); s.xy = make_float2(0.f, 0.f);
#pragma unroll
    for (int f = 0; f <= C; ++f) s.feat[f] = 0.f;
    if (valid) {
        const unsigned id = st.point_list[idx];
        s.id = id;
        s.co = reinterpret_cast<const float4 *>(st.conic_opacity)[id];
        s.xy = reinterpret_cast<const float2 *>(st.xy)[id];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) s.feat[ch] = colors[(size_t)id * C + ch];
        if (WITH_DEPTH) s.feat[C] = st.depth[id];
    }
}

// LDS slab of one wave: 64 staged Gaussians.  Features are [f][64] (conflict-free
// writes, broadcast reads) except for the 4-float case which uses one float4.
template <int C>
struct Slab {
    float4 co[64];
    float2 xy[64];
    unsigned id[64];
    float feat[(C + 1) * 64];
};

template <int C>
__device__ __forceinline__ void commit(Slab<C> &sl, const Staged<C> &s, int lane) {
    sl.co[lane] = s.co;
    sl.xy[lane] = s.xy;
    sl.id[lane] = s.id;
    if constexpr (C + 1 == 4) {
        reinterpret_cast<float4 *>(sl.feat)[lane] = make_float4(s.feat[0], s.feat[1], s.feat[2], s.feat[3]);
    } else {
#pragma unroll
        for (int f = 0; f <= C; ++f) sl.feat[f * 64 + lane] = s.feat[f];
    }
}

template <int C>
__device__ __forceinline__ void read_feat(const Slab<C> &sl, int j, float *feat) {
    if constexpr (C + 1 == 4) {
        const float4 v = reinterpret_cast<const float4 *>(sl.feat)[j];
        feat[0] = v.x; feat[1] = v.y; feat[2] = v.z; feat[3] = v.w;
    } else {
#pragma unroll
        for (int f = 0; f <= C; ++f) feat[f] = sl.feat[f * 64 + j];
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
    const unsigned lo = st.tile_base[tile], hi = st.tile_base[tile + 1];
    const int n = (int)(hi - lo);

    Staged<C> pre;
    gather<C, true>(pre, st, colors, lo + lane, lane < n);
    for (int base = 0; base < n; base += 64) {
        if (__all(done[0] && done[1] && done[2] && done[3])) break;
        commit<C>(sl, pre, lane);
        __syncthreads();
        if (base + 64 < n) gather<C, true>(pre, st, colors, lo + base + 64 + lane, base + 64 + lane < n);
        const int cnt = min(64, n - base);
        for (int j = 0; j < cnt; ++j) {
            const float4 co = sl.co[j];
            const float2 g = sl.xy[j];
            const float dx[2] = {g.x - fpx[0], g.x - fpx[1]}, dy[2] = {g.y - fpy[0], g.y - fpy[1]};
            float alpha[4];
            bool live[4], any_live = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ddx = dx[k & 1], ddy = dy[k >> 1];
                const float power = -0.5f * (co.x * ddx * ddx + co.z * ddy * ddy) - co.y * ddx * ddy;
                alpha[k] = fminf(kAlphaMax, co.w * fast_exp2(power * kLog2e));
                live[k] = !done[k] && power <= 0.f && alpha[k] >= kAlphaMin;
                any_live |= live[k];
            }
            if (!__any(any_live)) continue;
            float feat[C + 1];
            read_feat<C>(sl, j, feat);
            const unsigned pos = (unsigned)(base + j + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (live[k]) {
                    const float test_T = T[k] * (1.f - alpha[k]);
                    if (test_T < kTStop) {
                        done[k] = true;
                    } else {
                        const float w = alpha[k] * T[k];
#pragma unroll
                        for (int ch = 0; ch < C; ++ch) Cc[k][ch] += feat[ch] * w;
                        D[k] += feat[C] * w;
                        T[k] = test_T;
                        last[k] = pos;
                    }
                }
            }
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
    const size_t HW = (size_t)H * W;

    float T[4], Tfin[4], dpix[4][C], bgdot[4], arec[4][C], lcol[4][C], lalpha[4];
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
        bgdot[k] = 0.f; lalpha[k] = 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            dpix[k][ch] = inside ? dL_dcolor[ch * HW + pix] : 0.f;
            bgdot[k] += cam.bg[ch] * dpix[k][ch];
            arec[k][ch] = 0.f; lcol[k][ch] = 0.f;
        }
    }
    max_last = wave_max_u32(max_last);
    if (max_last == 0) return;
    const unsigned lo = st.tile_base[tile];
    const int nb = (int)((max_last + 63) / 64);

    Staged<C> pre;
    gather<C, false>(pre, st, colors, lo + (nb - 1) * 64 + lane, (unsigned)((nb - 1) * 64 + lane) < max_last);
    for (int b = nb - 1; b >= 0; --b) {
        commit<C>(sl, pre, lane);
        __syncthreads();
        if (b > 0) gather<C, false>(pre, st, colors, lo + (b - 1) * 64 + lane, true);
        const int jhi = min(64, (int)max_last - b * 64);
        for (int j = jhi - 1; j >= 0; --j) {
            const unsigned pos = (unsigned)(b * 64 + j + 1);
            const float4 co = sl.co[j];
            const float2 g = sl.xy[j];
            const float dx[2] = {g.x - fpx[0], g.x - fpx[1]}, dy[2] = {g.y - fpy[0], g.y - fpy[1]};
            float G[4], alpha[4];
            bool live[4], any_live = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ddx = dx[k & 1], ddy = dy[k >> 1];
                const float power = -0.5f * (co.x * ddx * ddx + co.z * ddy * ddy) - co.y * ddx * ddy;
                G[k] = fast_exp2(power * kLog2e);
                alpha[k] = fminf(kAlphaMax, co.w * G[k]);
                live[k] = pos <= last[k] && power <= 0.f && alpha[k] >= kAlphaMin;
                any_live |= live[k];
            }
            if (!__any(any_live)) continue;
            float feat[C + 1];
            read_feat<C>(sl, j, feat);
            float s[NG * 4];
#pragma unroll
            for (int v = 0; v < NG * 4; ++v) s[v] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (live[k]) {
                    const float ddx = dx[k & 1], ddy = dy[k >> 1];
                    const float rcp = __builtin_amdgcn_rcpf(1.f - alpha[k]);
                    T[k] = T[k] * rcp;
                    const float w = alpha[k] * T[k];
                    float dL_dalpha = 0.f;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        arec[k][ch] = lalpha[k] * lcol[k][ch] + (1.f - lalpha[k]) * arec[k][ch];
                        lcol[k][ch] = feat[ch];
                        dL_dalpha += (feat[ch] - arec[k][ch]) * dpix[k][ch];
                        s[6 + ch] += w * dpix[k][ch];
                    }
                    dL_dalpha *= T[k];
                    lalpha[k] = alpha[k];
                    if (has_bg) dL_dalpha += (-Tfin[k] * rcp) * bgdot[k];
                    const float q = co.w * dL_dalpha;
                    const float gdx = G[k] * ddx, gdy = G[k] * ddy;
                    s[0] += q * gdx;
                    s[1] += q * gdy;
                    s[2] += q * gdx * ddx;
                    s[3] += q * gdx * ddy;
                    s[4] += q * gdy * ddy;
                    s[5] += G[k] * dL_dalpha;
                }
            }
            float *dst = accum + (size_t)sl.id[j] * SPLAT_GRAD_STRIDE;
            const int rv = row_value(lane >> 4);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                const float r = wave_reduce4_packed(s[4 * grp], s[4 * grp + 1], s[4 * grp + 2], s[4 * grp + 3]);
                if ((lane & 15) == 0 && 4 * grp + rv < NV) atomicAdd(dst + 4 * grp + rv, r);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int C>
static void launch_fwd(const SplatCamera &cam, const float *colors, SplatState &st, float *oc, float *od, int T, hipStream_t s) {
    hipLaunchKernelGGL((render_forward_kernel<C>), dim3(T), dim3(64), 0, s, cam, colors, st, oc, od);
}
template <int C>
static void launch_bwd(const SplatCamera &cam, const float *colors, const SplatState &st, const float *dl, float *acc, int T,
                       hipStream_t s) {
    hipLaunchKernelGGL((render_backward_kernel<C>), dim3(T), dim3(64), 0, s, cam, colors, st, dl, acc);
}

static const float *colour_source(const SplatGaussians &g, const SplatState &st) {
    return g.shs ? st.rgb : g.colors_precomp;
}

hipError_t launch_render_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, float *out_color,
                                 float *out_depth, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    const float *col = colour_source(g, st);
    if (T == 0) return hipSuccess;
    switch (g.channels) {
        case 1: launch_fwd<1>(cam, col, st, out_color, out_depth, T, s); break;
        case 2: launch_fwd<2>(cam, col, st, out_color, out_depth, T, s); break;
        case 3: launch_fwd<3>(cam, col, st, out_color, out_depth, T, s); break;
        case 4: launch_fwd<4>(cam, col, st, out_color, out_depth, T, s); break;
        case 5: launch_fwd<5>(cam, col, st, out_color, out_depth, T, s); break;
        case 6: launch_fwd<6>(cam, col, st, out_color, out_depth, T, s); break;
        case 7: launch_fwd<7>(cam, col, st, out_color, out_depth, T, s); break;
        case 8: launch_fwd<8>(cam, col, st, out_color, out_depth, T, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_render_backward(const SplatCamera &cam, const SplatGaussians &g, const SplatState &st, SplatGrads &gr,
                                  hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    const float *col = colour_source(g, st);
    hipError_t e = hipMemsetAsync(gr.accum, 0, sizeof(float) * SPLAT_GRAD_STRIDE * (size_t)g.P, s);
    if (e != hipSuccess) return e;
    if (T == 0 || g.P == 0) return hipSuccess;
    switch (g.channels) {
        case 1: launch_bwd<1>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 2: launch_bwd<2>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 3: launch_bwd<3>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 4: launch_bwd<4>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 5: launch_bwd<5>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 6: launch_bwd<6>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 7: launch_bwd<7>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        case 8: launch_bwd<8>(cam, col, st, gr.dL_dcolor, gr.accum, T, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace splat
