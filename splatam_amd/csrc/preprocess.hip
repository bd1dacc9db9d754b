// preprocess.hip -- per-Gaussian kernels: forward projection (K1, fused with the
// per-tile instance count), the single-workgroup tile scan (K2), the adjoint
// chain rule (K8+K9 fused) and mark_visible (K10).  One lane per Gaussian; the
// [P][3]/[P][4] inputs are read straight from the caller's tensors (a wave
// touches one contiguous 768 B / 1 KiB span per attribute).
#include "splat_device.h"

namespace splat {

constexpr int kBlock = 256;
int g_debug_skip_count = 0;     // splat_debug_option(0, v): timing experiment only

// K1: Appendix A steps 1-9 for Gaussian i (geometry, SH colour); returns its visibility, `o` holds the tile rectangle.
__device__ __forceinline__ bool preprocess_forward_one(const SplatCamera &cam, const CamConst &c, const SplatGaussians &g, const SplatState &st, int i,
                                                       Projected &o) {
    const float p[3] = {g.means3D[3 * i], g.means3D[3 * i + 1], g.means3D[3 * i + 2]};
    float S6[6];
    if (g.cov3D_precomp) {
        for (int k = 0; k < 6; ++k) S6[k] = g.cov3D_precomp[6 * i + k];
    } else {
        const float s[3] = {g.scales[3 * i], g.scales[3 * i + 1], g.scales[3 * i + 2]};
        const float q[4] = {g.rotations[4 * i], g.rotations[4 * i + 1], g.rotations[4 * i + 2], g.rotations[4 * i + 3]};
        cov3d_from_scale_rot(s, c.scale_modifier, q, S6);
    }
    const bool vis = project_gaussian(c, p, S6, o);
    st.depth[i] = o.depth;
    reinterpret_cast<float2 *>(st.xy)[i] = make_float2(o.px, o.py);
    reinterpret_cast<float4 *>(st.conic_opacity)[i] = make_float4(o.conic[0], o.conic[1], o.conic[2], g.opacities[i]);
    reinterpret_cast<uint2 *>(st.rect)[i] = make_uint2((unsigned)o.x0 | ((unsigned)o.y0 << 16), (unsigned)o.x1 | ((unsigned)o.y1 << 16));
    st.radii[i] = o.radius;
    if (g.shs) {
        // colour from spherical harmonics, clamped at 0 (flags kept for the adjoint)
        float d[3] = {p[0] - cam.campos[0], p[1] - cam.campos[1], p[2] - cam.campos[2]};
        const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] *= inv; d[1] *= inv; d[2] *= inv;
        float basis[16];
        sh_basis(cam.sh_degree, d, basis, nullptr);
        const int nb = (cam.sh_degree + 1) * (cam.sh_degree + 1);
        const float *sh = g.shs + (size_t)i * g.sh_coeffs * 3;
        for (int ch = 0; ch < 3; ++ch) {
            float v = 0.f;
            for (int k = 0; k < nb; ++k) v += basis[k] * sh[3 * k + ch];
            v += 0.5f;
            st.clamped[3 * i + ch] = v < 0.f;
            st.rgb[3 * i + ch] = fmaxf(v, 0.f);
        }
    }
    return vis;
}

// K1 + one atomicAdd per touched tile.
__global__ __launch_bounds__(kBlock) void preprocess_forward_kernel(SplatCamera cam, SplatGaussians g, SplatState st, int skip_count) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.P) return;
    CamConst c;
    load_cam(c, cam);
    Projected o;
    const bool vis = preprocess_forward_one(cam, c, g, st, i, o);
    if (vis && !skip_count) {
        for (int y = o.y0; y < o.y1; ++y)
            for (int x = o.x0; x < o.x1; ++x) atomicAdd(&st.tile_count[sub_counter(st, y * c.gx + x, i)], 1u);
    }
}

// K1 with GROUP BINNING (SplatState.group_count / group_recs / group_stride, bucketed lists the forward composite sorts itself: the
// front end of the fused iteration, splat_device.h file_group_records): one 16-byte record per touched group of 2 x 2 tiles instead of a
// count per tile now and a scatter later -- no tile scan, no scatter pass, no sort launch (K2-K4), and nothing the host must read
// before it may launch the composite.  The caller knows (from an earlier call on this scene: SplatState.max_list_hint) that the lists
// are short; a list that has outgrown the hint is flagged (status[1] / status[3]) by the composite and the call must be repeated on
// exact lists.
__global__ __launch_bounds__(kGroupBlock) void preprocess_forward_group_kernel(SplatCamera cam, SplatGaussians g, SplatState st) {
    extern __shared__ unsigned s_grp[];
    const int i = blockIdx.x * kGroupBlock + threadIdx.x;
    const int ggx = tile_groups_x(cam.image_width), num_groups = tile_groups(cam.image_width, cam.image_height);
    group_hist_reset<kGroupBlock>(s_grp, num_groups);
    CamConst c;
    load_cam(c, cam);
    Projected o{};
    bool vis = false;
    if (i < g.P) {
        vis = preprocess_forward_one(cam, c, g, st, i, o);
        if (st.accum_to_zero) {         // the accumulator row the backward pass of this call will add into (SplatState.accum_to_zero)
            float4 *a4 = reinterpret_cast<float4 *>(st.accum_to_zero + (size_t)i * SPLAT_GRAD_STRIDE);
#pragma unroll
            for (int k = 0; k < SPLAT_GRAD_STRIDE / 4; ++k) a4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (vis) live_tile_rect(o.conic, g.opacities[i], o.px, o.py, o.lx0, o.ly0, o.lx1, o.ly1);      // the lists of this path are the library's own: the live part only
    file_group_records<kGroupBlock>(st, s_grp, i, vis && o.ly1 > o.ly0 && o.lx1 > o.lx0, o.lx0, o.ly0, o.lx1, o.ly1, o.depth, ggx, num_groups);
}

// K1 for exact lists known to be very long (dense_exact_lists, splat_device.h): the workgroup's instances are counted per tile in LDS and
// added to the tiles' counters (sub-bin: the workgroup's) with one atomic per non-empty tile -- 5 M clustered Gaussians on ~400 tiles took
// 497 us of same-address atomics with one per instance.
__global__ __launch_bounds__(kDenseThreads) void preprocess_forward_dense_kernel(SplatCamera cam, SplatGaussians g, SplatState st) {
    extern __shared__ unsigned s_tile[];
    const int tid = threadIdx.x;
    CamConst c;
    load_cam(c, cam);
    const int T = c.gx * c.gy;
    for (int t = tid; t < T; t += kDenseThreads) s_tile[t] = 0u;
    __syncthreads();
    for (int k = 0; k < kDenseGaussians / kDenseThreads; ++k) {
        const int i = blockIdx.x * kDenseGaussians + k * kDenseThreads + tid;
        if (i < g.P) {
            Projected o;
            if (preprocess_forward_one(cam, c, g, st, i, o))
                for (int y = o.y0; y < o.y1; ++y)
                    for (int x = o.x0; x < o.x1; ++x) atomicAdd(&s_tile[y * c.gx + x], 1u);
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += kDenseThreads) {
        const unsigned cnt = s_tile[t];
        if (cnt) atomicAdd(&st.tile_count[sub_counter(st, t, (int)blockIdx.x)], cnt);
    }
}

// K2: exclusive prefix sum of the per-tile instance counts, two launches of ceil(T / 256) workgroups (one thread per tile):
//   K2a tile_scan_partial_kernel   tile total = sum of its SUB-BIN counters, workgroup-local exclusive scan -> tile_base[t]
//                                  (workgroup-relative), workgroup total / maximum -> spare words of the cursor lines
//   K2b tile_scan_finish_kernel    workgroup offset = sum of the totals before it; tile_base[t] and the cursor of every sub-bin
//                                  made absolute, counters reset (the next call's K1 starts from zero without a memset), status
// Sub-bins (SplatState.sub_bins = S, a power of two): a tile's count / cursor atomics are spread over S counters, chosen by the
// low bits of the Gaussian index -- in the clustered stress scene (BASELINE config 5) 26 000 instances per tile hit ONE counter and
// the count and scatter passes were bound by same-address atomic serialisation (~34 ns each: 0.9 ms per pass at 11 M instances).
// A tile's list is still one contiguous range [tile_base[t], tile_base[t+1]): its sub-bins are laid end to end.
constexpr int kScanBlock = 256;
__device__ __forceinline__ int sub_bins_of(const SplatState &st) { return st.sub_bins > 1 ? st.sub_bins : 1; }
// spare words of workgroup b's first cursor line hold its total (word 1) and its longest list (word 2)
__device__ __forceinline__ uint32_t *scan_spare(const SplatState &st, int b, int S) { return st.tile_cursor + (size_t)b * kScanBlock * S * SPLAT_COUNTER_STRIDE; }

__device__ __forceinline__ unsigned block256_exclusive_scan(unsigned v, unsigned *s_tmp, unsigned *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) {
        const unsigned c = s_tmp[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void tile_scan_partial_kernel(SplatState st, int T) {
    __shared__ unsigned s_tmp[8];
    const int S = sub_bins_of(st);
    const int t = blockIdx.x * kScanBlock + threadIdx.x;
    unsigned n = 0;
    if (t < T)
        for (int k = 0; k < S; ++k) n += st.tile_count[((size_t)t * S + k) * SPLAT_COUNTER_STRIDE];
    unsigned total;
    const unsigned ex = block256_exclusive_scan(n, s_tmp, &total);
    if (t < T) st.tile_base[t] = ex;
    const unsigned mx = wave_max_u32(n);
    if ((threadIdx.x & 63) == 0) s_tmp[4 + (threadIdx.x >> 6)] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t *sp = scan_spare(st, blockIdx.x, S);
        sp[1] = total;
        sp[2] = max(max(s_tmp[4], s_tmp[5]), max(s_tmp[6], s_tmp[7]));
    }
}

__global__ __launch_bounds__(kScanBlock) void tile_scan_finish_kernel(SplatState st, int T) {
    __shared__ unsigned s_red[3][kScanBlock / 64];
    const int S = sub_bins_of(st);
    const int nb = (T + kScanBlock - 1) / kScanBlock;
    // offset of this workgroup, grand total and longest list: every workgroup reduces the (few dozen) partial records
    unsigned before = 0, all = 0, mx = 0;
    for (int b = threadIdx.x; b < nb; b += kScanBlock) {
        const uint32_t *sp = scan_spare(st, b, S);
        const unsigned v = sp[1];
        all += v;
        if (b < (int)blockIdx.x) before += v;
        mx = max(mx, sp[2]);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        before += (unsigned)__shfl_xor((int)before, m, 64);
        all += (unsigned)__shfl_xor((int)all, m, 64);
        mx = max(mx, (unsigned)__shfl_xor((int)mx, m, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = before; s_red[1][threadIdx.x >> 6] = all; s_red[2][threadIdx.x >> 6] = mx; }
    __syncthreads();
    before = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    all = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    mx = max(max(s_red[2][0], s_red[2][1]), max(s_red[2][2], s_red[2][3]));
    // lists that do not fit the caller's buffers are published as EMPTY (and flagged in status[1]): a composite
    // launched behind an overflowing binning must never index past keys / point_list
    const bool overflow = (long long)all > st.capacity;
    const int t = blockIdx.x * kScanBlock + threadIdx.x;
    if (t < T) {
        unsigned run = before + st.tile_base[t];
        st.tile_base[t] = overflow ? 0u : run;
        for (int k = 0; k < S; ++k) {
            const size_t c = ((size_t)t * S + k) * SPLAT_COUNTER_STRIDE;
            const unsigned n = st.tile_count[c];
            st.tile_cursor[c] = run;
            st.tile_count[c] = 0;            // consumed: the next call's K1 starts from zero without a memset
            run += n;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st.tile_base[T] = overflow ? 0u : all;
        st.status[0] = (int)all;
        st.status[1] = overflow ? 1 : 0;
        st.status[2] = (int)mx;
        st.status[3] = 0;
    }
}

// K8+K9 fused: from the per-Gaussian partial sums of the backward composite
// (accum[i] = {S1..S6, colour sums}) to every dL/d(input).
__global__ __launch_bounds__(kBlock) void preprocess_backward_kernel(SplatCamera cam, SplatGaussians g, SplatState st,
                                                                    SplatGrads gr) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.P) return;
    const int C = g.channels;
    const bool vis = st.radii[i] > 0;
    float acc[SPLAT_GRAD_STRIDE];
    {
        const float4 *a4 = reinterpret_cast<const float4 *>(gr.accum + (size_t)i * SPLAT_GRAD_STRIDE);
        for (int k = 0; k < SPLAT_GRAD_STRIDE / 4; ++k) {
            const float4 v = a4[k];
            acc[4 * k] = v.x; acc[4 * k + 1] = v.y; acc[4 * k + 2] = v.z; acc[4 * k + 3] = v.w;
        }
    }
    float dmean[3] = {0.f, 0.f, 0.f}, dS6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float g_ndc[2] = {0.f, 0.f};
    if (vis) {
        CamConst c;
        load_cam(c, cam);
        const float p[3] = {g.means3D[3 * i], g.means3D[3 * i + 1], g.means3D[3 * i + 2]};
        float S6[6];
        if (g.cov3D_precomp) {
            for (int k = 0; k < 6; ++k) S6[k] = g.cov3D_precomp[6 * i + k];
        } else {
            const float s[3] = {g.scales[3 * i], g.scales[3 * i + 1], g.scales[3 * i + 2]};
            const float q[4] = {g.rotations[4 * i], g.rotations[4 * i + 1], g.rotations[4 * i + 2], g.rotations[4 * i + 3]};
            cov3d_from_scale_rot(s, c.scale_modifier, q, S6);
        }
        const float4 co = reinterpret_cast<const float4 *>(st.conic_opacity)[i];
        // S1 = sum q*G*dx, S2 = sum q*G*dy, S3 = sum q*G*dx*dx, S4 = sum q*G*dx*dy, S5 = sum q*G*dy*dy
        g_ndc[0] = -(co.x * acc[0] + co.y * acc[1]) * 0.5f * c.W;
        g_ndc[1] = -(co.z * acc[1] + co.y * acc[0]) * 0.5f * c.H;
        const float g_conic[3] = {-0.5f * acc[2], -acc[3], -0.5f * acc[4]};
        project_gaussian_backward(c, p, S6, g_ndc, g_conic, dmean, dS6);
        if (gr.dL_dscales && g.scales) {
            const float s[3] = {g.scales[3 * i], g.scales[3 * i + 1], g.scales[3 * i + 2]};
            const float q[4] = {g.rotations[4 * i], g.rotations[4 * i + 1], g.rotations[4 * i + 2], g.rotations[4 * i + 3]};
            float ds[3], dq[4];
            cov3d_backward(s, c.scale_modifier, q, dS6, ds, dq, (gr.flags & SPLAT_GRADS_UPSTREAM_SCALE) != 0);
            for (int k = 0; k < 3; ++k) gr.dL_dscales[3 * i + k] = ds[k];
            for (int k = 0; k < 4; ++k) gr.dL_drotations[4 * i + k] = dq[k];
        }
        if (g.shs && gr.dL_dshs) {
            // adjoint of the SH colour: coefficients, and the centre through the view direction
            float d[3] = {p[0] - cam.campos[0], p[1] - cam.campos[1], p[2] - cam.campos[2]};
            const float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const float inv = 1.0f / sqrtf(n2);
            const float u[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
            float basis[16], db[3][16];
            sh_basis(cam.sh_degree, u, basis, db);
            const int nb = (cam.sh_degree + 1) * (cam.sh_degree + 1);
            const float *sh = g.shs + (size_t)i * g.sh_coeffs * 3;
            float *dsh = gr.dL_dshs + (size_t)i * g.sh_coeffs * 3;
            float ddir[3] = {0.f, 0.f, 0.f};
            for (int ch = 0; ch < 3; ++ch) {
                const float gc = st.clamped[3 * i + ch] ? 0.f : acc[6 + ch];
                for (int k = 0; k < nb; ++k) {
                    dsh[3 * k + ch] = basis[k] * gc;
                    const float w = sh[3 * k + ch] * gc;
                    ddir[0] += db[0][k] * w; ddir[1] += db[1][k] * w; ddir[2] += db[2][k] * w;
                }
                for (int k = nb; k < g.sh_coeffs; ++k) dsh[3 * k + ch] = 0.f;
            }
            // d(unit dir)/d(p) = (I - u u^T) / |d|
            const float dot = u[0] * ddir[0] + u[1] * ddir[1] + u[2] * ddir[2];
            for (int k = 0; k < 3; ++k) dmean[k] += (ddir[k] - u[k] * dot) * inv;
        }
    } else {
        if (gr.dL_dscales) {
            for (int k = 0; k < 3; ++k) gr.dL_dscales[3 * i + k] = 0.f;
            for (int k = 0; k < 4; ++k) gr.dL_drotations[4 * i + k] = 0.f;
        }
        if (g.shs && gr.dL_dshs) {
            float *dsh = gr.dL_dshs + (size_t)i * g.sh_coeffs * 3;
            for (int k = 0; k < g.sh_coeffs * 3; ++k) dsh[k] = 0.f;
        }
    }
    if ((gr.flags & SPLAT_GRADS_POISON_IF_FLAGGED) && st.tile_stride > 0 && (st.status[1] | st.status[3]) != 0) {
        // the forward pass ran on truncated lists and nobody has looked (SplatGrads.flags): nothing plausible leaves this call
        const float nan = __int_as_float(0x7fc00000);
        for (int k = 0; k < 3; ++k) dmean[k] = nan;
        for (int k = 0; k < 6; ++k) dS6[k] = nan;
        g_ndc[0] = g_ndc[1] = nan;
        for (int k = 0; k < SPLAT_GRAD_STRIDE; ++k) acc[k] = nan;
        if (gr.dL_dscales) {
            for (int k = 0; k < 3; ++k) gr.dL_dscales[3 * i + k] = nan;
            for (int k = 0; k < 4; ++k) gr.dL_drotations[4 * i + k] = nan;
        }
    }
    for (int k = 0; k < 3; ++k) gr.dL_dmeans3D[3 * i + k] = dmean[k];
    gr.dL_dmeans2D[3 * i] = g_ndc[0];
    gr.dL_dmeans2D[3 * i + 1] = g_ndc[1];
    gr.dL_dmeans2D[3 * i + 2] = 0.f;
    const bool poisoned = acc[5] != acc[5] && g_ndc[0] != g_ndc[0];     // (see above: a poisoned row is NaN whether or not the Gaussian was seen)
    gr.dL_dopacities[i] = (vis || poisoned) ? acc[5] : 0.f;
    if (gr.dL_dcolors)
        for (int ch = 0; ch < C; ++ch) gr.dL_dcolors[(size_t)i * C + ch] = (vis || poisoned) ? acc[6 + ch] : 0.f;
    if (gr.dL_dcov3D)
        for (int k = 0; k < 6; ++k) gr.dL_dcov3D[6 * i + k] = dS6[k];
}

__global__ __launch_bounds__(kBlock) void mark_visible_kernel(int P, const float *means3D, const float *view, uint8_t *present) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const float z = mat(view, 2, 0) * means3D[3 * i] + mat(view, 2, 1) * means3D[3 * i + 1] +
                    mat(view, 2, 2) * means3D[3 * i + 2] + mat(view, 2, 3);
    present[i] = z > kNearZ;
}

hipError_t launch_tile_count_reset(SplatState &st, int T, hipStream_t s) {
    const int S = st.sub_bins > 1 ? st.sub_bins : 1;
    return hipMemsetAsync(st.tile_count, 0, sizeof(uint32_t) * (size_t)T * S * SPLAT_COUNTER_STRIDE, s);
}

hipError_t launch_tile_scan(SplatState &st, int T, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    const int nb = (T + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(tile_scan_partial_kernel, dim3(nb), dim3(kScanBlock), 0, s, st, T);
    hipLaunchKernelGGL(tile_scan_finish_kernel, dim3(nb), dim3(kScanBlock), 0, s, st, T);
    return hipGetLastError();
}

hipError_t launch_preprocess_forward(const SplatCamera &cam, const SplatGaussians &g, SplatState &st, hipStream_t s) {
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    if (group_binning(st, cam.image_width, cam.image_height)) {
        // group counters and status words start at zero: ONE memset when the caller laid the state out with splat_state_layout
        // (SPLAT_LAYOUT_GROUPS puts the status words right behind the counters), two otherwise
        // (SPLAT_LAYOUT_GROUPS pads the counters to SPLAT_SLAB_ALIGN and puts the status words exactly behind them: the padding is the
        //  counter array's own, nothing foreign lies in the span)
        const size_t cbytes = sizeof(uint32_t) * (size_t)tile_groups(cam.image_width, cam.image_height) * SPLAT_COUNTER_STRIDE;
        const size_t padded = (cbytes + SPLAT_SLAB_ALIGN - 1) / SPLAT_SLAB_ALIGN * SPLAT_SLAB_ALIGN;
        const char *c0 = reinterpret_cast<const char *>(st.group_count), *s0 = reinterpret_cast<const char *>(st.status);
        hipError_t e;
        if (s0 == c0 + padded) {
            e = hipMemsetAsync(st.group_count, 0, padded + 4 * sizeof(int32_t), s);
        } else {
            e = hipMemsetAsync(st.group_count, 0, cbytes, s);
            if (e == hipSuccess) e = hipMemsetAsync(st.status, 0, 4 * sizeof(int32_t), s);
        }
        if (e != hipSuccess) return e;
        if (g.P > 0)
            hipLaunchKernelGGL(preprocess_forward_group_kernel, dim3((g.P + kGroupBlock - 1) / kGroupBlock), dim3(kGroupBlock),
                               sizeof(unsigned) * (size_t)tile_groups(cam.image_width, cam.image_height), s, cam, g, st);
        return hipGetLastError();
    }
    hipError_t e = launch_tile_count_reset(st, T, s);
    if (e != hipSuccess) return e;
    if (g.P > 0) {
        if (dense_exact_lists(st, g.P, T))
            hipLaunchKernelGGL(preprocess_forward_dense_kernel, dim3((g.P + kDenseGaussians - 1) / kDenseGaussians), dim3(kDenseThreads),
                               sizeof(unsigned) * (size_t)T, s, cam, g, st);
        else
            hipLaunchKernelGGL(preprocess_forward_kernel, dim3((g.P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, cam, g, st, g_debug_skip_count);
    }
    return launch_tile_scan(st, T, s);
}

hipError_t launch_preprocess_backward(const SplatCamera &cam, const SplatGaussians &g, const SplatState &st,
                                      SplatGrads &gr, hipStream_t s) {
    if (g.P > 0)
        hipLaunchKernelGGL(preprocess_backward_kernel, dim3((g.P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, cam, g, st, gr);
    return hipGetLastError();
}

// K11: do two rasterizer calls see the same geometry?  One lane per Gaussian compares the bits of opacity, scales (3), rotation (4)
// of call A and call B; any difference raises *differ.  (The drop-in path's geometry cache, rasterizer.py: the two renders of the
// reference's get_loss -- /root/reference/scripts/splatam.py:249,253 -- are given equal-valued but DISTINCT tensors.)
__global__ __launch_bounds__(kBlock) void same_geometry_kernel(int P, const uint32_t *oa, const uint32_t *ob, const uint32_t *sa, const uint32_t *sb,
                                                               const uint4 *ra, const uint4 *rb, int *differ) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const uint4 qa = ra[i], qb = rb[i];
    bool same = oa[i] == ob[i] && qa.x == qb.x && qa.y == qb.y && qa.z == qb.z && qa.w == qb.w;
#pragma unroll
    for (int k = 0; k < 3; ++k) same = same && sa[3 * (size_t)i + k] == sb[3 * (size_t)i + k];
    if (__builtin_amdgcn_ballot_w64(!same) != 0ull && (threadIdx.x & 63) == 0) atomicOr(differ, 1);
}

hipError_t launch_same_geometry(int P, const float *oa, const float *ob, const float *sa, const float *sb, const float *ra, const float *rb,
                                int *differ, hipStream_t s) {
    hipError_t e = hipMemsetAsync(differ, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    if (P > 0)
        hipLaunchKernelGGL(same_geometry_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, P, reinterpret_cast<const uint32_t *>(oa),
                           reinterpret_cast<const uint32_t *>(ob), reinterpret_cast<const uint32_t *>(sa), reinterpret_cast<const uint32_t *>(sb),
                           reinterpret_cast<const uint4 *>(ra), reinterpret_cast<const uint4 *>(rb), differ);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s) {
    if (P > 0) hipLaunchKernelGGL(mark_visible_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, P, means3D, view, present);
    return hipGetLastError();
}

}  // namespace splat
