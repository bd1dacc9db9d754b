// fused.hip -- the fused SplaTAM iteration (include/splat_hip.h, "Fused SplaTAM iteration"): the callers on
// either side of the rasterizer boundary as kernels, so that one optimisation iteration is 5 (tracking) or 7 (mapping)
// launches with no host synchronisation instead of ~150 PyTorch launches + MIOpen convolutions + a GEMM/GEMV pair.
//
//   F1 fused_preprocess_kernel   transform_to_frame + both render-variable dicts + K1 (one lane per Gaussian); files the
//                                instances: per-tile counts (exact lists), bucket slots, or one record per 2x2-tile group
//      K2..K4                    tile scan, scatter, per-tile sort (binning.hip) -- only while the list statistics are unknown or
//                                lists are too long for the composite's own sort
//      K6<6 channels>            r, g, b, z, 1, z^2 over shared geometry (render.hip); builds and sorts its tile's list itself;
//                                tracking: forms the masked L1 loss and its gradient planes in its epilogue
//   F3 track_loss_kernel         tracking with outlier rejection only: masked L1 sums + dL/d(out6) in one pass over the pixels
//   F4 ssim_forward_kernel       mapping: separable 11x11 SSIM statistics -> map sum + three partial-derivative maps,
//                                image L1 sum, masked depth L1 sum and mask count
//   F5 map_loss_backward_kernel  mapping: blur of the partial maps -> dL/d(rgb), depth gradient with the final count
//      K7<6 channels>            (render.hip)
//   F6 fused_backward_kernel     K8+K9 + adjoint of F1's glue + camera-pose partial sums (block reduce, f64 atomics);
//                                single-view mapping step: + the Adam step of the map
//   F7 pose_finish_kernel        pose partial sums -> dL/d(cam_unnorm_rots[..., t]), dL/d(cam_trans[..., t]), loss value,
//                                tracking: + the pose's Adam step and the best-candidate bookkeeping
//   F8 adam_map_kernel / adam_pose_kernel   (the steps as launches of their own: exchanged / batched mapping, hand-wired loops)
//
// Arithmetic lives in fused_math.h / splat_math.h (host-testable); reference lines are cited there and in
// include/splat_hip.h.
#include "splat_device.h"

#include "fused_math.h"

namespace splat {

namespace {

constexpr int kBlock = 256;
constexpr int kFlagSum = SPLAT_ITER_SUMS - 1;     // ws.sums slot that carries a rank's capacity flags through the all-reduce of sharded tracking

struct FusedArgs {
    SplatCamera cam;
    SplatMap map;
    SplatFrameData frame;
    SplatLossConfig cfg;
    SplatIterWorkspace ws;
    float win[11];      // SSIM window (host-built the way create_window builds it)
};

__device__ __forceinline__ int c_num_tiles(int W, int H) { return ((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile); }
__device__ __forceinline__ int c_num_groups(int W, int H) {
    return (((W + kTile - 1) / kTile + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES) * (((H + kTile - 1) / kTile + SPLAT_GROUP_TILES - 1) / SPLAT_GROUP_TILES);
}

__device__ __forceinline__ void load_pose(const SplatMap &m, int time_idx, Pose &P) {
    pose_from_params(m.cam_unnorm_rots + time_idx, m.cam_trans + time_idx, m.num_frames, P);
}

__device__ __forceinline__ void load_gaussian(const SplatMap &m, int i, float *p, float *u, float &logit, float *ls) {
    p[0] = m.means3D[3 * i]; p[1] = m.means3D[3 * i + 1]; p[2] = m.means3D[3 * i + 2];
    const float4 q = reinterpret_cast<const float4 *>(m.unnorm_rotations)[i];
    u[0] = q.x; u[1] = q.y; u[2] = q.z; u[3] = q.w;
    logit = m.logit_opacities[i];
    // (branch-free: a branch on the layout makes the compiler wait for the loads above before it requests the anisotropic scales)
    const bool iso = m.isotropic != 0;
    const float *s = m.log_scales + (iso ? (size_t)i : 3 * (size_t)i);
    ls[0] = s[0]; ls[1] = s[iso ? 0 : 1]; ls[2] = s[iso ? 0 : 2];
}

// ---------------------------------------------------------------------------------------------------------
// F1: per Gaussian: pose transform, activations, projection (K1), tile counts, feature record
// ---------------------------------------------------------------------------------------------------------
// Bucket slots of a workgroup's (Gaussian, tile) instances.  AGG (maps in creation order, SplatState.order_hint): the 256
// consecutive Gaussians of a workgroup lie along one image row and touch a few dozen tiles, ~13 lanes per tile counter; their
// returning global atomics serialise on those few addresses (116 us for 830 k Gaussians in the frame loop).  The workgroup
// counts per tile in an LDS table (open addressing on the tile id) and takes ONE returning global atomic per (workgroup, tile)
// for the whole group; a lane's slot is the group's base + its LDS rank.  Instances that do not find a table slot (or a
// Gaussian's tiles beyond the eighth) fall back to their own global atomic.
constexpr int kAggSlots = 128;
constexpr int kAggPerLane = 8;

// MODE 2, group binning (SplatState.group_count): splat_device.h, file_group_records

// One Gaussian of F1: pose transform, activations, projection; writes its geometry, feature record and (mapping) seen radius.
// Returns its visibility; `o` holds the tile rectangle (clipped to the launch's band of tile rows) and the depth.
// In two halves -- everything it reads, then everything else -- so that a thread that takes several Gaussians
// (fused_preprocess_dense_kernel) has all their loads in flight before the first one's stores.
struct GaussianIn {
    float rgb[3], seen_radius, p[3], u[4], logit, ls[3];
};

__device__ __forceinline__ void preprocess_load(const FusedArgs &a, int i, GaussianIn &g) {
    // (colour and seen radius requested with the parameters, not behind the geometry stores: the kernel lasts one wave lifetime = its
    //  chain of round trips)
    g.rgb[0] = a.map.rgb_colors[3 * i]; g.rgb[1] = a.map.rgb_colors[3 * i + 1]; g.rgb[2] = a.map.rgb_colors[3 * i + 2];
    g.seen_radius = a.ws.max_2D_radius ? a.ws.max_2D_radius[i] : 0.f;
    load_gaussian(a.map, i, g.p, g.u, g.logit, g.ls);
}

__device__ __forceinline__ bool preprocess_finish(const FusedArgs &a, const CamConst &c, const Pose &P, int i, const GaussianIn &g, Projected &o) {
    const SplatState &st = a.ws.st;
    Glue G;
    glue_forward(P, a.frame.w2c + 8, g.p, g.u, g.logit, g.ls, a.map.isotropic != 0, G);
    float S6[6];
    cov3d_from_scale_rot(G.s, c.scale_modifier, G.rq, S6);
    const bool vis = project_gaussian(c, G.Xc, S6, o);
    if (st.tile_row_end > st.tile_row_begin) {      // a band of tile rows is composited: instances outside it are not filed
        o.y0 = max(o.y0, st.tile_row_begin);
        o.y1 = max(o.y0, min(o.y1, st.tile_row_end));
    }
    o.lx0 = o.x0; o.ly0 = o.y0; o.lx1 = o.x1; o.ly1 = o.y1;
    if (vis && st.group_stride > 0) live_tile_rect(o.conic, G.op, o.px, o.py, o.lx0, o.ly0, o.lx1, o.ly1);      // group binning files the live part only
    st.depth[i] = o.depth;
    reinterpret_cast<float2 *>(st.xy)[i] = make_float2(o.px, o.py);
    reinterpret_cast<float4 *>(st.conic_opacity)[i] = make_float4(o.conic[0], o.conic[1], o.conic[2], G.op);
    reinterpret_cast<uint2 *>(st.rect)[i] = make_uint2((unsigned)o.x0 | ((unsigned)o.y0 << 16), (unsigned)o.x1 | ((unsigned)o.y1 << 16));
    st.radii[i] = o.radius;
    float4 *f = reinterpret_cast<float4 *>(a.ws.feat8) + 2 * (size_t)i;
    f[0] = make_float4(g.rgb[0], g.rgb[1], g.rgb[2], G.z);
    f[1] = make_float4(1.0f, G.z * G.z, 0.f, 0.f);
    if (vis && a.ws.max_2D_radius && (float)o.radius > g.seen_radius) a.ws.max_2D_radius[i] = (float)o.radius;
    return vis;
}

__device__ __forceinline__ bool preprocess_one(const FusedArgs &a, const CamConst &c, int i, Projected &o) {
    Pose P;
    load_pose(a.map, a.frame.time_idx, P);
    GaussianIn g;
    preprocess_load(a, i, g);
    return preprocess_finish(a, c, P, i, g, o);
}

template <int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fused_preprocess_kernel(FusedArgs a) {
    constexpr bool AGG = MODE == 1;
    constexpr bool GROUP = MODE == 2;
    __shared__ unsigned s_key[AGG ? kAggSlots : 1], s_cnt[AGG ? kAggSlots : 1], s_base[AGG ? kAggSlots : 1];
    extern __shared__ unsigned s_grp[];         // GROUP: records per group of this workgroup, then the group's base slot
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < a.map.P;
    if constexpr (MODE == 0)
        if (!active) return;
    if constexpr (AGG) {
        if (threadIdx.x < kAggSlots) { s_key[threadIdx.x] = 0xFFFFFFFFu; s_cnt[threadIdx.x] = 0u; }
        __syncthreads();
    }
    const int ggx = (((a.cam.image_width + kTile - 1) / kTile) + kGT - 1) / kGT;
    const int num_groups = ggx * ((((a.cam.image_height + kTile - 1) / kTile) + kGT - 1) / kGT);
    if constexpr (GROUP) group_hist_reset<BLOCK>(s_grp, num_groups);
    if (i == 0 && a.ws.st.tile_stride > 0) {
        // bucketed lists have no scan kernel: the per-iteration status words are reset here and re-accumulated by the
        // kernel that consumes the tile counters (fused_backward_kernel); [1] (overflow) stays sticky for the host
        a.ws.st.status[0] = 0; a.ws.st.status[2] = 0; a.ws.st.status[3] = 0;
    }
    SplatState &st = a.ws.st;
    Projected o{};
    bool vis = false;
    CamConst c;
    load_cam(c, a.cam);
    if (active) vis = preprocess_one(a, c, i, o);
    if (st.tile_stride == 0) {
        // exact path: count now, scan + scatter later
        if (vis && o.y1 > o.y0)
            for (int y = o.y0; y < o.y1; ++y)
                for (int x = o.x0; x < o.x1; ++x) atomicAdd(&st.tile_count[sub_counter(st, y * c.gx + x, i)], 1u);
        return;                                              // (uniform over the launch)
    }
    if constexpr (GROUP) {
        file_group_records<BLOCK>(st, s_grp, i, vis && o.ly1 > o.ly0 && o.lx1 > o.lx0, o.lx0, o.ly0, o.lx1, o.ly1, o.depth, ggx, num_groups);
        return;
    }
    // bucketed path: the returning atomic IS the slot
    const unsigned stride = (unsigned)st.tile_stride;
    const uint64_t key = ((uint64_t)__float_as_uint(o.depth) << 32) | (uint32_t)i;
    const int w = o.x1 - o.x0, nt = (vis && o.y1 > o.y0) ? w * (o.y1 - o.y0) : 0;
    bool spilled = false;
    int t_direct = 0;                                       // tiles [t_direct, nt) take their own global atomic
    if constexpr (AGG) {
        unsigned where[kAggPerLane];                        // slot << 24 | rank; 0xFFFFFFFF: own global atomic
        const int na = min(nt, kAggPerLane);
#pragma unroll
        for (int t = 0; t < kAggPerLane; ++t) {
            where[t] = 0xFFFFFFFFu;
            if (t < na) {
                const int yy = t / w, xx = t - yy * w;
                const unsigned tix = (unsigned)((o.y0 + yy) * c.gx + o.x0 + xx);
                unsigned h = (tix * 2654435761u) >> 25;
                for (int probe = 0; probe < 4; ++probe) {
                    const unsigned old = atomicCAS(&s_key[h], 0xFFFFFFFFu, tix);
                    if (old == 0xFFFFFFFFu || old == tix) {
                        where[t] = (h << 24) | atomicAdd(&s_cnt[h], 1u);
                        break;
                    }
                    h = (h + 1) & (kAggSlots - 1);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < kAggSlots && s_key[threadIdx.x] != 0xFFFFFFFFu)
            s_base[threadIdx.x] = atomicAdd(&st.tile_count[(size_t)s_key[threadIdx.x] * SPLAT_COUNTER_STRIDE], s_cnt[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kAggPerLane; ++t)
            if (t < na) {
                const int yy = t / w, xx = t - yy * w;
                const unsigned tix = (unsigned)((o.y0 + yy) * c.gx + o.x0 + xx);
                unsigned slot;
                if (where[t] != 0xFFFFFFFFu) slot = s_base[where[t] >> 24] + (where[t] & 0xFFFFFFu);
                else slot = atomicAdd(&st.tile_count[(size_t)tix * SPLAT_COUNTER_STRIDE], 1u);
                if (slot < stride) st.keys[(size_t)tix * stride + slot] = key;
                else spilled = true;
            }
        t_direct = na;
    }
    // up to four returning atomics in flight per lane
    for (int t0 = t_direct; t0 < nt; t0 += 4) {
        unsigned slot[4], tix[4];
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int t = t0 + uu;
            if (t < nt) {
                const int yy = t / w, xx = t - yy * w;
                tix[uu] = (unsigned)((o.y0 + yy) * c.gx + o.x0 + xx);
                slot[uu] = atomicAdd(&st.tile_count[(size_t)tix[uu] * SPLAT_COUNTER_STRIDE], 1u);
            }
        }
#pragma unroll
        for (int uu = 0; uu < 4; ++uu)
            if (t0 + uu < nt) {
                if (slot[uu] < stride) st.keys[(size_t)tix[uu] * stride + slot[uu]] = key;
                else spilled = true;
            }
    }
    if (spilled) st.status[1] = 1;
}

// F1 for LONG bucketed lists of a map in ANY order (BASELINE config E: millions of Gaussians on a few hundred tiles).  One returning
// global atomic per instance serialises on the tiles' counters (one L2 atomic unit per address, ~12 ns each: 1 M Gaussians on 400
// tiles -> 201 us, 5 M -> 900 us).  Here a 1024-thread workgroup takes kDensePerThread Gaussians per thread, counts their instances per
// TILE in an LDS table of the whole frame (4 bytes x tiles, dynamic LDS), reserves each non-empty tile's slots with ONE global atomic,
// and hands them out through the same table: 4096 Gaussians per global atomic round instead of one.
constexpr int kDenseBlock = 1024;
constexpr int kDensePerThread = 4;
constexpr int kDenseMaxTiles = 12 * 1024;          // 48 KB of LDS

__global__ __launch_bounds__(kDenseBlock) void fused_preprocess_dense_kernel(FusedArgs a) {
    extern __shared__ unsigned s_tile[];            // instances per tile of this workgroup, then the tile's next free slot
    const int tid = threadIdx.x;
    CamConst c;
    load_cam(c, a.cam);
    const int T = c.gx * c.gy;
    for (int t = tid; t < T; t += kDenseBlock) s_tile[t] = 0u;
    if (blockIdx.x == 0 && tid == 0) { a.ws.st.status[0] = 0; a.ws.st.status[2] = 0; a.ws.st.status[3] = 0; }     // (see fused_preprocess_kernel)
    __syncthreads();
    const SplatState &st = a.ws.st;
    unsigned r0[kDensePerThread], r1[kDensePerThread], dbits[kDensePerThread];
    Pose P;
    load_pose(a.map, a.frame.time_idx, P);
    GaussianIn in[kDensePerThread];
#pragma unroll
    for (int g = 0; g < kDensePerThread; ++g) {         // every Gaussian's loads in flight before the first one's stores
        const int i = (blockIdx.x * kDensePerThread + g) * kDenseBlock + tid;
        preprocess_load(a, min(i, a.map.P - 1), in[g]);
    }
#pragma unroll
    for (int g = 0; g < kDensePerThread; ++g) {
        const int i = (blockIdx.x * kDensePerThread + g) * kDenseBlock + tid;
        r0[g] = r1[g] = dbits[g] = 0u;
        if (i < a.map.P) {
            Projected o{};
            const bool vis = preprocess_finish(a, c, P, i, in[g], o);
            if (vis && o.y1 > o.y0 && o.x1 > o.x0) {
                r0[g] = (unsigned)o.x0 | ((unsigned)o.y0 << 16);
                r1[g] = (unsigned)o.x1 | ((unsigned)o.y1 << 16);
                dbits[g] = __float_as_uint(o.depth);
                for (int y = o.y0; y < o.y1; ++y)
                    for (int x = o.x0; x < o.x1; ++x) atomicAdd(&s_tile[y * c.gx + x], 1u);
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += kDenseBlock) {
        const unsigned cnt = s_tile[t];
        if (cnt) s_tile[t] = atomicAdd(&st.tile_count[(size_t)t * SPLAT_COUNTER_STRIDE], cnt);
    }
    __syncthreads();
    const unsigned stride = (unsigned)st.tile_stride;
    bool spilled = false;
#pragma unroll
    for (int g = 0; g < kDensePerThread; ++g) {
        const int i = (blockIdx.x * kDensePerThread + g) * kDenseBlock + tid;
        const int x0 = r0[g] & 0xFFFF, y0 = r0[g] >> 16, x1 = r1[g] & 0xFFFF, y1 = r1[g] >> 16;
        const uint64_t key = ((uint64_t)dbits[g] << 32) | (uint32_t)i;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const unsigned tix = (unsigned)(y * c.gx + x);
                const unsigned slot = atomicAdd(&s_tile[tix], 1u);
                if (slot < stride) st.keys[(size_t)tix * stride + slot] = key;
                else spilled = true;
            }
    }
    if (spilled) st.status[1] = 1;
}

// ---------------------------------------------------------------------------------------------------------
// block-level sums -> double atomics
// ---------------------------------------------------------------------------------------------------------
// ws.sums holds SPLAT_ITER_SUM_COPIES copies of the SPLAT_ITER_SUMS partial sums (one 256-byte pair of lines each):
// a workgroup adds to copy (its linear id % copies), so that the few thousand workgroups of a launch do not queue
// their atomics on ONE line (measured: ~12 ns per same-line atomic, i.e. 60 us for 4 900 workgroups).
__device__ __forceinline__ double *sum_copy(double *sums) {
    const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    return sums + (size_t)(b % SPLAT_ITER_SUM_COPIES) * SPLAT_ITER_SUMS;
}

template <int N>
__device__ __forceinline__ void block_sum_to(double *dst, const float (&v)[N], double *s_part /* [N][waves] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float x = v[k];
        for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
        if (lane == 0) s_part[k * nw + wave] = (double)x;
    }
    __syncthreads();
    if ((int)threadIdx.x < N) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += s_part[threadIdx.x * nw + w];
        if (t != 0.0) atomicAdd(dst + threadIdx.x, t);
    }
    __syncthreads();
}

// total of one partial sum over the copies: call with a full wave; every lane returns the total
__device__ __forceinline__ double sum_total(const double *sums, int k) {
    static_assert(SPLAT_ITER_SUM_COPIES == 64, "one copy per lane");
    double v = sums[(size_t)(threadIdx.x & 63) * SPLAT_ITER_SUMS + k];
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

struct Pixel {
    bool mask;          // the depth (and, for tracking with the silhouette, colour) loss mask
    float d_err;        // |gt_depth - depth| (finite when mask)
    float d_sign;       // d|gt - d| / dd = sign(d - gt)
};

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// median: of |gt - d| * (gt > 0) over the frame, only read with ignore_outlier_depth_loss
// (/root/reference/scripts/splatam.py:264-272: mask = (depth_error < 10 * median) & (gt > 0) & nan_mask [& silhouette])
__device__ __forceinline__ Pixel depth_pixel(const SplatLossConfig &cfg, float depth, float sil, float depth_sq, float gt, float median) {
    Pixel r;
    const float unc = depth_sq - depth * depth;
    const bool nan_ok = !(depth != depth) && !(unc != unc);
    bool m = gt > 0.f && nan_ok;
    if (cfg.ignore_outlier_depth_loss) m = m && (fabsf(gt - depth) < 10.f * median);
    if (cfg.tracking && cfg.use_sil_for_loss) m = m && (sil > cfg.sil_thres);
    r.mask = m;
    const float diff = gt - depth;
    r.d_err = m ? fabsf(diff) : 0.f;
    r.d_sign = m ? -sgn(diff) : 0.f;
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// F3: tracking loss (/root/reference/scripts/splatam.py:256-286 with tracking=True): one pass
// ---------------------------------------------------------------------------------------------------------
// Planes 4 (silhouette) and 5 (depth^2) never receive a gradient (the uncertainty is detached, the silhouette only
// masks): their dL_dout6 planes are zeroed once by the caller and never written.
template <int V>
__global__ __launch_bounds__(kBlock) void track_loss_kernel(FusedArgs a, int HW) {
    __shared__ double s_part[2 * (kBlock / 64)];
    const float *o = a.ws.out6;
    float *g = a.ws.dL_dout6;
    const bool masked_im = a.cfg.use_sil_for_loss || a.cfg.ignore_outlier_depth_loss;
    const float median = a.cfg.ignore_outlier_depth_loss ? a.ws.d_cam[13] : 0.f;
    float acc[2] = {0.f, 0.f};
    const int nvec = HW / V;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nvec; i += gridDim.x * kBlock) {
        float in[10][V], out[4][V];
        auto ld = [&](const float *p, float *dst) {
            if constexpr (V == 4) {
                const float4 t = reinterpret_cast<const float4 *>(p)[i];
                dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
            } else {
                dst[0] = p[i];
            }
        };
#pragma unroll
        for (int ch = 0; ch < 6; ++ch) ld(o + ch * (size_t)HW, in[ch]);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) ld(a.frame.im + ch * (size_t)HW, in[6 + ch]);
        ld(a.frame.depth, in[9]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const Pixel px = depth_pixel(a.cfg, in[3][v], in[4][v], in[5][v], in[9][v], median);
            acc[0] += px.d_err;
            out[3][v] = a.cfg.use_l1 ? a.cfg.w_depth * px.d_sign : 0.f;
            const bool cm = masked_im ? px.mask : true;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float diff = in[6 + ch][v] - in[ch][v];
                acc[1] += cm ? fabsf(diff) : 0.f;
                out[ch][v] = cm ? -a.cfg.w_im * sgn(diff) : 0.f;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if constexpr (V == 4) reinterpret_cast<float4 *>(g + ch * (size_t)HW)[i] = make_float4(out[ch][0], out[ch][1], out[ch][2], out[ch][3]);
            else g[ch * (size_t)HW + i] = out[ch][0];
        }
    }
    block_sum_to<2>(sum_copy(a.ws.sums), acc, s_part);
}

// ---------------------------------------------------------------------------------------------------------
// F4 / F5: SSIM (11x11, sigma 1.5, zero padding; /root/reference/utils/slam_external.py:54-97)
// ---------------------------------------------------------------------------------------------------------
constexpr int kSsimR = 5;                   // window radius
typedef float f2 __attribute__((ext_vector_type(2)));

// blockIdx -> (tile column, tile row, channel) of F4 / F5.  Workgroups are dealt to the 8 XCDs round robin and each XCD has its own
// L2: with a plain 3-d grid the eight neighbours of a tile run on eight OTHER XCDs, and every halo pixel (the window is 1.9x the tile) comes from
// memory again (F5: 167 MB of traffic for 75 MB of planes).  Here XCD x owns a contiguous run of the (channel, row, column) order, so a
// tile's halo was read by the workgroup before it or one tile row earlier, through the SAME L2.
template <int TW, int TH>
__device__ __forceinline__ bool ssim_tile(int W, int H, int &bx, int &by, int &ch) {
    const int ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH, total = 3 * ntx * nty, per = (total + 7) / 8;
    const int b = blockIdx.x, slot = b >> 3, t = (b & 7) * per + slot;
    if (slot >= per || t >= total) return false;
    ch = t / (ntx * nty);
    const int r = t - ch * ntx * nty;
    by = r / ntx;
    bx = r - by * ntx;
    return true;
}

// exp(-(x-5)^2 / (2 * 1.5^2)) normalised in float32, as create_window builds it (/root/reference/utils/slam_external.py:54-56)
void ssim_window_host(float *g) {
    float s = 0.f;
    for (int k = 0; k < 11; ++k) { g[k] = (float)exp(-(double)((k - 5) * (k - 5)) / 4.5); s += g[k]; }
    for (int k = 0; k < 11; ++k) g[k] /= s;
}

// ssim_pixel (fused_math.h) with hardware reciprocals (1 ulp) instead of four IEEE divisions
__device__ __forceinline__ float ssim_pixel_dev(float mu1, float mu2, float e11, float e22, float e12, float *dmu1, float *de11, float *de12) {
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A = 2.f * mu12 + kSsimC1, B = 2.f * s12 + kSsimC2;
    const float Cd = mu1_sq + mu2_sq + kSsimC1, Dd = s1 + s2 + kSsimC2;
    const float iC = __builtin_amdgcn_rcpf(Cd), iD = __builtin_amdgcn_rcpf(Dd);
    const float iCD = iC * iD;
    const float map = A * B * iCD;
    *dmu1 = (2.f * mu2) * (B - A) * iCD + map * (2.f * mu1) * (iD - iC);
    *de11 = -map * iD;
    *de12 = 2.f * A * iCD;
    return map;
}

// F4 / F5 are separable 11-tap passes, VERTICAL pass first and without LDS: a thread owns one column of the window (lanes =
// consecutive columns: the loads coalesce), holds kCR + 10 input rows in registers and forms kCR output rows of vertical sums; only
// those go through LDS (24 rows x 42 columns, no halo rows), and the horizontal pass reads 14 columns for 4 output pixels, which it
// finishes and stores as one float4 per plane: 10.5 LDS reads and 76 multiply-adds per pixel, 21 KB (F4) / 13 KB (F5) of LDS, one
// barrier.  (Rounds 2-4 ran the passes the other way round -- window staged in LDS, horizontal pass over 26 halo rows for 16 output
// rows, 12 rows of five sums read back per two output pixels: 29 LDS reads and 87 multiply-adds per pixel, 26 KB, two barriers:
// F4 31.2 -> 25.0 us, F5 22.2 -> 21.0 us at 1200x680, profiles/r04_experiments.md 10.)  The five window statistics travel as two
// float pairs + one float, so that the 11-tap sums are v_pk_fma_f32 (two statistics per instruction).
constexpr int kCW = 32, kCR = 4, kCG = 6;   // tile width; output rows per thread of the vertical pass; row groups per workgroup
constexpr int kCH = kCR * kCG;              // tile height 24
constexpr int kCCols = kCW + 2 * kSsimR;    // 42 columns with halo
constexpr int kCStride = 45;                // LDS row stride in elements: 1 mod 4, the 64-bit reads of the horizontal pass fall on distinct banks
constexpr int kCItems = kCH * (kCW / 4);    // horizontal work items: (row, group of 4 columns)
static_assert(kCCols * kCG <= kBlock && kCItems <= kBlock, "one trip per pass");

// four consecutive pixels of a plane row (clamped addresses; the caller discards what lies outside the image)
template <bool VEC>
__device__ __forceinline__ void load_px4(const float *plane, int yy, int xx, int W, int H, float *dst) {
    const size_t row = (size_t)min(yy, H - 1) * W;
    if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4 *>(plane + row + min(xx, W - 4));
        dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = plane[row + min(xx + j, W - 1)];
    }
}

template <bool VEC>
__device__ __forceinline__ void store_px4(float *plane, int yy, int xx, int W, int H, const float *v) {
    if (yy >= H) return;
    float *p = plane + (size_t)yy * W + xx;
    if constexpr (VEC) {
        if (xx < W) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (xx + j < W) p[j] = v[j];
    }
}

// plane[byte_off / 4] with a uniform base and a 32-bit byte offset: one address register per load instead of two
__device__ __forceinline__ float ld_off(const float *plane, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(plane) + byte_off);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// VEC: W % 4 == 0 and every plane 16-byte aligned (the launcher checks): whole float4 loads / stores of a thread's four pixels
template <bool VEC>
__global__ __launch_bounds__(kBlock) void ssim_forward_kernel(FusedArgs a, int W, int H) {
    __shared__ f2 svA[kCH][kCStride], svB[kCH][kCStride];      // vertical sums of (x, y) and (x x, y y)
    __shared__ float svC[kCH][kCStride];                       // ... of x y
    __shared__ double s_part[4 * (kBlock / 64)];
    float g[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) g[k] = a.win[k];
    int bx, by, ch;
    if (!ssim_tile<kCW, kCH>(W, H, bx, by, ch)) return;
    const int tid = threadIdx.x;
    const int x0 = bx * kCW, y0 = by * kCH;
    const size_t HW = (size_t)H * W;
    const float *X = a.ws.out6 + ch * HW, *Y = a.frame.im + ch * HW;
    float *M = a.ws.ssim_maps + (size_t)(3 * ch) * HW;
    // the horizontal pass' work item and its own pixels (image L1 term; z == 0 slice: the depth-loss inputs)
    const int hr = tid / (kCW / 4), hc = (tid & (kCW / 4 - 1)) * 4;
    float own_x[4], own_y[4];
    {   // vertical pass: thread = (column of the window, group of kCR output rows); every load in flight before the first sum
        const int grp = tid / kCCols, col = tid - grp * kCCols;
        if (grp < kCG) {
            const int xx = x0 - kSsimR + col;
            const bool cin = xx >= 0 && xx < W;
            const int xc = min(max(xx, 0), W - 1);
            float vx[kCR + 10], vy[kCR + 10];
#pragma unroll
            for (int t = 0; t < kCR + 10; ++t) {
                const int yy = y0 + grp * kCR - kSsimR + t;
                const bool in = cin && yy >= 0 && yy < H;
                // (any address inside the plane will do for a row outside the image: its value is discarded)
                const unsigned off = (unsigned)min(max(yy * (4 * W) + 4 * xc, 0), 4 * (H * W - 1));
                const float tx = ld_off(X, off), ty = ld_off(Y, off);
                vx[t] = in ? tx : 0.f;
                vy[t] = in ? ty : 0.f;
            }
            f2 vA[kCR], vB[kCR];
            float vC[kCR];
#pragma unroll
            for (int j = 0; j < kCR; ++j) { vA[j] = (f2)(0.f); vB[j] = (f2)(0.f); vC[j] = 0.f; }
#pragma unroll
            for (int t = 0; t < kCR + 10; ++t) {
                const f2 p = {vx[t], vy[t]};
                const f2 q = p * p;
                const float xy = p.x * p.y;
#pragma unroll
                for (int j = 0; j < kCR; ++j) {
                    const int tap = t - j;
                    if (tap >= 0 && tap < 11) {
                        const f2 w = (f2)(g[tap]);
                        vA[j] = __builtin_elementwise_fma(w, p, vA[j]);
                        vB[j] = __builtin_elementwise_fma(w, q, vB[j]);
                        vC[j] = fmaf(g[tap], xy, vC[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kCR; ++j) { svA[grp * kCR + j][col] = vA[j]; svB[grp * kCR + j][col] = vB[j]; svC[grp * kCR + j][col] = vC[j]; }
        }
    }
    // (requested here, when the window's registers are free again; they are used at the very end of the horizontal pass)
    if (tid < kCItems) {
        load_px4<VEC>(X, y0 + hr, x0 + hc, W, H, own_x);
        load_px4<VEC>(Y, y0 + hr, x0 + hc, W, H, own_y);
    }
    __syncthreads();
    const float median = a.cfg.ignore_outlier_depth_loss ? a.ws.d_cam[13] : 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};        // depth L1 (masked), image L1, mask count, SSIM map sum
    if (tid < kCItems) {                        // horizontal pass: 14 columns of sums feed 4 output pixels
        f2 oA[4], oB[4];
        float oC[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { oA[j] = (f2)(0.f); oB[j] = (f2)(0.f); oC[j] = 0.f; }
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const f2 pa = svA[hr][hc + t], pb = svB[hr][hc + t];
            const float pc = svC[hr][hc + t];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tap = t - j;
                if (tap >= 0 && tap < 11) {
                    const f2 w = (f2)(g[tap]);
                    oA[j] = __builtin_elementwise_fma(w, pa, oA[j]);
                    oB[j] = __builtin_elementwise_fma(w, pb, oB[j]);
                    oC[j] = fmaf(g[tap], pc, oC[j]);
                }
            }
        }
        const int yy = y0 + hr;
        float d0[4], d1[4], d2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float map = ssim_pixel_dev(oA[j].x, oA[j].y, oB[j].x, oB[j].y, oC[j], &d0[j], &d1[j], &d2[j]);
            // (selects, not branches: the compiler sinks a pixel's sums into its conditional block and keeps all 70 inputs alive)
            const bool ok = yy < H && x0 + hc + j < W;
            acc[3] += ok ? map : 0.f;
            acc[1] += ok ? fabsf(own_x[j] - own_y[j]) : 0.f;
        }
        store_px4<VEC>(M, yy, x0 + hc, W, H, d0);
        store_px4<VEC>(M + HW, yy, x0 + hc, W, H, d1);
        store_px4<VEC>(M + 2 * HW, yy, x0 + hc, W, H, d2);
    }
    if (ch == 0 && tid < kCItems) {
        // the depth loss of the z == 0 slice (a third of the workgroups), behind everything else: its sixteen input registers held
        // through the passes would cost the kernel a wave per SIMD; the planes come through L2 (the forward composite wrote them)
        const int yy = y0 + hr;
        float pre[4][4];
        load_px4<VEC>(a.ws.out6 + 3 * HW, yy, x0 + hc, W, H, pre[0]);
        load_px4<VEC>(a.ws.out6 + 4 * HW, yy, x0 + hc, W, H, pre[1]);
        load_px4<VEC>(a.ws.out6 + 5 * HW, yy, x0 + hc, W, H, pre[2]);
        load_px4<VEC>(a.frame.depth, yy, x0 + hc, W, H, pre[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = yy < H && x0 + hc + j < W;
            const Pixel px = depth_pixel(a.cfg, pre[0][j], pre[1][j], pre[2][j], pre[3][j], median);
            acc[0] += ok ? px.d_err : 0.f;
            acc[2] += ok && px.mask ? 1.f : 0.f;
        }
    }
    block_sum_to<4>(sum_copy(a.ws.sums), acc, s_part);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void map_loss_backward_kernel(FusedArgs a, int W, int H) {
    __shared__ f2 svA[kCH][kCStride];           // vertical sums of (d/dmu1, d/dE11)
    __shared__ float svC[kCH][kCStride];        // ... of d/dE12
    __shared__ float s_count;
    float g[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) g[k] = a.win[k];
    int bx, by, ch;
    if (!ssim_tile<kCW, kCH>(W, H, bx, by, ch)) return;
    const int tid = threadIdx.x;
    const int x0 = bx * kCW, y0 = by * kCH;
    const size_t HW = (size_t)H * W;
    const float *M = a.ws.ssim_maps + (size_t)(3 * ch) * HW;
    if (tid < 64) {
        const double c = sum_total(a.ws.sums, 2);
        if (tid == 0) s_count = (float)c;
    }
    const int hr = tid / (kCW / 4), hc = (tid & (kCW / 4 - 1)) * 4;
    float own_x[4], own_y[4];
    {   // vertical pass (see ssim_forward_kernel)
        const int grp = tid / kCCols, col = tid - grp * kCCols;
        if (grp < kCG) {
            const int xx = x0 - kSsimR + col;
            const bool cin = xx >= 0 && xx < W;
            const int xc = min(max(xx, 0), W - 1);
            float v0[kCR + 10], v1[kCR + 10], v2[kCR + 10];
#pragma unroll
            for (int t = 0; t < kCR + 10; ++t) {
                const int yy = y0 + grp * kCR - kSsimR + t;
                const bool in = cin && yy >= 0 && yy < H;
                const unsigned off = (unsigned)min(max(yy * (4 * W) + 4 * xc, 0), 4 * (H * W - 1));
                const float t0 = ld_off(M, off), t1 = ld_off(M + HW, off), t2 = ld_off(M + 2 * HW, off);
                v0[t] = in ? t0 : 0.f;
                v1[t] = in ? t1 : 0.f;
                v2[t] = in ? t2 : 0.f;
            }
            f2 vA[kCR];
            float vC[kCR];
#pragma unroll
            for (int j = 0; j < kCR; ++j) { vA[j] = (f2)(0.f); vC[j] = 0.f; }
#pragma unroll
            for (int t = 0; t < kCR + 10; ++t) {
                const f2 p = {v0[t], v1[t]};
#pragma unroll
                for (int j = 0; j < kCR; ++j) {
                    const int tap = t - j;
                    if (tap >= 0 && tap < 11) {
                        vA[j] = __builtin_elementwise_fma((f2)(g[tap]), p, vA[j]);
                        vC[j] = fmaf(g[tap], v2[t], vC[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kCR; ++j) { svA[grp * kCR + j][col] = vA[j]; svC[grp * kCR + j][col] = vC[j]; }
        }
    }
    if (tid < kCItems) {
        load_px4<VEC>(a.ws.out6 + ch * HW, y0 + hr, x0 + hc, W, H, own_x);
        load_px4<VEC>(a.frame.im + ch * HW, y0 + hr, x0 + hc, W, H, own_y);
    }
    __syncthreads();
    if (tid >= kCItems) return;
    const float inv_n = 1.0f / (3.0f * (float)HW);
    const float median = a.cfg.ignore_outlier_depth_loss ? a.ws.d_cam[13] : 0.f;
    const float count = s_count;
    f2 oA[4];
    float oC[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { oA[j] = (f2)(0.f); oC[j] = 0.f; }
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        const f2 pa = svA[hr][hc + t];
        const float pc = svC[hr][hc + t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tap = t - j;
            if (tap >= 0 && tap < 11) {
                oA[j] = __builtin_elementwise_fma((f2)(g[tap]), pa, oA[j]);
                oC[j] = fmaf(g[tap], pc, oC[j]);
            }
        }
    }
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xv = own_x[j], yv = own_y[j];
        const float dssim = oA[j].x + 2.f * xv * oA[j].y + yv * oC[j];
        out[j] = a.cfg.w_im * (0.8f * sgn(xv - yv) * inv_n - 0.2f * inv_n * dssim);
    }
    store_px4<VEC>(a.ws.dL_dout6 + ch * HW, y0 + hr, x0 + hc, W, H, out);
    if (ch == 0) {      // the depth plane's gradient (see the end of ssim_forward_kernel)
        float pre[4][4], dout[4];
        load_px4<VEC>(a.ws.out6 + 3 * HW, y0 + hr, x0 + hc, W, H, pre[0]);
        load_px4<VEC>(a.ws.out6 + 4 * HW, y0 + hr, x0 + hc, W, H, pre[1]);
        load_px4<VEC>(a.ws.out6 + 5 * HW, y0 + hr, x0 + hc, W, H, pre[2]);
        load_px4<VEC>(a.frame.depth, y0 + hr, x0 + hc, W, H, pre[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Pixel px = depth_pixel(a.cfg, pre[0][j], pre[1][j], pre[2][j], pre[3][j], median);
            dout[j] = a.cfg.use_l1 ? a.cfg.w_depth * px.d_sign / count : 0.f;
        }
        store_px4<VEC>(a.ws.dL_dout6 + 3 * HW, y0 + hr, x0 + hc, W, H, dout);
    }
}

// Adam step of the camera pose of one frame + the reference's best-candidate bookkeeping
// (/root/reference/scripts/splatam.py:704-711); state: m_q[0..3] m_t[4..6] v_q[7..10] v_t[11..13] min_loss[14] cand_q[15..18] cand_t[19..21]
struct PoseAdam {
    float *state;               // nullptr: no step (the separate splat_iter_adam_pose call takes it)
    float beta1, beta2, eps, bc2_sqrt, ss_rot, ss_trans;
};

// The pose's Adam step + the reference's best-candidate bookkeeping, by ONE thread: every load first (pose_adam_load: 22 values in
// flight; F7 issues them at the top of the kernel, ahead of its first store), then the arithmetic, then the stores -- parameter by
// parameter it was a chain of fourteen dependent round trips.
struct PoseAdamRegs {
    float *pp[7];
    float par[7], m[7], v[7], best;
};

__device__ __forceinline__ void pose_adam_load(const SplatMap &map, int time_idx, const PoseAdam &pa, PoseAdamRegs &r) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        r.pp[k] = k < 4 ? map.cam_unnorm_rots + k * map.num_frames + time_idx : map.cam_trans + (k - 4) * map.num_frames + time_idx;
        r.par[k] = *r.pp[k];
        r.m[k] = pa.state[k];
        r.v[k] = pa.state[7 + k];
    }
    r.best = pa.state[14];
}

__device__ __forceinline__ void pose_adam_apply(PoseAdamRegs &r, const float *g, float loss, const PoseAdam &pa) {
    float *state = pa.state;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        r.par[k] = adam_update(r.par[k], g[k], r.m[k], r.v[k], pa.beta1, pa.beta2, k < 4 ? pa.ss_rot : pa.ss_trans, pa.bc2_sqrt, pa.eps);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        *r.pp[k] = r.par[k];
        state[k] = r.m[k];
        state[7 + k] = r.v[k];
    }
    // the reference compares the loss of THIS iteration and stores the parameters AFTER the step
    if (loss < r.best) {
        state[14] = loss;
#pragma unroll
        for (int k = 0; k < 7; ++k) state[15 + k] = r.par[k];
    }
}

__device__ __forceinline__ void pose_adam_step(const SplatMap &map, int time_idx, const float *g, float loss, const PoseAdam &pa) {
    PoseAdamRegs r;
    pose_adam_load(map, time_idx, pa, r);
    pose_adam_apply(r, g, loss, pa);
}

// F7: one thread: pose partial sums -> gradients of the raw camera parameters; loss value (+ the pose's Adam step when the
// caller asked for the whole tracking step in one call: one launch less per iteration)
// Workgroups 1..8 (launched when the state carries SplatState.tile_work / tile_order) have nothing to do with the pose: each turns the
// forward composite's per-tile work estimates of ITS XCD band into the band's launch order for the next iteration's composites -- a
// 5 us kernel of its own on the iteration's critical path otherwise, here it runs beside the single workgroup that closes the iteration.
__global__ __launch_bounds__(256) void pose_finish_kernel(FusedArgs a, int HW, PoseAdam pa) {
    static_assert(SPLAT_ITER_SUMS * 8 == 256 && SPLAT_ITER_SUM_COPIES == 64, "thread t: sum k = t / 8, copies (t % 8) * 8 .. + 7");
    if (blockIdx.x > 0) {
        __shared__ unsigned s_order[513];
        const int T = c_num_tiles(a.cam.image_width, a.cam.image_height);
        tile_order_band(a.ws.st.tile_work, a.ws.st.tile_order, T, (T + 7) / 8, (int)blockIdx.x - 1, s_order);
        return;
    }
    __shared__ double S[SPLAT_ITER_SUMS];
    const int t = threadIdx.x, k = t >> 3, part = t & 7;
    // what the closing thread needs -- the pose, the Adam state, the capacity flags -- is requested here, ahead of the first store
    Pose P{};
    PoseAdamRegs adam{};
    int flagged = 0, stat[4] = {0, 0, 0, 0}, skipped = 0;
    float sticky = 0.f;
    if (t == 0) {
        if (a.cfg.camera_grad) load_pose(a.map, a.frame.time_idx, P);
        if (pa.state) pose_adam_load(a.map, a.frame.time_idx, pa, adam);
#pragma unroll
        for (int k = 0; k < 4; ++k) stat[k] = a.ws.st.status[k];
        flagged = stat[1] | stat[3];
        sticky = a.ws.d_cam[12];
        skipped = reinterpret_cast<const int *>(a.ws.d_cam)[21];
    }
    // (all eight loads first, then the resets: a store between two loads makes the second wait for the first)
    double part_sum[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) part_sum[c] = a.ws.sums[(size_t)(part * 8 + c) * SPLAT_ITER_SUMS + k];
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v += part_sum[c];
        a.ws.sums[(size_t)(part * 8 + c) * SPLAT_ITER_SUMS + k] = 0.0;      // reset for the next iteration (no memset launch)
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (part == 0) S[k] = v;
    __syncthreads();
    if (t != 0) return;
    // tile-row-sharded tracking: a rank whose band overflowed says so through the all-reduced sums (kFlagSum), so that every rank
    // takes the same decision about the step below
    if (S[kFlagSum] != 0.0) flagged = 1;
    float *out = a.ws.d_cam;
    float dq[4] = {0.f, 0.f, 0.f, 0.f}, dt[3] = {0.f, 0.f, 0.f};
    if (a.cfg.camera_grad) {
        float sums[kPoseSums];
        for (int k = 0; k < kPoseSums; ++k) sums[k] = (float)S[8 + k];
        pose_backward(P, sums, dq, dt);
    }
    for (int k = 0; k < 4; ++k) out[k] = dq[k];
    for (int k = 0; k < 3; ++k) out[4 + k] = dt[k];
    float loss, w_depth_term, w_im_term;
    const float l_depth = a.cfg.use_l1 ? (float)S[0] : 0.f;
    if (a.cfg.tracking) {
        w_depth_term = a.cfg.w_depth * l_depth;
        w_im_term = a.cfg.w_im * (float)S[1];
    } else {
        const float n = 3.0f * (float)HW;
        const float l_im = 0.8f * ((float)S[1] / n) + 0.2f * (1.0f - (float)S[3] / n);
        w_depth_term = a.cfg.w_depth * (a.cfg.use_l1 ? l_depth / (float)S[2] : 0.f);
        w_im_term = a.cfg.w_im * l_im;
    }
    loss = w_depth_term + w_im_term;
    out[7] = loss;
    for (int k = 0; k < 4; ++k) out[8 + k] = (float)S[k];       // raw sums, for inspection
    const bool gate_up = flagged != 0 || sticky != 0.f;
    if (flagged != 0) out[12] = 1.0f;     // sticky until the host clears it
    out[14] = w_depth_term;
    out[15] = w_im_term;
    int *outi = reinterpret_cast<int *>(out);
    for (int k = 0; k < 4; ++k) outi[16 + k] = stat[k];
    outi[20] = flagged != 0 ? 1 : 0;
    if (gate_up) outi[21] = skipped + 1;  // an iteration on truncated / unsorted lists: no Adam step moves anything (see SplatIterWorkspace.d_cam)
    if (pa.state && !gate_up) {
        float g[7];
        for (int k = 0; k < 4; ++k) g[k] = dq[k];
        for (int k = 0; k < 3; ++k) g[4 + k] = dt[k];
        pose_adam_apply(adam, g, loss, pa);
    }
}

// ---------------------------------------------------------------------------------------------------------
// F6: per Gaussian adjoint: K8+K9, glue adjoint, pose partial sums
// ---------------------------------------------------------------------------------------------------------
// ADAM (single-view mapping step, splat_iter_mapping_step): the Adam step of the five Gaussian groups on the gradients this
// thread has just formed -- parameters, gradients and moments of a Gaussian are touched once instead of in a second kernel
// (+1.3 % mapping iterations/s at B).  Folding F7 into the last workgroup to finish (a ticket) was measured too: no gain, the
// tail is as long as the separate launch.
// MAPGRADS = false (tracking: no per-Gaussian gradient is stored): only the camera partial sums are wanted, and with ISO (Sigma =
// s^2 I in any camera frame) they depend on dL/dX_c alone -- the covariance adjoints fall away at compile time.
template <bool ADAM, bool MAPGRADS, bool ISO>
__global__ __launch_bounds__(kBlock, ISO ? 5 : 4) void fused_backward_kernel(FusedArgs a, SplatAdamMap opt) {
    static_assert(MAPGRADS || !ADAM, "the Adam step consumes the map gradients");
    __shared__ double s_part[kPoseSums * (kBlock / 64)];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float pose[kPoseSums];
#pragma unroll
    for (int k = 0; k < kPoseSums; ++k) pose[k] = 0.f;
    const SplatIterWorkspace &ws = a.ws;
    constexpr bool iso = ISO;
    if (i < a.map.P) {
        // Everything this lane reads is requested HERE, before the first use: the kernel is one wave-lifetime long (4 688 waves at
        // workload B: 4.6 per SIMD), i.e. its duration is the length of its chain of dependent memory round trips.  Round 2 read the
        // accumulator row only after the radius, the parameters after that, and in the Adam step each of the 12 components' moments,
        // then its parameter, behind the previous component's stores (22 round trips; 28.5 us).  (A Gaussian outside the view reads
        // its zero accumulator row and its parameters for nothing: 92 B.)
        // (camera, pose: ahead of the first store of the kernel, where the compiler can still prove them unclobbered: scalar loads)
        CamConst c;
        load_cam(c, a.cam);
        Pose P;
        load_pose(a.map, a.frame.time_idx, P);
        const float w2c_row2[4] = {a.frame.w2c[8], a.frame.w2c[9], a.frame.w2c[10], a.frame.w2c[11]};
        // (MAPGRADS = false, camera tracking: the backward composite's tracking form publishes S1..S5 and the depth channel's colour sum
        //  in slots 0..5 -- render.hip kTrackSlots --: half the accumulator line is read and cleared)
        //  (MAPGRADS = true: the fused iteration's backward composites carry gradient in r, g, b, z only -- launch_render_backward_feat8,
        //  the full-gradient tracking composite -- i.e. S1..S6 and four colour sums in slots 0..9: three quarters of the line; slots 10..15
        //  are never written on this path and stay at the zero they were allocated with)
        constexpr int kRow4 = MAPGRADS ? 3 : 2;
        static_assert(SPLAT_GRAD_STRIDE >= 12, "slots 0..9 live in the first three float4 of the accumulator row");
        float acc[SPLAT_GRAD_STRIDE];
#pragma unroll
        for (int k = 0; k < SPLAT_GRAD_STRIDE; ++k) acc[k] = 0.f;
        float4 *a4 = reinterpret_cast<float4 *>(ws.accum + (size_t)i * SPLAT_GRAD_STRIDE);
#pragma unroll
        for (int k = 0; k < kRow4; ++k) {
            const float4 v = a4[k];
            acc[4 * k] = v.x; acc[4 * k + 1] = v.y; acc[4 * k + 2] = v.z; acc[4 * k + 3] = v.w;
        }
        float p[3], u[4], logit, ls[3];
        load_gaussian(a.map, i, p, u, logit, ls);
        const float4 co = reinterpret_cast<const float4 *>(ws.st.conic_opacity)[i];
        // the Adam step is skipped while a capacity flag is up (this iteration's lists, or an earlier iteration's the host has not
        // dealt with yet: SplatIterWorkspace.d_cam[12]); wave-uniform
        bool gate_up = false;
        if constexpr (ADAM) gate_up = (ws.st.status[1] | ws.st.status[3]) != 0 || ws.d_cam[12] != 0.f;
        // the row is consumed: the next iteration's K7 accumulates from zero without a memset.  (Unconditional -- rows outside the view
        // are zero already -- and AFTER every load above: a store the compiler can neither sink nor prove disjoint pins them up here,
        // ahead of the visibility test.)
#pragma unroll
        for (int k = 0; k < kRow4; ++k) a4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool vis = ws.st.radii[i] > 0;
        float dp[3] = {0.f, 0.f, 0.f}, du[4] = {0.f, 0.f, 0.f, 0.f}, dlogit = 0.f, dls[3] = {0.f, 0.f, 0.f};
        float drgb[3] = {0.f, 0.f, 0.f};
        if (vis) {
            Glue G;
            glue_forward(P, w2c_row2, p, u, logit, ls, iso, G);
            float S6[6];
            cov3d_from_scale_rot(G.s, c.scale_modifier, G.rq, S6);
            const float g_ndc[2] = {-(co.x * acc[0] + co.y * acc[1]) * 0.5f * c.W, -(co.z * acc[1] + co.y * acc[0]) * 0.5f * c.H};
            const float g_conic[3] = {-0.5f * acc[2], -acc[3], -0.5f * acc[4]};
            float dXc[3], dS6[6], ds[3] = {0.f, 0.f, 0.f}, drq[4] = {0.f, 0.f, 0.f, 0.f};
            project_gaussian_backward(c, G.Xc, S6, g_ndc, g_conic, dXc, dS6);
            if constexpr (MAPGRADS || !ISO) cov3d_backward(G.s, c.scale_modifier, G.rq, dS6, ds, drq);
            // colour channels: 6..8 rgb, 9 z, 10 silhouette (constant), 11 z^2
            drgb[0] = acc[6]; drgb[1] = acc[7]; drgb[2] = acc[8];
            const float dz = MAPGRADS ? acc[9] + 2.f * G.z * acc[11] : acc[5];     // (tracking form: the depth channel's sum sits in slot 5)
            glue_backward(P, w2c_row2, p, iso, G, dXc, dz, MAPGRADS ? acc[5] : 0.f, ds, drq, dp, du, &dlogit, dls, pose);
        }
        constexpr int kWidth[5] = {3, 3, 4, 1, ISO ? 1 : 3};
        // (the moments: one more round trip, all 27 loads at once, issued ahead of the gradient stores; holding them across the adjoint
        //  arithmetic costs 36 registers = one wave per SIMD less, measured slower)
        float mom1[5][4], mom2[5][4], rgb_old[3] = {0.f, 0.f, 0.f};
        if constexpr (ADAM) {
#pragma unroll
            for (int gidx = 0; gidx < 5; ++gidx)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool on = k < kWidth[gidx] && opt.grad[gidx] != nullptr;       // torch skips parameters without a gradient
                    mom1[gidx][k] = on ? opt.exp_avg[gidx][(size_t)i * kWidth[gidx] + k] : 0.f;
                    mom2[gidx][k] = on ? opt.exp_avg_sq[gidx][(size_t)i * kWidth[gidx] + k] : 0.f;
                }
            if (opt.grad[1]) { rgb_old[0] = a.map.rgb_colors[3 * i]; rgb_old[1] = a.map.rgb_colors[3 * i + 1]; rgb_old[2] = a.map.rgb_colors[3 * i + 2]; }
        }
        if constexpr (MAPGRADS) {
        if (a.cfg.gaussians_grad) {
            if (ws.d_means3D) { ws.d_means3D[3 * i] = dp[0]; ws.d_means3D[3 * i + 1] = dp[1]; ws.d_means3D[3 * i + 2] = dp[2]; }
            // isotropic map: Sigma = s^2 I does not depend on the quaternion -- the derivative is exactly zero (what autograd
            // leaves there is rounding noise); writing the exact zero lets the gradient exchange skip the four rotation floats
            if (ws.d_unnorm_rotations)
                reinterpret_cast<float4 *>(ws.d_unnorm_rotations)[i] = iso ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(du[0], du[1], du[2], du[3]);
        }
        if (ws.d_rgb_colors) { ws.d_rgb_colors[3 * i] = drgb[0]; ws.d_rgb_colors[3 * i + 1] = drgb[1]; ws.d_rgb_colors[3 * i + 2] = drgb[2]; }
        if (ws.d_logit_opacities) ws.d_logit_opacities[i] = dlogit;
        if (ws.d_log_scales) {
            if (iso) ws.d_log_scales[i] = dls[0];
            else { ws.d_log_scales[3 * i] = dls[0]; ws.d_log_scales[3 * i + 1] = dls[1]; ws.d_log_scales[3 * i + 2] = dls[2]; }
        }
        }
        if constexpr (ADAM) {
            // torch.optim.Adam over every row (a Gaussian outside the view has a zero gradient, but its moments still move it), on
            // the parameter values and moments read at the top
            const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
            const float *grads[5] = {dp, drgb, iso ? zero4 : du, &dlogit, dls};
            const float *olds[5] = {p, rgb_old, u, &logit, ls};
            float *params[5] = {a.map.means3D, a.map.rgb_colors, a.map.unnorm_rotations, a.map.logit_opacities, a.map.log_scales};
#pragma unroll
            for (int gidx = 0; gidx < 5; ++gidx) {
                if (!opt.grad[gidx]) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= kWidth[gidx]) continue;
                    float mm = mom1[gidx][k], vv = mom2[gidx][k];
                    const float gk = grads[gidx][k];
                    if ((gk == 0.f && mm == 0.f && vv == 0.f) || gate_up) continue;          // (see adam_map_kernel)
                    const size_t j = (size_t)i * kWidth[gidx] + k;
                    params[gidx][j] = adam_update(olds[gidx][k], gk, mm, vv, opt.beta1, opt.beta2, opt.step_size[gidx], opt.bc2_sqrt[gidx], opt.eps);
                    opt.exp_avg[gidx][j] = mm;
                    opt.exp_avg_sq[gidx][j] = vv;
                }
            }
        }
    }
    if (ws.st.tile_stride > 0) {
        // bucketed lists: this is the last kernel that needs the tile counters -- fold them into the status words
        // (num_rendered, longest list) and reset them for the next iteration's per-Gaussian kernel
        const int T = c_num_tiles(a.cam.image_width, a.cam.image_height);
        unsigned sum = 0, mx = 0;
        const int G = ws.st.group_count ? c_num_groups(a.cam.image_width, a.cam.image_height) : 0;      // G <= T
        for (int t = i; t < T; t += gridDim.x * kBlock) {
            const unsigned cnt = ws.st.tile_count[(size_t)t * SPLAT_COUNTER_STRIDE];
            ws.st.tile_count[(size_t)t * SPLAT_COUNTER_STRIDE] = 0;
            ws.st.tile_cursor[(size_t)t * SPLAT_COUNTER_STRIDE] = cnt;       // kept for a later pass over the same lists (splat_iter_means2d_accumulate)
            if (t < G) {
                // (the group's record count stays in word 1 of its counter line for a later pass over the same records:
                //  splat_iter_time_kernel times the sorting form of the forward composite on it)
                ws.st.group_count[(size_t)t * SPLAT_COUNTER_STRIDE + 1] = ws.st.group_count[(size_t)t * SPLAT_COUNTER_STRIDE];
                ws.st.group_count[(size_t)t * SPLAT_COUNTER_STRIDE] = 0;
            }
            sum += cnt;
            mx = max(mx, cnt);
        }
        if (blockIdx.x * kBlock < T) {              // uniform per block; later blocks hold no tile
            for (int m = 32; m >= 1; m >>= 1) { sum += (unsigned)__shfl_xor((int)sum, m, 64); mx = max(mx, (unsigned)__shfl_xor((int)mx, m, 64)); }
            if ((threadIdx.x & 63) == 0 && sum) { atomicAdd((unsigned *)&ws.st.status[0], sum); atomicMax((unsigned *)&ws.st.status[2], mx); }
        }
    }
    // a band of tile rows (the other ranks hold the other bands): this rank's capacity flags travel with the partial sums
    if (a.cfg.defer_finish && i == 0 && (ws.st.status[1] | ws.st.status[3]) != 0) atomicAdd(ws.sums + kFlagSum, 1.0);
    if (a.cfg.camera_grad) block_sum_to<kPoseSums>(sum_copy(ws.sums) + 8, pose, s_part);
}

// ---------------------------------------------------------------------------------------------------------
// F8: Adam
// ---------------------------------------------------------------------------------------------------------
struct AdamArgs {
    SplatMap map;
    SplatAdamMap opt;
};

// one lane per ELEMENT of the five groups laid end to end (coalesced 4-byte streams of param / grad / moments)
__global__ __launch_bounds__(kBlock) void adam_map_kernel(AdamArgs a, long long total) {
    const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total) return;
    if (a.opt.gate && a.opt.gate[12] != 0.f) return;            // the iteration that formed these gradients ran on truncated lists
    float *params[5] = {a.map.means3D, a.map.rgb_colors, a.map.unnorm_rotations, a.map.logit_opacities, a.map.log_scales};
    const int width[5] = {3, 3, 4, 1, a.map.isotropic ? 1 : 3};
    long long off = 0;
#pragma unroll
    for (int gidx = 0; gidx < 5; ++gidx) {
        const long long n = (long long)a.map.P * width[gidx];
        if (e >= off && e < off + n) {
            const float *grad = a.opt.grad[gidx];
            if (grad) {                                         // torch skips parameters without a gradient
                const long long j = e - off;
                float mm = a.opt.exp_avg[gidx][j], vv = a.opt.exp_avg_sq[gidx][j];
                const float gg = grad[j];
                // zero gradient on zero moments (Gaussians never seen since the optimizer was created; the rotations of an
                // isotropic map): the step leaves parameter and moments as they are -- nothing to write
                if (gg == 0.f && mm == 0.f && vv == 0.f) return;
                float *p = params[gidx];
                p[j] = adam_update(p[j], gg, mm, vv, a.opt.beta1, a.opt.beta2, a.opt.step_size[gidx], a.opt.bc2_sqrt[gidx], a.opt.eps);
                a.opt.exp_avg[gidx][j] = mm;
                a.opt.exp_avg_sq[gidx][j] = vv;
            }
        }
        off += n;
    }
}

// The 64 copies of the partial sums folded into copy 0 (the others zeroed): tile-row-sharded tracking then exchanges 256 bytes
// instead of 16 KB (pose_finish_kernel totals the copies either way)
__global__ __launch_bounds__(256) void fold_sums_kernel(double *sums) {
    static_assert(SPLAT_ITER_SUMS * 8 == 256 && SPLAT_ITER_SUM_COPIES == 64, "thread t: sum k = t / 8, copies (t % 8) * 8 .. + 7");
    const int t = threadIdx.x, k = t >> 3, part = t & 7;
    double part_sum[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) part_sum[c] = sums[(size_t)(part * 8 + c) * SPLAT_ITER_SUMS + k];
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v += part_sum[c];
        if (part * 8 + c != 0) sums[(size_t)(part * 8 + c) * SPLAT_ITER_SUMS + k] = 0.0;
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (part == 0) sums[k] = v;
}

__global__ void adam_pose_kernel(SplatMap map, int time_idx, const float *d_cam, float *state, float beta1, float beta2,
                                 float eps, float bc2_sqrt, float ss_rot, float ss_trans) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (d_cam[12] != 0.f) return;                               // (see SplatIterWorkspace.d_cam[12])
    const PoseAdam pa{state, beta1, beta2, eps, bc2_sqrt, ss_rot, ss_trans};
    pose_adam_step(map, time_idx, d_cam, d_cam[7], pa);
}

}  // namespace

hipError_t launch_depth_error_median(const float *out6, const float *depth, float *err, uint32_t *scratch, int HW, int32_t *counts,
                                     hipStream_t s);

static int num_tile_groups(const SplatCamera &cam) { return tile_groups(cam.image_width, cam.image_height); }
static bool group_binning(const SplatState &st, const SplatCamera &cam) { return group_binning(st, cam.image_width, cam.image_height); }

// workgroups of the iteration's last kernel: the one that closes the iteration + one per XCD band when the composites' launch order is kept
// (whole frames only: a band of tile rows is composited in the natural order)
static int finish_blocks(const SplatState &st) {
    return (st.tile_work && st.tile_order && st.tile_row_end <= st.tile_row_begin) ? 9 : 1;
}

// F1 in the mode the state asks for
static void launch_fused_preprocess(const FusedArgs &a, hipStream_t s) {
    const int P = a.map.P;
    if (P <= 0) return;
    const SplatState &st = a.ws.st;
    if (st.group_stride > 0)
        hipLaunchKernelGGL((fused_preprocess_kernel<2, kGroupBlock>), dim3((P + kGroupBlock - 1) / kGroupBlock), dim3(kGroupBlock),
                           sizeof(unsigned) * (size_t)num_tile_groups(a.cam), s, a);
    else if (st.order_hint && st.tile_stride > 0)
        hipLaunchKernelGGL((fused_preprocess_kernel<1, kBlock>), dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
    else if (st.tile_stride > 0 && !lists_sorted_by_composite(st) && P >= 8 * kDenseBlock * kDensePerThread &&
             (int)splat_num_tiles(a.cam.image_width, a.cam.image_height) <= kDenseMaxTiles)
        // long bucketed lists, no order known: the workgroup-level tile histogram (a map of a few thousand Gaussians stays on the
        // one-Gaussian-per-lane kernels: 4096 Gaussians per workgroup would leave most CUs idle)
        hipLaunchKernelGGL(fused_preprocess_dense_kernel, dim3((P + kDenseBlock * kDensePerThread - 1) / (kDenseBlock * kDensePerThread)),
                           dim3(kDenseBlock), sizeof(unsigned) * splat_num_tiles(a.cam.image_width, a.cam.image_height), s, a);
    else
        hipLaunchKernelGGL((fused_preprocess_kernel<0, kBlock>), dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a);
}

hipError_t launch_iter_loss_backward(const SplatCamera &cam, const SplatMap &map, const SplatFrameData &frame,
                                     const SplatLossConfig &cfg, SplatIterWorkspace &ws_in, hipStream_t s, const SplatPoseAdam *pose_adam,
                                     const SplatAdamMap *map_adam) {
    SplatIterWorkspace ws = ws_in;          // (copy: the binning mode is decided per call)
    if (!group_binning(ws.st, cam)) ws.st.group_stride = 0;
    FusedArgs a{cam, map, frame, cfg, ws, {}};
    ssim_window_host(a.win);
    const int W = cam.image_width, H = cam.image_height, HW = W * H;
    const int T = (int)splat_num_tiles(W, H);
    const int P = map.P;
    // no memsets: sums, tile counters and the accumulator rows are zeroed by the kernels that consume them
    // (pose_finish / tile_scan / fused_backward); the caller zero-initialises the workspace once
    hipError_t e = hipSuccess;
    const int gblocks = (P + kBlock - 1) / kBlock;
    launch_fused_preprocess(a, s);
    if (ws.st.tile_stride == 0) {
        e = launch_tile_scan(ws.st, T, s);
        if (e != hipSuccess) return e;
    }
    SplatGaussians g{};
    g.P = P;
    g.channels = 6;
    const bool sort_in_k6 = lists_sorted_by_composite(ws.st);
    e = launch_bin_forward(cam, g, ws.st, s, !sort_in_k6);
    if (e != hipSuccess) return e;
    // tracking without outlier rejection: the loss and its gradient planes are formed in the composite's epilogue
    TrackLossEpilogue ep{frame.im, frame.depth, ws.dL_dout6, ws.sums, cfg.sil_thres, cfg.w_im, cfg.w_depth, cfg.use_sil_for_loss, cfg.use_l1};
    bool loss_done = false;
    const bool fuse_loss = cfg.tracking && !cfg.ignore_outlier_depth_loss;
    // a band of tile rows (SplatState.tile_row_begin): only the loss that is formed per tile in the composite's epilogue is defined
    if (ws.st.tile_row_end > ws.st.tile_row_begin && !fuse_loss) return hipErrorInvalidValue;
    // ... and with short lists and nothing but the pose gradient wanted, the backward composite rides in the same kernel
    // (map gradients wanted as well -- the reference's backward() forms dL/d(rgb, opacity, scale) in tracking too --: the same kernel
    //  with the backward composite's mapping form inside)
    const bool one_kernel = fuse_loss && sort_in_k6 && cfg.fused_composite != 0;
    const bool full_sums = ws.d_rgb_colors != nullptr || ws.d_logit_opacities != nullptr;
    if (one_kernel) {
        e = launch_render_track_fused(cam, ws.feat8, ws.st, ws.out6, ws.accum, ep, cfg.fused_composite == 2, s, full_sums);
        if (e != hipSuccess) return e;
    } else {
        e = launch_render_forward_feat8(cam, ws.feat8, ws.st, ws.out6, sort_in_k6, s, fuse_loss ? &ep : nullptr, fuse_loss ? &loss_done : nullptr);
        if (e != hipSuccess) return e;
        if (cfg.ignore_outlier_depth_loss) {
            // torch.median of the depth error (exact radix selection, mapedit.hip) -> d_cam[13] (its bits through counts[4])
            e = launch_depth_error_median(ws.out6, frame.depth, ws.outlier_err, ws.outlier_scratch, HW,
                                          reinterpret_cast<int32_t *>(ws.d_cam) + 9, s);
            if (e != hipSuccess) return e;
        }
        if (cfg.tracking && loss_done) {
            // nothing: see the epilogue above
        } else if (cfg.tracking) {
            if (HW % 4 == 0) {
                const int blocks = min((HW / 4 + kBlock - 1) / kBlock, 2048);
                hipLaunchKernelGGL(track_loss_kernel<4>, dim3(blocks), dim3(kBlock), 0, s, a, HW);
            } else {
                const int blocks = min((HW + kBlock - 1) / kBlock, 2048);
                hipLaunchKernelGGL(track_loss_kernel<1>, dim3(blocks), dim3(kBlock), 0, s, a, HW);
            }
        } else {
            const int tiles = 3 * ((W + kCW - 1) / kCW) * ((H + kCH - 1) / kCH);
            const dim3 grid(8 * ((tiles + 7) / 8));                 // (ssim_tile: XCD x owns a contiguous run of tiles)
            if ((W & 3) == 0 && aligned16(ws.out6) && aligned16(frame.im) && aligned16(frame.depth) && aligned16(ws.ssim_maps) && aligned16(ws.dL_dout6)) {
                hipLaunchKernelGGL(ssim_forward_kernel<true>, grid, dim3(kBlock), 0, s, a, W, H);
                hipLaunchKernelGGL(map_loss_backward_kernel<true>, grid, dim3(kBlock), 0, s, a, W, H);
            } else {
                hipLaunchKernelGGL(ssim_forward_kernel<false>, grid, dim3(kBlock), 0, s, a, W, H);
                hipLaunchKernelGGL(map_loss_backward_kernel<false>, grid, dim3(kBlock), 0, s, a, W, H);
            }
        }
        // (which sums the backward composite forms: what a stored gradient or a stepped group needs)
        e = launch_render_backward_feat8(cam, ws.feat8, ws.st, ws.dL_dout6, ws.accum, P, false,
                                         ws.d_rgb_colors != nullptr || (map_adam && map_adam->grad[1]), s,
                                         ws.d_logit_opacities != nullptr || (map_adam && map_adam->grad[3]));
        if (e != hipSuccess) return e;
    }
    PoseAdam pa{};
    if (pose_adam)
        pa = PoseAdam{pose_adam->state, pose_adam->beta1, pose_adam->beta2, pose_adam->eps, pose_adam->bc2_sqrt, pose_adam->step_size_rot,
                      pose_adam->step_size_trans};
    if (P > 0) {
        const bool iso = map.isotropic != 0;
        const bool mapgrads = ws.d_means3D || ws.d_rgb_colors || ws.d_unnorm_rotations || ws.d_logit_opacities || ws.d_log_scales;
        const SplatAdamMap opt = map_adam ? *map_adam : SplatAdamMap{};
        const dim3 grid(gblocks), block(kBlock);
        if (map_adam && iso) hipLaunchKernelGGL((fused_backward_kernel<true, true, true>), grid, block, 0, s, a, opt);
        else if (map_adam) hipLaunchKernelGGL((fused_backward_kernel<true, true, false>), grid, block, 0, s, a, opt);
        else if (mapgrads && iso) hipLaunchKernelGGL((fused_backward_kernel<false, true, true>), grid, block, 0, s, a, opt);
        else if (mapgrads) hipLaunchKernelGGL((fused_backward_kernel<false, true, false>), grid, block, 0, s, a, opt);
        else if (iso) hipLaunchKernelGGL((fused_backward_kernel<false, false, true>), grid, block, 0, s, a, opt);
        else hipLaunchKernelGGL((fused_backward_kernel<false, false, false>), grid, block, 0, s, a, opt);
    }
    if (!cfg.defer_finish) hipLaunchKernelGGL(pose_finish_kernel, dim3(finish_blocks(ws.st)), dim3(256), 0, s, a, HW, pa);
    return hipGetLastError();
}

// F7 alone: the end of an iteration that ran with cfg.defer_finish (the caller has completed ws.sums in between)
hipError_t launch_iter_finish(const SplatCamera &cam, const SplatMap &map, const SplatFrameData &frame, const SplatLossConfig &cfg,
                              SplatIterWorkspace &ws, hipStream_t s, const SplatPoseAdam *pose_adam) {
    FusedArgs a{cam, map, frame, cfg, ws, {}};
    PoseAdam pa{};
    if (pose_adam)
        pa = PoseAdam{pose_adam->state, pose_adam->beta1, pose_adam->beta2, pose_adam->eps, pose_adam->bc2_sqrt, pose_adam->step_size_rot,
                      pose_adam->step_size_trans};
    hipLaunchKernelGGL(pose_finish_kernel, dim3(finish_blocks(ws.st)), dim3(256), 0, s, a, cam.image_width * cam.image_height, pa);
    return hipGetLastError();
}

// Forward half only (F1, lists, K6): the render of add_new_gaussians / evaluation.  Leaves every per-iteration
// scratch word the way a full iteration leaves it (tile counters zero, nothing accumulated).
hipError_t launch_fold_tile_counters(SplatState &st, int T, int G, hipStream_t s);

hipError_t launch_iter_render(const SplatCamera &cam, const SplatMap &map, const SplatFrameData &frame, SplatIterWorkspace &ws_in,
                              hipStream_t s) {
    SplatIterWorkspace ws = ws_in;
    if (!group_binning(ws.st, cam)) ws.st.group_stride = 0;
    FusedArgs a{cam, map, frame, SplatLossConfig{}, ws, {}};
    a.ws.max_2D_radius = nullptr;           // only get_loss updates variables['max_2D_radius'] (/root/reference/scripts/splatam.py:342)
    const int T = (int)splat_num_tiles(cam.image_width, cam.image_height);
    const int P = map.P;
    hipError_t e = hipSuccess;
    launch_fused_preprocess(a, s);
    if (ws.st.tile_stride == 0) {
        e = launch_tile_scan(ws.st, T, s);
        if (e != hipSuccess) return e;
    }
    SplatGaussians g{};
    g.P = P;
    g.channels = 6;
    const bool sort_in_k6 = lists_sorted_by_composite(ws.st);
    e = launch_bin_forward(cam, g, ws.st, s, !sort_in_k6);
    if (e != hipSuccess) return e;
    e = launch_render_forward_feat8(cam, ws.feat8, ws.st, ws.out6, sort_in_k6, s);
    if (e != hipSuccess) return e;
    if (ws.st.tile_stride > 0) return launch_fold_tile_counters(ws.st, T, num_tile_groups(cam), s);
    return hipGetLastError();
}

namespace {
// accumulate_mean2d_gradient (/root/reference/utils/slam_external.py:100-104) from the colour-only sums S1, S2 of a Gaussian:
// dL/dmeans2D (NDC, as the rasterizer returns it) = -(conic . S) * 0.5 * (W, H)   (K8: splat_math.h / fused_backward_kernel)
__global__ __launch_bounds__(kBlock) void means2d_accumulate_kernel(SplatIterWorkspace ws, int P, int W, int H, float *gaccum, float *denom, float *out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    float gx = 0.f, gy = 0.f;
    const bool seen = ws.st.radii[i] > 0;
    if (seen) {
        float4 *a4 = reinterpret_cast<float4 *>(ws.accum + (size_t)i * SPLAT_GRAD_STRIDE);
        const float4 s0 = a4[0], s1 = a4[1];
        a4[0] = make_float4(0.f, 0.f, 0.f, 0.f);               // consumed (slots 0..4 were written)
        a4[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        (void)s1;
        const float4 co = reinterpret_cast<const float4 *>(ws.st.conic_opacity)[i];
        gx = -(co.x * s0.x + co.y * s0.y) * 0.5f * (float)W;
        gy = -(co.z * s0.y + co.y * s0.x) * 0.5f * (float)H;
        if (gaccum) {                       // (NULL: the caller only wants the gradient itself, splatam_amd.plugin)
            gaccum[i] += sqrtf(gx * gx + gy * gy);
            denom[i] += 1.0f;
        }
    }
    if (out) { out[2 * (size_t)i] = gx; out[2 * (size_t)i + 1] = gy; }
}
}  // namespace

hipError_t launch_iter_means2d_accumulate(const SplatCamera &cam, const SplatMap &map, SplatIterWorkspace &ws, float *gaccum, float *denom,
                                          float *means2D_grad, hipStream_t s) {
    const int P = map.P;
    if (P <= 0) return hipSuccess;
    // bucketed lists: the iteration's last kernel folded and reset the tile counters, and left the counts in the cursor words
    SplatState st = ws.st;
    if (st.tile_stride > 0) st.tile_count = st.tile_cursor;
    st.tile_recs = nullptr;         // (whichever composite ran last on this state may not have left every batch's records: gather)
    hipError_t e = launch_render_backward_rgb_only(cam, ws.feat8, st, ws.dL_dout6, ws.accum, P, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(means2d_accumulate_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ws, P, cam.image_width, cam.image_height, gaccum,
                       denom, means2D_grad);
    return hipGetLastError();
}

hipError_t launch_iter_fold_sums(double *sums, hipStream_t s) {
    hipLaunchKernelGGL(fold_sums_kernel, dim3(1), dim3(256), 0, s, sums);
    return hipGetLastError();
}

hipError_t launch_iter_adam_map(const SplatMap &map, const SplatAdamMap &opt, hipStream_t s) {
    if (map.P <= 0) return hipSuccess;
    AdamArgs a{map, opt};
    const long long total = (long long)map.P * (3 + 3 + 4 + 1 + (map.isotropic ? 1 : 3));
    hipLaunchKernelGGL(adam_map_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, total);
    return hipGetLastError();
}

hipError_t launch_iter_adam_pose(const SplatMap &map, int time_idx, const float *d_cam, float *state, float beta1, float beta2,
                                 float eps, float bc2_sqrt, float ss_rot, float ss_trans, hipStream_t s) {
    hipLaunchKernelGGL(adam_pose_kernel, dim3(1), dim3(64), 0, s, map, time_idx, d_cam, state, beta1, beta2, eps, bc2_sqrt, ss_rot,
                       ss_trans);
    return hipGetLastError();
}

}  // namespace splat
