/* A binding in plain C: nothing but include/splat_hip.h and the HIP runtime API.
 *
 * What a maintainer of the reference would write to call this library from C / cgo / JNI: size the scratch with
 * splat_state_layout, hipMalloc ONE slab, splat_state_bind, then splat_preprocess_forward -> (read status[0]) -> splat_bin_forward ->
 * splat_render_forward -> splat_backward on 1 000 Gaussians, and compare every output with the CPU oracle through ITS C entry
 * points (oracle/raster_ref.c; test infrastructure: the checker, linked here only because this is a test).  Part 2: the same call with
 * group binning (no host read between the calls); part 3: the fused iteration through splat_iter_workspace_layout / _bind.
 *
 * Built by tests/test_gpu_capi_c.py (gcc, no hipcc needed):
 *   gcc -std=c99 -O1 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/capi_smoke.c \
 *       -L splatam_amd/lib -lsplat_hip -L oracle/_build -lraster_ref -L /opt/rocm/lib -lamdhip64 -lm
 * Exit code 0 and a line "capi_smoke ok ..." on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "splat_hip.h"

/* the oracle's C entry points (float32 build, oracle/raster_ref.c) */
typedef struct ref_ctx ref_ctx;
ref_ctx *ref_create(void);
void ref_destroy(ref_ctx *c);
int ref_forward(ref_ctx *c, int P, int C, int W, int H, const float *bg, const float *means3D, const float *colors, const float *opac,
                const float *scales, float mod, const float *rot, const float *cov3D_precomp, const float *view, const float *proj,
                float tanfovx, float tanfovy, float *out_color, float *out_depth, int *out_radii);
int ref_backward(const ref_ctx *c, const float *bg, const float *means3D, const float *colors, const float *scales, float mod,
                 const float *rot, const float *view, const float *proj, float tanfovx, float tanfovy, const float *dL_dpix,
                 float *dmeans3D, float *dmeans2D, float *dcolors, float *dopac, float *dscales, float *drot, float *dcov3D);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SPLAT(x) do { int rc_ = (x); if (rc_ != SPLAT_OK) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, splat_error_string(rc_)); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static float frand(void) { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 16777216.0f; }

static void *to_device(const void *host, size_t bytes) {
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

/* max |a - b| and max |b| over n floats; elements beyond `tol` are counted */
static int compare(const char *what, const float *a, const float *b, size_t n, float tol_abs, float tol_rel_of_max, size_t allowed) {
    float scale = 0.f, worst = 0.f;
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) if (fabsf(b[i]) > scale) scale = fabsf(b[i]);
    const float tol = tol_abs + tol_rel_of_max * scale;
    for (size_t i = 0; i < n; i++) {
        float d = fabsf(a[i] - b[i]);
        if (!(d <= tol)) bad++;
        if (d > worst) worst = d;
    }
    printf("  %-14s max err %.3e (scale %.3e), %zu of %zu beyond %.1e\n", what, worst, scale, bad, n, tol);
    return bad <= allowed ? 0 : 1;
}

int main(void) {
    enum { P = 1000, W = 160, H = 112, C = 3 };
    const float fx = 150.f, fy = 150.f, cx = W / 2 - 0.5f, cy = H / 2 - 0.5f, near_ = 0.01f, far_ = 100.f;
    if (splat_abi_version() != SPLAT_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", splat_abi_version(), SPLAT_ABI_VERSION); return 1; }
    if (splat_sizeof("SplatState") != sizeof(SplatState) || splat_sizeof("SplatGrads") != sizeof(SplatGrads) ||
        splat_sizeof("SplatArrayInfo") != sizeof(SplatArrayInfo)) { fprintf(stderr, "struct layout mismatch\n"); return 1; }

    /* ---- a seeded SplaTAM-like cloud (one Gaussian per random pixel, depth 1..4 m, ~1 px sigma) ---- */
    static float means[P * 3], colors[P * C], opac[P], scales[P * 3], rot[P * 4], gout[C * H * W];
    for (int i = 0; i < P; i++) {
        float u = frand() * W - 0.5f, v = frand() * H - 0.5f, z = 1.f + 3.f * frand();
        means[3 * i] = (u - cx) / fx * z; means[3 * i + 1] = (v - cy) / fy * z; means[3 * i + 2] = z;
        float s = (0.8f + 2.5f * frand()) * z / fx;
        scales[3 * i] = s; scales[3 * i + 1] = s * (0.7f + 0.6f * frand()); scales[3 * i + 2] = s * (0.7f + 0.6f * frand());
        float q[4] = {1.f, 0.3f * (frand() - .5f), 0.3f * (frand() - .5f), 0.3f * (frand() - .5f)};
        float qn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; k++) rot[4 * i + k] = q[k] / qn;
        opac[i] = 0.3f + 0.69f * frand();
        for (int k = 0; k < C; k++) colors[C * i + k] = frand();
    }
    for (int i = 0; i < C * H * W; i++) gout[i] = frand() - 0.5f;
    /* the settings tuple's matrices (/root/reference/utils/recon_helpers.py:8-13): viewmatrix = w2c^T (identity here),
     * projmatrix = (P w2c)^T, both flattened row-major */
    const float bg[3] = {0.1f, 0.2f, 0.3f};
    float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float Pm[16] = {2 * fx / W, 0, -(W - 2 * cx) / W, 0, 0, 2 * fy / H, -(H - 2 * cy) / H, 0, 0, 0, far_ / (far_ - near_), -(far_ * near_) / (far_ - near_), 0, 0, 1, 0};
    float proj[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) proj[4 * r + c] = Pm[4 * c + r];
    const float tanx = W / (2 * fx), tany = H / (2 * fy);

    /* ---- the oracle ---- */
    static float o_color[C * H * W], o_depth[H * W], o_dm3[P * 3], o_dm2[P * 3], o_dcol[P * C], o_dop[P], o_dsc[P * 3], o_drot[P * 4], o_dcov[P * 6];
    static int o_radii[P];
    ref_ctx *ref = ref_create();
    if (ref_forward(ref, P, C, W, H, bg, means, colors, opac, scales, 1.f, rot, NULL, view, proj, tanx, tany, o_color, o_depth, o_radii)) return 4;
    if (ref_backward(ref, bg, means, colors, scales, 1.f, rot, view, proj, tanx, tany, gout, o_dm3, o_dm2, o_dcol, o_dop, o_dsc, o_drot, o_dcov)) return 4;

    /* ---- the HIP library through its C ABI ---- */
    hipStream_t stream;
    HIP(hipStreamCreate(&stream));
    SplatCamera cam;
    memset(&cam, 0, sizeof cam);
    cam.image_height = H; cam.image_width = W; cam.tanfovx = tanx; cam.tanfovy = tany; cam.scale_modifier = 1.f;
    cam.bg = (const float *)to_device(bg, sizeof bg);
    cam.viewmatrix = (const float *)to_device(view, sizeof view);
    cam.projmatrix = (const float *)to_device(proj, sizeof proj);
    SplatGaussians g;
    memset(&g, 0, sizeof g);
    g.P = P; g.channels = C;
    g.means3D = (const float *)to_device(means, sizeof means); g.opacities = (const float *)to_device(opac, sizeof opac);
    g.colors_precomp = (const float *)to_device(colors, sizeof colors); g.scales = (const float *)to_device(scales, sizeof scales);
    g.rotations = (const float *)to_device(rot, sizeof rot);
    if (!cam.bg || !cam.viewmatrix || !cam.projmatrix || !g.means3D || !g.opacities || !g.colors_precomp || !g.scales || !g.rotations) return 2;

    /* scratch: geometry first (the number of instances is an OUTPUT of the first call, as in the reference), then the lists */
    SplatArrayInfo arrays[SPLAT_LAYOUT_MAX_ARRAYS];
    size_t total = 0;
    int n = splat_state_layout(P, W, H, 1, 0, SPLAT_LAYOUT_BACKWARD, arrays, SPLAT_LAYOUT_MAX_ARRAYS, &total);
    if (n <= 0 || n > SPLAT_LAYOUT_MAX_ARRAYS) return 1;
    void *slab = NULL;
    HIP(hipMalloc(&slab, total));
    SplatState st;
    SplatGrads gr;
    memset(&st, 0, sizeof st);
    memset(&gr, 0, sizeof gr);
    SPLAT(splat_state_bind(&st, &gr, slab, arrays, n, 1, 0));
    /* the lists do not exist yet: the tile scan of the first call publishes EMPTY lists when the count exceeds st.capacity, so a caller
     * that sizes its lists from status[0] (as the reference does) declares the capacity unbounded for this call (include/splat_hip.h) */
    st.capacity = INT64_MAX / 2;
    SPLAT(splat_preprocess_forward(&cam, &g, &st, stream));
    int32_t status[4];
    HIP(hipMemcpyAsync(status, st.status, sizeof status, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    const int64_t instances = status[0];
    SplatArrayInfo list_arrays[SPLAT_LAYOUT_MAX_ARRAYS];
    size_t list_total = 0;
    int ln = splat_state_layout(0, W, H, 1, instances, SPLAT_LAYOUT_LONG_LISTS, list_arrays, SPLAT_LAYOUT_MAX_ARRAYS, &list_total);
    void *lists = NULL;
    HIP(hipMalloc(&lists, list_total));
    for (int i = 0; i < ln; i++) {            /* only the list arrays of the second layout are wired */
        char *p = (char *)lists + list_arrays[i].offset;
        if (!strcmp(list_arrays[i].name, "keys")) st.keys = (uint64_t *)p;
        else if (!strcmp(list_arrays[i].name, "point_list")) st.point_list = (uint32_t *)p;
        else if (!strcmp(list_arrays[i].name, "keys_alt")) st.keys_alt = (uint64_t *)p;
        else if (!strcmp(list_arrays[i].name, "long_items")) st.long_items = (uint32_t *)p;
    }
    st.capacity = instances;
    float *d_color = NULL, *d_depth = NULL;
    HIP(hipMalloc((void **)&d_color, sizeof o_color));
    HIP(hipMalloc((void **)&d_depth, sizeof o_depth));
    SPLAT(splat_bin_forward(&cam, &g, &st, stream));
    SPLAT(splat_render_forward(&cam, &g, &st, d_color, d_depth, stream));
    gr.dL_dcolor = (const float *)to_device(gout, sizeof gout);
    HIP(hipMalloc((void **)&gr.dL_dmeans3D, sizeof o_dm3)); HIP(hipMalloc((void **)&gr.dL_dmeans2D, sizeof o_dm2));
    HIP(hipMalloc((void **)&gr.dL_dcolors, sizeof o_dcol)); HIP(hipMalloc((void **)&gr.dL_dopacities, sizeof o_dop));
    HIP(hipMalloc((void **)&gr.dL_dscales, sizeof o_dsc)); HIP(hipMalloc((void **)&gr.dL_drotations, sizeof o_drot));
    SPLAT(splat_backward(&cam, &g, &st, &gr, stream));
    static float h_color[C * H * W], h_depth[H * W], h_dm3[P * 3], h_dm2[P * 3], h_dcol[P * C], h_dop[P], h_dsc[P * 3], h_drot[P * 4];
    static int h_radii[P];
    HIP(hipStreamSynchronize(stream));
    HIP(hipMemcpy(h_color, d_color, sizeof h_color, hipMemcpyDeviceToHost)); HIP(hipMemcpy(h_depth, d_depth, sizeof h_depth, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_radii, st.radii, sizeof h_radii, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_dm3, gr.dL_dmeans3D, sizeof h_dm3, hipMemcpyDeviceToHost)); HIP(hipMemcpy(h_dm2, gr.dL_dmeans2D, sizeof h_dm2, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_dcol, gr.dL_dcolors, sizeof h_dcol, hipMemcpyDeviceToHost)); HIP(hipMemcpy(h_dop, gr.dL_dopacities, sizeof h_dop, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(h_dsc, gr.dL_dscales, sizeof h_dsc, hipMemcpyDeviceToHost)); HIP(hipMemcpy(h_drot, gr.dL_drotations, sizeof h_drot, hipMemcpyDeviceToHost));

    /* ---- compare: north-star tolerances (1e-4 colour / depth, 1e-3 of the maximum on gradients; radii exact); at most two pixels /
     *      rows may sit on a float32 threshold decision at this size ---- */
    int fail = 0, radii_bad = 0;
    for (int i = 0; i < P; i++) radii_bad += h_radii[i] != o_radii[i];
    printf("capi_smoke: %d Gaussians, %dx%d, %lld instances, slab %zu + lists %zu bytes, %d radii differ\n", P, W, H, (long long)instances, total, list_total, radii_bad);
    fail += radii_bad > 1;
    fail += compare("color", h_color, o_color, (size_t)C * H * W, 1e-4f, 0.f, 2);
    fail += compare("depth", h_depth, o_depth, (size_t)H * W, 1e-4f, 1e-4f, 2);
    fail += compare("dL/dmeans3D", h_dm3, o_dm3, P * 3, 0.f, 1e-3f, 2);
    fail += compare("dL/dmeans2D", h_dm2, o_dm2, P * 3, 0.f, 1e-3f, 2);
    fail += compare("dL/dcolors", h_dcol, o_dcol, P * C, 0.f, 1e-3f, 2);
    fail += compare("dL/dopacities", h_dop, o_dop, P, 0.f, 1e-3f, 2);
    fail += compare("dL/dscales", h_dsc, o_dsc, P * 3, 0.f, 1e-3f, 2);
    fail += compare("dL/drotations", h_drot, o_drot, P * 4, 0.f, 1e-3f, 2);
    /* ---- part 2: the same call with GROUP BINNING (include/splat_hip.h, ABI 10): the lists of this scene are known to be short
     *      (status[2] of the call above), so K1 files group records, the forward composite sorts its own lists and nothing has to be read
     *      between the calls; staged records handed to the backward composite (SPLAT_LAYOUT_RECS); the overflow flag in a host word ---- */
    {
        const int32_t longest = status[2];
        const int64_t tiles = (int64_t)splat_num_tiles(W, H), stride = 1024, cap2 = tiles * stride;
        SplatArrayInfo a2[SPLAT_LAYOUT_MAX_ARRAYS];
        size_t total2 = 0;
        int n2 = splat_state_layout(P, W, H, 1, cap2, SPLAT_LAYOUT_GROUPS | SPLAT_LAYOUT_BACKWARD | SPLAT_LAYOUT_RECS, a2, SPLAT_LAYOUT_MAX_ARRAYS, &total2);
        if (n2 <= 0 || n2 > SPLAT_LAYOUT_MAX_ARRAYS) return 1;
        void *slab2 = NULL;
        HIP(hipMalloc(&slab2, total2));
        SplatState s2;
        SplatGrads g2 = gr;                     /* the same gradient outputs, a scratch of its own */
        memset(&s2, 0, sizeof s2);
        SPLAT(splat_state_bind(&s2, &g2, slab2, a2, n2, 1, cap2));
        if (!s2.group_count || !s2.group_recs || !s2.tile_recs || s2.keys) { fprintf(stderr, "GROUPS layout: unexpected arrays\n"); return 1; }
        s2.tile_stride = (int32_t)stride;
        s2.group_stride = (int32_t)(SPLAT_GROUP_TILES * SPLAT_GROUP_TILES * stride);
        s2.max_list_hint = longest + longest / 2 + 1;
        int32_t *flag_host = NULL;
        HIP(hipHostMalloc((void **)&flag_host, sizeof(int32_t), hipHostMallocDefault));
        *flag_host = 0;
        s2.status_host = flag_host;
        HIP(hipMemsetAsync(d_color, 0, sizeof o_color, stream));
        SPLAT(splat_forward(&cam, &g, &s2, d_color, d_depth, stream));     /* two launches, no host read */
        SPLAT(splat_backward(&cam, &g, &s2, &g2, stream));
        HIP(hipStreamSynchronize(stream));
        if (*flag_host != 0) { fprintf(stderr, "group binning: a list outgrew its bucket (flag %d)\n", *flag_host); return 5; }
        HIP(hipMemcpy(h_color, d_color, sizeof h_color, hipMemcpyDeviceToHost)); HIP(hipMemcpy(h_depth, d_depth, sizeof h_depth, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(h_dm3, g2.dL_dmeans3D, sizeof h_dm3, hipMemcpyDeviceToHost)); HIP(hipMemcpy(h_dsc, g2.dL_dscales, sizeof h_dsc, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(h_dcol, g2.dL_dcolors, sizeof h_dcol, hipMemcpyDeviceToHost));
        printf("capi_smoke (group binning, longest list %d, slab %zu bytes):\n", longest, total2);
        fail += compare("color", h_color, o_color, (size_t)C * H * W, 1e-4f, 0.f, 2);
        fail += compare("depth", h_depth, o_depth, (size_t)H * W, 1e-4f, 1e-4f, 2);
        fail += compare("dL/dmeans3D", h_dm3, o_dm3, P * 3, 0.f, 1e-3f, 2);
        fail += compare("dL/dcolors", h_dcol, o_dcol, P * C, 0.f, 1e-3f, 2);
        fail += compare("dL/dscales", h_dsc, o_dsc, P * 3, 0.f, 1e-3f, 2);
    }

    /* ---- part 3: the FUSED iteration through splat_iter_workspace_layout / _bind: ONE slab, zeroed as a binding would zero
     *      it (a superset of what the layout marks zero_init -- the launch order among it: a zeroed tile_order IS the natural order), a
     *      render, one mapping iteration (which writes the launch order), the render again: bit-identical, and equal to the oracle ---- */
    {
        static float logit[P], logsc[P * 3], camq[4] = {1.f, 0.f, 0.f, 0.f}, camt[3] = {0.f, 0.f, 0.f}, w2c[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        static float zeros6[6];
        for (int i = 0; i < P; i++) {
            logit[i] = logf(opac[i] / (1.f - opac[i]));
            for (int k = 0; k < 3; k++) logsc[3 * i + k] = logf(scales[3 * i + k]);
        }
        SplatCamera cam0 = cam;
        cam0.bg = (const float *)to_device(zeros6, sizeof zeros6);          /* the fused iteration renders on black */
        SplatMap map;
        memset(&map, 0, sizeof map);
        map.P = P; map.isotropic = 0; map.num_frames = 1;
        map.means3D = (float *)to_device(means, sizeof means); map.rgb_colors = (float *)to_device(colors, sizeof colors);
        map.unnorm_rotations = (float *)to_device(rot, sizeof rot); map.logit_opacities = (float *)to_device(logit, sizeof logit);
        map.log_scales = (float *)to_device(logsc, sizeof logsc);
        map.cam_unnorm_rots = (float *)to_device(camq, sizeof camq); map.cam_trans = (float *)to_device(camt, sizeof camt);
        const int64_t cap3 = 4 * P + 65536;
        const int fl = SPLAT_LAYOUT_SSIM | SPLAT_LAYOUT_TILE_ORDER;
        SplatArrayInfo a3[SPLAT_LAYOUT_MAX_ARRAYS];
        size_t total3 = 0;
        int n3 = splat_iter_workspace_layout(P, W, H, cap3, 0, fl, a3, SPLAT_LAYOUT_MAX_ARRAYS, &total3);
        if (n3 <= 0 || n3 > SPLAT_LAYOUT_MAX_ARRAYS || total3 != splat_iter_workspace_bytes(P, W, H, cap3, 0, fl)) return 1;
        void *slab3 = NULL;
        HIP(hipMalloc(&slab3, total3));
        HIP(hipMemset(slab3, 0, total3));
        SplatIterWorkspace ws;
        memset(&ws, 0, sizeof ws);
        SPLAT(splat_iter_workspace_bind(&ws, slab3, a3, n3, cap3, 0));
        if (!ws.st.tile_order || !ws.st.tile_work || !ws.out6 || !ws.ssim_maps) { fprintf(stderr, "iteration layout: missing arrays\n"); return 1; }
        SplatFrameData frame;
        memset(&frame, 0, sizeof frame);
        frame.w2c = (const float *)to_device(w2c, sizeof w2c);
        frame.time_idx = 0;
        SPLAT(splat_iter_render(&cam0, &map, &frame, &ws, stream));
        static float r_a[6 * H * W], r_b[6 * H * W], o_black[C * H * W], o_bdepth[H * W];
        static int o_bradii[P];
        HIP(hipStreamSynchronize(stream));
        HIP(hipMemcpy(r_a, ws.out6, sizeof r_a, hipMemcpyDeviceToHost));
        /* a frame to fit: the render itself, brightened (a non-zero loss with a gradient) */
        static float im[3 * H * W], dep[H * W];
        for (int i = 0; i < 3 * H * W; i++) im[i] = 0.9f * r_a[i] + 0.05f;
        for (int i = 0; i < H * W; i++) dep[i] = r_a[3 * H * W + i] * 1.02f;
        frame.im = (const float *)to_device(im, sizeof im);
        frame.depth = (const float *)to_device(dep, sizeof dep);
        SplatLossConfig cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.gaussians_grad = 1; cfg.use_l1 = 1; cfg.sil_thres = 0.5f; cfg.w_im = 0.5f; cfg.w_depth = 1.0f;
        HIP(hipMalloc((void **)&ws.d_means3D, sizeof o_dm3));
        SPLAT(splat_iter_loss_backward(&cam0, &map, &frame, &cfg, &ws, stream));      /* its last kernel writes the launch order (tile + 1) */
        SPLAT(splat_iter_render(&cam0, &map, &frame, &ws, stream));                   /* ... which this render's composite follows */
        float d_cam[SPLAT_ITER_DCAM];
        HIP(hipStreamSynchronize(stream));
        HIP(hipMemcpy(r_b, ws.out6, sizeof r_b, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(d_cam, ws.d_cam, sizeof d_cam, hipMemcpyDeviceToHost));
        static uint32_t order[8 * ((W / 16 + 1) * (H / 16 + 1) / 8 + 1)];
        const size_t order_words = 8 * ((splat_num_tiles(W, H) + 7) / 8);
        HIP(hipMemcpy(order, ws.st.tile_order, sizeof(uint32_t) * order_words, hipMemcpyDeviceToHost));
        size_t written = 0;
        for (size_t i = 0; i < order_words; i++) written += order[i] != 0;
        int same = memcmp(r_a, r_b, sizeof r_a) == 0;
        if (ref_forward(ref, P, C, W, H, zeros6, means, colors, opac, scales, 1.f, rot, NULL, view, proj, tanx, tany, o_black, o_bdepth, o_bradii)) return 4;
        printf("capi_smoke (fused iteration through the layout, slab %zu bytes): loss %.6f, flag %.0f, %zu of %zu order words written, renders %s\n",
               total3, d_cam[7], d_cam[12], written, order_words, same ? "bit-identical" : "DIFFER");
        fail += !same || !(d_cam[7] > 0.f) || d_cam[12] != 0.f || written != order_words;
        fail += compare("rgb planes", r_b, o_black, (size_t)C * H * W, 1e-4f, 0.f, 2);
        fail += compare("depth plane", r_b + 3 * H * W, o_bdepth, (size_t)H * W, 1e-4f, 1e-4f, 2);
    }
    ref_destroy(ref);
    if (fail) { printf("capi_smoke FAILED (%d checks)\n", fail); return 10; }
    printf("capi_smoke ok\n");
    return 0;
}
