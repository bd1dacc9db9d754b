/*
 * CPU oracle #2 for the Gaussian-splat rasterizer: plain C, float32 arithmetic,
 * forward AND hand-written backward.
 *
 * TEST INFRASTRUCTURE ONLY -- only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load the library built from this file.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the un-vendored
 * dependency JonathonLuiten/diff-gaussian-rasterization-w-depth @ cb65e4b8
 * (/root/reference/requirements.txt:15); the reference carries no golden
 * vectors.  This file restates the published algorithm
 *  step by step; it is cross-checked against the
 * autograd oracle oracle/raster_ref.py in tests/test_oracle.py.
 *
 * Reference call sites this restates the callee of:
 *   forward  .. /root/reference/scripts/splatam.py:249,253,384
 *   backward .. /root/reference/scripts/splatam.py:702,854 (loss.backward())
 *   settings .. /root/reference/utils/recon_helpers.py:14-26 (matrix layout :8-13)
 *
 * Build: see oracle/Makefile  ->  oracle/_build/libraster_ref.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define MAXC 8

/* Working precision.  Default: float32, the arithmetic type of the path (every operation of Appendix A rounds as the
 * reference's float kernels round).  -DREF_DOUBLE builds the SAME statements in float64 (library
 * _build/libraster_ref_f64.so): the "exact" evaluation of the algorithm that the tests use to calibrate how far two correct
 * float32 implementations may differ (pixel coordinates of ~1e3 carry 1e-4 px of float32 rounding, which the backward
 * chain amplifies through cancelling sums).  Threshold constants keep their float32 VALUES in both builds (0.99f, 1.f/255.f,
 * 0.0001f, 0.2f, 0.3f ... are float literals promoted exactly), and the depth sort key is always the float32 bit pattern. */
#ifdef REF_DOUBLE
typedef double real;
#define R_EXP exp
#define R_SQRT sqrt
#define R_MIN fmin
#define R_MAX fmax
#define R_CEIL ceil
#else
typedef float real;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_MIN fminf
#define R_MAX fmaxf
#define R_CEIL ceilf
#endif
int ref_real_bytes(void) { return (int)sizeof(real); }

typedef struct {
    int P, C, W, H, gx, gy;
    /* per-Gaussian geometry (Appendix A, preprocess outputs) */
    real *depth, *xy, *conic_op, *cov3d, *cov2d;
    int *radii, *rect;          /* rect: minx,miny,maxx,maxy (tiles) */
    /* binning */
    int64_t R;
    int64_t pairs;              /* live (pixel, Gaussian) pairs composited by the last forward */
    int upstream_scale;         /* ref_set_upstream_scale: dscales without the scale_modifier factor (the CUDA original's numbers) */
    int *range;                 /* [tiles+1] */
    int *list;                  /* [R] Gaussian ids, per tile sorted by (depth bits, id) */
    /* per-pixel */
    real *final_T;
    int *n_contrib;
} ref_ctx;

ref_ctx *ref_create(void) { return (ref_ctx *)calloc(1, sizeof(ref_ctx)); }

static void ctx_release(ref_ctx *c) {
    free(c->depth); free(c->xy); free(c->conic_op); free(c->cov3d); free(c->cov2d);
    free(c->radii); free(c->rect); free(c->range); free(c->list);
    free(c->final_T); free(c->n_contrib);
    memset(c, 0, sizeof(*c));
}
void ref_destroy(ref_ctx *c) { if (c) { ctx_release(c); free(c); } }
/* The CUDA original's computeCov3D adjoint returns dL/d(mod * s) as dL/dscale ([UPSTREAM-RECALLED] dL_dscale->x = dot(Rt[0],
 * dL_dMt[0]) with M = S R built from s = mod * scale): without the factor `mod` of the chain rule.  on != 0: ref_backward hands out
 * those numbers (the twin of SplatGrads.flags SPLAT_GRADS_UPSTREAM_SCALE, include/splat_hip.h); 0 (default): the true gradient w.r.t.
 * the scales the caller passed.  Equal at mod = 1 (every SplaTAM configuration). */
void ref_set_upstream_scale(ref_ctx *c, int on) { if (c) c->upstream_scale = on; }
int64_t ref_num_rendered(const ref_ctx *c) { return c->R; }
int64_t ref_num_pairs(const ref_ctx *c) { return c->pairs; }
const int *ref_ranges(const ref_ctx *c) { return c->range; }
const int *ref_list(const ref_ctx *c) { return c->list; }
const real *ref_final_T(const ref_ctx *c) { return c->final_T; }
const int *ref_n_contrib(const ref_ctx *c) { return c->n_contrib; }
const real *ref_geom_xy(const ref_ctx *c) { return c->xy; }
const real *ref_geom_conic_op(const ref_ctx *c) { return c->conic_op; }
const real *ref_geom_depth(const ref_ctx *c) { return c->depth; }

/* 4x4 given as 16 floats, element (row r, col c) at m[c*4+r] (Appendix A conventions). */
static inline real m4(const real *m, int r, int c) { return m[c * 4 + r]; }

static void quat_rot(const real *q, real R[3][3]) {
    /* same polynomial as /root/reference/utils/slam_external.py:33-41, no renormalisation */
    real r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

static void cov3d_of(const real *scale, real mod, const real *q, real *out6) {
    real R[3][3], M[3][3];
    quat_rot(q, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = R[i][j] * (mod * scale[j]);
    real S[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        real a = 0.f; for (int k = 0; k < 3; k++) a += M[i][k] * M[j][k]; S[i][j] = a;
    }
    out6[0] = S[0][0]; out6[1] = S[0][1]; out6[2] = S[0][2]; out6[3] = S[1][1]; out6[4] = S[1][2]; out6[5] = S[2][2];
}

typedef struct { uint32_t key; int id; } kv_t;
static int kv_cmp(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* Forward.  cov3D_precomp may be NULL.  Returns 0. */
int ref_forward(ref_ctx *c, int P, int C, int W, int H,
                const real *bg, const real *means3D, const real *colors, const real *opac,
                const real *scales, real mod, const real *rot, const real *cov3D_precomp,
                const real *view, const real *proj, real tanfovx, real tanfovy,
                real *out_color, real *out_depth, int *out_radii)
{
    if (C > MAXC) return 1;
    ctx_release(c);
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, tiles = gx * gy;
    c->P = P; c->C = C; c->W = W; c->H = H; c->gx = gx; c->gy = gy;
    c->depth = (real *)calloc(P, sizeof(real)); c->xy = (real *)calloc(P, 2 * sizeof(real)); c->conic_op = (real *)calloc(P, 4 * sizeof(real));
    c->cov3d = (real *)calloc(P, 6 * sizeof(real)); c->cov2d = (real *)calloc(P, 3 * sizeof(real));
    c->radii = (int *)calloc(P, 4); c->rect = (int *)calloc(P, 16);
    c->range = (int *)calloc(tiles + 1, 4);
    c->final_T = (real *)malloc((size_t)W * H * sizeof(real)); c->n_contrib = (int *)calloc((size_t)W * H, 4);
    const real fx = W / (2.f * tanfovx), fy = H / (2.f * tanfovy);
    int *count = (int *)calloc(tiles, 4);

    /* ---- preprocess (Appendix A steps 1-9) ---- */
    for (int i = 0; i < P; i++) {
        const real *p = means3D + 3 * i;
        real tv[3];
        for (int r = 0; r < 3; r++) tv[r] = m4(view, r, 0) * p[0] + m4(view, r, 1) * p[1] + m4(view, r, 2) * p[2] + m4(view, r, 3);
        out_radii[i] = 0;
        if (tv[2] <= 0.2f) continue;
        real hom[4];
        for (int r = 0; r < 4; r++) hom[r] = m4(proj, r, 0) * p[0] + m4(proj, r, 1) * p[1] + m4(proj, r, 2) * p[2] + m4(proj, r, 3);
        real pw = 1.f / (hom[3] + 0.0000001f);
        real ndcx = hom[0] * pw, ndcy = hom[1] * pw;
        real *S6 = c->cov3d + 6 * i;
        if (cov3D_precomp) memcpy(S6, cov3D_precomp + 6 * i, 6 * sizeof(real));
        else cov3d_of(scales + 3 * i, mod, rot + 4 * i, S6);
        /* EWA */
        real limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        real txtz = tv[0] / tv[2], tytz = tv[1] / tv[2];
        real tx = R_MIN(limx, R_MAX(-limx, txtz)) * tv[2];
        real ty = R_MIN(limy, R_MAX(-limy, tytz)) * tv[2];
        real tz = tv[2];
        real J[2][3] = {{fx / tz, 0.f, -(fx * tx) / (tz * tz)}, {0.f, fy / tz, -(fy * ty) / (tz * tz)}};
        real T[2][3];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++)
            T[r][k] = J[r][0] * m4(view, 0, k) + J[r][1] * m4(view, 1, k) + J[r][2] * m4(view, 2, k);
        real Sg[3][3] = {{S6[0], S6[1], S6[2]}, {S6[1], S6[3], S6[4]}, {S6[2], S6[4], S6[5]}};
        real TS[2][3];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++) TS[r][k] = T[r][0] * Sg[0][k] + T[r][1] * Sg[1][k] + T[r][2] * Sg[2][k];
        real a = TS[0][0] * T[0][0] + TS[0][1] * T[0][1] + TS[0][2] * T[0][2] + 0.3f;
        real b = TS[0][0] * T[1][0] + TS[0][1] * T[1][1] + TS[0][2] * T[1][2];
        real cc = TS[1][0] * T[1][0] + TS[1][1] * T[1][1] + TS[1][2] * T[1][2] + 0.3f;
        real det = a * cc - b * b;
        if (det == 0.f) continue;
        real di = 1.f / det;
        real mid = 0.5f * (a + cc);
        real disc = R_SQRT(R_MAX(0.1f, mid * mid - det));
        real lam = R_MAX(mid + disc, mid - disc);
        real radius = R_CEIL(3.f * R_SQRT(lam));
        real px = ((ndcx + 1.f) * W - 1.f) * 0.5f, py = ((ndcy + 1.f) * H - 1.f) * 0.5f;
        int x0 = (int)((px - radius) / TILE), y0 = (int)((py - radius) / TILE);
        int x1 = (int)((px + radius + TILE - 1) / TILE), y1 = (int)((py + radius + TILE - 1) / TILE);
        x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
        y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        c->depth[i] = tv[2]; c->radii[i] = out_radii[i] = (int)radius;
        c->xy[2 * i] = px; c->xy[2 * i + 1] = py;
        c->conic_op[4 * i] = cc * di; c->conic_op[4 * i + 1] = -b * di; c->conic_op[4 * i + 2] = a * di; c->conic_op[4 * i + 3] = opac[i];
        c->cov2d[3 * i] = a; c->cov2d[3 * i + 1] = b; c->cov2d[3 * i + 2] = cc;
        int *rc = c->rect + 4 * i; rc[0] = x0; rc[1] = y0; rc[2] = x1; rc[3] = y1;
        for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) count[y * gx + x]++;
    }
    /* ---- binning: per tile, ascending (real bits of depth, id) ---- */
    for (int t = 0; t < tiles; t++) c->range[t + 1] = c->range[t] + count[t];
    c->R = c->range[tiles];
    c->list = (int *)malloc((size_t)(c->R ? c->R : 1) * 4);
    kv_t *kv = (kv_t *)malloc((size_t)(c->R ? c->R : 1) * sizeof(kv_t));
    memset(count, 0, (size_t)tiles * 4);
    for (int i = 0; i < P; i++) {
        if (c->radii[i] <= 0) continue;
        const int *rc = c->rect + 4 * i;
        uint32_t key; { const float d32 = (float)c->depth[i]; memcpy(&key, &d32, 4); }
        for (int y = rc[1]; y < rc[3]; y++) for (int x = rc[0]; x < rc[2]; x++) {
            int t = y * gx + x; kv_t *e = kv + c->range[t] + count[t]++;
            e->key = key; e->id = i;
        }
    }
    #pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < tiles; t++) {
        int n = c->range[t + 1] - c->range[t];
        if (n > 1) qsort(kv + c->range[t], n, sizeof(kv_t), kv_cmp);
        for (int k = 0; k < n; k++) c->list[c->range[t] + k] = kv[c->range[t] + k].id;
    }
    free(kv); free(count);

    /* ---- composite (Appendix A "Forward composite (K6)") ---- */
    int64_t pairs = 0;
    #pragma omp parallel for schedule(dynamic, 4) reduction(+ : pairs)
    for (int t = 0; t < tiles; t++) {
        int tx = t % gx, ty = t / gx, s = c->range[t], e = c->range[t + 1];
        for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
            int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px >= W || py >= H) continue;
            real T = 1.f, D = 0.f, Cc[MAXC] = {0};
            int contributor = 0, last = 0;
            for (int k = s; k < e; k++) {
                int id = c->list[k]; contributor++;
                real dx = c->xy[2 * id] - (real)px, dy = c->xy[2 * id + 1] - (real)py;
                const real *co = c->conic_op + 4 * id;
                real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.f) continue;
                real alpha = R_MIN(0.99f, co[3] * R_EXP(power));
                if (alpha < 1.f / 255.f) continue;
                real test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) break;
                for (int ch = 0; ch < C; ch++) Cc[ch] += colors[(size_t)id * C + ch] * alpha * T;
                D += c->depth[id] * alpha * T;
                T = test_T; last = contributor; pairs++;
            }
            size_t pix = (size_t)py * W + px;
            c->final_T[pix] = T; c->n_contrib[pix] = last;
            for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * W * H + pix] = Cc[ch] + T * bg[ch];
            out_depth[pix] = D;
        }
    }
    c->pairs = pairs;
    return 0;
}

/* Classifier of float32 decision flips (tests/util.py: explain_outliers).  Replays the composite of the last forward and, for every
 * pixel, looks at each DECISION the loop takes -- `power > 0`, `alpha < 1/255`, `T (1 - alpha) < 1e-4`, and the order of two
 * consecutive contributors whose depths agree to `tie_tol` (relative) -- and asks how close it was: |lhs - rhs| / rhs (for `power`:
 * relative to the magnitude of its terms).  A decision within `tol` may fall on the other side in another correct float32
 * evaluation (projected centres of ~1e3 px carry ~1e-4 px of float32 rounding, which moves `power` by ~1e-4 absolute, i.e. alpha and T
 * by ~1e-4 relative).  For every such decision the function adds, per output channel, a bound on what taking it the other way can
 * change in the pixel:
 *     alpha threshold / power sign:  alpha T (|c_k| + cmax)       (the contribution itself + the (1 - alpha) factor on all later ones)
 *     termination:                   T cmax                       (everything that could still be added)
 *     depth tie (j before k):        alpha_j alpha_k T_j |c_j - c_k|
 * with cmax the largest |colour| of the tile's list.  bound: [(C + 1), H, W] (channel C is the depth output); margin: [H, W], the
 * smallest relative margin of any decision of the pixel (reported, to show how close the flipped decisions were).  Meant for the
 * float64 build: margins measured there hold for every float32 evaluation. */
void ref_flip_bounds(const ref_ctx *c, const real *colors, real tol, real tie_tol, real *bound, real *margin, real ulps, real *noise)
{
    const int C = c->C, W = c->W, H = c->H, gx = c->gx, tiles = c->gx * c->gy;
    const size_t HW = (size_t)W * H;
    #pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles; t++) {
        int tx = t % gx, ty = t / gx, s = c->range[t], e = c->range[t + 1];
        real cmax[MAXC + 1] = {0};
        for (int k = s; k < e; k++) {
            int id = c->list[k];
            for (int ch = 0; ch < C; ch++) { real v = colors[(size_t)id * C + ch]; v = v < 0 ? -v : v; if (v > cmax[ch]) cmax[ch] = v; }
            if (c->depth[id] > cmax[C]) cmax[C] = c->depth[id];
        }
        for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
            int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px >= W || py >= H) continue;
            real T = 1.f, mmin = 1e30f, b[MAXC + 1] = {0};
            int prev = -1; real prev_alpha = 0.f, prev_T = 0.f;
            for (int k = s; k < e; k++) {
                int id = c->list[k];
                real dx = c->xy[2 * id] - (real)px, dy = c->xy[2 * id + 1] - (real)py;
                const real *co = c->conic_op + 4 * id;
                real t0 = 0.5f * co[0] * dx * dx, t1 = 0.5f * co[2] * dy * dy, t2 = co[1] * dx * dy;
                real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                real mag = (t0 < 0 ? -t0 : t0) + (t1 < 0 ? -t1 : t1) + (t2 < 0 ? -t2 : t2) + 1e-30f;
                real mp = (power < 0 ? -power : power) / mag;
                real a_raw = co[3] * R_EXP(power > 0.f ? 0.f : power);
                real alpha = R_MIN(0.99f, a_raw);
                #define ADD_FLIP(scale_self, scale_rest) do { \
                    for (int ch = 0; ch < C; ch++) { real v = colors[(size_t)id * C + ch]; v = v < 0 ? -v : v; b[ch] += (scale_self) * v + (scale_rest) * cmax[ch]; } \
                    b[C] += (scale_self) * c->depth[id] + (scale_rest) * cmax[C]; } while (0)
                if (mp < tol && alpha >= 1.f / 255.f) { if (mp < mmin) mmin = mp; ADD_FLIP(alpha * T, alpha * T); }
                if (power > 0.f) continue;
                real ma = (alpha > 1.f / 255.f ? alpha - 1.f / 255.f : 1.f / 255.f - alpha) * 255.f;
                if (ma < mmin) mmin = ma;
                if (ma < tol) ADD_FLIP(alpha * T, alpha * T);
                if (alpha < 1.f / 255.f) continue;
                real test_T = T * (1.f - alpha);
                real mt = (test_T > 0.0001f ? test_T - 0.0001f : 0.0001f - test_T) * 10000.f;
                if (mt < mmin) mmin = mt;
                if (mt < tol) ADD_FLIP(0.f, T);
                if (test_T < 0.0001f) break;
                if (prev >= 0) {
                    real dd = c->depth[id] - c->depth[prev]; dd = dd < 0 ? -dd : dd;
                    real md = dd / c->depth[id];
                    if (md <= tie_tol) {
                        if (md < mmin) mmin = md;
                        real w = prev_alpha * alpha * prev_T;
                        for (int ch = 0; ch < C; ch++) { real v = colors[(size_t)id * C + ch] - colors[(size_t)prev * C + ch]; b[ch] += w * (v < 0 ? -v : v); }
                        b[C] += w * dd;
                    }
                }
                prev = id; prev_alpha = alpha; prev_T = T;
                T = test_T;
                #undef ADD_FLIP
            }
            size_t pix = (size_t)py * W + px;
            margin[pix] = mmin;
            for (int ch = 0; ch <= C; ch++) bound[(size_t)ch * HW + pix] = b[ch];
            if (noise) {
                /* Sensitivity to the float32 rounding of the projected CENTRES (no decision involved).  A centre at pixel coordinate
                 * ~1e3 is known to `ulps` x 2^-23 x 1e3 ~ 1e-4 px in float32 (projection: a quotient and an affine map; the fused
                 * path's in-kernel transform rounds differently from torch's); that moves power_k by |grad power| d ~ 1e-4 and alpha_k
                 * by as much relatively, and the pixel by  |dC/dalpha_k| alpha_k dpower_k  with  dC/dalpha_k = T_k c_k - after_k / (1 -
                 * alpha_k)  (what the backward pass calls dL/dalpha): weight moves between Gaussians of different colour / depth while
                 * the silhouette hardly changes.  noise[ch] = the sum over the pixel's contributors: how far two CORRECT float32
                 * evaluations may differ at this pixel without any decision flipping. */
                real tot[MAXC + 1] = {0};
                real T2 = 1.f;
                int stop = e;
                for (int k = s; k < e; k++) {               /* pass 1: the pixel's totals */
                    int id = c->list[k];
                    real dx = c->xy[2 * id] - (real)px, dy = c->xy[2 * id + 1] - (real)py;
                    const real *co = c->conic_op + 4 * id;
                    real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.f) continue;
                    real alpha = R_MIN(0.99f, co[3] * R_EXP(power));
                    if (alpha < 1.f / 255.f) continue;
                    real test_T = T2 * (1.f - alpha);
                    if (test_T < 0.0001f) { stop = k; break; }
                    for (int ch = 0; ch < C; ch++) tot[ch] += colors[(size_t)id * C + ch] * alpha * T2;
                    tot[C] += c->depth[id] * alpha * T2;
                    T2 = test_T;
                }
                real pre[MAXC + 1] = {0}, nz[MAXC + 1] = {0};
                T2 = 1.f;
                const real eps = ulps * 1.1920929e-7f;
                for (int k = s; k < stop; k++) {            /* pass 2: every contributor's share */
                    int id = c->list[k];
                    real dx = c->xy[2 * id] - (real)px, dy = c->xy[2 * id + 1] - (real)py;
                    const real *co = c->conic_op + 4 * id;
                    real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.f) continue;
                    real a_raw = co[3] * R_EXP(power);
                    real alpha = R_MIN(0.99f, a_raw);
                    if (alpha < 1.f / 255.f) continue;
                    real gxp = co[0] * dx + co[1] * dy, gyp = co[2] * dy + co[1] * dx;
                    real ax = c->xy[2 * id], ay = c->xy[2 * id + 1];
                    ax = ax < 0 ? -ax : ax; ay = ay < 0 ? -ay : ay;
                    real dpow = (gxp < 0 ? -gxp : gxp) * eps * R_MAX(ax, 1.f) + (gyp < 0 ? -gyp : gyp) * eps * R_MAX(ay, 1.f);
                    real dalpha = a_raw < 0.99f ? alpha * dpow : 0.f;
                    for (int ch = 0; ch <= C; ch++) {
                        real col = ch < C ? colors[(size_t)id * C + ch] : c->depth[id];
                        real own = col * alpha * T2;
                        real after = tot[ch] - pre[ch] - own;
                        real dC = T2 * col - after / (1.f - alpha);
                        nz[ch] += (dC < 0 ? -dC : dC) * dalpha;
                        pre[ch] += own;
                    }
                    T2 *= (1.f - alpha);
                }
                for (int ch = 0; ch <= C; ch++) noise[(size_t)ch * HW + pix] = nz[ch];
            }
        }
    }
    /* Per-GAUSSIAN decisions: the tile rectangle.  radius = ceil(3 sqrt(lambda)) and the rectangle's edges (int)((p -+ radius [+ 15]) /
     * 16) are step functions of float32 quantities: a centre within `geo` pixels of the value where an edge moves (or 3 sqrt(lambda)
     * within `geo` of an integer) puts the Gaussian into one more / one fewer row or column of tiles in another float32 evaluation,
     * i.e. adds / removes it at EVERY pixel of those tiles it reaches.  For the pixels of the tiles between the smallest and the
     * largest rectangle the decision allows: bound += alpha (|c| + cmax_global)  (T <= 1). */
    {
        const real geo = 2e-3f;
        real gmax[MAXC + 1] = {0};
        for (int i = 0; i < c->P; i++) {
            if (c->radii[i] <= 0) continue;
            for (int ch = 0; ch < C; ch++) { real v = colors[(size_t)i * C + ch]; v = v < 0 ? -v : v; if (v > gmax[ch]) gmax[ch] = v; }
            if (c->depth[i] > gmax[C]) gmax[C] = c->depth[i];
        }
        for (int i = 0; i < c->P; i++) {
            if (c->radii[i] <= 0) continue;
            const real px = c->xy[2 * i], py = c->xy[2 * i + 1];
            const real a = c->cov2d[3 * i], bb = c->cov2d[3 * i + 1], cc = c->cov2d[3 * i + 2];
            const real det = a * cc - bb * bb, mid = 0.5f * (a + cc);
            const real lam = mid + R_SQRT(R_MAX(0.1f, mid * mid - det));
            const real r3 = 3.f * R_SQRT(lam), rad = (real)c->radii[i];
            const real r_lo = (rad - r3 > 1.f - geo) ? rad - 1.f : rad;       /* 3 sqrt(lambda) just above an integer: ceil may give one less */
            const real r_hi = (rad - r3 < geo) ? rad + 1.f : rad;             /* just below: one more */
            int lo[4], hi[4];
            lo[0] = (int)((px - r_lo + geo) / TILE); hi[0] = (int)((px - r_hi - geo) / TILE);
            lo[1] = (int)((py - r_lo + geo) / TILE); hi[1] = (int)((py - r_hi - geo) / TILE);
            lo[2] = (int)((px + r_lo - geo + TILE - 1) / TILE); hi[2] = (int)((px + r_hi + geo + TILE - 1) / TILE);
            lo[3] = (int)((py + r_lo - geo + TILE - 1) / TILE); hi[3] = (int)((py + r_hi + geo + TILE - 1) / TILE);
            const int lim[4] = {gx, c->gy, gx, c->gy};
            int same = 1;
            for (int k = 0; k < 4; k++) {
                lo[k] = lo[k] < 0 ? 0 : (lo[k] > lim[k] ? lim[k] : lo[k]);
                hi[k] = hi[k] < 0 ? 0 : (hi[k] > lim[k] ? lim[k] : hi[k]);
                if (lo[k] != hi[k]) same = 0;
            }
            if (same) continue;
            const real *co = c->conic_op + 4 * i;
            for (int tyy = hi[1]; tyy < hi[3]; tyy++) for (int txx = hi[0]; txx < hi[2]; txx++) {
                if (txx >= lo[0] && txx < lo[2] && tyy >= lo[1] && tyy < lo[3]) continue;       /* in the smallest rectangle: certain */
                for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
                    int qx = txx * TILE + lx, qy = tyy * TILE + ly;
                    if (qx >= W || qy >= H) continue;
                    real dx = px - (real)qx, dy = py - (real)qy;
                    real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.f) continue;
                    real alpha = R_MIN(0.99f, co[3] * R_EXP(power));
                    if (alpha < (1.f - tol) / 255.f) continue;
                    size_t pix = (size_t)qy * W + qx;
                    for (int ch = 0; ch < C; ch++) { real v = colors[(size_t)i * C + ch]; v = v < 0 ? -v : v; bound[(size_t)ch * HW + pix] += alpha * (v + gmax[ch]); }
                    bound[(size_t)C * HW + pix] += alpha * (c->depth[i] + gmax[C]);
                    if (margin[pix] > geo) margin[pix] = geo;
                }
            }
        }
    }
}

/* Backward.  dL_dpix: [C,H,W].  Outputs (all overwritten):
 * dmeans3D[P,3] dmeans2D[P,3] dcolors[P,C] dopac[P] dscales[P,3] drot[P,4] dcov3D[P,6]
 * (dscales/drot are skipped when scales==NULL, i.e. cov3D_precomp was used). */
int ref_backward(const ref_ctx *c, const real *bg, const real *means3D, const real *colors,
                 const real *scales, real mod, const real *rot,
                 const real *view, const real *proj, real tanfovx, real tanfovy,
                 const real *dL_dpix,
                 real *dmeans3D, real *dmeans2D, real *dcolors, real *dopac,
                 real *dscales, real *drot, real *dcov3D)
{
    const int P = c->P, C = c->C, W = c->W, H = c->H, gx = c->gx, tiles = c->gx * c->gy;
    /* double accumulators keep the OpenMP result order-independent to real precision */
    double *acc = (double *)calloc((size_t)P * (6 + C), sizeof(double));
    const int NA = 6 + C; /* 0,1 mean2D(ndc)  2,3,4 conic (true derivative)  5 opacity  6.. colour */

    #pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles; t++) {
        int tx = t % gx, ty = t / gx, s = c->range[t];
        for (int ly = 0; ly < TILE; ly++) for (int lx = 0; lx < TILE; lx++) {
            int px = tx * TILE + lx, py = ty * TILE + ly;
            if (px >= W || py >= H) continue;
            size_t pix = (size_t)py * W + px;
            const real T_final = c->final_T[pix];
            real T = T_final;
            int last = c->n_contrib[pix];
            real dpix[MAXC], accum[MAXC] = {0}, last_col[MAXC] = {0}, last_alpha = 0.f, bgdot = 0.f;
            for (int ch = 0; ch < C; ch++) { dpix[ch] = dL_dpix[(size_t)ch * W * H + pix]; bgdot += bg[ch] * dpix[ch]; }
            for (int k = s + last - 1; k >= s; k--) {
                int id = c->list[k];
                real dx = c->xy[2 * id] - (real)px, dy = c->xy[2 * id + 1] - (real)py;
                const real *co = c->conic_op + 4 * id;
                real power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.f) continue;
                real G = R_EXP(power);
                real alpha = R_MIN(0.99f, co[3] * G);
                if (alpha < 1.f / 255.f) continue;
                T = T / (1.f - alpha);
                real w = alpha * T, dL_dalpha = 0.f;
                double *a = acc + (size_t)id * NA;
                for (int ch = 0; ch < C; ch++) {
                    real col = colors[(size_t)id * C + ch];
                    accum[ch] = last_alpha * last_col[ch] + (1.f - last_alpha) * accum[ch];
                    last_col[ch] = col;
                    dL_dalpha += (col - accum[ch]) * dpix[ch];
                    #pragma omp atomic
                    a[6 + ch] += (double)(w * dpix[ch]);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bgdot;
                real dL_dG = co[3] * dL_dalpha;
                real gdx = G * dx, gdy = G * dy;
                real dG_ddx = -gdx * co[0] - gdy * co[1], dG_ddy = -gdy * co[2] - gdx * co[1];
                #pragma omp atomic
                a[0] += (double)(dL_dG * dG_ddx * 0.5f * W);
                #pragma omp atomic
                a[1] += (double)(dL_dG * dG_ddy * 0.5f * H);
                #pragma omp atomic
                a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                #pragma omp atomic
                a[3] += (double)(-gdx * dy * dL_dG);
                #pragma omp atomic
                a[4] += (double)(-0.5f * gdy * dy * dL_dG);
                #pragma omp atomic
                a[5] += (double)(G * dL_dalpha);
            }
        }
    }

    /* ---- per-Gaussian backward (Appendix A "Backward preprocess (K8+K9)") ---- */
    const real fx = W / (2.f * tanfovx), fy = H / (2.f * tanfovy);
    for (int i = 0; i < P; i++) {
        real *gm = dmeans3D + 3 * i; gm[0] = gm[1] = gm[2] = 0.f;
        dmeans2D[3 * i] = dmeans2D[3 * i + 1] = dmeans2D[3 * i + 2] = 0.f;
        for (int ch = 0; ch < C; ch++) dcolors[(size_t)i * C + ch] = 0.f;
        dopac[i] = 0.f;
        if (dscales) { for (int k = 0; k < 3; k++) dscales[3 * i + k] = 0.f; for (int k = 0; k < 4; k++) drot[4 * i + k] = 0.f; }
        if (dcov3D) for (int k = 0; k < 6; k++) dcov3D[6 * i + k] = 0.f;
        if (c->radii[i] <= 0) continue;
        const double *a = acc + (size_t)i * NA;
        real g2x = (real)a[0], g2y = (real)a[1], gcx = (real)a[2], gcy = (real)a[3], gcz = (real)a[4];
        dmeans2D[3 * i] = g2x; dmeans2D[3 * i + 1] = g2y;
        dopac[i] = (real)a[5];
        for (int ch = 0; ch < C; ch++) dcolors[(size_t)i * C + ch] = (real)a[6 + ch];

        const real *p = means3D + 3 * i;
        real tv[3];
        for (int r = 0; r < 3; r++) tv[r] = m4(view, r, 0) * p[0] + m4(view, r, 1) * p[1] + m4(view, r, 2) * p[2] + m4(view, r, 3);
        real limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
        real txtz = tv[0] / tv[2], tytz = tv[1] / tv[2];
        real tx = R_MIN(limx, R_MAX(-limx, txtz)) * tv[2], ty = R_MIN(limy, R_MAX(-limy, tytz)) * tv[2], tz = tv[2];
        real xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f, ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        real J[2][3] = {{fx / tz, 0.f, -(fx * tx) / (tz * tz)}, {0.f, fy / tz, -(fy * ty) / (tz * tz)}};
        real T[2][3];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++)
            T[r][k] = J[r][0] * m4(view, 0, k) + J[r][1] * m4(view, 1, k) + J[r][2] * m4(view, 2, k);
        const real *S6 = c->cov3d + 6 * i;
        real Sg[3][3] = {{S6[0], S6[1], S6[2]}, {S6[1], S6[3], S6[4]}, {S6[2], S6[4], S6[5]}};
        real ca = c->cov2d[3 * i], cb = c->cov2d[3 * i + 1], cc = c->cov2d[3 * i + 2];
        real det = ca * cc - cb * cb;
        real d2 = 1.f / (det * det + 0.0000001f);
        real dLa = d2 * (-cc * cc * gcx + cb * cc * gcy - cb * cb * gcz);
        real dLc = d2 * (-cb * cb * gcx + ca * cb * gcy - ca * ca * gcz);
        real dLb = d2 * (2.f * cb * cc * gcx - (det + 2.f * cb * cb) * gcy + 2.f * ca * cb * gcz);
        /* symmetric gradient of the 2x2 covariance */
        real G2[2][2] = {{dLa, 0.5f * dLb}, {0.5f * dLb, dLc}};
        /* dL/dSigma = T^T G2 T */
        real GT[2][3];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++) GT[r][k] = G2[r][0] * T[0][k] + G2[r][1] * T[1][k];
        real dS[3][3];
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) dS[r][k] = T[0][r] * GT[0][k] + T[1][r] * GT[1][k];
        real g6[6] = {dS[0][0], 2.f * dS[0][1], 2.f * dS[0][2], dS[1][1], 2.f * dS[1][2], dS[2][2]};
        if (dcov3D) memcpy(dcov3D + 6 * i, g6, 6 * sizeof(real));
        /* dL/dT = 2 G2 T Sigma */
        real TS[2][3], dT[2][3];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++) TS[r][k] = T[r][0] * Sg[0][k] + T[r][1] * Sg[1][k] + T[r][2] * Sg[2][k];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++) dT[r][k] = 2.f * (G2[r][0] * TS[0][k] + G2[r][1] * TS[1][k]);
        /* dL/dJ = dT W3^T */
        real dJ[2][3];
        for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++)
            dJ[r][k] = dT[r][0] * m4(view, k, 0) + dT[r][1] * m4(view, k, 1) + dT[r][2] * m4(view, k, 2);
        real iz = 1.f / tz, iz2 = iz * iz, iz3 = iz2 * iz;
        real dtx = xmul * -fx * iz2 * dJ[0][2];
        real dty = ymul * -fy * iz2 * dJ[1][2];
        real dtz = -fx * iz2 * dJ[0][0] - fy * iz2 * dJ[1][1] + 2.f * fx * tx * iz3 * dJ[0][2] + 2.f * fy * ty * iz3 * dJ[1][2];
        for (int k = 0; k < 3; k++) gm[k] = m4(view, 0, k) * dtx + m4(view, 1, k) * dty + m4(view, 2, k) * dtz;
        /* projection: NDC gradient -> centre */
        real hom[4];
        for (int r = 0; r < 4; r++) hom[r] = m4(proj, r, 0) * p[0] + m4(proj, r, 1) * p[1] + m4(proj, r, 2) * p[2] + m4(proj, r, 3);
        real pw = 1.f / (hom[3] + 0.0000001f);
        for (int k = 0; k < 3; k++) {
            real dndcx = m4(proj, 0, k) * pw - hom[0] * pw * pw * m4(proj, 3, k);
            real dndcy = m4(proj, 1, k) * pw - hom[1] * pw * pw * m4(proj, 3, k);
            gm[k] += dndcx * g2x + dndcy * g2y;
        }
        /* Sigma = M M^T, M = R diag(mod*s)  ->  scale and quaternion */
        if (dscales && scales) {
            real R[3][3], Gs[3][3] = {{g6[0], 0.5f * g6[1], 0.5f * g6[2]}, {0.5f * g6[1], g6[3], 0.5f * g6[4]}, {0.5f * g6[2], 0.5f * g6[4], g6[5]}};
            const real *q = rot + 4 * i; quat_rot(q, R);
            real sv[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
            real dM[3][3], A[3][3];
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) {
                real v = 0.f; for (int m = 0; m < 3; m++) v += Gs[r][m] * R[m][k] * sv[k];
                dM[r][k] = 2.f * v;
            }
            for (int k = 0; k < 3; k++) {
                dscales[3 * i + k] = (c->upstream_scale ? (real)1 : mod) * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
                for (int r = 0; r < 3; r++) A[r][k] = dM[r][k] * sv[k];
            }
            real r_ = q[0], x = q[1], y = q[2], z = q[3];
            drot[4 * i + 0] = 2.f * (-z * A[0][1] + y * A[0][2] + z * A[1][0] - x * A[1][2] - y * A[2][0] + x * A[2][1]);
            drot[4 * i + 1] = 2.f * (y * A[0][1] + z * A[0][2] + y * A[1][0] - 2.f * x * A[1][1] - r_ * A[1][2] + z * A[2][0] + r_ * A[2][1] - 2.f * x * A[2][2]);
            drot[4 * i + 2] = 2.f * (-2.f * y * A[0][0] + x * A[0][1] + r_ * A[0][2] + x * A[1][0] + z * A[1][2] - r_ * A[2][0] + z * A[2][1] - 2.f * y * A[2][2]);
            drot[4 * i + 3] = 2.f * (-2.f * z * A[0][0] - r_ * A[0][1] + x * A[0][2] + r_ * A[1][0] - 2.f * z * A[1][1] + y * A[1][2] + x * A[2][0] + y * A[2][1]);
        }
    }
    free(acc);
    return 0;
}

/* markVisible of the boundary */
void ref_mark_visible(int P, const real *means3D, const real *view, unsigned char *present) {
    for (int i = 0; i < P; i++) {
        const real *p = means3D + 3 * i;
        real z = m4(view, 2, 0) * p[0] + m4(view, 2, 1) * p[1] + m4(view, 2, 2) * p[2] + m4(view, 2, 3);
        present[i] = z > 0.2f;
    }
}
