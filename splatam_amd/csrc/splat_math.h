// splat_math.h -- per-Gaussian arithmetic of the rasterizer (projection, EWA
// covariance, tile rect and their hand-derived adjoints), written once as
// host/device inline functions: the HIP kernels call them per lane, and
// tests/test_host_math.py compiles the very same header with g++ to check the
// arithmetic against the oracle on a machine without a GPU.
//
// Restates the published algorithm summarised in SURVEY.md Appendix A (the
// callee of /root/reference/scripts/splatam.py:249,253); matrix layout as in
// /root/reference/utils/recon_helpers.py:8-13, quaternion polynomial as in
// /root/reference/utils/slam_external.py:33-41.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SPLAT_HD __host__ __device__ __forceinline__
#else
#define SPLAT_HD inline
#endif

namespace splat {

constexpr float kNearZ = 0.2f;        // near cull: view-space z <= 0.2 is dropped
constexpr float kDilation = 0.3f;     // low-pass added to the 2D covariance diagonal
constexpr float kFovGuard = 1.3f;     // frustum clamp of x/z, y/z before the Jacobian
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTStop = 0.0001f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kTile = 16;

// Camera constants shared by every Gaussian of a launch.  m(r,c) = flat[c*4+r].
struct CamConst {
    float view[16];
    float proj[16];
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy;
};

SPLAT_HD float mat(const float *m, int r, int c) { return m[c * 4 + r]; }

SPLAT_HD void init_cam(CamConst &c, const float *view, const float *proj, int W, int H,
                       float tanfovx, float tanfovy, float scale_modifier) {
    for (int i = 0; i < 16; ++i) { c.view[i] = view[i]; c.proj[i] = proj[i]; }
    c.tanfovx = tanfovx; c.tanfovy = tanfovy;
    c.focal_x = W / (2.0f * tanfovx); c.focal_y = H / (2.0f * tanfovy);
    c.scale_modifier = scale_modifier;
    c.W = W; c.H = H; c.gx = (W + kTile - 1) / kTile; c.gy = (H + kTile - 1) / kTile;
}

// Rotation matrix (row-major R[3*r+c]) of an un-normalised quaternion (r,x,y,z).
SPLAT_HD void quat_to_rot(const float *q, float *R) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = M M^T with M = R diag(mod*s); six upper-triangular entries (00,01,02,11,12,22).
SPLAT_HD void cov3d_from_scale_rot(const float *s, float mod, const float *q, float *S6) {
    float R[9];
    quat_to_rot(q, R);
    float M[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[3 * i + j] = R[3 * i + j] * (mod * s[j]);
    S6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    S6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    S6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    S6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    S6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    S6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// What the forward keeps per Gaussian.
struct Projected {
    float depth;        // view-space z
    float px, py;       // pixel centre
    float conic[3];     // inverse of the dilated 2D covariance (xx, xy, yy)
    float cov2d[3];     // (a, b, c) after dilation
    int radius;         // ceil(3 sigma_max); 0 when culled
    int x0, y0, x1, y1; // tile rect, max exclusive
    int lx0, ly0, lx1, ly1; // the part of it group binning files (live_tile_rect below; the whole rect until a caller tightens it)
};

// EWA linearisation shared by forward and backward: the 2x3 matrix T = J W
// (J = perspective Jacobian at the guard-band clamped view-space centre, W =
// rotation of the view matrix), plus the pieces the adjoint needs.
struct Ewa {
    float T[6];             // row-major 2x3
    float tx, ty, tz;       // clamped view-space centre
    float xmul, ymul;       // 0 where the guard-band clamp is active
};

SPLAT_HD void ewa_setup(const CamConst &c, const float *tv, Ewa &e) {
    const float limx = kFovGuard * c.tanfovx, limy = kFovGuard * c.tanfovy;
    const float txtz = tv[0] / tv[2], tytz = tv[1] / tv[2];
    e.tz = tv[2];
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * tv[2];
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * tv[2];
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float j00 = c.focal_x / e.tz, j02 = -(c.focal_x * e.tx) / (e.tz * e.tz);
    const float j11 = c.focal_y / e.tz, j12 = -(c.focal_y * e.ty) / (e.tz * e.tz);
    for (int k = 0; k < 3; ++k) {
        e.T[k] = j00 * mat(c.view, 0, k) + j02 * mat(c.view, 2, k);
        e.T[3 + k] = j11 * mat(c.view, 1, k) + j12 * mat(c.view, 2, k);
    }
}

SPLAT_HD void view_transform(const CamConst &c, const float *p, float *tv) {
    for (int r = 0; r < 3; ++r)
        tv[r] = mat(c.view, r, 0) * p[0] + mat(c.view, r, 1) * p[1] + mat(c.view, r, 2) * p[2] + mat(c.view, r, 3);
}

// cov2d = T Sigma T^T (+ dilation).  TS (2x3) is returned for the adjoint.
SPLAT_HD void ewa_cov2d(const Ewa &e, const float *S6, float *abc, float *TS) {
    const float S[9] = {S6[0], S6[1], S6[2], S6[1], S6[3], S6[4], S6[2], S6[4], S6[5]};
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 3; ++k)
            TS[3 * r + k] = e.T[3 * r] * S[k] + e.T[3 * r + 1] * S[3 + k] + e.T[3 * r + 2] * S[6 + k];
    abc[0] = TS[0] * e.T[0] + TS[1] * e.T[1] + TS[2] * e.T[2] + kDilation;
    abc[1] = TS[0] * e.T[3] + TS[1] * e.T[4] + TS[2] * e.T[5];
    abc[2] = TS[3] * e.T[3] + TS[4] * e.T[4] + TS[5] * e.T[5] + kDilation;
}

SPLAT_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Forward of one Gaussian.  Returns false (radius 0) when culled.
SPLAT_HD bool project_gaussian(const CamConst &c, const float *p, const float *S6, Projected &o) {
    o.radius = 0; o.x0 = o.y0 = o.x1 = o.y1 = 0;
    o.lx0 = o.ly0 = o.lx1 = o.ly1 = 0;
    o.depth = 0.f; o.px = o.py = 0.f;
    o.conic[0] = o.conic[1] = o.conic[2] = 0.f;
    o.cov2d[0] = o.cov2d[1] = o.cov2d[2] = 0.f;
    float tv[3];
    view_transform(c, p, tv);
    if (!(tv[2] > kNearZ)) return false;
    float hom[4];
    for (int r = 0; r < 4; ++r)
        hom[r] = mat(c.proj, r, 0) * p[0] + mat(c.proj, r, 1) * p[1] + mat(c.proj, r, 2) * p[2] + mat(c.proj, r, 3);
    const float pw = 1.0f / (hom[3] + 0.0000001f);
    const float ndcx = hom[0] * pw, ndcy = hom[1] * pw;
    Ewa e;
    ewa_setup(c, tv, e);
    float abc[3], TS[6];
    ewa_cov2d(e, S6, abc, TS);
    const float det = abc[0] * abc[2] - abc[1] * abc[1];
    if (det == 0.0f) return false;
    const float di = 1.0f / det;
    const float mid = 0.5f * (abc[0] + abc[2]);
    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lam = fmaxf(mid + disc, mid - disc);
    const float radius = ceilf(3.0f * sqrtf(lam));
    const float px = ((ndcx + 1.0f) * c.W - 1.0f) * 0.5f;
    const float py = ((ndcy + 1.0f) * c.H - 1.0f) * 0.5f;
    const int x0 = clampi((int)((px - radius) / kTile), 0, c.gx);
    const int y0 = clampi((int)((py - radius) / kTile), 0, c.gy);
    const int x1 = clampi((int)((px + radius + kTile - 1) / kTile), 0, c.gx);
    const int y1 = clampi((int)((py + radius + kTile - 1) / kTile), 0, c.gy);
    if ((x1 - x0) * (y1 - y0) == 0) return false;
    o.depth = tv[2]; o.px = px; o.py = py;
    o.conic[0] = abc[2] * di; o.conic[1] = -abc[1] * di; o.conic[2] = abc[0] * di;
    o.cov2d[0] = abc[0]; o.cov2d[1] = abc[1]; o.cov2d[2] = abc[2];
    o.radius = (int)radius; o.x0 = x0; o.y0 = y0; o.x1 = x1; o.y1 = y1;
    o.lx0 = x0; o.ly0 = y0; o.lx1 = x1; o.ly1 = y1;
    return true;
}

// The part of a Gaussian's tile rectangle [x0, x1) x [y0, y1) whose tiles can hold a pixel with alpha >= 1/255 (the composites' blend
// threshold, Appendix A "skip if alpha < 1/255"): the live region {(p - mu)^T Q (p - mu) <= 2 ln(255 o)} has the half extents
// sqrt(2 ln(255 o) Q^-1_ii); a tile of pixel centres 16 t .. 16 t + 15 outside that box blends nothing of this Gaussian.  The reference's
// rectangle comes from ceil(3 sigma_max) and holds 12-13 % such tiles at the SplaTAM workloads (scripts/fwd_balance_stats.py): list
// entries that are sorted, staged and culled for nothing.  Used where the lists are this library's own business (GROUP BINNING: the
// forward composite builds, sorts and publishes the list, the backward composite replays it); the exact path keeps the reference's
// lists entry for entry.  Conservative by the same margins as the composites' staging cull (render.hip gather()); NaN geometry or
// opacity leaves the rectangle alone (the NaN must reach the image as it does in the reference).
SPLAT_HD void live_tile_rect(const float *conic, float opacity, float px, float py, int &x0, int &y0, int &x1, int &y1) {
    if (opacity < kAlphaMin) { x1 = x0; y1 = y0; return; }       // alpha <= opacity < 1/255 at every pixel: nothing passes the alpha test
    if (!(opacity >= kAlphaMin)) return;                         // NaN
    // (the threshold itself is decided on the opacity, as the composite decides it; 255 o may round below 1 for o == 1/255)
    const float tau2 = fmaxf(2.0f * logf(255.0f * opacity), 0.f);
    const float det = conic[0] * conic[2] - conic[1] * conic[1];
    const float hx = sqrtf(tau2 * conic[2] / det) * 1.00001f + 0.01f;
    const float hy = sqrtf(tau2 * conic[0] / det) * 1.00001f + 0.01f;
    if (!(hx == hx) || !(hy == hy)) return;
    const float inv = 1.0f / (float)kTile, big = 1.0e6f;
    const int tx0 = (int)ceilf(fminf(fmaxf((px - hx - (float)(kTile - 1)) * inv, -1.f), big));
    const int ty0 = (int)ceilf(fminf(fmaxf((py - hy - (float)(kTile - 1)) * inv, -1.f), big));
    const int tx1 = (int)floorf(fminf(fmaxf((px + hx) * inv, -2.f), big)) + 1;
    const int ty1 = (int)floorf(fminf(fmaxf((py + hy) * inv, -2.f), big)) + 1;
    x0 = x0 > tx0 ? x0 : tx0;
    y0 = y0 > ty0 ? y0 : ty0;
    x1 = x1 < tx1 ? x1 : tx1;
    y1 = y1 < ty1 ? y1 : ty1;
    if (x1 < x0) x1 = x0;
    if (y1 < y0) y1 = y0;
}

// Adjoint of project_gaussian for one visible Gaussian.
//   g_ndc  : dL/d(NDC x,y) of the projected centre
//   g_conic: TRUE derivatives dL/d(conic xx, xy, yy)
// Outputs dL/dmean (3) and dL/dSigma as the 6-vector in which an off-diagonal
// entry carries the gradient of BOTH symmetric positions.
SPLAT_HD void project_gaussian_backward(const CamConst &c, const float *p, const float *S6,
                                        const float *g_ndc, const float *g_conic,
                                        float *dmean, float *dS6) {
    float tv[3];
    view_transform(c, p, tv);
    Ewa e;
    ewa_setup(c, tv, e);
    float abc[3], TS[6];
    ewa_cov2d(e, S6, abc, TS);
    const float a = abc[0], b = abc[1], cc = abc[2];
    const float det = a * cc - b * b;
    const float d2 = 1.0f / (det * det + 0.0000001f);
    // conic = (c, -b, a) / det
    const float dLa = d2 * (-cc * cc * g_conic[0] + b * cc * g_conic[1] - b * b * g_conic[2]);
    const float dLc = d2 * (-b * b * g_conic[0] + a * b * g_conic[1] - a * a * g_conic[2]);
    const float dLb = d2 * (2.f * b * cc * g_conic[0] - (det + 2.f * b * b) * g_conic[1] + 2.f * a * b * g_conic[2]);
    const float G2[4] = {dLa, 0.5f * dLb, 0.5f * dLb, dLc};       // symmetric 2x2 gradient
    // dL/dSigma = T^T G2 T
    float GT[6];
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 3; ++k) GT[3 * r + k] = G2[2 * r] * e.T[k] + G2[2 * r + 1] * e.T[3 + k];
    float dS[9];
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) dS[3 * r + k] = e.T[r] * GT[k] + e.T[3 + r] * GT[3 + k];
    dS6[0] = dS[0]; dS6[1] = 2.f * dS[1]; dS6[2] = 2.f * dS[2];
    dS6[3] = dS[4]; dS6[4] = 2.f * dS[5]; dS6[5] = dS[8];
    // dL/dT = 2 G2 (T Sigma);  dL/dJ = dL/dT W^T  (only J00, J02, J11, J12 are live)
    float dT[6];
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 3; ++k) dT[3 * r + k] = 2.f * (G2[2 * r] * TS[k] + G2[2 * r + 1] * TS[3 + k]);
    const float dJ00 = dT[0] * mat(c.view, 0, 0) + dT[1] * mat(c.view, 0, 1) + dT[2] * mat(c.view, 0, 2);
    const float dJ02 = dT[0] * mat(c.view, 2, 0) + dT[1] * mat(c.view, 2, 1) + dT[2] * mat(c.view, 2, 2);
    const float dJ11 = dT[3] * mat(c.view, 1, 0) + dT[4] * mat(c.view, 1, 1) + dT[5] * mat(c.view, 1, 2);
    const float dJ12 = dT[3] * mat(c.view, 2, 0) + dT[4] * mat(c.view, 2, 1) + dT[5] * mat(c.view, 2, 2);
    const float iz = 1.0f / e.tz, iz2 = iz * iz, iz3 = iz2 * iz;
    const float dtx = e.xmul * -c.focal_x * iz2 * dJ02;
    const float dty = e.ymul * -c.focal_y * iz2 * dJ12;
    const float dtz = -c.focal_x * iz2 * dJ00 - c.focal_y * iz2 * dJ11
                      + 2.f * c.focal_x * e.tx * iz3 * dJ02 + 2.f * c.focal_y * e.ty * iz3 * dJ12;
    for (int k = 0; k < 3; ++k)
        dmean[k] = mat(c.view, 0, k) * dtx + mat(c.view, 1, k) * dty + mat(c.view, 2, k) * dtz;
    // perspective divide
    float hom[4];
    for (int r = 0; r < 4; ++r)
        hom[r] = mat(c.proj, r, 0) * p[0] + mat(c.proj, r, 1) * p[1] + mat(c.proj, r, 2) * p[2] + mat(c.proj, r, 3);
    const float pw = 1.0f / (hom[3] + 0.0000001f);
    const float mx = hom[0] * pw * pw, my = hom[1] * pw * pw;
    for (int k = 0; k < 3; ++k) {
        const float w3 = mat(c.proj, 3, k);
        dmean[k] += (mat(c.proj, 0, k) * pw - w3 * mx) * g_ndc[0] + (mat(c.proj, 1, k) * pw - w3 * my) * g_ndc[1];
    }
}

// Adjoint of cov3d_from_scale_rot.  dscale is the gradient w.r.t. the scales the caller passed (it carries the factor `mod`);
// upstream_scale: WITHOUT that factor -- what the CUDA original hands out (its computeCov3D adjoint forms dL/d(mod * s) =
// dot(R^T[k], dL/dM^T[k]) and returns it as dL/dscale: SURVEY.md Appendix A, SplatGrads.flags SPLAT_GRADS_UPSTREAM_SCALE)
SPLAT_HD void cov3d_backward(const float *s, float mod, const float *q, const float *dS6,
                             float *dscale, float *dq, bool upstream_scale = false) {
    float R[9];
    quat_to_rot(q, R);
    const float sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
    const float G[9] = {dS6[0], 0.5f * dS6[1], 0.5f * dS6[2], 0.5f * dS6[1], dS6[3], 0.5f * dS6[4],
                        0.5f * dS6[2], 0.5f * dS6[4], dS6[5]};
    float A[9];   // dL/dR
    for (int k = 0; k < 3; ++k) {
        float col[3];
        for (int r = 0; r < 3; ++r)   // dL/dM[:,k] = 2 G R[:,k] s_k
            col[r] = 2.f * sv[k] * (G[3 * r] * R[k] + G[3 * r + 1] * R[3 + k] + G[3 * r + 2] * R[6 + k]);
        dscale[k] = (upstream_scale ? 1.0f : mod) * (col[0] * R[k] + col[1] * R[3 + k] + col[2] * R[6 + k]);
        for (int r = 0; r < 3; ++r) A[3 * r + k] = col[r] * sv[k];
    }
    const float r_ = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2.f * (-z * A[1] + y * A[2] + z * A[3] - x * A[5] - y * A[6] + x * A[7]);
    dq[1] = 2.f * (y * A[1] + z * A[2] + y * A[3] - 2.f * x * A[4] - r_ * A[5] + z * A[6] + r_ * A[7] - 2.f * x * A[8]);
    dq[2] = 2.f * (-2.f * y * A[0] + x * A[1] + r_ * A[2] + x * A[3] + z * A[5] - r_ * A[6] + z * A[7] - 2.f * y * A[8]);
    dq[3] = 2.f * (-2.f * z * A[0] - r_ * A[1] + x * A[2] + r_ * A[3] - 2.f * z * A[4] + y * A[5] + x * A[6] + y * A[7]);
}

// ---- spherical harmonics (degrees 0..3), colour = max(0, SH(dir) + 0.5) ----
constexpr float kSH0 = 0.28209479177387814f;
constexpr float kSH1 = 0.4886025119029199f;
constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                           -1.0925484305920792f, 0.5462742152960396f};
constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                           -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// basis[k] for k < (deg+1)^2 at unit direction d; optionally d(basis)/d(dir) in db[3][16].
SPLAT_HD void sh_basis(int deg, const float *d, float *basis, float (*db)[16]) {
    const float x = d[0], y = d[1], z = d[2];
    basis[0] = kSH0;
    if (db) { db[0][0] = db[1][0] = db[2][0] = 0.f; }
    if (deg < 1) return;
    basis[1] = -kSH1 * y; basis[2] = kSH1 * z; basis[3] = -kSH1 * x;
    if (db) {
        db[0][1] = 0.f; db[1][1] = -kSH1; db[2][1] = 0.f;
        db[0][2] = 0.f; db[1][2] = 0.f; db[2][2] = kSH1;
        db[0][3] = -kSH1; db[1][3] = 0.f; db[2][3] = 0.f;
    }
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    basis[4] = kSH2[0] * xy; basis[5] = kSH2[1] * yz; basis[6] = kSH2[2] * (2.f * zz - xx - yy);
    basis[7] = kSH2[3] * xz; basis[8] = kSH2[4] * (xx - yy);
    if (db) {
        db[0][4] = kSH2[0] * y; db[1][4] = kSH2[0] * x; db[2][4] = 0.f;
        db[0][5] = 0.f; db[1][5] = kSH2[1] * z; db[2][5] = kSH2[1] * y;
        db[0][6] = kSH2[2] * -2.f * x; db[1][6] = kSH2[2] * -2.f * y; db[2][6] = kSH2[2] * 4.f * z;
        db[0][7] = kSH2[3] * z; db[1][7] = 0.f; db[2][7] = kSH2[3] * x;
        db[0][8] = kSH2[4] * 2.f * x; db[1][8] = kSH2[4] * -2.f * y; db[2][8] = 0.f;
    }
    if (deg < 3) return;
    basis[9] = kSH3[0] * y * (3.f * xx - yy);
    basis[10] = kSH3[1] * xy * z;
    basis[11] = kSH3[2] * y * (4.f * zz - xx - yy);
    basis[12] = kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    basis[13] = kSH3[4] * x * (4.f * zz - xx - yy);
    basis[14] = kSH3[5] * z * (xx - yy);
    basis[15] = kSH3[6] * x * (xx - 3.f * yy);
    if (db) {
        db[0][9] = kSH3[0] * 6.f * xy;            db[1][9] = kSH3[0] * (3.f * xx - 3.f * yy);   db[2][9] = 0.f;
        db[0][10] = kSH3[1] * yz;                 db[1][10] = kSH3[1] * xz;                     db[2][10] = kSH3[1] * xy;
        db[0][11] = kSH3[2] * -2.f * xy;          db[1][11] = kSH3[2] * (4.f * zz - xx - 3.f * yy); db[2][11] = kSH3[2] * 8.f * yz;
        db[0][12] = kSH3[3] * -6.f * xz;          db[1][12] = kSH3[3] * -6.f * yz;              db[2][12] = kSH3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
        db[0][13] = kSH3[4] * (4.f * zz - 3.f * xx - yy); db[1][13] = kSH3[4] * -2.f * xy;      db[2][13] = kSH3[4] * 8.f * xz;
        db[0][14] = kSH3[5] * 2.f * xz;           db[1][14] = kSH3[5] * -2.f * yz;              db[2][14] = kSH3[5] * (xx - yy);
        db[0][15] = kSH3[6] * (3.f * xx - 3.f * yy); db[1][15] = kSH3[6] * -6.f * xy;           db[2][15] = 0.f;
    }
}

}  // namespace splat
