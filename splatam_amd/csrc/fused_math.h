// fused_math.h -- per-Gaussian / per-pixel arithmetic of the fused SplaTAM iteration (the callers on either side
// of the rasterizer boundary), written once as host/device inline functions: fused.hip calls them per lane and
// tests/test_host_math.py compiles the very same header with g++ to check them against torch autograd of the
// reference-shaped Python (splatam_amd/slam.py, pinned to the reference by tests/golden/).
//
// Restates:
//   transform_to_frame                       /root/reference/utils/slam_helpers.py:252-304
//   transformed_params2rendervar             /root/reference/utils/slam_helpers.py:124-139
//   transformed_params2depthplussilhouette   /root/reference/utils/slam_helpers.py:234-249 (+ :196-213)
//   build_rotation                           /root/reference/utils/slam_external.py:25-42
//   quat_mult                                /root/reference/utils/slam_helpers.py:21-29
//   _ssim (per-pixel part)                   /root/reference/utils/slam_external.py:75-97
// and their hand-derived adjoints (what torch.autograd computes for the reference).
#pragma once

#include "splat_math.h"

namespace splat {

constexpr float kNormEps = 1e-12f;       // torch.nn.functional.normalize: v / max(|v|, eps)
constexpr float kSsimC1 = 0.01f * 0.01f;
constexpr float kSsimC2 = 0.03f * 0.03f;

// y = v / max(|v|, eps); returns 1 / max(|v|, eps)
SPLAT_HD float normalize4(const float *v, float *y) {
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    const float inv = 1.0f / fmaxf(n, kNormEps);
    for (int k = 0; k < 4; ++k) y[k] = v[k] * inv;
    return inv;
}

// adjoint of normalize4 at output y (= v * inv): dv = (g - y (y.g)) * inv   (|v| > eps)
SPLAT_HD void normalize4_backward(const float *y, float inv, const float *g, float *dv) {
    const float d = y[0] * g[0] + y[1] * g[1] + y[2] * g[2] + y[3] * g[3];
    for (int k = 0; k < 4; ++k) dv[k] = (g[k] - y[k] * d) * inv;
}

// Hamilton product c = a (x) b, components (w, x, y, z)
SPLAT_HD void quat_mult(const float *a, const float *b, float *c) {
    c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    c[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    c[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    c[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

// adjoints of quat_mult: da = g (x) conj(b), db = conj(a) (x) g
SPLAT_HD void quat_mult_backward(const float *a, const float *b, const float *g, float *da, float *db) {
    const float bc[4] = {b[0], -b[1], -b[2], -b[3]};
    const float ac[4] = {a[0], -a[1], -a[2], -a[3]};
    quat_mult(g, bc, da);
    quat_mult(ac, g, db);
}

// dL/dq of R = quat_to_rot(q) given A = dL/dR (row-major)
SPLAT_HD void quat_to_rot_backward(const float *q, const float *A, float *dq) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2.f * (-z * A[1] + y * A[2] + z * A[3] - x * A[5] - y * A[6] + x * A[7]);
    dq[1] = 2.f * (y * A[1] + z * A[2] + y * A[3] - 2.f * x * A[4] - r * A[5] + z * A[6] + r * A[7] - 2.f * x * A[8]);
    dq[2] = 2.f * (-2.f * y * A[0] + x * A[1] + r * A[2] + x * A[3] + z * A[5] - r * A[6] + z * A[7] - 2.f * y * A[8]);
    dq[3] = 2.f * (-2.f * z * A[0] - r * A[1] + x * A[2] + r * A[3] - 2.f * z * A[4] + y * A[5] + x * A[6] + y * A[7]);
}

// Camera pose of one frame as transform_to_frame builds it: cam_rot = normalize(q_raw) (used by quat_mult),
// R = build_rotation(cam_rot) (which normalises once more), t.
struct Pose {
    float q1[4];        // normalize(q_raw)
    float inv1;         // 1 / max(|q_raw|, eps)
    float q2[4];        // q1 / |q1|   (build_rotation's own normalisation)
    float inv2;
    float R[9];         // row-major
    float t[3];
};

// q_raw[k] at q_ptr[k * stride], t_raw[k] at t_ptr[k * stride]: params['cam_unnorm_rots'][0, k, time_idx] of a
// [1, 4, num_frames] tensor has stride num_frames.
SPLAT_HD void pose_from_params(const float *q_ptr, const float *t_ptr, int stride, Pose &P) {
    const float q[4] = {q_ptr[0], q_ptr[stride], q_ptr[2 * stride], q_ptr[3 * stride]};
    P.inv1 = normalize4(q, P.q1);
    const float n = sqrtf(P.q1[0] * P.q1[0] + P.q1[1] * P.q1[1] + P.q1[2] * P.q1[2] + P.q1[3] * P.q1[3]);
    P.inv2 = 1.0f / n;
    for (int k = 0; k < 4; ++k) P.q2[k] = P.q1[k] * P.inv2;
    quat_to_rot(P.q2, P.R);
    for (int k = 0; k < 3; ++k) P.t[k] = t_ptr[k * stride];
}

// Per-Gaussian pose partial sums: [0..8] dL/dR (row-major), [9..11] dL/dt, [12..15] dL/dq1 (through quat_mult, anisotropic maps)
constexpr int kPoseSums = 16;

// dL/dq_raw (4) and dL/dt_raw (3) from the summed partials
SPLAT_HD void pose_backward(const Pose &P, const float *sums, float *dq_raw, float *dt_raw) {
    float dq2[4], dq1[4];
    quat_to_rot_backward(P.q2, sums, dq2);
    normalize4_backward(P.q2, P.inv2, dq2, dq1);                    // build_rotation's q / norm
    for (int k = 0; k < 4; ++k) dq1[k] += sums[12 + k];
    normalize4_backward(P.q1, P.inv1, dq1, dq_raw);                 // F.normalize(cam_unnorm_rots[..., t])
    for (int k = 0; k < 3; ++k) dt_raw[k] = sums[9 + k];
}

// What the rasterizer is fed for one Gaussian (the two render-variable dicts share everything but the colours).
struct Glue {
    float Xc[3];        // centre in the camera frame of time_idx
    float z;            // depth-silhouette "colour": (w2c @ [Xc; 1])[2]
    float op;           // sigmoid(logit_opacity)
    float s[3];         // exp(log_scales) (tiled for isotropic maps)
    float rq[4];        // normalize(transformed unnorm rotation)
    // kept for the adjoint
    float un[4], inv_un;    // normalize(unnorm_rotation) (anisotropic only)
    float inv_rq;
};

// w2c_row2 = curr_data['w2c'][2, :] (row-major 4x4): z = row2[:3] . Xc + row2[3]
SPLAT_HD void glue_forward(const Pose &P, const float *w2c_row2, const float *p, const float *u, float logit,
                           const float *ls, bool iso, Glue &G) {
    for (int r = 0; r < 3; ++r) G.Xc[r] = P.R[3 * r] * p[0] + P.R[3 * r + 1] * p[1] + P.R[3 * r + 2] * p[2] + P.t[r];
    G.z = w2c_row2[0] * G.Xc[0] + w2c_row2[1] * G.Xc[1] + w2c_row2[2] * G.Xc[2] + w2c_row2[3];
    G.op = 1.0f / (1.0f + expf(-logit));
    for (int k = 0; k < 3; ++k) G.s[k] = expf(ls[iso ? 0 : k]);
    if (iso) {
        G.inv_un = 0.f;
        for (int k = 0; k < 4; ++k) G.un[k] = 0.f;
        G.inv_rq = normalize4(u, G.rq);
    } else {
        G.inv_un = normalize4(u, G.un);
        float qm[4];
        quat_mult(P.q1, G.un, qm);
        G.inv_rq = normalize4(qm, G.rq);
    }
}

// Adjoint of glue_forward.  Cotangents: dXc (3, geometry part only), dz, drgb is passed through by the caller,
// dop, ds (3), drq (4).  Outputs: dp (3), du (4), dlogit, dls (1 or 3), pose partial sums (kPoseSums).
SPLAT_HD void glue_backward(const Pose &P, const float *w2c_row2, const float *p, bool iso, const Glue &G,
                            const float *dXc_geom, float dz, float dop, const float *ds, const float *drq,
                            float *dp, float *du, float *dlogit, float *dls, float *pose_sums) {
    float dXc[3];
    for (int k = 0; k < 3; ++k) dXc[k] = dXc_geom[k] + w2c_row2[k] * dz;
    for (int c = 0; c < 3; ++c) dp[c] = P.R[c] * dXc[0] + P.R[3 + c] * dXc[1] + P.R[6 + c] * dXc[2];     // R^T dXc
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) pose_sums[3 * r + c] = dXc[r] * p[c];
    for (int k = 0; k < 3; ++k) pose_sums[9 + k] = dXc[k];
    *dlogit = dop * G.op * (1.0f - G.op);
    if (iso) {
        dls[0] = ds[0] * G.s[0] + ds[1] * G.s[1] + ds[2] * G.s[2];
        normalize4_backward(G.rq, G.inv_rq, drq, du);
        for (int k = 0; k < 4; ++k) pose_sums[12 + k] = 0.f;
    } else {
        for (int k = 0; k < 3; ++k) dls[k] = ds[k] * G.s[k];
        float dqm[4], dun[4];
        normalize4_backward(G.rq, G.inv_rq, drq, dqm);
        quat_mult_backward(P.q1, G.un, dqm, pose_sums + 12, dun);
        normalize4_backward(G.un, G.inv_un, dun, du);
    }
}

// ---- SSIM, per pixel (window sums already taken) -----------------------------------------------------------
// mu1 = G*x, mu2 = G*y, e11 = G*(x x), e22 = G*(y y), e12 = G*(x y).  Returns the SSIM map value and its partial
// derivatives w.r.t. mu1, e11, e12 (x is the rendered image; y, the target, carries no gradient).
SPLAT_HD float ssim_pixel(float mu1, float mu2, float e11, float e22, float e12, float *dmu1, float *de11, float *de12) {
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A = 2.f * mu12 + kSsimC1, B = 2.f * s12 + kSsimC2;
    const float Cd = mu1_sq + mu2_sq + kSsimC1, Dd = s1 + s2 + kSsimC2;
    const float iCD = 1.0f / (Cd * Dd);
    const float map = A * B * iCD;
    // d/dmu1 with e11, e12 held fixed: A' = 2 mu2, B' = -2 mu2, Cd' = 2 mu1, Dd' = -2 mu1
    *dmu1 = (2.f * mu2 * B - 2.f * mu2 * A) * iCD - map * (2.f * mu1 / Cd) + map * (2.f * mu1 / Dd);
    *de11 = -map / Dd;
    *de12 = 2.f * A * iCD;
    return map;
}

// Adam, one element (torch.optim.Adam, amsgrad=False, weight_decay=0): step_size = lr / (1 - beta1^t),
// bc2_sqrt = sqrt(1 - beta2^t), both formed on the host in double as torch does.
SPLAT_HD float adam_update(float param, float grad, float &m, float &v, float beta1, float beta2, float step_size,
                           float bc2_sqrt, float eps) {
    m = m + (1.0f - beta1) * (grad - m);
    v = v * beta2 + (1.0f - beta2) * grad * grad;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    return param - step_size * (m / denom);
}

}  // namespace splat
