// Compiles splatam_amd/csrc/splat_math.h for the HOST (g++) so that the exact
// per-Gaussian arithmetic the HIP kernels execute can be checked against the
// oracle on a machine without a GPU (tests/test_host_math.py).
#include "../splatam_amd/csrc/splat_math.h"
#include "../splatam_amd/csrc/fused_math.h"

using namespace splat;

extern "C" {

// live_tile_rect (group binning files only the tiles that can blend): rect[4] = x0, y0, x1, y1 in, tightened in place
void hm_live_tile_rect(int P, const float *conic, const float *opacity, const float *xy, int *rect) {
    for (int i = 0; i < P; ++i)
        live_tile_rect(conic + 3 * i, opacity[i], xy[2 * i], xy[2 * i + 1], rect[4 * i], rect[4 * i + 1], rect[4 * i + 2], rect[4 * i + 3]);
}


void hm_forward(int P, const float *means, const float *scales, const float *rots, const float *view, const float *proj,
                int W, int H, float tfx, float tfy, float mod, float *depth, float *xy, float *conic, int *radii, int *rect) {
    CamConst c;
    init_cam(c, view, proj, W, H, tfx, tfy, mod);
    for (int i = 0; i < P; ++i) {
        float S6[6];
        cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, S6);
        Projected o;
        project_gaussian(c, means + 3 * i, S6, o);
        depth[i] = o.depth; xy[2 * i] = o.px; xy[2 * i + 1] = o.py;
        for (int k = 0; k < 3; ++k) conic[3 * i + k] = o.conic[k];
        radii[i] = o.radius;
        rect[4 * i] = o.x0; rect[4 * i + 1] = o.y0; rect[4 * i + 2] = o.x1; rect[4 * i + 3] = o.y1;
    }
}

void hm_backward(int P, const float *means, const float *scales, const float *rots, const float *view, const float *proj,
                 int W, int H, float tfx, float tfy, float mod, const float *g_ndc, const float *g_conic,
                 float *dmean, float *dscale, float *drot, float *dcov) {
    CamConst c;
    init_cam(c, view, proj, W, H, tfx, tfy, mod);
    for (int i = 0; i < P; ++i) {
        float S6[6];
        cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, S6);
        project_gaussian_backward(c, means + 3 * i, S6, g_ndc + 2 * i, g_conic + 3 * i, dmean + 3 * i, dcov + 6 * i);
        cov3d_backward(scales + 3 * i, mod, rots + 4 * i, dcov + 6 * i, dscale + 3 * i, drot + 4 * i);
    }
}

// colour = max(0, sum_k basis_k(dir) sh_k + 0.5); also d(colour)/d(sh) contraction and d/d(dir) for a given dL/dcolour
void hm_sh(int P, int deg, int M, const float *dirs, const float *sh, const float *gcol, float *col, float *dsh, float *ddir) {
    for (int i = 0; i < P; ++i) {
        float basis[16], db[3][16];
        sh_basis(deg, dirs + 3 * i, basis, db);
        const int nb = (deg + 1) * (deg + 1);
        ddir[3 * i] = ddir[3 * i + 1] = ddir[3 * i + 2] = 0.f;
        for (int ch = 0; ch < 3; ++ch) {
            float v = 0.5f;
            for (int k = 0; k < nb; ++k) v += basis[k] * sh[((size_t)i * M + k) * 3 + ch];
            col[3 * i + ch] = v > 0.f ? v : 0.f;
            const float g = v > 0.f ? gcol[3 * i + ch] : 0.f;
            for (int k = 0; k < M; ++k) dsh[((size_t)i * M + k) * 3 + ch] = k < nb ? basis[k] * g : 0.f;
            for (int k = 0; k < nb; ++k) {
                const float w = sh[((size_t)i * M + k) * 3 + ch] * g;
                ddir[3 * i] += db[0][k] * w; ddir[3 * i + 1] += db[1][k] * w; ddir[3 * i + 2] += db[2][k] * w;
            }
        }
    }
}

// ---- fused_math.h: the glue around the rasterizer (pose transform, activations) and its adjoint -------------
// q_raw[4], t_raw[3] contiguous; outputs per Gaussian: Xc(3) z(1) op(1) s(3) rq(4) = 12 floats
void hm_glue_forward(int P, int iso, const float *q_raw, const float *t_raw, const float *w2c_row2, const float *means,
                     const float *urot, const float *logit, const float *ls, float *out) {
    Pose Ps;
    pose_from_params(q_raw, t_raw, 1, Ps);
    for (int i = 0; i < P; ++i) {
        Glue G;
        glue_forward(Ps, w2c_row2, means + 3 * i, urot + 4 * i, logit[i], ls + (iso ? 1 : 3) * i, iso != 0, G);
        float *o = out + 12 * i;
        for (int k = 0; k < 3; ++k) o[k] = G.Xc[k];
        o[3] = G.z; o[4] = G.op;
        for (int k = 0; k < 3; ++k) o[5 + k] = G.s[k];
        for (int k = 0; k < 4; ++k) o[8 + k] = G.rq[k];
    }
}

// cotangents per Gaussian in the layout of hm_glue_forward's output; outputs: dmeans(3) durot(4) dlogit(1) dls(1|3),
// and the camera gradient dq_raw(4), dt_raw(3) (partial sums accumulated in double like the kernel's atomics)
void hm_glue_backward(int P, int iso, const float *q_raw, const float *t_raw, const float *w2c_row2, const float *means,
                      const float *urot, const float *logit, const float *ls, const float *cot,
                      float *dmeans, float *durot, float *dlogit, float *dls, float *dq_raw, float *dt_raw) {
    Pose Ps;
    pose_from_params(q_raw, t_raw, 1, Ps);
    double acc[kPoseSums];
    for (int k = 0; k < kPoseSums; ++k) acc[k] = 0.0;
    for (int i = 0; i < P; ++i) {
        Glue G;
        glue_forward(Ps, w2c_row2, means + 3 * i, urot + 4 * i, logit[i], ls + (iso ? 1 : 3) * i, iso != 0, G);
        const float *c = cot + 12 * i;
        float pose[kPoseSums];
        glue_backward(Ps, w2c_row2, means + 3 * i, iso != 0, G, c, c[3], c[4], c + 5, c + 8, dmeans + 3 * i, durot + 4 * i,
                      dlogit + i, dls + (iso ? 1 : 3) * i, pose);
        for (int k = 0; k < kPoseSums; ++k) acc[k] += pose[k];
    }
    float sums[kPoseSums];
    for (int k = 0; k < kPoseSums; ++k) sums[k] = (float)acc[k];
    pose_backward(Ps, sums, dq_raw, dt_raw);
}

void hm_ssim_pixel(int n, const float *mu1, const float *mu2, const float *e11, const float *e22, const float *e12,
                   float *map, float *dmu1, float *de11, float *de12) {
    for (int i = 0; i < n; ++i) map[i] = ssim_pixel(mu1[i], mu2[i], e11[i], e22[i], e12[i], dmu1 + i, de11 + i, de12 + i);
}

void hm_adam(int n, float *param, const float *grad, float *m, float *v, float beta1, float beta2, float step_size,
             float bc2_sqrt, float eps) {
    for (int i = 0; i < n; ++i) param[i] = adam_update(param[i], grad[i], m[i], v[i], beta1, beta2, step_size, bc2_sqrt, eps);
}
}
