// Compiles splatam_amd/csrc/splat_math.h for the HOST (g++) so that the exact
// per-Gaussian arithmetic the HIP kernels execute can be checked against the
// oracle on a machine without a GPU (tests/test_host_math.py).
#include "../splatam_amd/csrc/splat_math.h"

using namespace splat;

extern "C" {

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
}
