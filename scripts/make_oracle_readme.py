#!/usr/bin/env python3
"""Writes oracle/README.md: the clause-by-clause audit table of the UNPINNED restatement -- every statement of
SURVEY.md Appendix A (the recalled arithmetic of the un-vendored CUDA rasterizer) beside the line of oracle/raster_ref.c that restates
it and the line of the product (splat_math.h / the .hip kernels) that computes it.  Line numbers are looked up from the statements'
text, so the table cannot drift from the sources: tests/test_oracle_readme.py re-runs this and compares.
usage: python scripts/make_oracle_readme.py [--check]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = "oracle/raster_ref.c"
M = "splatam_amd/csrc/splat_math.h"
RH = "splatam_amd/csrc/render.hip"
BH = "splatam_amd/csrc/binning.hip"
PH = "splatam_amd/csrc/preprocess.hip"

# (Appendix A step, clause, token in the oracle, product file, token in the product)
CLAUSES = [
    ("K1.1", "p_view = W [p; 1]", "tv[r] = m4(view, r, 0) * p[0]", M, "tv[r] = mat(c.view, r, 0) * p[0]"),
    ("K1.1", "cull when p_view.z <= 0.2 (radius stays 0)", "if (tv[2] <= 0.2f) continue;", M, "if (!(tv[2] > kNearZ)) return false;"),
    ("K1.2", "p_hom = (P W)[p; 1]; p_w = 1 / (p_hom.w + 1e-7)", "real pw = 1.f / (hom[3] + 0.0000001f);", M, "const float pw = 1.0f / (hom[3] + 0.0000001f);"),
    ("K1.3", "Sigma = M M^T, M = R diag(scale_modifier * s); quaternion (r, x, y, z) NOT renormalised", "M[i][j] = R[i][j] * (mod * scale[j]);", M, "M[3 * i + j] = R[3 * i + j] * (mod * s[j]);"),
    ("K1.4", "guard band: x/z, y/z clamped to +-1.3 tanfov before the Jacobian", "real limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;", M, "const float limx = kFovGuard * c.tanfovx, limy = kFovGuard * c.tanfovy;"),
    ("K1.4", "J = [[fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2]], T = J W", "real J[2][3] = {{fx / tz, 0.f, -(fx * tx) / (tz * tz)}", M, "const float j00 = c.focal_x / e.tz, j02 = -(c.focal_x * e.tx) / (e.tz * e.tz);"),
    ("K1.4", "cov2D = T Sigma T^T; + 0.3 on both diagonal entries", "+ TS[0][2] * T[0][2] + 0.3f;", M, "abc[0] = TS[0] * e.T[0] + TS[1] * e.T[1] + TS[2] * e.T[2] + kDilation;"),
    ("K1.5", "det = a c - b^2; det == 0 -> skip; conic = (c, -b, a) / det", "if (det == 0.f) continue;", M, "if (det == 0.0f) return false;"),
    ("K1.6", "lambda = mid + sqrt(max(0.1, mid^2 - det)); radius = ceil(3 sqrt(lambda))", "real disc = R_SQRT(R_MAX(0.1f, mid * mid - det));", M, "const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));"),
    ("K1.7", "pixel centre = ((ndc + 1) S - 1) / 2", "real px = ((ndcx + 1.f) * W - 1.f) * 0.5f", M, "const float px = ((ndcx + 1.0f) * c.W - 1.0f) * 0.5f;"),
    ("K1.8", "tile rectangle [int((xy - r)/16), int((xy + r + 15)/16)) clamped to the grid; empty -> skip", "int x1 = (int)((px + radius + TILE - 1) / TILE)", M, "const int x1 = clampi((int)((px + radius + kTile - 1) / kTile), 0, c.gx);"),
    ("K2-K5", "per tile: ascending (float bits of depth, Gaussian id)", "uint32_t key; { const float d32 = (float)c->depth[i]; memcpy(&key, &d32, 4); }", BH, "const uint64_t key = ((uint64_t)__float_as_uint(st.depth[i]) << 32) | (uint32_t)i;"),
    ("K6", "power = -0.5 (cxx dx^2 + cyy dy^2) - cxy dx dy; skip when power > 0 (the product folds log2 e into the staged conic)", "if (power > 0.f) continue;", RH, "const float p2 = dx * (cur.a.x * dx + cur.a.y * dy) + cur.a.z * dy * dy;     // power * log2(e)"),
    ("K6", "alpha = min(0.99, opacity exp(power)); skip when alpha < 1/255", "if (alpha < 1.f / 255.f) continue;", RH, "const float alpha = fminf(kAlphaMax, cur.a.w * fast_exp2(p2));"),
    ("K6", "test_T = T (1 - alpha); test_T < 1e-4 -> pixel done, this Gaussian NOT accumulated", "if (test_T < 0.0001f) break;", RH, "const unsigned long long stop_m = __builtin_amdgcn_ballot_w64(test_T < kTStop) & live_m;"),
    ("K6", "C += colour alpha T; D += depth alpha T; T = test_T; last contributor = list position (1-based, skipped ones counted)", "D += c->depth[id] * alpha * T;", RH, "const float wgt = upd ? alpha * Tr : 0.f;"),
    ("K6", "out = C + T bg; depth without background; final_T and n_contrib kept for the backward pass", "out_color[(size_t)ch * W * H + pix] = Cc[ch] + T * bg[ch];", RH, "o[ch] = Cc[ch] + Tr * bg_of(cam, ch);"),
    ("K7", "back to front from n_contrib: T = T / (1 - alpha)", "T = T / (1.f - alpha);", RH, "const float Tn = Tr * rcp;                     // transmittance in front of this Gaussian"),
    ("K7", "accum = last_alpha last_colour + (1 - last_alpha) accum; dL/dalpha += (colour - accum) dL/dC  (the product carries the same recursion as the running sum R)", "accum[ch] = last_alpha * last_col[ch] + (1.f - last_alpha) * accum[ch];", RH, "const float dL_dalpha = fmaf(cdot, Tn, -(R * rcp));"),
    ("K7", "dL/dalpha += -T_final / (1 - alpha) * sum_ch bg dL/dC", "dL_dalpha += (-T_final / (1.f - alpha)) * bgdot;", RH, "if constexpr (BG) R += Tfin * bg_of(cam, ch) * dpix[ch];"),
    ("K7", "dL/dG = opacity dL/dalpha; dL/dmean2D (NDC) = dL/dG dG/dd 0.5 (W, H)", "a[0] += (double)(dL_dG * dG_ddx * 0.5f * W);", PH, "g_ndc[0] = -(co.x * acc[0] + co.y * acc[1]) * 0.5f * c.W;"),
    ("K7", "dL/dconic = (-0.5 gdx dx, -gdx dy [the two symmetric halves], -0.5 gdy dy) dL/dG", "a[3] += (double)(-gdx * dy * dL_dG);", PH, "const float g_conic[3] = {-0.5f * acc[2], -acc[3], -0.5f * acc[4]};"),
    ("K7", "dL/dopacity += G dL/dalpha; dL/dcolour += alpha T dL/dC", "a[5] += (double)(G * dL_dalpha);", RH, "const float ww = al * Tn;"),
    ("K8", "conic -> cov2D with 1 / (det^2 + 1e-7)", "real d2 = 1.f / (det * det + 0.0000001f);", M, "const float d2 = 1.0f / (det * det + 0.0000001f);"),
    ("K8", "dL/da, dL/dc, dL/db of the 2x2 covariance", "real dLb = d2 * (2.f * cb * cc * gcx - (det + 2.f * cb * cb) * gcy + 2.f * ca * cb * gcz);", M, "const float dLb = d2 * (2.f * b * cc * g_conic[0] - (det + 2.f * b * b) * g_conic[1] + 2.f * a * b * g_conic[2]);"),
    ("K8", "dL/dSigma = T^T G T; dL/dT = 2 G T Sigma; dL/dJ; guard-band masks on dL/dt", "real dtx = xmul * -fx * iz2 * dJ[0][2];", M, "const float dtx = e.xmul * -c.focal_x * iz2 * dJ02;"),
    ("K8", "dL/dmean3D += quotient rule of the perspective divide (rows 0, 1, 3 of P W)", "real dndcx = m4(proj, 0, k) * pw - hom[0] * pw * pw * m4(proj, 3, k);", M, "dmean[k] += (mat(c.proj, 0, k) * pw - w3 * mx) * g_ndc[0] + (mat(c.proj, 1, k) * pw - w3 * my) * g_ndc[1];"),
    ("K9", "Sigma = M M^T -> dL/dscale (with the factor scale_modifier; SPLAT_GRADS_UPSTREAM_SCALE / ref_set_upstream_scale: without, as the CUDA original), dL/dquat without normalisation Jacobian", "dscales[3 * i + k] = (c->upstream_scale ? (real)1 : mod) *", M, "dscale[k] = (upstream_scale ? 1.0f : mod) * (col[0] * R[k] + col[1] * R[3 + k] + col[2] * R[6 + k]);"),
]

HEAD = """# oracle/ -- the CPU restatement of the rasterizer, and how to audit it

TEST INFRASTRUCTURE ONLY: `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg load what is built from this directory;
the product (`splatam_amd/`, `diff_gaussian_rasterization/`) never does.

* `raster_ref.c` -- plain C, float32 (`_build/libraster_ref.so`) and the same statements in float64 (`_build/libraster_ref_f64.so`),
  forward and hand-written backward, OpenMP; `c_ref.py` is its ctypes face (`CRef`, and `CRasterizer`: the oracle behind the
  reference's `Renderer` surface).  `raster_ref.py` -- an independent dense PyTorch restatement whose gradients come from autograd.
* **Parity unpinned.**  The arithmetic of this path lives in `JonathonLuiten/diff-gaussian-rasterization-w-depth @ cb65e4b8`
  (`/root/reference/requirements.txt:15`; the submodule directory is empty, `.gitmodules:1-3`), which cannot be built here (CUDA) and
  for which the reference holds no golden vectors.  What the restatement follows is SURVEY.md Appendix A (recalled from the published
  algorithm); what pins it is: analytic known answers, float64 central differences on all six inputs, the C file against the autograd
  file (`tests/test_oracle.py`), and the reference's own call sites and behavioural constraints (3 outputs, radii 0 = culled,
  silhouette > 0.99 where covered, `C + T bg`: SURVEY.md 8c).  The table below is for a reader who wants to check the restatement
  against the CUDA original in one sitting: one row per statement of Appendix A, the line that restates it, the line of the product
  that computes it.
* One deliberate, switchable difference from the original as recalled: `dL/dscales` carries the factor `scale_modifier` (the gradient
  w.r.t. the scales the caller passed); `SplatGrads.flags = SPLAT_GRADS_UPSTREAM_SCALE` / `rasterizer.set_upstream_scale_gradient(True)` /
  `ref_set_upstream_scale` hand out the original's numbers (identical at modifier 1, i.e. for every call SplaTAM makes;
  `tests/test_gpu_dropin_policy.py::test_upstream_scale_gradient_switch`).

## Clause by clause (generated by `scripts/make_oracle_readme.py`; `tests/test_oracle_readme.py` keeps the line numbers true)

| Appendix A | statement | oracle/raster_ref.c | product |
|---|---|---|---|
"""


def line_of(path, token):
    hits = [i + 1 for i, ln in enumerate(open(os.path.join(ROOT, path))) if token in ln]
    if not hits:
        raise SystemExit(f"{path}: statement not found: {token}")
    return hits[0]


def render():
    rows = []
    for step, clause, otok, pfile, ptok in CLAUSES:
        rows.append(f"| {step} | {clause} | `:{line_of(ORACLE, otok)}` | `{pfile}:{line_of(pfile, ptok)}` |")
    return HEAD + "\n".join(rows) + "\n"


def main():
    text = render()
    path = os.path.join(ROOT, "oracle", "README.md")
    if "--check" in sys.argv:
        return 0 if os.path.exists(path) and open(path).read() == text else 1
    with open(path, "w") as f:
        f.write(text)
    print(path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
