"""CPU oracle for the Gaussian-splat rasterizer (forward + gradients).

TEST INFRASTRUCTURE ONLY.  Nothing under ``splatam_amd/`` or
``diff_gaussian_rasterization/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the timed CPU baseline.

PARITY UNPINNED.  The arithmetic lives in a third-party dependency that is not
vendored in the reference tree: ``JonathonLuiten/diff-gaussian-rasterization-w-depth``
pinned at commit ``cb65e4b86bc3bd8ed42174b72a62e8d3a3a71110``
(/root/reference/requirements.txt:15, /root/reference/.gitmodules:1-3; the
submodule directory is empty).  The reference holds no golden vectors or tests
for this path, so this file restates the *published*
3D-Gaussian-splatting rasterizer algorithm and anchors
on the reference's own call sites:

* call signature / outputs ........ /root/reference/scripts/splatam.py:249,253
* settings tuple .................. /root/reference/utils/recon_helpers.py:14-26
* matrix conventions .............. /root/reference/utils/recon_helpers.py:8-13
* quaternion -> rotation .......... /root/reference/utils/slam_external.py:25-42
* silhouette = 1 - T_final ........ /root/reference/scripts/splatam.py:254-256

Formulation: a dense, differentiable, per-tile vectorised restatement in
PyTorch (float32 by default, float64 for finite-difference checks).  Every
gradient comes from autograd over the forward arithmetic, with the two places
where the published backward deliberately differs from the true derivative
modelled explicitly (straight-through 0.99 opacity clamp; frozen frustum clamp
of the view-space centre).  The independent C restatement in
``oracle/raster_ref.c`` writes the backward pass out by hand; the two are
cross-checked in ``tests/test_oracle.py``.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

TILE = 16
NEAR_Z = 0.2            # Appendix A, preprocess step 1
DILATION = 0.3          # Appendix A, preprocess step 4
ALPHA_MIN = 1.0 / 255.0  # Appendix A, composite skip rule
ALPHA_MAX = 0.99
T_STOP = 1e-4
FOV_GUARD = 1.3

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


class Settings(NamedTuple):
    """Same 11 fields as the reference's settings tuple
    (/root/reference/utils/recon_helpers.py:14-26)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """(r,x,y,z) -> 3x3, *no* renormalisation (the caller normalises:
    /root/reference/utils/slam_helpers.py:134).  Same polynomial as
    /root/reference/utils/slam_external.py:33-41."""
    r, x, y, z = q.unbind(-1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=-1).reshape(q.shape[:-1] + (3, 3))


def cov3d_from_scale_rot(scales, rotations, scale_modifier):
    """Sigma = R diag((mod*s)^2) R^T, returned as the 6 upper-triangular
    entries (00,01,02,11,12,22).  Appendix A preprocess step 3."""
    R = quat_to_rot(rotations)
    s = scales * scale_modifier
    M = R * s[:, None, :]              # R @ diag(s)
    Sig = M @ M.transpose(1, 2)
    return torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2],
                        Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics, degrees 0..3, the 3DGS convention: colour =
    max(0, SH(dir) + 0.5); the clamp zeroes the gradient (autograd's relu does
    the same).  sh: [N, M, 3], dirs: [N, 3] unit."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.relu(res + 0.5)


class Geometry(NamedTuple):
    depth: torch.Tensor        # [N] view-space z (no grad)
    radii: torch.Tensor        # [N] int32, 0 = culled
    xy: torch.Tensor           # [N,2] pixel centre (differentiable)
    conic: torch.Tensor        # [N,3] inverse 2D covariance (differentiable)
    rect_min: torch.Tensor     # [N,2] int tile rect (x,y), inclusive
    rect_max: torch.Tensor     # [N,2] int tile rect (x,y), exclusive
    cov2d: torch.Tensor        # [N,3] (a,b,c) after dilation
    cov3d: torch.Tensor        # [N,6]


def preprocess(means3D, means2D, scales, rotations, cov3D_precomp, settings) -> Geometry:
    """Appendix A 'Preprocess (K1)'.  Differentiable in means3D, scales,
    rotations, cov3D_precomp and means2D (the NDC-space gradient sink)."""
    dt = means3D.dtype
    N = means3D.shape[0]
    W_img, H_img = int(settings.image_width), int(settings.image_height)
    V = settings.viewmatrix.reshape(4, 4).to(dt).t()     # w2c          (column-major flat = w2c)
    PV = settings.projmatrix.reshape(4, 4).to(dt).t()    # P @ w2c
    W3, tv = V[:3, :3], V[:3, 3]

    # step 1: view space + near cull
    p_view = means3D @ W3.t() + tv
    depth = p_view[:, 2].detach()
    visible = depth > NEAR_Z

    # step 2: clip space -> NDC
    ones = torch.ones(N, 1, dtype=dt)
    p_hom = torch.cat([means3D, ones], dim=1) @ PV.t()
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :2] * p_w[:, None]
    if means2D is not None:
        p_proj = p_proj + means2D[:, :2]     # zeros; .grad == dL/d(NDC)

    # step 3
    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        cov3d = cov3D_precomp
    else:
        cov3d = cov3d_from_scale_rot(scales, rotations, float(settings.scale_modifier))
    Sig = torch.stack([cov3d[:, 0], cov3d[:, 1], cov3d[:, 2],
                       cov3d[:, 1], cov3d[:, 3], cov3d[:, 4],
                       cov3d[:, 2], cov3d[:, 4], cov3d[:, 5]], dim=-1).reshape(N, 3, 3)

    # step 4: EWA projection.  The frustum clamp freezes the clamped
    # coordinate in the published backward (x_grad_mul = 0, and no extra
    # d/dz term), hence the detach on the clamped branch.
    tz = p_view[:, 2]
    tz_safe = torch.where(visible, tz, torch.ones_like(tz))
    limx, limy = FOV_GUARD * settings.tanfovx, FOV_GUARD * settings.tanfovy
    txtz, tytz = p_view[:, 0] / tz_safe, p_view[:, 1] / tz_safe
    cx_ = (txtz < -limx) | (txtz > limx)
    cy_ = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx_, (txtz.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
    ty = torch.where(cy_, (tytz.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    fx = W_img / (2.0 * settings.tanfovx)
    fy = H_img / (2.0 * settings.tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -fx * tx / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -fy * ty / (tz_safe * tz_safe)], dim=-1).reshape(N, 2, 3)
    Tm = J @ W3                                   # [N,2,3]
    cov = Tm @ Sig @ Tm.transpose(1, 2)           # [N,2,2]
    a = cov[:, 0, 0] + DILATION
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + DILATION

    # step 5
    det = a * c - b * b
    ok = visible & (det.detach() != 0)
    det_safe = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=-1)

    # step 6
    with torch.no_grad():
        mid = 0.5 * (a + c)
        disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        lam = torch.maximum(mid + disc, mid - disc)
        radius = torch.ceil(3.0 * torch.sqrt(lam))

    # step 7
    px = ((p_proj[:, 0] + 1.0) * W_img - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H_img - 1.0) * 0.5
    xy = torch.stack([px, py], dim=-1)

    # step 8 (C-style int cast truncates toward zero)
    gx, gy = (W_img + TILE - 1) // TILE, (H_img + TILE - 1) // TILE
    with torch.no_grad():
        def _rect(p, r, g, off):
            v = torch.trunc((p + r + off) / TILE)
            v = torch.nan_to_num(v, nan=0.0, posinf=float(g), neginf=0.0)
            return v.clamp(0, g).to(torch.int64)
        r = torch.where(ok, radius, torch.zeros_like(radius))
        rmin = torch.stack([_rect(px, -r, gx, 0.0), _rect(py, -r, gy, 0.0)], dim=-1)
        rmax = torch.stack([_rect(px, r, gx, TILE - 1.0), _rect(py, r, gy, TILE - 1.0)], dim=-1)
        area = (rmax[:, 0] - rmin[:, 0]) * (rmax[:, 1] - rmin[:, 1])
        ok = ok & (area > 0)
        radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
        rmin = torch.where(ok[:, None], rmin, torch.zeros_like(rmin))
        rmax = torch.where(ok[:, None], rmax, torch.zeros_like(rmax))
    return Geometry(depth, radii, xy, conic, rmin, rmax,
                    torch.stack([a, b, c], dim=-1), cov3d)


class RenderAux(NamedTuple):
    final_T: torch.Tensor      # [H,W]
    n_contrib: torch.Tensor    # [H,W] int32, 1-based position of last contributor in the tile list
    tile_counts: torch.Tensor  # [gy*gx] int64, length of every tile list
    geom: Geometry


def rasterize(means3D, means2D, opacities, colors_precomp, scales, rotations, settings,
              cov3D_precomp=None, shs=None, return_aux: bool = False):
    """Forward of the reference boundary (/root/reference/scripts/splatam.py:249):
    returns (color[C,H,W], radii[N] int32, depth[1,H,W]) and is differentiable
    w.r.t. means3D, means2D (NDC sink), opacities, colors_precomp / shs, scales,
    rotations, cov3D_precomp.  The depth image carries no gradient (the
    published backward ignores it)."""
    dt = means3D.dtype
    N = means3D.shape[0]
    W_img, H_img = int(settings.image_width), int(settings.image_height)
    geom = preprocess(means3D, means2D, scales, rotations, cov3D_precomp, settings)

    if shs is not None and shs.numel() > 0:
        dirs = means3D - settings.campos.to(dt)[None, :]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        colors = eval_sh(int(settings.sh_degree), shs, dirs)
    else:
        colors = colors_precomp
    C = colors.shape[1]
    bg = settings.bg.to(dt).reshape(-1)
    opac = opacities.reshape(-1)

    gx, gy = (W_img + TILE - 1) // TILE, (H_img + TILE - 1) // TILE
    out_color = torch.zeros(C, gy * TILE, gx * TILE, dtype=dt)
    out_depth = torch.zeros(gy * TILE, gx * TILE, dtype=dt)
    final_T = torch.ones(gy * TILE, gx * TILE, dtype=dt)
    n_contrib = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int32)
    tile_counts = torch.zeros(gy * gx, dtype=torch.int64)

    # Binning (K2-K5): per tile, the Gaussians whose rect covers it, ordered by
    # (float32 depth bits, index) == stable sort of index-ordered emission.
    rmin, rmax = geom.rect_min, geom.rect_max
    alive = torch.nonzero(geom.radii > 0).reshape(-1)
    key_depth = geom.depth.to(torch.float32)
    ly, lx = torch.meshgrid(torch.arange(TILE, dtype=dt), torch.arange(TILE, dtype=dt), indexing="ij")
    # bucket Gaussians per tile row to keep the python loop cheap
    for ty_ in range(gy):
        row = alive[(rmin[alive, 1] <= ty_) & (rmax[alive, 1] > ty_)]
        if row.numel() == 0:
            continue
        for tx_ in range(gx):
            ids = row[(rmin[row, 0] <= tx_) & (rmax[row, 0] > tx_)]
            G = ids.numel()
            tile_counts[ty_ * gx + tx_] = G
            if G == 0:
                continue
            order = torch.argsort(key_depth[ids], stable=True)
            ids = ids[order]
            pixx = (lx + tx_ * TILE).reshape(-1, 1)       # [256,1]
            pixy = (ly + ty_ * TILE).reshape(-1, 1)
            dx = geom.xy[ids, 0][None, :] - pixx           # [256,G]
            dy = geom.xy[ids, 1][None, :] - pixy
            con = geom.conic[ids]
            power = -0.5 * (con[:, 0][None] * dx * dx + con[:, 2][None] * dy * dy) - con[:, 1][None] * dx * dy
            a_raw = opac[ids][None, :] * torch.exp(power)
            # straight-through clamp: the published backward differentiates
            # alpha = opacity * G even where the 0.99 clamp is active.
            alpha = a_raw + (torch.clamp(a_raw, max=ALPHA_MAX) - a_raw).detach()
            skip = (power.detach() > 0) | (alpha.detach() < ALPHA_MIN)
            a_eff = torch.where(skip, torch.zeros_like(alpha), alpha)
            one_m = 1.0 - a_eff
            T_incl = torch.cumprod(one_m, dim=1)                                   # T after each Gaussian
            T_before = torch.cat([torch.ones(T_incl.shape[0], 1, dtype=dt), T_incl[:, :-1]], dim=1)
            stop = (T_incl.detach() < T_STOP) & ~skip
            keep = (torch.cumsum(stop.to(torch.int32), dim=1) == 0)
            w = a_eff * T_before * keep.to(dt)                                     # [256,G]
            col = w @ colors[ids]                                                  # [256,C]
            # transmittance after the last kept Gaussian
            kept_one_m = torch.where(keep, one_m, torch.ones_like(one_m))
            Tf = torch.prod(kept_one_m, dim=1)
            with torch.no_grad():
                dep = w.detach() @ geom.depth[ids].to(dt)
                contrib = keep & ~skip
                pos = torch.arange(1, G + 1, dtype=torch.int32)[None, :].expand_as(contrib)
                nc = torch.where(contrib, pos, torch.zeros_like(pos)).amax(dim=1)
            ys, xs = slice(ty_ * TILE, (ty_ + 1) * TILE), slice(tx_ * TILE, (tx_ + 1) * TILE)
            out_color[:, ys, xs] = col.t().reshape(C, TILE, TILE)
            out_depth[ys, xs] = dep.reshape(TILE, TILE)
            final_T[ys, xs] = Tf.reshape(TILE, TILE)
            n_contrib[ys, xs] = nc.reshape(TILE, TILE)

    out_color = out_color[:, :H_img, :W_img]
    final_T = final_T[:H_img, :W_img]
    out_color = out_color + final_T[None] * bg[:, None, None]
    out_depth = out_depth[:H_img, :W_img].detach()[None]
    n_contrib = n_contrib[:H_img, :W_img]
    if return_aux:
        return out_color, geom.radii, out_depth, RenderAux(final_T, n_contrib, tile_counts, geom)
    return out_color, geom.radii, out_depth


def mark_visible(means3D, settings) -> torch.Tensor:
    """Visibility mask of the boundary's ``markVisible``."""
    V = settings.viewmatrix.reshape(4, 4).to(means3D.dtype).t()
    z = means3D @ V[2, :3] + V[2, 3]
    return z > NEAR_Z


class OracleRasterizer(torch.nn.Module):
    """Same call surface as the reference's ``Renderer``
    (/root/reference/scripts/splatam.py:249) backed by the dense CPU oracle."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None,
                scales=None, rotations=None, cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize(means3D, means2D, opacities, colors_precomp, scales, rotations,
                         self.raster_settings, cov3D_precomp=cov3D_precomp, shs=shs)


# --------------------------------------------------------------------------
# Synthetic SplaTAM-like scenes (SURVEY.md section 8d "Synthetic inputs")
# --------------------------------------------------------------------------

def make_camera(width, height, fx, fy, cx, cy, w2c=None, near=0.01, far=100.0, bg=(0.0, 0.0, 0.0),
                dtype=torch.float32) -> Settings:
    """Restates /root/reference/utils/recon_helpers.py:4-27 (setup_camera) on CPU."""
    w2c_t = torch.eye(4, dtype=dtype) if w2c is None else torch.as_tensor(w2c, dtype=dtype)
    cam_center = torch.inverse(w2c_t)[:3, 3]
    P = torch.tensor([[2 * fx / width, 0.0, -(width - 2 * cx) / width, 0.0],
                      [0.0, 2 * fy / height, -(height - 2 * cy) / height, 0.0],
                      [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                      [0.0, 0.0, 1.0, 0.0]], dtype=dtype)
    view = w2c_t.unsqueeze(0).transpose(1, 2)
    full = view.bmm(P.unsqueeze(0).transpose(1, 2))
    return Settings(image_height=height, image_width=width,
                    tanfovx=width / (2 * fx), tanfovy=height / (2 * fy),
                    bg=torch.tensor(bg, dtype=dtype), scale_modifier=1.0,
                    viewmatrix=view, projmatrix=full, sh_degree=0,
                    campos=cam_center, prefiltered=False)


def synthetic_cloud(n, width, height, fx, fy, cx, cy, seed=0, anisotropic=False, dtype=torch.float32, region=None):
    """Seeded SplaTAM-like cloud: one Gaussian per random (sub-)pixel back-projected
    at z~U[1,4] (mirrors /root/reference/scripts/splatam.py:76-99,120-128).
    ``region`` = (u0, v0, u1, v1) as fractions of the image: all Gaussians project inside that window (the
    clustered variant of SURVEY.md 8d: per-tile lists far longer than LDS)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, generator=g, dtype=torch.float64) * width - 0.5
    v = torch.rand(n, generator=g, dtype=torch.float64) * height - 0.5
    if region is not None:
        u0, v0, u1, v1 = region
        u = (u + 0.5) * (u1 - u0) + u0 * width - 0.5
        v = (v + 0.5) * (v1 - v0) + v0 * height - 0.5
    z = 1.0 + 3.0 * torch.rand(n, generator=g, dtype=torch.float64)
    means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], dim=-1)
    log_s = torch.log(z / ((fx + fy) / 2)) + 0.3 * torch.randn(n, generator=g, dtype=torch.float64)
    if anisotropic:
        log_scales = log_s[:, None] + 0.3 * torch.randn(n, 3, generator=g, dtype=torch.float64)
        rots = torch.randn(n, 4, generator=g, dtype=torch.float64)
    else:
        log_scales = log_s[:, None]
        rots = torch.zeros(n, 4, dtype=torch.float64)
        rots[:, 0] = 1.0
    logit_op = 2.0 + torch.randn(n, 1, generator=g, dtype=torch.float64)
    rgb = torch.rand(n, 3, generator=g, dtype=torch.float64)
    return {
        'means3D': means.to(dtype), 'rgb_colors': rgb.to(dtype),
        'unnorm_rotations': rots.to(dtype), 'logit_opacities': logit_op.to(dtype),
        'log_scales': log_scales.to(dtype),
    }


def cloud_to_rendervar(params):
    """Restates /root/reference/utils/slam_helpers.py:110-121 (params2rendervar)."""
    ls = params['log_scales']
    if ls.shape[1] == 1:
        ls = ls.repeat(1, 3)
    return {
        'means3D': params['means3D'],
        'colors_precomp': params['rgb_colors'],
        'rotations': torch.nn.functional.normalize(params['unnorm_rotations']),
        'opacities': torch.sigmoid(params['logit_opacities']),
        'scales': torch.exp(ls),
        'means2D': torch.zeros_like(params['means3D']),
    }
