"""Map export next to ``params.npz`` ("the on-disk format ... and PLY export live next to it").

``save_ply`` writes what /root/reference/scripts/export_ply.py:20-42 writes -- a 3D-Gaussian-Splatting ``splat.ply`` with one float32
vertex element of 17 properties -- without the ``plyfile`` package (not installed here): the header is assembled by hand and the
table written in one piece.  Not on the hot path; kept because a user of the reference expects the file the viewers open."""
from __future__ import annotations

import numpy as np

PLY_ATTRS = ('x', 'y', 'z', 'nx', 'ny', 'nz', 'f_dc_0', 'f_dc_1', 'f_dc_2', 'opacity', 'scale_0', 'scale_1', 'scale_2',
             'rot_0', 'rot_1', 'rot_2', 'rot_3')
SH_C0 = 0.28209479177387814


def save_ply(path, means, scales, rotations, rgbs, opacities, normals=None):
    """Binary little-endian PLY, vertex properties ``PLY_ATTRS``: colours as the DC spherical-harmonic coefficient
    (rgb - 0.5) / C0, an isotropic [N, 1] scale column repeated to three, normals zero unless given; opacities stay logits and
    scales logs (the callers pass the raw parameters, as the reference does)."""
    means = np.asarray(means, dtype=np.float32)
    normals = np.zeros_like(means) if normals is None else np.asarray(normals, dtype=np.float32)
    scales = np.asarray(scales, dtype=np.float32)
    if scales.shape[1] == 1:
        scales = np.tile(scales, (1, 3))
    colors = (np.asarray(rgbs, dtype=np.float32) - 0.5) / SH_C0
    table = np.concatenate((means, normals, colors, np.asarray(opacities, dtype=np.float32).reshape(-1, 1), scales,
                            np.asarray(rotations, dtype=np.float32)), axis=1).astype('<f4')
    if table.shape[1] != len(PLY_ATTRS):
        raise ValueError(f"expected {len(PLY_ATTRS)} columns, got {table.shape[1]}")
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join(f"property float {a}\n" for a in PLY_ATTRS) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())
    return path


def export_params_ply(params, path):
    """``params`` (the reference's dict, tensors or arrays: /root/reference/scripts/export_ply.py:63-73) -> ``path``."""
    def a(k):
        v = params[k]
        return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    return save_ply(path, a('means3D'), a('log_scales'), a('unnorm_rotations'), a('rgb_colors'), a('logit_opacities'))
