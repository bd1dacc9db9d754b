"""Drop-in for the reference's un-vendored ``diff_gaussian_rasterization`` package.

``from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings``
(/root/reference/scripts/splatam.py:37, /root/reference/utils/recon_helpers.py:2)
resolves here when this repository is on ``sys.path``; the implementation is the
MI355X HIP rasterizer in ``splatam_amd`` (libsplat_hip.so).
"""
from splatam_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                    rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
