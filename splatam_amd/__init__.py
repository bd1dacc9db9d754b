"""splatam_amd -- MI355X (gfx950) Gaussian-splat rasterizer for SplaTAM's hot path.

``splatam_amd.rasterizer`` mirrors the reference's ``diff_gaussian_rasterization``
package on top of libsplat_hip.so (C ABI: include/splat_hip.h); the top-level
``diff_gaussian_rasterization`` package in this repository re-exports it under
the name the reference imports.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, get_sync_mode,  # noqa: F401
                         rasterize_gaussians, set_sync_mode)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians",
           "set_sync_mode", "get_sync_mode"]
