"""Drop-in for diff-gaussian-rasterization-light's `diff_gaussian_rasterization` module.

Put this directory's parent (`.../diff-gaussian-rasterization_amd/light`) and the package root
(`.../diff-gaussian-rasterization_amd`) on PYTHONPATH; CG-SLAM's
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer` then
resolves to the MI355X implementation.
"""
from dgr_amd.light import (GaussianRasterizationSettings, GaussianRasterizer, _C, _RasterizeGaussians,  # noqa: F401
                           cpu_deep_copy_tuple, rasterize_gaussians)
