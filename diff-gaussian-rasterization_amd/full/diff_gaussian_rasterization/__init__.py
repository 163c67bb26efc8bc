"""Drop-in for diff-gaussian-rasterization-full's `diff_gaussian_rasterization` module.

Put `.../diff-gaussian-rasterization_amd/full` and the package root on PYTHONPATH (instead of `.../light`)."""
from dgr_amd.full import (GaussianRasterizationSettings, GaussianRasterizer, _C, _RasterizeGaussians,  # noqa: F401
                          rasterize_gaussians)
