"""MI355X-native differentiable Gaussian rasterizer: host side above the C ABI (include/dgr_hip.h).

`dgr_amd.light` mirrors diff-gaussian-rasterization-light/diff_gaussian_rasterization/__init__.py.
The drop-in module name `diff_gaussian_rasterization` is provided by the sibling directory
`light/` (add it to sys.path / PYTHONPATH).  There is no CPU fallback: importing the binding
without the built HIP library raises.
"""
