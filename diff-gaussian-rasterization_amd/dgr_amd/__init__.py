"""MI355X-native differentiable Gaussian rasterizer: host side above the C ABI (include/dgr_hip.h).

  dgr_amd.light      mirror of diff-gaussian-rasterization-light/diff_gaussian_rasterization/__init__.py
  dgr_amd.full       mirror of diff-gaussian-rasterization-full/diff_gaussian_rasterization/__init__.py
  dgr_amd.slam       CG-SLAM's render() call and differentiable pose -> camera-matrix helpers
  dgr_amd.multiview  views in flight on several streams, hipGraph capture of a step, fused gradient all-reduce
  dgr_amd.optim      fused sparse Adam for the Gaussian parameters
  dgr_amd.synth      the seeded synthetic scenes tests and bench.py share

The drop-in module name `diff_gaussian_rasterization` is provided by the sibling directories `light/` and `full/`
(add one of them to sys.path / PYTHONPATH).  There is no CPU fallback: the bindings raise if the HIP library
(`lib/libdgr_hip.so`, built by `make`) is missing.
"""
