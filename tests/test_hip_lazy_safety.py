"""Lazy status mode (DGR_SYNC_MODE=lazy: no host wait inside a forward) when the instance count GROWS under it.

The capacity a lazy forward renders with is a guess (1.5 x the largest count seen for the shape).  A frame past it has empty tile
lists; before round 9 it looked like a frame (background colour, zero depth) for up to lazy_depth + 1 forwards.  Now
  * its images are NaN (the blend kernels see the overflow flag), its backward yields zero gradients,
  * check_async_errors() -- the call to make before optimizer.step() -- raises,
  * the shape runs strict until its count has settled, and a count that grows by more than 25 % between two reads (or comes
    within 20 % of the capacity) sends the shape to strict BEFORE anything overflows.
Reference behaviour for comparison: the reference sizes its binning buffer after a blocking read of num_rendered in every
forward (L/cuda_rasterizer/rasterizer_impl.cu:287-296) and so never overflows.
"""
import numpy as np
import pytest
import torch

from util import make_scene
import hip_helpers as hh

pytestmark = pytest.mark.gpu


def _leaves(s, scale=1.0):
    T = hh.T
    return [T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales * np.float32(scale), s.rots, s.view)]


def _step(rast, leaves, s, backward=True):
    m2 = torch.zeros((s.P, 3), device=hh.dev(), requires_grad=True)
    o = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
             viewmatrix=leaves[5], gt_depth=hh.T(s.gt))
    if backward:
        torch.autograd.backward([o[0], o[2]], [hh.T(s.gC), hh.T(s.gD[None])])
    return o


@pytest.fixture
def lazy(monkeypatch):
    from dgr_amd import light as L
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")
    yield L
    L._pending_status.clear()
    L._unsettled.clear()


def test_an_overflowed_lazy_forward_is_nan_not_an_empty_frame_and_raises_before_the_step(lazy):
    L = lazy
    from dgr_amd.multiview import make_settings
    s = make_scene(4000, 96, 64, 11)
    key = (hh.dev().index, s.P, s.H, s.W)
    rast = L.GaussianRasterizer(make_settings(s, 3, hh.dev()))
    for _ in range(3):                                   # the first call is strict and teaches the capacity; then lazy
        o = _step(rast, _leaves(s), s)
    L.check_async_errors()
    assert key not in L._unsettled and torch.isfinite(o[0]).all()
    R0 = L._capacity_cache[key]
    # the same P, every splat three times as large: three times the instances on this small frame, past 1.5 R0 + 4096
    big = _leaves(s, 3.0)
    o = _step(rast, big, s)
    assert torch.isnan(o[0]).all() and torch.isnan(o[2]).all() and torch.isnan(o[5]).all()      # colour, depth, alpha image
    for leaf in big[:5]:
        assert leaf.grad is not None and not leaf.grad.any()          # empty lists: nothing accumulates
    with pytest.raises(RuntimeError, match="overflow"):                # ... and this is what stands in front of optimizer.step()
        L.check_async_errors()
    assert key in L._unsettled
    # the shape is unsettled: the next forwards are strict (exact count, retried inside the call) and therefore right
    o1 = _step(rast, _leaves(s, 3.0), s)
    assert torch.isfinite(o1[0]).all() and L._capacity_cache[key] > 1.5 * R0 + 4096
    _, d_ref = hh.hip_forward(s._replace(scales=s.scales * np.float32(3.0)), 3)   # (the `_C` call: strict as well, here)
    assert np.array_equal(o1[0].detach().cpu().numpy(), d_ref["color"])
    for _ in range(4):                                    # steady count: three settled reads and the shape is lazy again
        _step(rast, _leaves(s, 3.0), s)
    L.check_async_errors()
    assert key not in L._unsettled
    n_before = len(L._pending_status)
    _step(rast, _leaves(s, 3.0), s)
    assert len(L._pending_status) == n_before + 1          # a lazy forward leaves a status word to be read later


def test_a_growing_count_sends_the_shape_to_strict_before_it_overflows(lazy):
    L = lazy
    from dgr_amd.multiview import make_settings
    s = make_scene(4000, 256, 192, 12)
    key = (hh.dev().index, s.P, s.H, s.W)
    rast = L.GaussianRasterizer(make_settings(s, 3, hh.dev()))
    scale = 1.0
    counts, strict_at = [], []
    for it in range(7):
        o = _step(rast, _leaves(s, scale), s)
        assert torch.isfinite(o[0]).all(), f"iteration {it}: a frame rendered past its capacity"
        L.check_async_errors()                           # (before the optimiser step of a real loop)
        counts.append(L._last_status[key][0] if key in L._last_status else L._capacity_cache[key])
        if key in L._unsettled:
            strict_at.append(it)
        scale *= 1.25                                     # ~50 % more instances per iteration
    assert counts[-1] > 3 * counts[0]
    assert strict_at and strict_at[0] <= 2, strict_at      # the growth was noticed at the first lazy read
