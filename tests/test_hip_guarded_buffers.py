"""Memory safety of the kernels on awkward shapes: every device buffer the ctypes binding allocates for a call -- the three
opaque state buffers, the output images, the backward scratch, the gradient arena -- is placed between two 64 KiB guard
regions, and after forward + backward no guard byte may have changed.  (Round 2 found an out-of-bounds store this way,
tests/test_hip_edge_cases.py::test_overflow_writes_stay_inside_the_state_buffers; a memory fault only shows when the
overrun leaves the allocator's slack.)  Shapes: images that are not multiples of the 16 x 16 tile, fewer Gaussians than one
256-thread block, one more than a block, lists longer than a staging batch, an empty view."""
import numpy as np
import pytest
import torch

from util import make_scene
import hip_helpers as hh

pytestmark = pytest.mark.gpu
G = 1 << 16


class GuardedTorch:
    """Stands in for the `torch` module inside dgr_amd.light / dgr_amd.full: empty() and zeros() hand out views into
    guard-padded allocations and remember them."""

    def __init__(self):
        self.live = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _alloc(self, shape, zero, dtype=torch.float32, device=None):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        whole = torch.full((nbytes + 2 * G,), 0xAB, dtype=torch.uint8, device=device)
        inner = whole[G:G + nbytes]
        if zero:
            inner.zero_()
        self.live.append((whole, nbytes, tuple(shape), dtype))
        return inner.view(dtype).view(tuple(shape))

    def empty(self, *shape, **kw):
        return self._alloc(shape, False, **kw)

    def zeros(self, *shape, **kw):
        return self._alloc(shape, True, **kw)

    def check(self):
        torch.cuda.synchronize()
        assert self.live
        for whole, nbytes, shape, dtype in self.live:
            lo, hi = whole[:G], whole[G + nbytes:]
            assert int((lo != 0xAB).sum()) == 0, f"bytes written below a {dtype} buffer of shape {shape}"
            assert int((hi != 0xAB).sum()) == 0, f"bytes written above a {dtype} buffer of shape {shape}"
        n = len(self.live)
        self.live.clear()
        return n


SHAPES = [  # P, W, H, seed, scale_modifier
    (1, 7, 5, 1, 1.0), (255, 17, 33, 2, 1.0), (257, 100, 47, 3, 2.5), (5000, 321, 200, 4, 1.0), (3000, 48, 48, 5, 6.0),
    (20000, 31, 16, 6, 1.0), (777, 803, 64, 7, 0.3),
]


@pytest.mark.parametrize("P,W,H,seed,sm", SHAPES)
@pytest.mark.parametrize("mode", [(False, False), (True, False), (False, True)])
def test_light_forward_and_backward_stay_inside_their_buffers(monkeypatch, P, W, H, seed, sm, mode):
    from dgr_amd import light as L
    gt_ = GuardedTorch()
    monkeypatch.setattr(L, "torch", gt_)
    monkeypatch.setattr(L, "_C", L._CtypesC)  # the binding that allocates in Python (the compiled one allocates in C++)
    L._capacity_cache.pop((hh.dev().index, P, H, W), None)
    s = make_scene(P, W, H, seed)
    out, d = hh.hip_forward(s, 3, scale_modifier=sm)
    n_fwd = gt_.check()
    assert n_fwd >= 11 and np.all(np.isfinite(d["color"]))
    g = hh.hip_backward(s, 3, out, track_off=mode[0], map_off=mode[1], scale_modifier=sm)
    assert gt_.check() >= 3  # gradient arena, dL_dview, scratch
    assert all(np.all(np.isfinite(v)) for v in g.values())
    # an empty view: every Gaussian behind the camera
    s0 = s._replace(means=(s.means - 2.0 * (s.means - s.campos)).astype(np.float32))
    out, d = hh.hip_forward(s0, 3, scale_modifier=sm)
    gt_.check()
    hh.hip_backward(s0, 3, out, track_off=mode[0], map_off=mode[1], scale_modifier=sm)
    gt_.check()


@pytest.mark.parametrize("P,W,H,seed,sm", SHAPES[:5])
def test_full_forward_and_backward_stay_inside_their_buffers(monkeypatch, P, W, H, seed, sm):
    from dgr_amd import full as F
    from dgr_amd import light as L
    gt_ = GuardedTorch()
    monkeypatch.setattr(F, "torch", gt_)
    monkeypatch.setattr(L, "torch", gt_)  # (helpers of the light module allocate for both variants)
    monkeypatch.setattr(F, "_C", F._CtypesC)
    s = make_scene(P, W, H, seed)
    out, d = hh.hip_full_forward(s, 3)
    assert gt_.check() >= 6 and np.all(np.isfinite(d["color"]))
    g = hh.hip_full_backward(s, 3, out)
    assert gt_.check() >= 2
    assert all(np.all(np.isfinite(v)) for v in g.values())
