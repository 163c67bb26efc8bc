"""Per-thread options (include/dgr_hip.h: dgr_set_thread_option): two threads of one process render with different alpha modes at
the same time, and a backward runs under its forward's options on whatever thread the autograd engine picks."""
import threading

import numpy as np
import pytest
import torch

from util import make_scene
import hip_helpers as hh
from dgr_amd import _capi

pytestmark = pytest.mark.gpu


def _render(s, n=1):
    from dgr_amd import light as L
    from dgr_amd.multiview import make_settings
    rast = L.GaussianRasterizer(make_settings(s, 3, hh.dev()))
    out = None
    for _ in range(n):
        leaves = [hh.T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
        m2 = torch.zeros((s.P, 3), device=hh.dev(), requires_grad=True)
        o = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                 viewmatrix=leaves[5], gt_depth=hh.T(s.gt))
        torch.autograd.backward([o[0], o[2], o[3], o[4]], [hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gM[None]), hh.T(s.gV[None])])
        torch.cuda.synchronize()
        out = (o[0].detach().cpu().numpy(), leaves[5].grad.cpu().numpy(), leaves[0].grad.cpu().numpy())
    return out


def test_two_threads_hold_different_alpha_modes():
    s = make_scene(20000, 320, 240, 3)
    ref_exact = _render(s)
    _capi.set_option("alpha_mode", 1)
    try:
        ref_fast = _render(s)
    finally:
        _capi.set_option("alpha_mode", 0)
    assert not np.array_equal(ref_exact[0], ref_fast[0])               # the two modes differ in the last bits of the colour
    got = {}

    def worker(name, mode):
        torch.cuda.set_device(hh.dev())
        with torch.cuda.stream(torch.cuda.Stream(device=hh.dev())):
            with _capi.thread_options(alpha_mode=mode):
                got[name] = _render(s, n=6)                             # several rounds, so that the two threads overlap

    ta, tb = threading.Thread(target=worker, args=("fast", 1)), threading.Thread(target=worker, args=("exact", 0))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert _capi.get_option("alpha_mode") == 0
    assert np.array_equal(got["fast"][0], ref_fast[0]) and np.array_equal(got["exact"][0], ref_exact[0])
    # backward under the forward's mode (the engine's thread never called thread_options itself): float atomics order only
    for name, ref in (("fast", ref_fast), ("exact", ref_exact)):
        assert np.abs(got[name][1] - ref[1]).max() <= 2e-6 * np.abs(ref[1]).max(), name
    assert np.abs(ref_fast[1] - ref_exact[1]).max() > 2e-6 * np.abs(ref_exact[1]).max()   # ... which does tell the modes apart


def test_deterministic_gradients_for_one_thread_only():
    s = make_scene(20000, 320, 240, 4)
    with _capi.thread_options(deterministic_grads=1):
        a, b = _render(s), _render(s)
    assert _capi.get_option("deterministic_grads") == 0
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
