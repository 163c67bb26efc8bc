"""The blend kernels' lane mappings and alpha modes in their combinations, end to end against the oracle.

Round 8's default: the forward walks one list per half-wave and tags what it blended per half of a quadrant; the tracking backward
walks half-wave lists from those tags, the mapping backward pairs list entries by them.  Two combinations do not get the default
backward and are not reached by the other parity tests:
  * alpha_mode 2 (glibc's expf in the double pipe, kept for A/B): half-wave forward, but a backward with one list per quadrant
    wave from the 4-bit tags (the half-wave forms spill a register there) -- held against the oracle in ITS exp mode 1;
  * one list per quadrant wave in forward and backward (rounds 1-7's mapping): since round 9 the choice of the FRAME -- its binning
    kernel flags a frame of big splats in the frame's state, forward and backward branch on the flag (option "lane_lists":
    0 / 1 force one, 2 = by the frame; DGR_FWD_HALVES = 0 / 1 sets the initial value, covered through a child process).
Every mode of the backward (mapping + tracking, mapping only, tracking only) at the bars of tests/test_hip_light_parity.py:
threshold-carrying images bit for bit, gradients at 1e-5 of scale with no outlier rows."""
import os
import subprocess
import sys

import pytest

from util import make_scene
import hip_helpers as hh
from test_hip_light_parity import assert_images_carry_the_references_bits, check_backward

pytestmark = pytest.mark.gpu
MODES = [dict(), dict(map_off=True), dict(track_off=True)]


@pytest.fixture
def glibc_alpha(oracle):
    from dgr_amd import _capi
    _capi.load()
    _capi.set_option("alpha_mode", 2)
    oracle.set_exp_mode(1)
    yield
    oracle.set_exp_mode(0)
    _capi.set_option("alpha_mode", 0)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", [(10000, 256, 256, 3, 0), (100000, 640, 480, 3, 0)])
def test_glibc_alpha_with_the_half_wave_forward(oracle, glibc_alpha, case, mode):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    d, st, ref = check_backward(oracle, s, deg, what=f"glibc alpha P={P}", **mode)
    assert_images_carry_the_references_bits(d, st, ref, s)


@pytest.mark.parametrize("mode", MODES)
def test_heavy_tailed_scene_in_every_backward_mode(oracle, mode):
    """Big splats: lists of more than 64 entries per quadrant and batch (the mapping backward leaves those unpaired), entries that
    live in both halves everywhere."""
    from dgr_amd.synth import heavy_tail_scene
    s = heavy_tail_scene(make_scene(30000, 640, 480, 4), frac=0.05, sigma_px=(10, 200), seed=9)
    check_backward(oracle, s, 2, what="heavy tail, lane mappings", **mode)


@pytest.fixture
def lane_lists():
    from dgr_amd import _capi
    _capi.load()
    yield lambda v: _capi.set_option("lane_lists", v)
    _capi.set_option("lane_lists", 2)


def quadrant_flag(s, d):
    return int(hh.hip_state("sched_flag", s, d)[0] >> 2) & 1


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("lists", [0, 1])
def test_either_lane_mapping_forced(oracle, lane_lists, lists, mode):
    """lane_lists = 0 / 1: the frame's flag says what was asked for, and forward and backward that branch on it meet the bars."""
    s = make_scene(20000, 320, 200, 3)
    lane_lists(lists)
    d, st, ref = check_backward(oracle, s, 3, what=f"lane_lists={lists}", **mode)
    assert quadrant_flag(s, d) == 1 - lists
    assert_images_carry_the_references_bits(d, st, ref, s)


def test_the_frame_decides_its_lane_lists(oracle, lane_lists):
    """lane_lists = 2 (the default): small splats -> half-wave lists, big splats -> quadrant lists, decided by the binning kernel for
    the frame at hand; the images are the same bits whichever it takes."""
    import numpy as np
    from dgr_amd.synth import heavy_tail_scene
    small = make_scene(100000, 640, 480, 0)
    big = heavy_tail_scene(make_scene(30000, 640, 480, 4), frac=0.05, sigma_px=(10, 200), seed=9)
    for s, want in ((small, 0), (big, 1), (small, 0)):
        lane_lists(2)
        _, d = hh.hip_forward(s, 3)
        assert quadrant_flag(s, d) == want
        lane_lists(want)                                   # the OTHER mapping, forced (option 0 = quadrant lists = flag 1)
        _, e = hh.hip_forward(s, 3)
        assert quadrant_flag(s, e) == 1 - want
        for k in ("color", "depth", "depth_median", "opacity_map", "gau_related_pixels"):
            assert np.array_equal(d[k], e[k]), k
        for name in ("n_contrib", "point_list", "contribution_tags"):
            assert np.array_equal(hh.hip_state(name, s, d), hh.hip_state(name, s, e)), name
    lane_lists(2)
    check_backward(oracle, big, 2, what="big splats, the frame's own choice")
    check_backward(oracle, small, 3, what="small splats, the frame's own choice")


def test_a_replayed_graph_follows_the_frames_lane_lists(monkeypatch, lane_lists):
    """The choice is made on the device, per frame: a step recorded into a hipGraph takes whichever mapping the frame at hand asks for on
    every replay -- no re-capture when the splats grow past the threshold or shrink below it (a host-side policy would have frozen the
    captured step's kernels)."""
    import numpy as np
    import torch
    from dgr_amd import light as D
    from dgr_amd.multiview import CapturedStep, make_settings
    from util import assert_grad_close
    lane_lists(2)
    s = make_scene(20000, 320, 240, 6)
    flags = {}
    for f in (1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0):   # where does this scene cross ten tiles per Gaussian on screen?
        sf = s._replace(scales=s.scales * np.float32(f))
        _, d = hh.hip_forward(sf, 3)
        flags[f] = (quadrant_flag(sf, d), d["num_rendered"])
    lo = max(f for f, (q, _) in flags.items() if q == 0)
    hi = min(f for f, (q, _) in flags.items() if q == 1 and f > lo)
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")           # (the strict forwards above taught the shape its largest count)
    dev = hh.dev()
    rast = D.GaussianRasterizer(make_settings(s, 3, dev))
    means3D, shs, opac = hh.T(s.means).requires_grad_(), hh.T(s.shs).requires_grad_(), hh.T(s.opac).requires_grad_()
    scales, rots, view = hh.T(s.scales * np.float32(hi)).requires_grad_(), hh.T(s.rots).requires_grad_(), hh.T(s.view).requires_grad_()
    means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
    gt, gC, gD = hh.T(s.gt), hh.T(s.gC), hh.T(s.gD[None])
    leaves = [means3D, shs, opac, scales, rots, view]

    def step():
        for t in leaves + [means2D]:
            t.grad = None
        outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
        torch.autograd.backward([outs[0], outs[2]], [gC, gD])
        return [outs[0].detach()] + [t.grad for t in leaves]

    def snapshot(tensors):
        torch.cuda.synchronize()
        return [t.cpu().numpy().copy() for t in tensors]

    cap = CapturedStep(step)                               # recorded on the frame of BIG splats (quadrant lists)
    for f in (hi, lo, hi, lo):
        with torch.no_grad():
            scales.copy_(hh.T(s.scales * np.float32(f)))
        want = snapshot(step())                            # eager: this frame's own choice
        got = snapshot(cap.replay())
        cap.check()
        assert np.array_equal(got[0], want[0]), f
        for a, b in zip(got[1:], want[1:]):
            assert_grad_close(a, b, f"replay at scale x {f}", rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)
    D.check_async_errors()


def test_the_old_lane_mapping_in_a_child_process():
    """DGR_FWD_HALVES=0 is read once per process: one scene, all three backward modes, in a child."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import conftest  # (puts the package and the oracle on sys.path)\n"
            "from oracle import oracle as o\n"
            "from util import make_scene\n"
            "from test_hip_light_parity import check_backward, assert_images_carry_the_references_bits\n"
            "o.build()\n"
            "s = make_scene(20000, 320, 200, 3)\n"
            "for m in (dict(), dict(map_off=True), dict(track_off=True)):\n"
            "    d, st, ref = check_backward(o, s, 3, what='old lane mapping', **m)\n"
            "assert_images_carry_the_references_bits(d, st, ref, s)\n"
            "print('old mapping ok')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, DGR_FWD_HALVES="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "old mapping ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
