"""Parity on a HEAVY-TAILED scene (dgr_amd.synth.heavy_tail_scene): 1 % of the Gaussians with an on-screen sigma of 20 .. 150 px --
3-sigma rectangles of hundreds to thousands of tiles -- among synth-v1's small splats.  This is what a SLAM map hands the
front end: the reference's duplicateWithKeys walks a whole rectangle with one thread for exactly that case
(L/cuda_rasterizer/rasterizer_impl.cu:70-111; the rectangle: */cuda_rasterizer/forward.cu:229-237, auxiliary.h:46-56).  At 1080p /
500 k those 1 % own two thirds of the 5.07 M tile instances and every tile list holds 620 entries on average (203 on synth-v1).

Same bars as the uniform scenes: radii, num_rendered, ranges, point_list and the threshold-carrying images bit for bit, colour and
depth to 1e-6 on every value, gradients at 1e-5 of each tensor's scale stage-isolated and end to end; both variants; both binning
paths agree."""
import numpy as np
import pytest

from dgr_amd.synth import heavy_tail_scene
from util import assert_grad_close, make_scene
import hip_helpers as hh
from test_hip_light_parity import assert_images_carry_the_references_bits, check_backward

pytestmark = pytest.mark.gpu

SIZES = [(100000, 1920, 1080, 3, 0), (500000, 1920, 1080, 3, 0)]


def test_the_scene_is_heavy_tailed():
    s = heavy_tail_scene(make_scene(100000, 1920, 1080, 0))
    _, d = hh.hip_forward(s, 1)
    r = d["radii"]
    big = r > 60
    assert 500 < int(big.sum()) < 1500 and int(r.max()) > 400
    rect_tiles = lambda rr: (2.0 * rr / 16.0 + 1.0) ** 2  # noqa: E731  (rectangle of a splat in the middle of the frame)
    assert rect_tiles(r[big].astype(np.float64)).sum() > 2.0 * rect_tiles(r[(r > 0) & ~big].astype(np.float64)).sum()


@pytest.mark.parametrize("case", SIZES + [(20000, 256, 256, 3, 1)])
def test_heavy_tail_forward_against_the_oracle(oracle, case):
    P, W, H, deg, seed = case
    s = heavy_tail_scene(make_scene(P, W, H, seed))
    _, d = hh.hip_forward(s, deg)
    st, ref = hh.oracle_forward(oracle, s, deg)
    assert d["num_rendered"] == ref["num_rendered"] and np.array_equal(d["radii"], ref["radii"])
    assert np.array_equal(hh.hip_state("ranges", s, d), st.get("ranges"))
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert_images_carry_the_references_bits(d, st, ref, s)


@pytest.mark.parametrize("case", SIZES)
@pytest.mark.parametrize("mode", [(False, False), (False, True)])
def test_heavy_tail_backward_against_the_oracle(oracle, case, mode):
    P, W, H, deg, seed = case
    s = heavy_tail_scene(make_scene(P, W, H, seed))
    check_backward(oracle, s, deg, track_off=mode[0], map_off=mode[1])


@pytest.mark.parametrize("case", SIZES)
def test_heavy_tail_full_variant_against_the_oracle(oracle, case):
    P, W, H, deg, seed = case
    s = heavy_tail_scene(make_scene(P, W, H, seed))
    grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gV))
    out, d = hh.hip_full_forward(s, deg)
    g = hh.hip_full_backward(s, deg, out, grads=grads)
    st, ref, gr = hh.oracle_full(oracle, s, deg, grads=grads)
    assert np.array_equal(d["radii"], ref["radii"]) and d["num_rendered"] == ref["num_rendered"]
    assert d["num_related"] == ref["num_related"]
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert np.array_equal(hh.hip_state("n_contrib", s, d), st.get("n_contrib"))
    assert np.array_equal(d["uncertainty"], ref["uncertainty"])
    for k in ("color", "depth"):
        a, b = d[k].astype(np.float64), ref[k].astype(np.float64)
        assert np.all(np.abs(a - b) <= 1e-6 * np.maximum(1.0, np.abs(b))), k
    tol = dict(rel_to_max=1e-5, elem_rtol=2e-3, elem_frac=1e-3)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert_grad_close(g[k], gr[k], k, **tol)
    assert_grad_close(g["dL_dview"], gr["dL_dview"], "dL_dview", rel_to_max=5e-5, elem_rtol=5e-3, elem_frac=0.1)


def test_heavy_tail_both_binning_paths_and_the_callback_path_agree(monkeypatch):
    """The segment binning (default), the global-counter binning (lds_count = 0) and the callback entry points on one frame."""
    from dgr_amd import _capi
    s = heavy_tail_scene(make_scene(100000, 1920, 1080, 0))
    _, d = hh.hip_forward(s, 3)
    _capi.set_option("lds_count", 0)
    try:
        _, dg = hh.hip_forward(s, 3)
    finally:
        _capi.set_option("lds_count", 1)
    monkeypatch.setenv("DGR_FORWARD_MODE", "callback")
    _, dc = hh.hip_forward(s, 3)
    monkeypatch.delenv("DGR_FORWARD_MODE")
    for other in (dg, dc):
        assert other["num_rendered"] == d["num_rendered"]
        for name in ("ranges", "point_list", "n_contrib"):
            assert np.array_equal(hh.hip_state(name, s, other), hh.hip_state(name, s, d)), name
        for k in ("color", "depth", "depth_median", "opacity_map"):
            assert np.array_equal(other[k], d[k]), k
