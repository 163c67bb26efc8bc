"""The blend kernels' tile schedule (csrc/binning.hip: tile_schedule_kernel) and parity on a NON-UNIFORM scene.

synth-v1 spreads the Gaussians evenly (tile lists of 203 +- 20 % at config 3); a mapped room does not.  The schedule hands
the blend kernels their tiles heaviest first; it changes nothing but the order in which tiles are worked on, so every
parity bar of the uniform scenes must hold on a clustered one too (dgr_amd.synth.cluster_scene: lists from a few dozen to
over a thousand entries), and the table itself must be a permutation of the tiles, carry each tile's own range, be ordered
by descending list-length class and keep a class's tiles in image order, one contiguous part per XCD.
"""
import numpy as np
import pytest

from dgr_amd.synth import cluster_scene
from util import make_scene
import hip_helpers as hh
from test_hip_light_parity import assert_images_carry_the_references_bits, check_backward

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def schedule_always():
    """This file checks the schedule itself: build it for every frame (the default, "tile_schedule" = 2, skips it for a shape
    whose last reported frame had even lists -- tests/test_hip_front_end.py covers that policy)."""
    from dgr_amd import _capi
    _capi.set_option("tile_schedule", 1)
    yield
    _capi.set_option("tile_schedule", 2)


def sched_class(n):
    """csrc/dgr_common.h: sched_class"""
    if n == 0:
        return 0
    l = int(n).bit_length() - 1
    half = (n >> (l - 1)) & 1 if l else 0
    return min(31, 1 + 2 * l + half)


def check_schedule(s, d):
    tiles = ((s.W + 15) // 16) * ((s.H + 15) // 16)
    rg = hh.hip_state("ranges", s, d).reshape(tiles, 2).astype(np.int64)
    sc = hh.hip_state("tile_sched", s, d).reshape(tiles, 4).astype(np.int64)
    assert np.array_equal(np.sort(sc[:, 0]), np.arange(tiles)), "not a permutation of the tiles"
    assert np.array_equal(sc[:, 1:3], rg[sc[:, 0]]), "a slot does not carry its tile's range"
    n = sc[:, 2] - sc[:, 1]
    cls = np.array([sched_class(int(v)) for v in n])
    assert np.all(cls[:-1] >= cls[1:]), "slots are not ordered by descending list-length class"
    # inside a class every XCD (slot mod 8) holds a contiguous part of the class in image order, and the parts of successive
    # XCDs follow each other: walking the XCDs' parts one after the other gives the class's tiles in ascending order (up to the
    # order inside one 512-tile block of the schedule kernel's waves)
    slot = np.arange(tiles)
    per_wave = (((tiles + 15) // 16) + 511) // 512 * 512
    for c in np.unique(cls):
        sel = cls == c
        walk = np.concatenate([sc[sel & (slot % 8 == x), 0] for x in range(8)])
        group = (walk // per_wave) * per_wave + ((walk % per_wave) // 512) * 512
        assert np.all(group[:-1] <= group[1:]), f"class {c}: the XCDs' parts are not the class in image order"
    return n


@pytest.mark.parametrize("case", [(2000, 64, 48, 0, 1), (2000, 70, 45, 3, 2), (100000, 640, 480, 3, 0), (500000, 1920, 1080, 3, 0)])
@pytest.mark.parametrize("clustered", [False, True])
def test_schedule_is_a_heaviest_first_permutation(case, clustered):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    if clustered:
        s = cluster_scene(s)
    _, d = hh.hip_forward(s, deg)
    n = check_schedule(s, d)
    if clustered and P >= 100000:
        assert n.max() > 3 * max(1.0, n.mean()), "the clustered scene is supposed to have long lists"


def test_schedule_of_a_frame_with_more_tiles_than_the_kernel_keeps_in_registers():
    """3840x2160 has 32 400 tiles: the schedule kernel's second half (tiles beyond 8 per thread) re-reads the ranges."""
    s = make_scene(50000, 3840, 2160, 3)
    _, d = hh.hip_forward(s, 1)
    check_schedule(s, d)


def test_schedule_of_an_empty_frame():
    """No visible Gaussian: every list is empty, the schedule is still a permutation and the blend writes the background."""
    s = make_scene(500, 100, 60, 4)
    s = s._replace(means=(s.means * np.float32(0) + np.array([0, 0, -5], np.float32)))  # behind the camera
    _, d = hh.hip_forward(s, 0)
    assert d["num_rendered"] == 0
    check_schedule(s, d)
    assert np.allclose(d["color"], np.asarray(s.bg)[:, None, None])


@pytest.mark.parametrize("case", [(10000, 256, 256, 3, 0), (100000, 640, 480, 3, 0), (500000, 1920, 1080, 3, 0)])
def test_clustered_scene_forward_against_the_oracle(oracle, case):
    P, W, H, deg, seed = case
    s = cluster_scene(make_scene(P, W, H, seed))
    _, d = hh.hip_forward(s, deg)
    st, ref = hh.oracle_forward(oracle, s, deg)
    assert d["num_rendered"] == ref["num_rendered"] and np.array_equal(d["radii"], ref["radii"])
    assert np.array_equal(hh.hip_state("ranges", s, d), st.get("ranges"))
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert_images_carry_the_references_bits(d, st, ref, s)


@pytest.mark.parametrize("case", [(10000, 256, 256, 3, 0), (100000, 640, 480, 3, 0)])
@pytest.mark.parametrize("mode", [(False, False), (False, True)])
def test_clustered_scene_backward_against_the_oracle(oracle, case, mode):
    P, W, H, deg, seed = case
    s = cluster_scene(make_scene(P, W, H, seed))
    check_backward(oracle, s, deg, track_off=mode[0], map_off=mode[1])


@pytest.mark.parametrize("case", [(10000, 256, 256, 0, 0), (100000, 640, 480, 3, 0)])
def test_clustered_scene_full_variant_against_the_oracle(oracle, case):
    """The -full variant on a non-uniform frame (its blend kernels walk the same tile schedule): the bars of
    tests/test_hip_full_parity.py."""
    from util import assert_grad_close
    P, W, H, deg, seed = case
    s = cluster_scene(make_scene(P, W, H, seed))
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


def test_clustered_scene_batch_views_are_the_one_view_calls():
    """The batched entry points on a non-uniform map: every view's images, lists and schedule state as a one-view call's."""
    import torch
    import test_hip_batch as tb
    P, W, H, deg, V = 60000, 480, 320, 3, 3
    ss = [cluster_scene(x) for x in tb.scenes(P, W, H, V, 0)]
    out, _ = tb.batch_forward(ss, deg)
    for v, s in enumerate(ss):
        one, d1 = hh.hip_forward(s, deg)
        ov = tb.one_view_dict(out, v)
        assert ov[0] == one[0]
        for k in (1, 2, 3, 5, 6):  # colour, depth, median, alpha, radii
            assert torch.equal(ov[k], one[k]), (v, k)
        dv = {"num_rendered": ov[0], "geom": ov[7], "binning": ov[8], "img": ov[9]}
        for name in ("ranges", "point_list", "n_contrib"):
            assert np.array_equal(hh.hip_state(name, s, dv), hh.hip_state(name, s, d1)), (v, name)
        check_schedule(s, dv)


def test_clustered_scene_with_tight_culling_and_through_the_callback_path(oracle, monkeypatch):
    """The two other ways into the binning on a non-uniform frame: alpha-aware tile rectangles (fewer, shorter lists) and the
    callback entry points (binning buffer sized after a host read: COUNT_LDS_CALLBACK)."""
    from dgr_amd import _capi
    s = cluster_scene(make_scene(100000, 640, 480, 0))
    st, ref = hh.oracle_forward(oracle, s, 3)
    monkeypatch.setenv("DGR_FORWARD_MODE", "callback")
    _, d = hh.hip_forward(s, 3)
    monkeypatch.delenv("DGR_FORWARD_MODE")
    assert d["num_rendered"] == ref["num_rendered"]
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert_images_carry_the_references_bits(d, st, ref, s)
    check_schedule(s, d)
    _capi.set_option("tight_cull", 1)
    try:
        _, dt = hh.hip_forward(s, 3)
    finally:
        _capi.set_option("tight_cull", 0)
    assert dt["num_rendered"] < d["num_rendered"]
    for k in ("color", "depth", "depth_median", "opacity_map"):
        assert np.array_equal(dt[k], d[k]), k
    check_schedule(s, dt)
