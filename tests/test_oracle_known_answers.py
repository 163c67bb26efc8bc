"""Pins the CPU oracle to the known-answer table of SURVEY.md Appendix C.

Those values were recorded by the survey from a CPU execution of the reference's own sources on the
synth-v1 inputs (the reference ships no tests or golden vectors of its own).  The survey's build bound
the reference's unqualified exp/sqrt/ceil to the C double functions; the oracle's C-math build restates
exactly that and must reproduce every digit.  The float-math build (what nvcc does, and what the GPU
tests compare against) must agree with it to rounding noise.
"""
import numpy as np
import pytest

from dgr_amd.synth import make_scene, sha16

# (P, W, H, deg) -> Appendix C columns (light variant)
KNOWN = {
    (10000, 256, 256, 0): dict(
        means_sha="dd02a0f8f00c0442", visible=9384, sum_radii=80637, max_radius=16, radii_sha="059d16ea0b759cbd",
        R=33704, list_mean=131.7, list_max=175, sum_n_contrib=7517087, color=87338.847, depth=150035.715,
        alpha=53164.300, median=184287.010, c000=0.56643301,
        dview=[+4.726215e-01, +3.502292e-01, +1.156665e-01, 0, +1.211684e-01, -2.053160e-01, -1.180594e-01, 0,
               -2.573290e-01, +5.135866e-01, -2.080060e-01, 0, -2.883756e-01, +1.813904e-01, -1.363746e-01, 0],
        gmax=dict(dL_dmeans3D=9.7e-2, dL_dopacity=2.0e-3, dL_dscales=1.0e-1, dL_drotations=3.8e-3, dL_dsh=5.3e-5)),
    (100000, 640, 480, 3): dict(
        means_sha="be3b5ffff70d420b", visible=87665, sum_radii=737518, max_radius=15, radii_sha="a20a1623bad630ac",
        R=332318, list_mean=276.9, list_max=336, sum_n_contrib=79527935, color=452621.003, depth=678213.701,
        alpha=298156.216, median=647604.466, c000=0.62521714,
        dview=[-3.450621e-01, +2.552776e-01, +1.151826e-01, 0, -1.766948e-01, -1.966203e-02, +8.297658e-02, 0,
               +2.150890e-02, -6.519964e-01, +3.327241e-01, 0, -5.430901e-02, -4.144245e-01, +2.108066e-01, 0],
        gmax=dict(dL_dmeans3D=7.1e-2, dL_dopacity=4.0e-4, dL_dscales=5.9e-2, dL_drotations=7.3e-4, dL_dsh=3.3e-5)),
    (500000, 1920, 1080, 3): dict(
        means_sha="f1c9f89692e11a8a", visible=425824, sum_radii=3533817, max_radius=15, radii_sha="e5c9b48ef4f1074e",
        R=1654310, list_mean=202.7, list_max=261, sum_n_contrib=383941016, color=2964281.851, depth=4848955.363,
        alpha=1911324.401, median=5131457.176, c000=0.52706647,
        dview=[-3.277759e-01, -1.357393e-01, +9.185782e-02, 0, -1.309201e-01, +8.697420e-02, +2.854166e-03, 0,
               +4.315028e-01, -1.640987e+00, +1.460000e-01, 0, +3.842236e-01, -1.504349e+00, +1.134152e-01, 0],
        gmax=dict(dL_dmeans3D=3.2e-2, dL_dopacity=7.3e-5, dL_dscales=2.7e-2, dL_drotations=1.8e-4, dL_dsh=5.3e-6)),
}


def run_light(O, s, deg, **kw):
    st, out = O.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                              s.tanfovx, s.tanfovy, s.H, s.W, s.shs, deg, s.campos)
    g = O.light_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.proj, s.tanfovx, s.tanfovy,
                         s.gC, s.gD, s.gM, s.gV, s.gt, s.shs, deg, s.campos, out["opacity_map"], s.persp, **kw)
    return st, out, g


@pytest.mark.parametrize("cfg", list(KNOWN))
def test_appendix_c_known_answers(oracle, cfg):
    P, W, H, deg = cfg
    k = KNOWN[cfg]
    s = make_scene(P, W, H, 0)
    assert sha16(s.means) == k["means_sha"]  # generator itself
    oracle.use_cmath(True)
    try:
        st, out, g = run_light(oracle, s, deg)
    finally:
        oracle.use_cmath(False)
    r = out["radii"]
    assert int((r > 0).sum()) == k["visible"]
    assert int(r.sum()) == k["sum_radii"] and int(r.max()) == k["max_radius"]
    assert sha16(r) == k["radii_sha"]
    assert out["num_rendered"] == k["R"]
    rg = st.get("ranges").reshape(-1, 2)
    ln = rg[:, 1].astype(np.int64) - rg[:, 0]
    assert round(float(ln.mean()), 1) == k["list_mean"] and int(ln.max()) == k["list_max"]
    assert int(st.get("n_contrib").astype(np.int64).sum()) == k["sum_n_contrib"]
    for name, key in (("color", "color"), ("depth", "depth"), ("opacity_map", "alpha"), ("depth_median", "median")):
        assert f"{out[name].astype(np.float64).sum():.3f}" == f"{k[key]:.3f}", name
    assert f"{out['color'][0, 0, 0]:.8f}" == f"{k['c000']:.8f}"
    assert np.all(out["depth_var"] == 0)
    # pose gradient: the table prints 7 significant digits (half a unit of the last one = 5e-7 relative)
    np.testing.assert_allclose(g["dL_dview"].reshape(-1), np.array(k["dview"]), rtol=6e-7, atol=0)
    for name, v in k["gmax"].items():
        assert abs(np.abs(g[name]).max() - v) <= 0.06 * v, name  # table holds 2 significant digits


def test_float_math_build_agrees_with_cmath_build(oracle):
    """exp/sqrt binding moves no integer and only rounding-level floats on config 1."""
    s = make_scene(10000, 256, 256, 0)
    oracle.use_cmath(True)
    try:
        st_c, out_c, g_c = run_light(oracle, s, 0)
    finally:
        oracle.use_cmath(False)
    st_f, out_f, g_f = run_light(oracle, s, 0)
    assert np.array_equal(out_c["radii"], out_f["radii"])
    assert np.array_equal(st_c.get("point_list"), st_f.get("point_list"))
    assert np.array_equal(st_c.get("n_contrib"), st_f.get("n_contrib"))
    for k in ("color", "depth", "opacity_map", "depth_median"):
        assert np.abs(out_c[k] - out_f[k]).max() <= 1e-5, k
    np.testing.assert_allclose(g_f["dL_dview"], g_c["dL_dview"], rtol=1e-4, atol=1e-6)


# What the float build (= the bits the HIP kernels carry) may differ by from the C-math build (= Appendix C's digits), per
# config: radii that differ (float vs double sqrt / ceil), pixels whose median depth lands on another Gaussian (a T within
# an ulp of 0.5 -- the distance between two correct expf), measured on this repository's two builds.
DISTANCE = {
    (100000, 640, 480, 3): dict(radii=0, median_pixels=1),
    (500000, 1920, 1080, 3): dict(radii=1, median_pixels=0),
}
DISTANCE_REPORT = []


@pytest.mark.parametrize("cfg", list(DISTANCE))
def test_float_math_build_distance_at_configs_2_and_3(oracle, cfg):
    """The second hop of the parity chain HIP <-> float-math oracle <-> C-math oracle <-> Appendix C, at the sizes the
    benchmark runs: the integer path is the same (but for config 3's one radius, which touches no further tile), no
    termination (`n_contrib`) flips, images agree to 2e-6, the median depth flips on at most one pixel, the pose gradient
    agrees to 1e-4 of its scale and every per-Gaussian gradient to 3e-4 of its tensor's scale (the survey measured 5e-4
    between two builds of the reference itself)."""
    P, W, H, deg = cfg
    s = make_scene(P, W, H, 0)
    oracle.use_cmath(True)
    try:
        st_c, out_c, g_c = run_light(oracle, s, deg)
    finally:
        oracle.use_cmath(False)
    st_f, out_f, g_f = run_light(oracle, s, deg)
    want = DISTANCE[cfg]
    n_radii = int((out_c["radii"] != out_f["radii"]).sum())
    assert n_radii == want["radii"]
    assert np.array_equal(st_c.get("tiles_touched"), st_f.get("tiles_touched"))
    assert out_c["num_rendered"] == out_f["num_rendered"]
    assert np.array_equal(st_c.get("point_list"), st_f.get("point_list"))
    assert np.array_equal(st_c.get("ranges"), st_f.get("ranges"))
    flipped = int((st_c.get("n_contrib") != st_f.get("n_contrib")).sum())
    assert flipped == 0
    line = [f"{cfg}: radii differing {n_radii}, n_contrib flips {flipped}"]
    for k in ("color", "depth", "opacity_map"):
        dmax = float(np.abs(out_c[k].astype(np.float64) - out_f[k]).max())
        line.append(f"{k} max |d| {dmax:.1e}")
        assert dmax <= 2e-6, k
    med = np.abs(out_c["depth_median"].astype(np.float64) - out_f["depth_median"])
    n_med = int((med > 0).sum())
    line.append(f"median-depth pixels differing {n_med} (max |d| {float(med.max()):.2f})")
    assert n_med <= want["median_pixels"]
    scale = np.abs(g_c["dL_dview"]).max()
    dv = float(np.abs(g_f["dL_dview"].astype(np.float64) - g_c["dL_dview"]).max() / scale)
    line.append(f"dL_dview max |d| / scale {dv:.1e}")
    assert dv <= 1e-4
    for name in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"):
        dg = float(np.abs(g_f[name].astype(np.float64) - g_c[name]).max() / np.abs(g_c[name]).max())
        line.append(f"{name} {dg:.1e}")
        assert dg <= 3e-4, name
    DISTANCE_REPORT.append("  ".join(line))
    print(DISTANCE_REPORT[-1])


# ---- full variant (NG column and the "Full dL_dview" block, gU := gV)
KNOWN_FULL = {
    (10000, 256, 256, 0): dict(NG=472252, dview=[+2.444670e-02, +1.515201e-01, -4.218012e-02, 0, -3.048362e-02, -4.470351e-02,
                                                -6.645557e-03, 0, +1.255521e-01, +1.000811e-01, +9.958681e-03, 0,
                                                +4.999591e-02, +6.168350e-02, +7.114520e-03, 0]),
    (100000, 640, 480, 3): dict(NG=4602988, dview=[-1.162237e-02, +3.704931e-02, -3.354471e-02, 0, -2.120929e-02, +3.775639e-02,
                                                  +7.021485e-04, 0, +2.316311e-01, +1.868812e-02, -4.177905e-02, 0,
                                                  +1.500740e-01, +2.374977e-03, -4.642588e-02, 0]),
}


def run_full(O, s, deg, **kw):
    st, out = O.full_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                             s.tanfovy, s.H, s.W, s.shs, deg, s.campos)
    g = O.full_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx, s.tanfovy,
                        s.gC, s.gD, s.gV, s.shs, deg, s.campos, s.persp, **kw)
    return st, out, g


@pytest.mark.parametrize("cfg", list(KNOWN_FULL))
def test_appendix_c_full_variant(oracle, cfg):
    """The survey's full-variant numbers come from a run in which ComputePG's early-returning threads (pixels
    without a valid contributor, F/cr/backward.cu:875-878,935-938) left their __shared__ slot stale -- undefined
    behaviour that its CPU execution resolved deterministically.  `emulate_dropout` restates exactly that, and
    only with it do the survey's digits come out on the sparse config 1; the survey's own matrix was summed by
    concurrent float atomics, so its last 1-2 digits are noise (Appendix C)."""
    P, W, H, deg = cfg
    k = KNOWN_FULL[cfg]
    s = make_scene(P, W, H, 0)
    oracle.use_cmath(True)
    try:
        st, out, g = run_full(oracle, s, deg, emulate_dropout=True)
        _, _, g_defined = run_full(oracle, s, deg)
    finally:
        oracle.use_cmath(False)
    assert out["num_rendered"] == KNOWN[cfg]["R"] and out["num_related"] == k["NG"]
    assert sha16(out["radii"]) == KNOWN[cfg]["radii_sha"]
    ref = np.array(k["dview"])
    scale = np.abs(ref).max()
    assert np.abs(g["dL_dview"].reshape(-1) - ref).max() <= 2e-5 * scale
    # the well-defined semantics (every recorded pair consumed) differ from that run only through the artifact
    d = np.abs(g_defined["dL_dview"].reshape(-1) - ref).max() / scale
    assert d <= (0.2 if cfg[0] == 10000 else 2e-5)
    assert not g_defined["dL_dview"].reshape(-1)[[3, 7, 11, 15]].any()


def test_full_forward_outputs_differ_from_light_only_as_documented(oracle):
    """full blends the terminating Gaussian before stopping (F/cr/forward.cu:370-381), light drops it (L:368-373)."""
    s = make_scene(10000, 256, 256, 0)
    _, lo = oracle.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                                 s.tanfovy, s.H, s.W, s.shs, 0, s.campos)
    _, fo = oracle.full_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                                s.tanfovy, s.H, s.W, s.shs, 0, s.campos)
    assert np.array_equal(lo["radii"], fo["radii"]) and lo["num_rendered"] == fo["num_rendered"]
    assert np.all(fo["uncertainty"] >= lo["opacity_map"] - 1e-6)   # silhouette can only gain the dropped Gaussian
    assert np.mean(fo["uncertainty"] != lo["opacity_map"]) < 0.05
