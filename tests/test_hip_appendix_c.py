"""The HIP kernels against SURVEY.md Appendix C, directly (no oracle in between).

Appendix C's table is the one artefact in this repository that descends from an execution of the reference's own sources
(`L/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu` and the full variant's, run on a CPU by the survey with the C
double `exp / sqrt / ceil` bound to the unqualified calls).  Every other GPU test compares the kernels with `oracle/`'s
float-math build; this one takes the table's digits (`KNOWN`, `KNOWN_FULL` of tests/test_oracle_known_answers.py) and holds
the HIP outputs against them, at configs 1-3 (light) and 1-2 (full):

* integer path -- visible count, sum / max of radii, sha256 of `radii`, `num_rendered`, tile-list mean / max, the sum of
  `n_contrib`, full: `num_related` (NG) -- EXACTLY.  One documented exception: at config 3 the radius of Gaussian 360449 is 5
  under float `sqrt / ceil` (what nvcc binds for float arguments, `L/cuda_rasterizer/forward.cu:221-224`; what the kernels
  and the oracle's float build evaluate) and 6 under the survey's double binding; the test asserts that this is the ONLY
  radius that differs (patching it reproduces the table's sha) and that it moves nothing else of the integer path.
* the four image sums (fp64 sums of the fp32 images): deviation printed, bar 2e-9 of the sum + the table's own 5e-4 of
  print resolution, except where a median-depth threshold falls differently (config 2: ONE pixel, the distance between
  two `expf` -- tests/test_oracle_known_answers.py::test_float_math_build_distance_at_configs_2_and_3 measures the same
  pixel between the two oracle builds).
* `color[0,0,0]`: 8 printed digits, bar 2e-7.
* the 12 pose-gradient entries: 1e-4 of the matrix's scale (the survey measured 5e-4 between an FMA and a non-FMA build of
  the reference itself); measured deviation printed per config.
* full `dL_dview`: the table's run resolved ComputePG's undefined behaviour one particular way (see
  test_appendix_c_full_variant); the kernels implement the well-defined reading, which coincides with it at config 2
  (2e-4 of scale: float atomics in the table's own run, "last 1-2 digits vary") and differs on the sparse config 1 (bar
  0.2 of scale, as for the oracle's well-defined mode).
"""
import numpy as np
import pytest

from dgr_amd.synth import make_scene, sha16
import hip_helpers as hh
from test_oracle_known_answers import KNOWN, KNOWN_FULL

pytestmark = pytest.mark.gpu

# the one radius on which float and double sqrt / ceil disagree at config 3: (Gaussian, float value, double value)
RADIUS_EXCEPTION = {(500000, 1920, 1080, 3): (360449, 5, 6)}
# pixels whose median depth falls on another Gaussian than in the table's run (a T within an ulp of 0.5)
MEDIAN_FLIPS = {(100000, 640, 480, 3): 1}
REPORT = []


def check_integer_path(cfg, radii, num_rendered, ranges):
    k = KNOWN[cfg]
    r = radii.astype(np.int64)
    assert int((r > 0).sum()) == k["visible"]
    assert int(r.max()) == k["max_radius"]
    exc = RADIUS_EXCEPTION.get(cfg)
    if exc is None:
        assert int(r.sum()) == k["sum_radii"]
        assert sha16(radii) == k["radii_sha"]
    else:
        g, float_value, double_value = exc
        assert radii[g] == float_value and int(r.sum()) == k["sum_radii"] - (double_value - float_value)
        patched = radii.copy()
        patched[g] = double_value
        assert sha16(patched) == k["radii_sha"], "more than the one documented radius differs from Appendix C"
    assert num_rendered == k["R"]
    rg = ranges.reshape(-1, 2).astype(np.int64)
    ln = rg[:, 1] - rg[:, 0]
    assert round(float(ln.mean()), 1) == k["list_mean"] and int(ln.max()) == k["list_max"]
    assert int(ln.sum()) == k["R"]


@pytest.mark.parametrize("cfg", list(KNOWN))
def test_light_kernels_reproduce_appendix_c(cfg):
    P, W, H, deg = cfg
    k = KNOWN[cfg]
    s = make_scene(P, W, H, 0)
    assert sha16(s.means) == k["means_sha"]
    out, d = hh.hip_forward(s, deg)
    check_integer_path(cfg, d["radii"], d["num_rendered"], hh.hip_state("ranges", s, d))
    assert int(hh.hip_state("n_contrib", s, d).astype(np.int64).sum()) == k["sum_n_contrib"]
    line = [f"light {cfg}:"]
    for name, key in (("color", "color"), ("depth", "depth"), ("opacity_map", "alpha"), ("depth_median", "median")):
        got = float(d[name].astype(np.float64).sum())
        dev = abs(got - k[key])
        line.append(f"{key} sum {got:.3f} (table {k[key]:.3f}, |d| {dev:.1e})")
        bar = 2e-9 * abs(k[key]) + 5e-4
        if name == "depth_median" and cfg in MEDIAN_FLIPS:
            bar += 5.0 * MEDIAN_FLIPS[cfg]       # depths lie in [1, 6]: one flipped pixel moves the sum by < 5
        assert dev <= bar, (name, got, k[key])
    assert abs(float(d["color"][0, 0, 0]) - k["c000"]) <= 2e-7
    assert np.all(d["depth_var"] == 0)
    g = hh.hip_backward(s, deg, out)
    ref = np.array(k["dview"])
    dv = np.abs(g["dL_dview"].reshape(-1).astype(np.float64) - ref)
    scale = np.abs(ref).max()
    nz = ref != 0
    line.append(f"dL_dview max |d| / scale {dv.max() / scale:.2e}, worst entry relative {(dv[nz] / np.abs(ref[nz])).max():.2e}")
    assert dv.max() <= 1e-4 * scale, line[-1]
    assert not g["dL_dview"].reshape(-1)[[3, 7, 11, 15]].any()
    for name, v in k["gmax"].items():
        assert abs(np.abs(g[name]).max() - v) <= 0.06 * v, name      # the table holds two significant digits
    REPORT.append("  ".join(line))
    print(REPORT[-1])


@pytest.mark.parametrize("cfg", list(KNOWN_FULL))
def test_full_kernels_reproduce_appendix_c(cfg):
    P, W, H, deg = cfg
    k = KNOWN_FULL[cfg]
    s = make_scene(P, W, H, 0)
    out, d = hh.hip_full_forward(s, deg)
    check_integer_path(cfg, d["radii"], d["num_rendered"], hh.hip_state("ranges", s, d))
    assert d["num_related"] == k["NG"]
    assert int(hh.hip_state("n_valid", s, d).astype(np.int64).sum()) == k["NG"]
    g = hh.hip_full_backward(s, deg, out)
    ref = np.array(k["dview"])
    scale = np.abs(ref).max()
    dev = np.abs(g["dL_dview"].reshape(-1).astype(np.float64) - ref).max() / scale
    REPORT.append(f"full {cfg}: NG {d['num_related']}  dL_dview max |d| / scale {dev:.2e}")
    print(REPORT[-1])
    assert dev <= (0.2 if cfg[0] == 10000 else 2e-4), REPORT[-1]
    assert not g["dL_dview"].reshape(-1)[[3, 7, 11, 15]].any()
