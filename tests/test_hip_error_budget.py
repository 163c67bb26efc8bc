"""Error budget of the end-to-end gradients: how much of the difference to the oracle comes from the
fast alpha path (alpha = o * 2^p2 on a log2(e)-scaled conic with v_exp_f32, T / (1 - alpha) with v_rcp_f32) and how much
is inherent in the reference algorithm (T_final = 1 - alpha_image, hard thresholds).  The same scene is run through the
shipped library and through the exact-alpha MEASUREMENT build (lib/libdgr_hip_exact.so: expf on the reference's own
expression, IEEE division), each in its own process, with BASELINE's loss scaling (pixel gradients N(0,1)/(H W))."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = os.path.join(ROOT, "diff-gaussian-rasterization_amd", "lib", "libdgr_hip_exact.so")


def run(lib, P, W, H):
    env = dict(os.environ)
    env.pop("DGR_HIP_LIB", None)
    if lib:
        env["DGR_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "error_budget.py"), str(P), str(W), str(H)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("shape", [(100000, 640, 480), (500000, 1920, 1080)])
def test_fast_alpha_stays_inside_the_references_own_spread(shape):
    assert os.path.exists(EXACT), "make -C diff-gaussian-rasterization_amd builds the measurement library"
    fast, exact = run(None, *shape), run(EXACT, *shape)
    assert fast["lib"] == "default" and exact["lib"].endswith("libdgr_hip_exact.so")
    for d in (fast, exact):
        assert d["integer_path_exact"] and d["n_contrib_mismatch"] <= 2
    print("\n[error budget]", shape, {v: {k: "%.1e" % x["max_abs"] for k, x in d["end_to_end"].items()}
                                       for v, d in (("fast", fast), ("exact", exact))})
    # measured: config 3 fast 5.8e-5 / exact 6.7e-6 abs on dL_dview (scale 1.6); 640x480 5.2e-5 / 1.5e-5.
    # The reference itself moves by 5e-4 there between an FMA and a non-FMA build of its own sources.
    for d in (fast, exact):
        assert d["end_to_end"]["dL_dview"]["max_abs"] < 2e-4
        for k in ("dL_dmeans3D", "dL_dscales"):
            assert d["end_to_end"][k]["max_abs"] < 2e-4 and d["isolated"][k]["max_abs"] < 1e-4
    # with alpha evaluated as the reference writes it the stage-isolated backward agrees to rounding
    for k, x in exact["isolated"].items():
        assert x["max_abs"] <= 1e-5 * max(1.0, x["scale"]), k
    assert exact["end_to_end"]["dL_dview"]["max_abs"] <= fast["end_to_end"]["dL_dview"]["max_abs"]
