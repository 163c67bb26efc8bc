#!/usr/bin/env python3
"""Full-size parity beyond the suite's config 3 / view 0 case: other views of config 3, config 4's 2 M Gaussians.
Same checks and bars as tests/test_hip_light_parity.py at full size.  usage: python tests/tools/fullsize_parity.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import hip_helpers as hh  # noqa: E402
from util import assert_grad_close, assert_image_close  # noqa: E402
from dgr_amd.synth import make_scene  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.use_cmath(False)
CASES = [(500000, 1920, 1080, 1), (500000, 1920, 1080, 2), (500000, 1920, 1080, 5), (2000000, 1920, 1080, 0), (2000000, 1920, 1080, 3)]
bad = 0
for P, W, H, view in CASES:
    t0 = time.time()
    s = make_scene(P, W, H, seed=0, view_index=view)
    try:
        out, d = hh.hip_forward(s, 3)
        st, ref = hh.oracle_forward(O, s, 3)
        assert d["num_rendered"] == ref["num_rendered"] and np.array_equal(d["radii"], ref["radii"])
        assert np.array_equal(hh.hip_state("ranges", s, d), st.get("ranges"))
        assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
        for k in ("color", "depth", "depth_median", "opacity_map"):
            assert_image_close(d[k], ref[k], k)
        flips = float(np.mean(hh.hip_state("n_contrib", s, d) != st.get("n_contrib")))
        assert flips <= 1e-4
        grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
        gr = hh.oracle_backward(O, st, s, 3, ref["opacity_map"], grads=grads)
        g = hh.hip_backward(s, 3, out, grads=grads, alphas=ref["opacity_map"])
        for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
            assert_grad_close(g[k], gr[k], k, rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-4, outlier_rows=max(2, P // 20000))  # (a flipped pair perturbs the handful of Gaussians behind it)
        assert_grad_close(g["dL_dview"], gr["dL_dview"], "dL_dview", rel_to_max=5e-4, elem_rtol=1e-2, elem_frac=0.25)
        print(f"ok   P={P} {W}x{H} view {view}: R={d['num_rendered']}, n_contrib differs on {flips:.1e} of the pixels, {time.time() - t0:.0f} s", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"FAIL P={P} {W}x{H} view {view}: {str(e)[:300]}", flush=True)
# the full variant at the light variant's headline size (its own BASELINE config is config 2, covered by the suite)
for P, W, H, view in [(500000, 1920, 1080, 0)]:
    t0 = time.time()
    s = make_scene(P, W, H, seed=0, view_index=view)
    try:
        grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gV))
        out, d = hh.hip_full_forward(s, 3)
        g = hh.hip_full_backward(s, 3, out, grads=grads)
        st, ref, gr = hh.oracle_full(O, s, 3, grads=grads)
        assert np.array_equal(d["radii"], ref["radii"]) and d["num_rendered"] == ref["num_rendered"]
        assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
        for k in ("color", "depth", "uncertainty"):
            assert_image_close(d[k], ref[k], k)
        assert np.mean(hh.hip_state("n_contrib", s, d) != st.get("n_contrib")) <= 1e-4
        assert np.mean(hh.hip_state("n_valid", s, d) != st.get("n_valid_contrib")) <= 1e-4
        for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
            assert_grad_close(g[k], gr[k], k, rel_to_max=2e-5, elem_rtol=2e-3, elem_frac=1e-3, outlier_rows=max(2, P // 20000))
        assert_grad_close(g["dL_dview"], gr["dL_dview"], "dL_dview", rel_to_max=5e-4, elem_rtol=1e-2, elem_frac=0.25)
        print(f"ok   full variant P={P} {W}x{H} view {view}: R={d['num_rendered']}, NG={d['num_related']}, {time.time() - t0:.0f} s", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"FAIL full variant P={P} {W}x{H} view {view}: {str(e)[:300]}", flush=True)
sys.exit(1 if bad else 0)
