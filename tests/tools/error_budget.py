#!/usr/bin/env python3
"""End-to-end gradient error of the HIP path against the oracle for one scene, as JSON on the last line.  The alpha mode
comes from the environment (DGR_FAST_ALPHA: 0 = the default, the reference's expression with the host's bits; 1 = the
fast_alpha option: v_exp_f32 on a log2(e)-scaled conic, v_rcp_f32), one process per mode.  usage: error_budget.py P W H [deg]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import hip_helpers as hh  # noqa: E402
from dgr_amd.synth import make_scene  # noqa: E402
from oracle import oracle as O  # noqa: E402

P, W, H = (int(x) for x in sys.argv[1:4])
deg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
O.use_cmath(False)
s = make_scene(P, W, H, 0)
out, d = hh.hip_forward(s, deg)
st, ref = hh.oracle_forward(O, s, deg)
from dgr_amd import _capi  # noqa: E402
res = {"lib": os.environ.get("DGR_HIP_LIB", "default"), "fast_alpha": _capi.get_option("fast_alpha"), "P": P, "W": W, "H": H,
       "integer_path_exact": bool(d["num_rendered"] == ref["num_rendered"] and np.array_equal(d["radii"], ref["radii"])
                                  and np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))),
       "n_contrib_mismatch": int((hh.hip_state("n_contrib", s, d) != st.get("n_contrib")).sum())}
for k in ("color", "depth", "depth_median", "opacity_map"):
    a, b = d[k].astype(np.float64), ref[k].astype(np.float64)
    res["img_" + k] = {"max_abs": float(np.abs(a - b).max()), "differing_values": int((d[k] != ref[k]).sum()),
                       "frac_over_1e-5": float(np.mean(np.abs(a - b) > 1e-5 * np.maximum(1.0, np.abs(b))))}
# the metric's loss scaling: pixel-gradient images N(0,1)/(H W), as bench.py's grad_max_abs_err
gr = hh.oracle_backward(O, st, s, deg, ref["opacity_map"])
for label, alphas in (("end_to_end", None), ("isolated", ref["opacity_map"])):
    g = hh.hip_backward(s, deg, out, alphas=alphas)
    res[label] = {k: {"max_abs": float(np.abs(g[k].astype(np.float64) - gr[k]).max()), "scale": float(np.abs(gr[k]).max())}
                  for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dview")}
print(json.dumps(res))
