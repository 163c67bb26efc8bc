#!/usr/bin/env python3
"""Who is closer to float64 where the HIP backward and the float32 oracle disagree by more than the soak's bar?

  python tests/tools/arbitrate_fp64.py <seed> <light draw index> [...more indices]

Re-creates light draws of tests/tools/soak_parity.py (same random stream), runs the HIP forward + backward and the oracle, then
the float64 autograd formulation of tests/test_oracle_autograd.py (written from SURVEY.md's formulas, not from the oracle) on
the oracle's integer path, and prints per gradient tensor  max |HIP - f64|, max |oracle - f64|, max |HIP - oracle|  over the
tensor's scale -- overall and on the row where HIP and oracle differ most.  The scale modifier is folded into the scales
(exact for 8.0; the reference's dL_dscales is the derivative by the MODIFIED scale, so nothing is scaled back).  Draws with precomputed colours /
covariances hand the float64 formulation the same arrays (leaves `colors` / `cov3D`).  DGR_SOAK_HEAVY=1 as in the soak run.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "tools")]
import numpy as np  # noqa: E402

import hip_helpers as hh  # noqa: E402
from oracle import oracle as O  # noqa: E402
from soak_draws import Draws  # noqa: E402
from test_oracle_autograd import torch_light  # noqa: E402

O.build()
O.use_cmath(False)
seed, wanted = int(sys.argv[1]), sorted(int(a) for a in sys.argv[2:])
draws = Draws(seed)
for i in range(wanted[-1] + 1):
    s, deg, sm, mode, pre = draws.light(i)
    if i not in wanted:
        continue
    tag = f"light#{i} (seed {seed}) P={s.P} {s.W}x{s.H} deg={deg} sm={sm} {mode}"
    kw = {}
    if pre in (1, 3, 5):  # (as tests/tools/soak_parity.py hands them in: the oracle's own colours / covariances)
        st0, _ = hh.oracle_forward(O, s, deg, scale_modifier=sm)
        if pre in (1, 5):
            kw["colors_precomp"] = st0.get("rgb").reshape(-1, 3).copy()
        if pre in (3, 5):
            kw["cov3D_precomp"] = st0.get("cov3D").reshape(-1, 6).copy()
        tag += f" precomp={sorted(kw)}"
    modes = [(False, False), (True, False), (False, True)][i % 3]
    npx = s.W * s.H
    grads = tuple(g * npx ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    out, d = hh.hip_forward(s, deg, scale_modifier=sm, **kw)
    st, ref = hh.oracle_forward(O, s, deg, scale_modifier=sm, **kw)
    assert np.array_equal(d["opacity_map"], ref["opacity_map"]) and np.array_equal(hh.hip_state("n_contrib", s, d), st.get("n_contrib"))
    g = hh.hip_backward(s, deg, out, grads=grads, scale_modifier=sm, track_off=modes[0], map_off=modes[1], **kw)
    gr = hh.oracle_backward(O, st, s, deg, ref["opacity_map"], grads=grads, scale_modifier=sm, track_off=modes[0], map_off=modes[1], **kw)
    s64 = s._replace(scales=(s.scales.astype(np.float64) * sm))
    loss, leaves, img = torch_light(s64, deg, ref["radii"] > 0, st.get("point_list"), st.get("ranges"), st.get("n_contrib"),
                                    tuple(np.asarray(x, np.float64) for x in grads), **kw)
    fwd_err = max(float(np.abs(img[k].reshape(-1) - ref[k].astype(np.float64).reshape(-1)).max()) for k in ("color", "depth", "opacity_map"))
    loss.backward()
    m2 = np.zeros((s.P, 3))
    m2[img["_idx"], 0] = img["_pix"].grad[:, 0].numpy() * 0.5 * s.W   # (L/cr/backward.cu:583-584: d/d(ndc) = d/d(pixel) * 0.5 W)
    m2[img["_idx"], 1] = img["_pix"].grad[:, 1].numpy() * 0.5 * s.H
    gradient = lambda n: None if leaves[n].grad is None else leaves[n].grad.numpy()  # noqa: E731  (None: a leaf the draw's inputs bypass)
    # (the reference returns d/d(mod * scale): L/cuda_rasterizer/backward.cu:297,324-327 apply no `mod`)
    f64 = dict(dL_dmeans2D=m2, dL_dmeans3D=gradient("means3D"), dL_dscales=gradient("scales"), dL_drotations=gradient("rotations"),
               dL_dopacity=gradient("opacities"), dL_dsh=gradient("shs"))
    if "colors_precomp" in kw:
        f64["dL_dcolors"] = gradient("colors")
    if "cov3D_precomp" in kw:
        f64["dL_dcov3D"] = gradient("cov3D")
    f64 = {k: v for k, v in f64.items() if v is not None}
    print(tag, f"modes track_off={modes[0]} map_off={modes[1]}; float64 forward vs oracle images: {fwd_err:.1e}")
    if not modes[0]:  # (the two pose paths of forward.cu:196-234 -- projection and depth -- sum into one dL_dviewmatrix)
        f64["dL_dview"] = (leaves["view_ndc"].grad + leaves["view_depth"].grad).numpy()
    for k, t in f64.items():
        if modes[1] and k != "dL_dview":
            continue
        rows = 4 if k == "dL_dview" else s.P
        a, b, c = (np.asarray(x, np.float64).reshape(rows, -1) for x in (g[k], gr[k], t))
        scale = max(np.abs(c).max(), 1e-300)
        row = int(np.abs(a - b).max(1).argmax())
        print(f"  {k:14s} of scale: HIP-f64 {np.abs(a - c).max() / scale:.2e}  oracle-f64 {np.abs(b - c).max() / scale:.2e}  HIP-oracle "
              f"{np.abs(a - b).max() / scale:.2e} | on row {row} (largest HIP-oracle): HIP-f64 {np.abs(a[row] - c[row]).max() / scale:.2e}  "
              f"oracle-f64 {np.abs(b[row] - c[row]).max() / scale:.2e}")
