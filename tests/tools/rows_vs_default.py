#!/usr/bin/env python3
"""Runs one light scene through the default and the opt-in rows backward and lists the gradient rows on which they differ.
usage: rows_vs_default.py P W H seed deg scale_modifier [translucent] [precomp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import hip_helpers as hh
from util import make_scene
from oracle import oracle as O
from dgr_amd import _capi
O.use_cmath(False)
P, W, H, seed, deg = (int(x) for x in sys.argv[1:6]); sm = float(sys.argv[6])
s = make_scene(P, W, H, seed)
if "translucent" in sys.argv:
    s = s._replace(opac=(s.opac * 0.12).astype(np.float32))
kw = {}
if "precomp" in sys.argv:
    st0, _ = hh.oracle_forward(O, s, deg, scale_modifier=sm)
    kw = dict(colors_precomp=st0.get("rgb").reshape(-1, 3).copy(), cov3D_precomp=st0.get("cov3D").reshape(-1, 6).copy())
elif "precomp_colors" in sys.argv:
    st0, _ = hh.oracle_forward(O, s, deg, scale_modifier=sm)
    kw = dict(colors_precomp=st0.get("rgb").reshape(-1, 3).copy())
grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
res = {}
for rows in (0, 1):
    _capi.set_option("bwd_rows", rows)
    out, d = hh.hip_forward(s, deg, scale_modifier=sm, **kw)
    res[rows] = hh.hip_backward(s, deg, out, grads=grads, alphas=d["opacity_map"], scale_modifier=sm, **kw)
    print("rows", rows, "R", d["num_rendered"])
st, ref = hh.oracle_forward(O, s, deg, scale_modifier=sm, **kw)
gr = hh.oracle_backward(O, st, s, deg, d["opacity_map"], grads=grads, scale_modifier=sm, **kw)
for k in res[0]:
    a, b, r = (np.asarray(x[k], dtype=np.float64) for x in (res[0], res[1], gr)) if k in gr else (None, None, None)
    if a is None or a.size == 0:
        continue
    sc = np.abs(r).max() + 1e-30
    a2, b2, r2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1), r.reshape(r.shape[0], -1)
    ea, eb = np.abs(a2 - r2).max(1) / sc, np.abs(b2 - r2).max(1) / sc
    bad = np.nonzero(eb > 2e-5)[0]
    print(f"{k:14s} scale {sc:.3e} default worst {ea.max():.2e} rows worst {eb.max():.2e} bad rows {bad[:8]}")
    for i in bad[:4]:
        print("     row", i, "default", a2[i], "rows", b2[i], "oracle", r2[i])
bad = np.nonzero(np.abs(np.asarray(res[1]["dL_dmeans3D"]) - gr["dL_dmeans3D"]).max(1) / np.abs(gr["dL_dmeans3D"]).max() > 2e-5)[0]
for i in bad[:4]:
    print("Gaussian", i, "radius", d["radii"][i], "depth", st.get("depths")[i] if "depths" in st.names() else "?")
