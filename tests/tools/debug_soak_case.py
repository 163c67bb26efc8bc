"""Looks at one failing draw of tests/tools/soak_parity.py (full variant): which rows differ, and whether the forward's
per-pixel valid-contributor counts differ (an alpha-threshold flip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import hip_helpers as hh
from util import make_scene
from oracle import oracle as O
O.use_cmath(False)
rng = np.random.default_rng(7)
def draw_scene(i):
    W = int(rng.choice([7, 16, 31, 64, 100, 129, 250, 321, 400])); H = int(rng.choice([5, 16, 47, 64, 97, 200, 300]))
    P = int(rng.integers(1, 30000)); s = make_scene(P, W, H, 1000 + i)
    mode = rng.choice(["as drawn", "translucent", "opaque"])
    if mode == "translucent": s = s._replace(opac=(s.opac * 0.12).astype(np.float32))
    elif mode == "opaque": s = s._replace(opac=np.minimum(1.0, s.opac * 0.2 + 0.85).astype(np.float32))
    return s, int(rng.integers(0, 4)), float(rng.choice([0.3, 1.0, 1.0, 2.5, 8.0])), mode
for i in range(150): draw_scene(i)
for i in range(73): s, deg, sm, mode = draw_scene(10000 + i)
npx = s.W * s.H
grads = tuple(g * npx ** 0.5 for g in (s.gC, s.gD, s.gV))
out, d = hh.hip_full_forward(s, deg)
g = hh.hip_full_backward(s, deg, out, grads=grads)
st, ref, gr = hh.oracle_full(O, s, deg, grads=grads)
nv_h, nv_o = hh.hip_state("n_valid", s, d), st.get("n_valid_contrib")
print("P", s.means.shape[0], s.W, s.H, "pixels with different n_valid:", int((nv_h != nv_o).sum()), "of", npx)
print("final_T max diff", float(np.abs(hh.hip_state("final_T", s, d) - st.get("final_T")).max()))
for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D"):
    a, b = g[k].reshape(len(g[k]), -1), gr[k].reshape(len(gr[k]), -1)
    err = np.abs(a - b).max(1) / np.abs(b).max()
    bad = np.nonzero(err > 2e-5)[0]
    print(k, "rows over the bar:", bad.tolist(), "errors", err[bad].round(6).tolist())
