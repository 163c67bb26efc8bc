import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/diff-gaussian-rasterization_amd", "/root/repo/tests"]
import numpy as np
from util import make_scene
import hip_helpers as hh
from oracle import oracle as O
P, W, H, deg = 500000, 1920, 1080, 3
s = make_scene(P, W, H, 0)
grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
out, d = hh.hip_forward(s, deg)
st, ref = hh.oracle_forward(O, s, deg)
gr = hh.oracle_backward(O, st, s, deg, ref["opacity_map"], grads=grads)
nc_h, nc_o = hh.hip_state("n_contrib", s, d), st.get("n_contrib")
print("n_contrib mismatches", int((nc_h != nc_o).sum()))
g = hh.hip_backward(s, deg, out, grads=grads, alphas=ref["opacity_map"])
for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh"):
    a, b = np.asarray(g[k], np.float64), np.asarray(gr[k], np.float64)
    err = np.abs(a - b).reshape(P, -1).max(1)
    scale = np.abs(b).max()
    idx = np.argsort(-err)[:5]
    print(k, "scale", scale, "worst rel", err[idx] / scale, "ids", idx, "count >1e-5:", int((err > 1e-5 * scale).sum()))
i = int(np.argmax(np.abs(np.asarray(g["dL_dmeans2D"]) - gr["dL_dmeans2D"]).reshape(P, -1).max(1)))
print("worst Gaussian", i, "radius", d["radii"][i], "means2D", hh.hip_state("means2D", s, d).reshape(P, 2)[i], "opac", s.opac[i])
print("hip", np.asarray(g["dL_dmeans2D"])[i], "ref", gr["dL_dmeans2D"][i])
# second run of the HIP backward: atomic-order noise?
g2 = hh.hip_backward(s, deg, out, grads=grads, alphas=ref["opacity_map"])
print("run-to-run", np.abs(np.asarray(g2["dL_dmeans2D"]) - np.asarray(g["dL_dmeans2D"])).max())
