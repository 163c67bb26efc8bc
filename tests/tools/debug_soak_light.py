"""Re-runs chosen light draws of tests/tools/soak_parity.py (seed, indices) and reports which check fails."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import hip_helpers as hh
from util import make_scene
from oracle import oracle as O
O.use_cmath(False)
seed = int(sys.argv[1]); want = set(int(x) for x in sys.argv[2:])
rng = np.random.default_rng(seed)
def draw_scene(i):
    W = int(rng.choice([7, 16, 31, 64, 100, 129, 250, 321, 400])); H = int(rng.choice([5, 16, 47, 64, 97, 200, 300]))
    P = int(rng.integers(1, 30000)); s = make_scene(P, W, H, 1000 + i)
    mode = rng.choice(["as drawn", "translucent", "opaque"])
    if mode == "translucent": s = s._replace(opac=(s.opac * 0.12).astype(np.float32))
    elif mode == "opaque": s = s._replace(opac=np.minimum(1.0, s.opac * 0.2 + 0.85).astype(np.float32))
    return s, int(rng.integers(0, 4)), float(rng.choice([0.3, 1.0, 1.0, 2.5, 8.0])), mode
for i in range(max(want) + 1):
    s, deg, sm, mode = draw_scene(i)
    if i not in want: continue
    out, d = hh.hip_forward(s, deg, scale_modifier=sm)
    st, ref = hh.oracle_forward(O, s, deg, scale_modifier=sm)
    print(f"#{i} P={s.means.shape[0]} {s.W}x{s.H} deg={deg} sm={sm} {mode}: R hip {d['num_rendered']} oracle {ref['num_rendered']}; radii equal",
          np.array_equal(d["radii"], ref["radii"]))
    if not np.array_equal(d["radii"], ref["radii"]):
        bad = np.nonzero(d["radii"] != ref["radii"])[0]
        print("   radii differ at", bad[:10], d["radii"][bad[:10]], ref["radii"][bad[:10]])
        print("   scales", s.scales[bad[:3]] * sm, "means", s.means[bad[:3]])
    if d["num_rendered"] == ref["num_rendered"]:
        print("   ranges equal", np.array_equal(hh.hip_state("ranges", s, d), st.get("ranges")), "point_list equal",
              np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list")))
    for k in ("color", "depth", "depth_median", "opacity_map"):
        print("   ", k, "max abs diff", float(np.abs(d[k] - ref[k]).max()))
    print("    n_contrib differing pixels", int((hh.hip_state("n_contrib", s, d) != st.get("n_contrib")).sum()))
    npx = s.W * s.H
    grads = tuple(g * npx ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    for modes in ((False, False), (False, True)):
        g = hh.hip_backward(s, deg, out, grads=grads, alphas=ref["opacity_map"], scale_modifier=sm, track_off=modes[0], map_off=modes[1])
        gr = hh.oracle_backward(O, st, s, deg, ref["opacity_map"], grads=grads, scale_modifier=sm, track_off=modes[0], map_off=modes[1])
        dv = np.abs(g["dL_dview"] - gr["dL_dview"]).max() / np.abs(gr["dL_dview"]).max()
        line = f"    track_off={modes[0]} map_off={modes[1]}: dL_dview rel err {dv:.2e}"
        for k in ("dL_dmeans2D", "dL_dopacity"):
            a2, b2 = g[k].reshape(len(g[k]), -1), gr[k].reshape(len(gr[k]), -1)
            err = np.abs(a2 - b2).max(1) / max(np.abs(b2).max(), 1e-30)
            line += f"; {k} rows > 2e-5: {int((err > 2e-5).sum())} (worst {err.max():.1e})"
        print(line)
    # where are the rows over the bar?  (a flipped pair perturbs the Gaussians BEHIND it at one pixel: they all cover that pixel)
    g = hh.hip_backward(s, deg, out, grads=grads, alphas=ref["opacity_map"], scale_modifier=sm)
    gr = hh.oracle_backward(O, st, s, deg, ref["opacity_map"], grads=grads, scale_modifier=sm)
    a2, b2 = g["dL_dsh"].reshape(len(g["dL_dsh"]), -1), gr["dL_dsh"].reshape(len(gr["dL_dsh"]), -1)
    err = np.abs(a2 - b2).max(1) / np.abs(b2).max()
    bad = np.nonzero(err > 2e-5)[0]
    m2 = np.asarray(st.get("means2D")).reshape(-1, 2)[bad]
    rad = d["radii"][bad]
    print(f"    dL_dsh rows over the bar: {len(bad)}; their 2-D centres span x {m2[:,0].min():.0f}..{m2[:,0].max():.0f}, y {m2[:,1].min():.0f}..{m2[:,1].max():.0f}; radii {rad.min()}..{rad.max()}")
    # common pixels: pixels inside every bad Gaussian's 3-sigma circle
    ys, xs = np.mgrid[0:s.H, 0:s.W]
    cover = np.ones((s.H, s.W), bool)
    cnt = np.zeros((s.H, s.W), int)
    for (cx, cy), r in zip(m2, rad):
        inside = (xs - cx) ** 2 + (ys - cy) ** 2 <= float(r) ** 2
        cnt += inside
    print(f"    pixels covered by ALL {len(bad)} of them: {int((cnt == len(bad)).sum())}; by at least 80 %: {int((cnt >= 0.8 * len(bad)).sum())} of {s.W * s.H}")
