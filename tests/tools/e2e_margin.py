#!/usr/bin/env python3
"""Worst end-to-end gradient error (relative to each tensor's scale) over the suite's parity cases, with the flipped
pixels masked as the tests do: the number the end-to-end bar of tests/test_hip_light_parity.py is derived from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import hip_helpers as hh
from util import make_scene, mask_flipped_pixels
from oracle import oracle as O
from test_hip_light_parity import CASES, IMAGES, GRAD_NAMES
worst = {}
for P, W, H, deg, seed in CASES:
    s = make_scene(P, W, H, seed)
    grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    out, d = hh.hip_forward(s, deg)
    st, ref = hh.oracle_forward(O, s, deg)
    grads, n = mask_flipped_pixels(grads, hh.hip_state("n_contrib", s, d), st.get("n_contrib"), W, H, "",
                                   images=[(d[k], ref[k]) for k in IMAGES])
    gr = hh.oracle_backward(O, st, s, deg, ref["opacity_map"], grads=grads)
    g = hh.hip_backward(s, deg, out, grads=grads)
    row = {}
    for k in GRAD_NAMES + ("dL_dview",):
        sc = np.abs(gr[k]).max()
        if sc > 0:
            row[k] = float(np.abs(g[k].astype(np.float64) - gr[k]).max() / sc)
            worst[k] = max(worst.get(k, 0), row[k])
    print((P, W, H, deg), "masked", n, {k: "%.1e" % v for k, v in row.items()}, flush=True)
print("worst", {k: "%.1e" % v for k, v in worst.items()})
