#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small regression vectors produced by the CPU oracle (float-math build).

These are NOT outputs of the reference (it cannot be built in this image and ships no vectors of its own); they
freeze the oracle -- which is pinned to SURVEY.md Appendix C by tests/test_oracle_known_answers.py -- on tiny
scenes so that (a) a later edit of the oracle cannot drift silently and (b) the GPU path has fixed expected values
that travel with the repository.  Inputs are regenerated from (P, W, H, seed) by dgr_amd.synth.make_scene.

  python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
from dgr_amd.synth import make_scene  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [  # name, variant, P, W, H, seed, deg, track_off, map_off
    ("light_deg0", "light", 1500, 64, 48, 11, 0, False, False),
    ("light_deg3", "light", 1500, 64, 48, 12, 3, False, False),
    ("light_deg3_track_off", "light", 1500, 64, 48, 12, 3, True, False),
    ("light_deg3_map_off", "light", 1500, 64, 48, 12, 3, False, True),
    ("full_deg0", "full", 1500, 64, 48, 13, 0, False, False),
    ("full_deg3", "full", 1500, 64, 48, 14, 3, False, False),
]


def main():
    for name, variant, P, W, H, seed, deg, toff, moff in CASES:
        s = make_scene(P, W, H, seed)
        grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
        d = dict(P=P, W=W, H=H, seed=seed, deg=deg, track_off=toff, map_off=moff)
        if variant == "light":
            st, out = O.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                      s.tanfovx, s.tanfovy, H, W, s.shs, deg, s.campos)
            g = O.light_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.proj, s.tanfovx,
                                 s.tanfovy, *grads, s.gt, s.shs, deg, s.campos, out["opacity_map"], s.persp,
                                 track_off=toff, map_off=moff)
        else:
            st, out = O.full_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                     s.tanfovx, s.tanfovy, H, W, s.shs, deg, s.campos)
            g = O.full_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                                s.tanfovy, grads[0], grads[1], grads[3], s.shs, deg, s.campos, s.persp)
        for k, v in out.items():
            d["out_" + k] = np.asarray(v)
        for k, v in g.items():
            d["grad_" + k] = v.astype(np.float32) if k != "dL_dsh" else v.astype(np.float32)
        d["point_list"] = st.get("point_list")
        d["ranges"] = st.get("ranges")
        d["n_contrib"] = st.get("n_contrib")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
