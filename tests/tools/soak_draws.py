"""The random draws of tests/tools/soak_parity.py, importable (tests/tools/arbitrate_fp64.py re-creates single draws)."""
import os

import numpy as np

from util import make_scene


class Draws:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        # DGR_SOAK_HEAVY=1 (round 8): a third of the draws get a heavy tail -- a few percent of the Gaussians with an on-screen sigma
        # of up to 3x the frame (dgr_amd.synth.heavy_tail_scene): big rectangles for bin_segments' queue walk, clipped rectangles on
        # tiny frames, full queues.  Drawn from a generator of its own, so that the draws of older seeds stay what they were.
        self.heavy = np.random.default_rng(seed + 77777) if os.environ.get("DGR_SOAK_HEAVY") == "1" else None

    def scene(self, i):
        rng = self.rng
        W = int(rng.choice([7, 16, 31, 64, 100, 129, 250, 321, 400, 803]))
        H = int(rng.choice([5, 16, 47, 64, 97, 200, 300, 611]))
        P = int(rng.integers(1, 30000)) if rng.random() < 0.9 else int(rng.integers(30000, 200000))
        s = make_scene(P, W, H, 1000 + i)
        if rng.random() < 0.4:  # round 6: non-uniform scenes (dense segments, helper workgroups, long-list merges, the tile schedule)
            from dgr_amd.synth import cluster_scene
            s = cluster_scene(s, frac=float(rng.uniform(0.3, 0.95)), shrink=float(rng.uniform(0.02, 0.5)),
                              shift=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.3, 0.3))), seed=2000 + i)
        if self.heavy is not None and self.heavy.random() < 0.34:
            from dgr_amd.synth import heavy_tail_scene
            hi = float(self.heavy.uniform(8.0, 3.0 * max(W, H)))
            s = heavy_tail_scene(s, frac=float(self.heavy.uniform(0.002, 0.08)), sigma_px=(min(4.0, hi / 2), hi), seed=3000 + i)
        mode = rng.choice(["as drawn", "translucent", "opaque"])
        if mode == "translucent":
            s = s._replace(opac=(s.opac * 0.12).astype(np.float32))
        elif mode == "opaque":
            s = s._replace(opac=np.minimum(1.0, s.opac * 0.2 + 0.85).astype(np.float32))
        return s, int(rng.integers(0, 4)), float(rng.choice([0.3, 1.0, 1.0, 2.5, 8.0])), mode

    def light(self, i):
        """(scene, degree, scale modifier, mode, which inputs are precomputed) of light draw i; call with i = 0, 1, 2, ..."""
        s, deg, sm, mode = self.scene(i)
        return s, deg, sm, mode, int(self.rng.integers(0, 6))
