"""What a lane-private bit-mask walk of the forward blend could save: its second phase runs
max-over-lanes(popcount) steps per half-wave instead of the half-list's length.  Here: the per-pixel number of blended Gaussians
(n_valid of the full variant's forward, = the light forward's count but for the terminating Gaussian) at config 3, and its maximum
over each half of a quadrant (8 x 4 pixels = the 32 lanes that share a half-wave list)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from dgr_amd.synth import make_scene
import hip_helpers as hh

s = make_scene(500000, 1920, 1080, 0)
out, d = hh.hip_full_forward(s, 3)
nv = hh.hip_state("n_valid", s, d).reshape(s.H, s.W).astype(np.int64)
H8, W8 = (s.H // 8) * 8, (s.W // 8) * 8
blk = nv[:H8 - H8 % 4, :W8].reshape(-1, 4, W8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 32)   # halves of quadrants: 4 rows x 8 columns
mx, mean = blk.max(1), blk.mean(1)
print(f"config 3: blended Gaussians per pixel mean {nv.mean():.2f}; per half-quadrant (32 lanes): mean of the lane maximum {mx.mean():.2f}, "
      f"median {np.median(mx):.0f}, p90 {np.percentile(mx, 90):.0f}; max / mean inside a half {np.mean(mx / np.maximum(mean, 1e-9)):.2f}")
q = nv[:H8, :W8].reshape(H8 // 8, 8, W8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
print(f"per quadrant (64 lanes): mean of the lane maximum {q.max(1).mean():.2f}")
# entries per (half, tile list): the forward's tags per half give the number of list entries SOME lane of the half blended
tags = hh.hip_state("contribution_tags", s, d) if False else None
