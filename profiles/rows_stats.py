"""Iteration statistics of the rows backward from the forward's own 16-bit tags (config 3): what the kernel's loop
count should be."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import hip_helpers as hh
from dgr_amd.synth import make_scene
s = make_scene(500000, 1920, 1080, 0)
out, d = hh.hip_forward(s, 3)
t16 = hh.hip_state("contribution_tags16", s, d).astype(np.uint32)
rg = hh.hip_state("ranges", s, d).reshape(-1, 2)
nc = hh.hip_state("n_contrib", s, d).reshape(s.H, s.W)
gx = (s.W + 15) // 16
bits = ((t16[:, None] >> np.arange(16)[None]) & 1).astype(np.int32)  # [R,16]
print("instances", len(t16), "tagged", int((t16 != 0).sum()), "(block,G) pairs", int(bits.sum()))
it_tile = it_batch = rounds = 0
for tile, (lo, hi) in enumerate(rg.astype(np.int64).tolist()):
    tx, ty = tile % gx, tile // gx
    total = int(min(hi - lo, int(nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].max(initial=0))))
    if total <= 0:
        continue
    b = bits[lo:lo + total]
    it_tile += int(b.sum(0).reshape(4, 4).max(1).sum())
    h = total
    while h > 0:
        l = max(0, h - 128)
        seg = b[l:h]
        # entries cut (RB_E = 384)
        cs = np.cumsum(seg.sum(1)[::-1])
        keep = int((cs <= 384).sum())
        seg = seg[len(seg) - keep:]
        it_batch += int(seg.sum(0).reshape(4, 4).max(1).sum())
        rounds += 1
        h -= keep
print("iterations: whole-tile lists", it_tile, " per-round", it_batch, " rounds", rounds, " tiles", len(rg))
