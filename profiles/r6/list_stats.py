"""Tile-list statistics of BASELINE configs 4 and 5 (per-GPU views): mean / max list, share of lists above 512 / 1024 entries, and per segment
size (4 / 8 / 16 tiles) the share of segments holding a list above 1024 entries or more than 6144 keys.  Usage (GPU box): python profiles/r6/list_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from dgr_amd.synth import make_scene
import hip_helpers as hh
for (P,W,H) in ((2000000,1920,1080),(5000000,3840,2160)):
    s = make_scene(P, W, H, 0)
    _, d = hh.hip_forward(s, 3)
    tiles = ((W+15)//16)*((H+15)//16)
    rg = hh.hip_state("ranges", s, d).reshape(tiles,2).astype(np.int64)
    n = rg[:,1]-rg[:,0]
    gx=(W+15)//16
    print(P, W, H, "R", d["num_rendered"], "mean", n.mean(), "max", n.max(), "frac>1024", (n>1024).mean(), "frac>512", (n>512).mean())
    for seg in (4,8,16):
        rows = n.reshape(-1, gx)
        pad = (-gx) % seg
        rr = np.pad(rows, ((0,0),(0,pad)))
        sg = rr.reshape(rr.shape[0], -1, seg)
        print("  seg", seg, "segments with a list >1024:", (sg.max(axis=2)>1024).mean(), "keys>6144:", (sg.sum(axis=2)>6144).mean())
