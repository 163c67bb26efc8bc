"""Where the lane lists' per-frame choice (option "lane_lists" = 2; bin_tiles' threshold: mean run of tiles per Gaussian, tile row and
segment > 2.5 -> quadrant lists) stands against the measured better mapping: synth-v1 at config 3 with every Gaussian's scale multiplied
by f, and the heavy-tailed scene with a growing share of big splats.  Per scene: instances per visible Gaussian, the blend kernels'
stage times under lane_lists = 1 (half-wave / paired) and 0 (quadrant), and what the frame chose by itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from dgr_amd import _capi
from dgr_amd.synth import make_scene, heavy_tail_scene
import hip_helpers as hh

_capi.load()
_capi.set_option("profile_every", 1)


def blend_times(s, lists, reps=6):
    _capi.set_option("lane_lists", lists)
    out, d = hh.hip_forward(s, 3)
    hh.hip_backward(s, 3, out)                      # warm
    _capi.profile_select("all")
    for st in _capi.profile_stages():
        _capi.profile_read(st)
    for _ in range(reps):
        out, d = hh.hip_forward(s, 3)
        hh.hip_backward(s, 3, out)
    torch.cuda.synchronize()
    t = {}
    for st in _capi.profile_stages():
        tot, n = _capi.profile_read(st)
        if n:
            t[st] = 1e3 * tot / n
    _capi.profile_select("")
    flag = int(hh.hip_state("sched_flag", s, d)[0] >> 2) & 1
    return t["render_fwd"], t["render_bwd"], flag, d


base = make_scene(500000, 1920, 1080, 0)
scenes = [(f"synth-v1, scales x {f}", base._replace(scales=base.scales * np.float32(f))) for f in (0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0)]
scenes += [(f"heavy tail, {100 * fr:g} % big", heavy_tail_scene(base, frac=fr)) for fr in (0.001, 0.003, 0.01, 0.03)]
print(f"{'scene':28s} {'R':>9s} {'R/visible':>9s} | half-wave fwd  bwd | quadrant fwd  bwd | better    | the frame chose")
for name, s in scenes:
    f1, b1, _, d = blend_times(s, 1)
    f0, b0, _, _ = blend_times(s, 0)
    _, _, flag, _ = blend_times(s, 2, reps=1)
    vis = int((d["radii"] > 0).sum())
    better = "quadrant" if f0 + b0 < f1 + b1 else "half-wave"
    chose = "quadrant" if flag else "half-wave"
    print(f"{name:28s} {d['num_rendered']:9d} {d['num_rendered'] / max(vis, 1):9.2f} | {f1:13.1f} {b1:5.1f} | {f0:12.1f} {b0:5.1f} | {better:9s} "
          f"({100 * ((f0 + b0) / (f1 + b1) - 1):+.1f} %) | {chose}{'' if chose == better else '   <-- not the better one'}")
_capi.set_option("lane_lists", 2)
