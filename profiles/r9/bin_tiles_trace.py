"""Phase time stamps of every bin_tiles workgroup on one frame (dgr_debug_bin_tiles_trace): where a dense / heavy segment's
time goes.  Usage: python profiles/r9/bin_tiles_trace.py [clustered|synth-v1|heavy_tail]   (env DGR_SEG_SHIFT / DGR_BT_* apply)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

from dgr_amd import _capi
from dgr_amd.synth import make_scene, cluster_scene, heavy_tail_scene
import hip_helpers as hh

scene = sys.argv[1] if len(sys.argv) > 1 else "clustered"
s = make_scene(500000, 1920, 1080, 0)
if scene == "clustered":
    s = cluster_scene(s)
elif scene == "heavy_tail":
    s = heavy_tail_scene(s)
lib = _capi.load()
hh.hip_forward(s, 3)            # warm-up (and the frame's capacity / hints)
hh.hip_forward(s, 3)
nwg = 68 * 30 * 4 + 64
buf = torch.zeros(3 * 2176 * 8, dtype=torch.int64, device="cuda")
lib.dgr_debug_bin_tiles_trace(buf.data_ptr())
hh.hip_forward(s, 3)
torch.cuda.synchronize()
lib.dgr_debug_bin_tiles_trace(None)
allw = buf.cpu().numpy().view(np.uint64).reshape(-1, 8)
t = allw[:2176]
per_wave = allw[2176:2 * 2176]
arrive = allw[2 * 2176:]
used = t[:, 0] != 0
per_wave = per_wave[used]
arrive = arrive[used]
t = t[used]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0                    # us
end = np.where(t[:, 4] != 0, t[:, 4], np.where(t[:, 1] != 0, t[:, 1], t[:, 0]))
print(f"{scene}: {len(t)} workgroups, kernel span {(end.max() - t0) / 100.0:.1f} us (first start to last stamp)")
gcount = (t[:, 6] & 0xffffffff).astype(np.int64)
npairs = (t[:, 6] >> 32).astype(np.int64)
tmax = (t[:, 7] & 0xffffffff).astype(np.int64)
dense = (t[:, 7] >> 63).astype(bool)
full = t[:, 4] != 0
d = lambda a, b: (t[:, b].astype(np.int64) - t[:, a].astype(np.int64)) / 100.0
print(f"workgroups that did work: {int(full.sum())}, dense: {int((dense & full).sum())}")
order = np.argsort(-(end.astype(np.int64) - t[:, 0].astype(np.int64)))
print("longest workgroups: start us | prologue+scan | load+count | ranges..place | sort | total | gcount n_pairs tmax dense")
for i in order[:16]:
    if not full[i]:
        continue
    print(f"  {start[i]:7.1f} | {d(0, 1)[i]:6.1f} | {d(1, 2)[i]:6.1f} | {d(2, 3)[i]:6.1f} | {d(3, 4)[i]:6.1f} | {d(0, 4)[i]:6.1f} | {gcount[i]} {npairs[i]} {tmax[i]} {int(dense[i])}" + (f"  (parts sorted after {d(3, 5)[i]:.1f} us of the sort phase)" if t[i, 5] else ""))
    if t[i, 5]:
        print("      per wave (us, entries; p = a 512-entry part): " + "  ".join(f"{(int(x) >> 16) / 100.0:.1f}/{'p' if int(x) & 0x8000 else ''}{int(x) & 0x7fff}" for x in per_wave[i] if x))
        print("      waves reach the barrier behind the sorts (us after the phase began): " + " ".join(f"{(int(x) - int(t[i, 3])) / 100.0:.1f}" for x in arrive[i] if x))
for name, m in (("all working", full), ("heavy (gcount > 3000)", full & (gcount > 3000)), ("light", full & (gcount <= 3000))):
    if m.sum():
        print(f"{name}: n {int(m.sum())}  mean us: prologue {d(0, 1)[m].mean():.1f} load+count {d(1, 2)[m].mean():.1f} place {d(2, 3)[m].mean():.1f} sort {d(3, 4)[m].mean():.1f} total {d(0, 4)[m].mean():.1f};  last start {start[m].max():.1f} last end {((end[m] - t0) / 100.0).max():.1f}")
