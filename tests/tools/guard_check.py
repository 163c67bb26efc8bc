#!/usr/bin/env python3
"""Runs one presized light forward on caller-owned state buffers that sit between guard regions and reports which guard
(if any) a kernel wrote into.  usage: guard_check.py P W H seed scale_modifier capacity [translucent]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from dgr_amd import _capi
from dgr_amd.synth import make_scene
import hip_helpers as hh
P, W, H, seed = (int(x) for x in sys.argv[1:5]); sm = float(sys.argv[5]); cap = int(sys.argv[6])
s = make_scene(P, W, H, seed)
if len(sys.argv) > 7:
    s = s._replace(opac=(s.opac * 0.12).astype(np.float32))
lib = _capi.load(); dev = hh.dev(); G = 1 << 20
def guarded(n):
    t = torch.full((n + 2 * G,), 0xAB, dtype=torch.uint8, device=dev)
    return t, t[G:G + n]
gb, geom = guarded(lib.dgr_geometry_bytes(P)); bb, binning = guarded(lib.dgr_binning_bytes(cap, W, H)); ib, img = guarded(lib.dgr_image_bytes(W, H))
f = lambda *sh: torch.empty(sh, device=dev)
color, depth, median, var, alpha = f(3, H, W), f(1, H, W), f(1, H, W), f(1, H, W), f(1, H, W)
radii = torch.empty(P, dtype=torch.int32, device=dev); unc = f(P, 1); px = torch.empty((P, 1), dtype=torch.int32, device=dev)
status = torch.zeros(4, dtype=torch.int32, device=dev)
T = hh.T; p = _capi.ptr
colors = torch.rand((P, 3), device=dev)
args = (P, 1, 0, p(T(s.bg)), W, H, p(T(s.means)), None, p(colors), p(T(s.opac)), p(T(s.scales)), sm, p(T(s.rots)), None,
        p(T(s.view)), p(T(s.proj)), p(T(s.campos)), s.tanfovx, s.tanfovy, 0, p(color), p(depth), p(median), p(alpha), p(T(s.gt)),
        p(var), p(unc), p(px), p(radii))
keep = [T(s.bg), T(s.means), T(s.opac), T(s.scales), T(s.rots), T(s.view), T(s.proj), T(s.campos), T(s.gt)]
args = (P, 1, 0, p(keep[0]), W, H, p(keep[1]), None, p(colors), p(keep[2]), p(keep[3]), sm, p(keep[4]), None, p(keep[5]), p(keep[6]),
        p(keep[7]), s.tanfovx, s.tanfovy, 0, p(color), p(depth), p(median), p(alpha), p(keep[8]), p(var), p(unc), p(px), p(radii))
rc = lib.dgr_light_forward_presized(_capi.stream_handle(), p(geom), p(binning), cap, p(img), p(status), *args)
torch.cuda.synchronize()
print("rc", rc, "status", status.tolist())
for name, whole, n in (("geometry", gb, geom.numel()), ("binning", bb, binning.numel()), ("image", ib, img.numel())):
    lo, hi = whole[:G].cpu().numpy(), whole[G + n:].cpu().numpy()
    bad_lo, bad_hi = np.nonzero(lo != 0xAB)[0], np.nonzero(hi != 0xAB)[0]
    if len(bad_hi):
        w = hi[: (bad_hi.max() // 4 + 1) * 4].view(np.uint32)
        print("   words written above:", len(bad_hi), "last byte", bad_hi.max(), "first words", w[:8], "max word", w[w != 0xABABABAB].max())
    print(name, "bytes", n, "guard below: ", (len(bad_lo), bad_lo[:4] - G if len(bad_lo) else ""), " guard above:", (len(bad_hi), bad_hi[:4] if len(bad_hi) else ""))
