"""How well do a view's forward (binning + forward blend) and another view's backward share the GPU?
Stream A loops the backward of a rendered view, stream B loops forwards; each alone, then together.
Usage (GPU box): python profiles/overlap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
os.environ["DGR_SYNC_MODE"] = "lazy"
import numpy as np, torch
from dgr_amd import light as L
from dgr_amd.synth import make_scene
import hip_helpers as hh

P, W, H, deg = 500000, 1920, 1080, 3
s = make_scene(P, W, H, 0)
out, d = hh.hip_forward(s, deg)
T, E = hh.T, hh.E
fargs = (T(s.bg), T(s.means), E(), T(s.opac), T(s.scales), T(s.rots), 1.0, E(), T(s.view), T(s.gt), T(s.proj), s.tanfovx,
         s.tanfovy, s.H, s.W, T(s.shs), deg, T(s.campos), False, False)
(R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
bargs = (T(s.bg), T(s.means), radii, E(), T(s.scales), T(s.rots), 1.0, E(), T(s.view), T(s.proj), s.tanfovx, s.tanfovy,
         T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None]), T(s.gt), T(s.shs), deg, T(s.campos), geom, R, binning, img,
         alpha, False, T(s.persp), False, False)
fwd = lambda: L._C.rasterize_gaussians(*fargs)
bwd = lambda: L._C.rasterize_gaussians_backward(*bargs)
for _ in range(5):
    fwd(); bwd()
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
n = 100


def run(do_a, do_b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if do_a:
            with torch.cuda.stream(A):
                bwd()
        if do_b:
            with torch.cuda.stream(B):
                fwd()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ta, tb, tab = run(True, False), run(False, True), run(True, True)
print(f"backward alone {ta:.3f} ms, forward alone {tb:.3f} ms, sum {ta + tb:.3f} ms, together {tab:.3f} ms per (fwd+bwd)")
