"""Pure host cost of the autograd surface: a tiny scene (GPU work negligible), wall-clock around forward and backward.
Usage (GPU box): python profiles/host_breakdown.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
os.environ.setdefault("DGR_SYNC_MODE", "lazy")
import numpy as np
import torch
from dgr_amd import _capi, light
from dgr_amd.multiview import make_settings
from dgr_amd.synth import make_scene

P, W, H, deg = 2000, 64, 64, 3
dev = torch.device("cuda:0")
s = make_scene(P, W, H, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means3D, shs, opac = t(s.means).requires_grad_(), t(s.shs).requires_grad_(), t(s.opac).requires_grad_()
scales, rots, view = t(s.scales).requires_grad_(), t(s.rots).requires_grad_(), t(s.view).requires_grad_()
means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
gt = t(s.gt)
gC, gD, gM, gV = t(s.gC), t(s.gD[None]), t(s.gM[None]), t(s.gV[None])
rast = light.GaussianRasterizer(make_settings(s, deg, dev))
params = [means3D, means2D, shs, opac, scales, rots, view]
n = 500
tf = tb = tc = 0.0
for i in range(n + 50):
    for p_ in params:
        p_.grad = None
    t0 = time.perf_counter()
    outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
    t1 = time.perf_counter()
    torch.autograd.backward([outs[0], outs[2], outs[3], outs[4]], [gC, gD, gM, gV])
    t2 = time.perf_counter()
    if i >= 50:
        tf += t1 - t0
        tb += t2 - t1
    if i % 50 == 49:
        torch.cuda.synchronize()
print(f"forward  {1e6 * tf / n:7.1f} us/call (host)")
print(f"backward {1e6 * tb / n:7.1f} us/call (host, incl. the autograd engine's thread hand-off)")
# the C calls alone
lib = _capi.load()
args = (t(s.bg), means3D.detach(), torch.empty(0, device=dev), opac.detach(), scales.detach(), rots.detach(), 1.0, torch.empty(0, device=dev),
        view.detach(), gt, t(s.proj), s.tanfovx, s.tanfovy, H, W, shs.detach(), deg, t(s.campos), False, False)
for _ in range(20):
    out = light._C.rasterize_gaussians(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    out = light._C.rasterize_gaussians(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"_C.rasterize_gaussians alone {1e6 * (t1 - t0) / n:7.1f} us/call")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    out = light._C.rasterize_gaussians(*args)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
