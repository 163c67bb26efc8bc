"""Host-side cost of one forward+backward through the autograd surface (cProfile), config 2 light by default.
Usage (GPU box): python profiles/host_profile.py [workload] [variant]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
os.environ.setdefault("DGR_SYNC_MODE", "lazy")
import numpy as np
import torch
from dgr_amd import light
from dgr_amd.multiview import make_settings
from dgr_amd.synth import make_scene

P, W, H, deg = {"config2": (100000, 640, 480, 3), "config3": (500000, 1920, 1080, 3)}[sys.argv[1] if len(sys.argv) > 1 else "config2"]
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


def step():
    for p_ in params:
        p_.grad = None
    color, radii, depth, median, var, alpha, unc, px = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs,
                                                             scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
    torch.autograd.backward([color, depth, median, var], [gC, gD, gM, gV])


for _ in range(20):
    step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e3 * (t1 - t0) / n:.3f} ms/step, with drain {1e3 * (t2 - t0) / n:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
