"""Views in flight: K independent views (forward+backward each) issued round-robin on K HIP streams.
Usage (GPU box): python profiles/streams_experiment.py [K ...]"""
import os
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

P, W, H, deg = 500000, 1920, 1080, 3
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
scenes = [make_scene(P, W, H, seed=0, view_index=k) for k in range(2)]
s = scenes[0]
means3D, shs, opac = t(s.means).requires_grad_(), t(s.shs).requires_grad_(), t(s.opac).requires_grad_()
scales, rots = t(s.scales).requires_grad_(), t(s.rots).requires_grad_()
means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
gt = t(s.gt)
gC, gD, gM, gV = t(s.gC), t(s.gD[None]), t(s.gM[None]), t(s.gV[None])
views = [t(sc.view).requires_grad_() for sc in scenes]
rasts = [light.GaussianRasterizer(make_settings(sc, deg, dev)) for sc in scenes]
params = [means3D, means2D, shs, opac, scales, rots]


def step(k):
    for p_ in params + views:
        p_.grad = None
    color, radii, depth, median, var, alpha, unc, px = rasts[k](means3D=means3D, means2D=means2D, opacities=opac, shs=shs,
                                                                scales=scales, rotations=rots, viewmatrix=views[k], gt_depth=gt)
    torch.autograd.backward([color, depth, median, var], [gC, gD, gM, gV])


for K in [int(x) for x in sys.argv[1:]] or [1, 2, 3]:
    streams = [torch.cuda.Stream() for _ in range(K)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for i in range(20):
        with torch.cuda.stream(streams[i % K]):
            step(i % 2)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % K]):
            step(i % 2)
    light.check_async_errors()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"K={K}: {1e3 * dt / n:.4f} ms/view, {n / dt:.0f} views/s")
