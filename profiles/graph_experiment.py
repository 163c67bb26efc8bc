"""hipGraph capture of one view (forward + backward through the autograd surface) with torch.cuda.graph, config 2.
Usage (GPU box): python profiles/graph_experiment.py [light|full]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
os.environ["DGR_SYNC_MODE"] = "lazy"
import numpy as np
import torch
from dgr_amd import light, full
from dgr_amd.multiview import make_settings
from dgr_amd.synth import make_scene

variant = sys.argv[1] if len(sys.argv) > 1 else "light"
P, W, H, deg = (100000, 640, 480, 3) if (len(sys.argv) < 3 or sys.argv[2] == "config2") else (500000, 1920, 1080, 3)
dev = torch.device("cuda:0")
s = make_scene(P, W, H, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means3D, shs, opac = t(s.means).requires_grad_(), t(s.shs).requires_grad_(), t(s.opac).requires_grad_()
scales, rots, view = t(s.scales).requires_grad_(), t(s.rots).requires_grad_(), t(s.view).requires_grad_()
means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
gt = t(s.gt)
gC, gD, gM, gV = t(s.gC), t(s.gD[None]), t(s.gM[None]), t(s.gV[None])
if variant == "light":
    rast = light.GaussianRasterizer(make_settings(s, deg, dev))
else:
    tt = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    rast = full.GaussianRasterizer(full.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=tt(s.bg), scale_modifier=1.0,
        viewmatrix=tt(s.view), projmatrix=tt(s.proj), sh_degree=deg, campos=tt(s.campos), prefiltered=False,
        perspec_matrix=tt(s.persp)))
params = [means3D, means2D, shs, opac, scales, rots, view]


def step():
    for p_ in params:
        p_.grad = None
    outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
    if variant == "light":
        torch.autograd.backward([outs[0], outs[2], outs[3], outs[4]], [gC, gD, gM, gV])
    else:
        torch.autograd.backward([outs[0], outs[2], outs[3]], [gC, gD, gV])
    return outs[0]


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(5):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager_color = step().detach().clone()
eager_grad = view.grad.clone()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"eager  {1e3 * (time.perf_counter() - t0) / n:.4f} ms/view")

g = torch.cuda.CUDAGraph()
for p_ in params:
    p_.grad = None
with torch.cuda.graph(g):
    color = step()
g.replay()
torch.cuda.synchronize()
print("graph replay: colour equal", bool(torch.equal(color, eager_color)), " pose grad max diff", float((view.grad - eager_grad).abs().max()))
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print(f"graph  {1e3 * (time.perf_counter() - t0) / n:.4f} ms/view")

# K independent views inside ONE graph, each on its own stream (fork / join inside the capture)
from dgr_amd.multiview import ViewStreams
for K in (2, 3):
    views = ViewStreams(K, dev)
    gk = torch.cuda.CUDAGraph()
    for p_ in params:
        p_.grad = None
    with torch.cuda.graph(gk):
        for k in range(K):
            with views.next():
                step()
        views.join()
    gk.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // K):
        gk.replay()
    torch.cuda.synchronize()
    print(f"graph of {K} views on {K} streams  {1e3 * (time.perf_counter() - t0) / (n // K * K):.4f} ms/view")

# K separately captured views, each replayed on its own stream (graphs on different streams may overlap each other)
for K in (2, 3):
    streams = [torch.cuda.Stream() for _ in range(K)]
    graphs = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        g_ = torch.cuda.CUDAGraph()
        for p_ in params:
            p_.grad = None
        with torch.cuda.graph(g_, stream=st):
            step()
        graphs.append(g_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % K]):
            graphs[i % K].replay()
    torch.cuda.synchronize()
    print(f"{K} graphs replayed round-robin on {K} streams  {1e3 * (time.perf_counter() - t0) / n:.4f} ms/view")
