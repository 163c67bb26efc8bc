import os, sys
os.environ["DGR_SYNC_MODE"] = sys.argv[1]
sys.path[:0] = ["/root/repo", "/root/repo/diff-gaussian-rasterization_amd", "/root/repo/tests"]
import torch, numpy as np
from dgr_amd import light
from dgr_amd.multiview import make_settings, ViewStreams
from dgr_amd.synth import make_scene
dev = torch.device("cuda:0")
s = make_scene(20000, 320, 240, 1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
leaves = [t(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
means3D, shs, opac, scales, rots, view = leaves
means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
rast = light.GaussianRasterizer(make_settings(s, 3, dev))
gt, gC, gD = t(s.gt), t(s.gC), t(s.gD[None])
views = ViewStreams(3, dev)
def step():
    for p in leaves + [means2D]: p.grad = None
    o = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
    torch.autograd.backward([o[0], o[2]], [gC, gD])
for i in range(6001):
    with views.next(): step()
    if i in (500, 3000, 6000):
        views.join(); torch.cuda.synchronize()
        import resource
        print(sys.argv[1], i, "allocated MB", torch.cuda.memory_allocated() >> 20, "reserved MB", torch.cuda.memory_reserved() >> 20,
              "pending", len(light._pending_status), "host RSS MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)
light.check_async_errors()
