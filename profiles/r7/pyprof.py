"""cProfile of the Python side of a forward + backward through the compiled node (tiny scene).  GPU box."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
os.environ.setdefault("DGR_SYNC_MODE", "lazy")
import numpy as np, torch
from dgr_amd import light
from dgr_amd.multiview import make_settings
from dgr_amd.synth import make_scene
torch.autograd.set_multithreading_enabled(False)
P, W, H, deg = 2000, 64, 64, 3
dev = torch.device("cuda:0")
s = make_scene(P, W, H, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
L = [t(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
gt = t(s.gt); g = [t(s.gC), t(s.gD[None]), t(s.gM[None]), t(s.gV[None])]
rast = light.GaussianRasterizer(make_settings(s, deg, dev))
def step():
    for p_ in L: p_.grad = None
    m2.grad = None
    o = rast(means3D=L[0], means2D=m2, opacities=L[2], shs=L[1], scales=L[3], rotations=L[4], viewmatrix=L[5], gt_depth=gt)
    torch.autograd.backward([o[0], o[2], o[3], o[4]], g)
for i in range(200): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(2000):
    step()
    if i % 20 == 19: torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
