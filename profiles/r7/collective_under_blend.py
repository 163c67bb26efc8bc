"""does a collective issued under running blend kernels make progress?  One GPU, a one-rank RCCL group.
Times (HIP events on a side stream) (i) dist.all_reduce of the 124 MB gradient span and (ii) a stand-in of the same size that
certainly launches a kernel (an out-of-place add: 2 x 124 MB read, 124 MB written -- a one-rank in-place all-reduce may be a
no-op inside RCCL), each ALONE and while seven views are in flight on seven streams, with the blend kernels uncapped and capped
at 7 workgroups per CU (dgr_set_option "blend_wgs_per_cu").  Usage (GPU box): python profiles/r7/collective_under_blend.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
os.environ["DGR_SYNC_MODE"] = "lazy"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
from dgr_amd import _capi, light
from dgr_amd.multiview import ViewStreams, make_settings
from dgr_amd.synth import make_scene
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P, W, H, deg = 500_000, 1920, 1080, 3
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
n = 248 * P // 4
buf, buf2, out = torch.zeros(n, device=dev), torch.ones(n, device=dev), torch.empty(n, device=dev)
side = torch.cuda.Stream(device=dev)
def coll(kind):
    if kind == "allreduce":
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    else:
        torch.add(buf, buf2, out=out)
def timed(kind, views, reps=12):
    vs = ViewStreams(7, dev) if views else None
    if vs:
        for _ in range(14):
            with vs.next(): step()
    torch.cuda.synchronize()
    evs = []
    for r in range(reps):
        if vs:
            for _ in range(7):
                with vs.next(): step()
        with torch.cuda.stream(side):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); coll(kind); e1.record()
            evs.append((e0, e1))
        if vs:
            for _ in range(7):
                with vs.next(): step()
    torch.cuda.synchronize()
    light.check_async_errors()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2], ms[0], ms[-1]
for _ in range(5): step()
print(f"payload {4 * n / 1e6:.1f} MB, one-rank RCCL group ({dist.get_backend()})")
for kind in ("allreduce", "stand-in add"):
    a = timed(kind, False)
    print(f"{kind:13s} alone                          median {a[0]:.3f} ms (min {a[1]:.3f}, max {a[2]:.3f})")
    for cap in (0, 7):
        _capi.set_option("blend_wgs_per_cu", cap)
        b = timed(kind, True)
        print(f"{kind:13s} under 7 views, blend cap {cap}: median {b[0]:.3f} ms (min {b[1]:.3f}, max {b[2]:.3f})")
    _capi.set_option("blend_wgs_per_cu", 0)
dist.destroy_process_group()
