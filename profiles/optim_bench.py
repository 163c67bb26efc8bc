"""Optimiser step over the config-3 parameter set (500 k Gaussians: 59 floats each): fused sparse Adam vs torch.optim.Adam.
Usage (GPU box): python profiles/optim_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
import torch
from dgr_amd.optim import SparseAdam
dev = torch.device("cuda:0")
P = 500000
shapes = [(P, 3), (P, 16, 3), (P, 1), (P, 3), (P, 4)]
def params():
    return [torch.randn(s, device=dev).requires_grad_() for s in shapes]
radii = (torch.rand(P, device=dev) < 0.85).to(torch.int32)
def timeit(opt, ps, **kw):
    for p in ps:
        p.grad = torch.randn_like(p)
    for _ in range(5):
        opt.step(**kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        opt.step(**kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100 * 1e3
ps = params(); print(f"torch.optim.Adam (foreach)      {timeit(torch.optim.Adam(ps, lr=1e-3), ps):.3f} ms/step")
try:
    ps = params(); print(f"torch.optim.Adam (fused=True)   {timeit(torch.optim.Adam(ps, lr=1e-3, fused=True), ps):.3f} ms/step")
except Exception as ex:
    print("torch fused Adam unavailable:", ex)
ps = params(); print(f"dgr SparseAdam, every row       {timeit(SparseAdam(ps, lr=1e-3), ps):.3f} ms/step")
ps = params(); print(f"dgr SparseAdam, 85 % rows seen  {timeit(SparseAdam(ps, lr=1e-3), ps, visible=radii):.3f} ms/step")
