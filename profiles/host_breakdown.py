"""Pure host cost of the autograd surface: a tiny scene (GPU work negligible), wall-clock around forward and backward.
Usage (GPU box): python profiles/host_breakdown.py [light|full]   (DGR_AUTOGRAD=python: the Python autograd.Function
over the compiled `_C` instead of the compiled node; DGR_BINDING=ctypes: the ctypes binding)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]
os.environ.setdefault("DGR_SYNC_MODE", "lazy")
import numpy as np
import torch
from dgr_amd import _capi, light, full
from dgr_amd.multiview import make_settings
from dgr_amd.synth import make_scene

variant = sys.argv[1] if len(sys.argv) > 1 else "light"
P, W, H, deg = 2000, 64, 64, 3
dev = torch.device("cuda:0")
s = make_scene(P, W, H, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
means3D, shs, opac = t(s.means).requires_grad_(), t(s.shs).requires_grad_(), t(s.opac).requires_grad_()
scales, rots, view = t(s.scales).requires_grad_(), t(s.rots).requires_grad_(), t(s.view).requires_grad_()
means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
gt = t(s.gt)
gC, gD, gM, gV = t(s.gC), t(s.gD[None]), t(s.gM[None]), t(s.gV[None])
if variant == "full":
    tt = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    rast = full.GaussianRasterizer(full.GaussianRasterizationSettings(
        image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=tt(s.bg), scale_modifier=1.0,
        viewmatrix=tt(s.view), projmatrix=tt(s.proj), sh_degree=deg, campos=tt(s.campos), prefiltered=False,
        perspec_matrix=tt(s.persp)))
else:
    rast = light.GaussianRasterizer(make_settings(s, deg, dev))
params = [means3D, means2D, shs, opac, scales, rots, view]
n = 1000


def one():
    outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
    return outs


def back(outs):
    if variant == "full":
        torch.autograd.backward([outs[0], outs[2], outs[3]], [gC, gD, gV])
    else:
        torch.autograd.backward([outs[0], outs[2], outs[3], outs[4]], [gC, gD, gM, gV])


def measure(label):
    tf = tb = 0.0
    for i in range(n + 100):
        for p_ in params:
            p_.grad = None
        t0 = time.perf_counter()
        outs = one()
        t1 = time.perf_counter()
        back(outs)
        t2 = time.perf_counter()
        if i >= 100:
            tf += t1 - t0
            tb += t2 - t1
        if i % 20 == 19:
            torch.cuda.synchronize()
    print(f"[{label}] forward {1e6 * tf / n:6.1f} us  backward {1e6 * tb / n:6.1f} us  (host, per view; binding={light._C.__name__}, "
          f"autograd={'compiled node' if light._USE_NODE and light._C is light._CompiledC else 'python Function'})", flush=True)


measure(variant)
torch.autograd.set_multithreading_enabled(False)
if light._C is light._CompiledC:
    light._CompiledC.ext.host_prof_dump(True)
measure(variant + ", autograd multithreading off")
if light._C is light._CompiledC and os.environ.get("DGR_HOST_PROF") == "1":
    print(light._CompiledC.ext.host_prof_dump(True))
torch.autograd.set_multithreading_enabled(True)

# ---- the pieces
lib = _capi.load()


def timeit(fn, n=2000, sync_every=50):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % sync_every == sync_every - 1:
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t0 += time.perf_counter() - t1
    return 1e6 * (time.perf_counter() - t0) / n


u8 = dict(dtype=torch.uint8, device=dev)
print(f"torch.empty(1 MB)                          {timeit(lambda: torch.empty((1 << 20,), **u8)):6.2f} us")
status = torch.zeros(4, dtype=torch.int32, device=dev)
st = _capi.stream_handle(0)
buf = (C.c_int * 4)()


def post_poll():
    tk = lib.dgr_status_post(st, status.data_ptr())
    lib.dgr_status_poll(tk, 1, buf)


print(f"dgr_status_post + poll (ctypes)            {timeit(post_poll, 500, 1000):6.2f} us")
x = torch.zeros(64, device=dev)
print(f"one tiny torch kernel (x.add_(1))          {timeit(lambda: x.add_(1)):6.2f} us")
ev = torch.cuda.Event()
print(f"event record                               {timeit(lambda: ev.record()):6.2f} us")

# C ABI alone: presized forward + backward with every buffer allocated once (the launches and nothing else)
p = _capi.ptr
bg, proj, campos, persp = t(s.bg), t(s.proj), t(s.campos), t(s.persp)
geom = torch.empty((lib.dgr_geometry_bytes(P),), **u8)
img = torch.empty((lib.dgr_image_bytes(W, H),), **u8)
cap = 8 * P
binning = torch.empty((lib.dgr_binning_bytes(cap, W, H),), **u8)
f32 = dict(dtype=torch.float32, device=dev)
color, depth, median, var, alpha = (torch.empty((c_, H, W), **f32) for c_ in (3, 1, 1, 1, 1))
radii = torch.empty((P,), dtype=torch.int32, device=dev)
unc, px = torch.empty((P, 1), **f32), torch.empty((P, 1), dtype=torch.int32, device=dev)
m3, sh_, op_, sc_, ro_, vw_ = (a.detach() for a in (means3D, shs, opac, scales, rots, view))


def c_forward():
    rc = lib.dgr_light_forward_presized(st, p(geom), p(binning), cap, p(img), p(status), P, deg, 16, p(bg), W, H, p(m3), p(sh_), None,
                                        p(op_), p(sc_), 1.0, p(ro_), None, p(vw_), p(proj), p(campos), s.tanfovx, s.tanfovy, 0,
                                        p(color), p(depth), p(median), p(alpha), p(gt), p(var), p(unc), p(px), p(radii))
    assert rc >= 0


print(f"dgr_light_forward_presized alone (ctypes)  {timeit(c_forward):6.2f} us  (5 launches)")
scratch = torch.empty((lib.dgr_light_backward_scratch_bytes(P, W, H),), **u8)
g3, g2, gsh, gop, gsc, gro, gcov, gcol = (torch.empty(sz, **f32) for sz in ((P, 3), (P, 3), (P, 16, 3), (P, 1), (P, 3), (P, 4), (P, 6), (P, 3)))
dview = torch.empty((16,), **f32)


def c_backward():
    rc = lib.dgr_light_backward(st, P, deg, 16, cap, p(bg), W, H, p(m3), p(sh_), None, p(alpha), p(sc_), 1.0, p(ro_), None, p(vw_),
                                p(proj), p(campos), s.tanfovx, s.tanfovy, p(radii), p(geom), p(binning), p(img), p(gC), p(gD), p(gM),
                                p(gV), p(g2), None, p(gop), p(gcol), None, p(g3), p(gcov), p(gsh), p(gsc), p(gro), 0, None, p(persp),
                                p(dview), None, p(gt), 0, 0, p(scratch), scratch.numel())
    assert rc >= 0


c_forward()
print(f"dgr_light_backward alone (ctypes)          {timeit(c_backward):6.2f} us  (3 launches)")
if light._C is light._CompiledC:
    ext = light._CompiledC.ext
    E = light._EMPTY
    a_f = (bg, m3, E, op_, sc_, ro_, 1.0, E, vw_, gt, proj, s.tanfovx, s.tanfovy, H, W, sh_, deg, campos, False, False, cap, 2)

    def ext_forward():
        o = ext.light_forward(*a_f)
        lib.dgr_status_poll(o[1], 1, buf)
        return o

    print(f"ext.light_forward + status poll            {timeit(ext_forward, 500, 1000):6.2f} us  (allocations + launches + status post)")
    o = ext_forward()
    a_b = (bg, m3, o[9], E, sc_, ro_, 1.0, E, vw_, proj, s.tanfovx, s.tanfovy, gC, gD, gM, gV, gt, sh_, deg, campos, o[10], cap, o[11],
           o[12], o[8], False, persp, False, False, True)
    print(f"ext.light_backward                         {timeit(lambda: ext.light_backward(*a_b)):6.2f} us  (allocations + launches)")
