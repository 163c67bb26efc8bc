"""One-view-per-rank sharding on CPU: world size 2, gloo.

Each rank produces the per-Gaussian gradients of ITS view (with the CPU oracle standing in for the GPU kernels --
this file tests the exchange step, not the kernels), lays them out in the flat gradient arena exactly as
dgr_amd.light._C.rasterize_gaussians_backward does, and `GradientArena.all_reduce` sums them with one collective.
The result must equal the serial sum over views; pose gradients stay per view."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diff-gaussian-rasterization_amd")

NAMES = ("means3D", "means2D", "sh", "opacity", "scales", "rotations")
GKEY = dict(means3D="dL_dmeans3D", means2D="dL_dmeans2D", sh="dL_dsh", opacity="dL_dopacity", scales="dL_dscales",
            rotations="dL_drotations")


def view_grads(P, W, H, deg, k):
    from dgr_amd.synth import make_scene
    from oracle import oracle as O
    s = make_scene(P, W, H, 5, view_index=k)
    grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    st, out = O.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                              s.tanfovy, H, W, s.shs, deg, s.campos)
    return O.light_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.proj, s.tanfovx, s.tanfovy,
                            *grads, s.gt, s.shs, deg, s.campos, out["opacity_map"], s.persp)


def worker(rank, world, port, fused, q, async_op=False):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgr_amd import light
    from dgr_amd.multiview import GradientArena
    P, W, H, deg = 800, 48, 32, 2
    g = view_grads(P, W, H, deg, rank)
    f32 = dict(dtype=torch.float32, device="cpu")
    seg = light._grad_arena(P, 16, f32)  # same arena layout the backward uses
    params = []
    for n in NAMES:
        seg[n].copy_(torch.from_numpy(g[GKEY[n]]))
        p = torch.zeros_like(seg[n], requires_grad=True)
        p.grad = seg[n] if fused else seg[n].clone()  # aliasing (autograd stole the view) vs copied gradient
        params.append(p)
    arena = GradientArena(params)
    if async_op:  # overlapped form (bench.py --allreduce overlap): issue, do other work, wait
        pending = arena.all_reduce(dist, async_op=True)
        for p in params:
            p.grad = None  # the next view's step drops the references; the pending reduce keeps the buffers alive
        ncoll = len(pending)
        reduced = pending.wait()
        if fused:
            views = {n: seg[n] for n in NAMES}
            assert reduced[0].data_ptr() == seg["means3D"].data_ptr()
        else:
            views = dict(zip(NAMES, reduced))
        for n, p in zip(NAMES, params):
            p.grad = views[n]
    else:
        ncoll = arena.all_reduce(dist)
    q.put((rank, ncoll, {n: p.grad.numpy().copy() for n, p in zip(NAMES, params)}, g["dL_dview"]))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("fused,async_op", [(True, False), (False, False), (True, True), (False, True)])
def test_gradient_all_reduce_equals_serial_sum_over_views(oracle, fused, async_op):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, fused, q, async_op)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path[:0] = [ROOT, PKG]
    serial = [view_grads(800, 48, 32, 2, k) for k in range(world)]
    for rank, ncoll, grads, dview in res:
        assert ncoll == (1 if fused else len(NAMES))  # fused: ONE collective for all per-Gaussian gradients
        for n in NAMES:
            want = serial[0][GKEY[n]].astype(np.float64) + serial[1][GKEY[n]]
            np.testing.assert_allclose(grads[n], want.reshape(grads[n].shape), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(dview, serial[rank]["dL_dview"], rtol=1e-6)  # pose gradient is per view, not reduced


def worker_grouped(rank, world, port, q):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgr_amd import light
    from dgr_amd.multiview import GradientArena, GroupedReduce
    P, W, H, deg = 800, 48, 32, 2
    f32 = dict(dtype=torch.float32, device="cpu")
    shapes = dict(means3D=(P, 3), means2D=(P, 3), sh=(P, 16, 3), opacity=(P, 1), scales=(P, 3), rotations=(P, 4))
    params = [torch.zeros(shapes[n], requires_grad=True) for n in NAMES]  # real leaf parameters
    opt = torch.optim.SGD(params, lr=0.5)
    arena = GradientArena(params)
    grouped = GroupedReduce(arena, dist, group_size=2)
    for local in range(2):  # two local views per rank: views 2*rank, 2*rank + 1
        g = view_grads(P, W, H, deg, 2 * rank + local)
        for p in params:
            p.grad = None
        seg = light._grad_arena(P, 16, f32)  # what a backward does: a fresh arena ...
        for n, p in zip(NAMES, params):
            seg[n].copy_(torch.from_numpy(g[GKEY[n]]).reshape(shapes[n]))
            p.grad = seg[n]                   # ... whose views autograd hands to .grad without copying
        grouped.add_view()
        if local == 0:
            assert grouped.collectives == 0  # nothing reduced before the group is complete
    assert grouped.collectives == 1 and not grouped.pending
    # the batch gradient is where an optimiser looks for it
    batch = {n: p.grad.numpy().copy() for n, p in zip(NAMES, params)}
    opt.step()
    stepped = {n: p.detach().numpy().copy() for n, p in zip(NAMES, params)}
    # a view that accumulated into an existing .grad instead of aliasing a fresh arena must be refused, not double counted
    refused = False
    seg = light._grad_arena(P, 16, f32)
    for n, p in zip(NAMES, params):
        p.grad = p.grad + seg[n].zero_()  # (an accumulated gradient: its own storage)
    try:
        grouped.add_view()
    except RuntimeError:
        refused = True
    q.put((rank, batch, stepped, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_grouped_reduce_sums_local_views_then_one_collective(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker_grouped, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path[:0] = [ROOT, PKG]
    serial = [view_grads(800, 48, 32, 2, k) for k in range(4)]
    for rank, batch, stepped, refused in res:
        assert refused
        for n in NAMES:
            want = sum(sv[GKEY[n]].astype(np.float64) for sv in serial).reshape(batch[n].shape)
            np.testing.assert_allclose(batch[n], want, rtol=2e-5, atol=1e-8)  # float32 sums of four views
            np.testing.assert_allclose(stepped[n], -0.5 * want, rtol=2e-5, atol=1e-8)  # SGD step from zeros, lr 0.5
