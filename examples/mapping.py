#!/usr/bin/env python3
"""Mapping against a window of keyframes (CG-SLAM's mapping loop in miniature).

A synthetic map rendered from K poses gives the keyframes' observed colour and depth; a degraded copy of the map
(jittered positions, flattened colours, wrong opacities) is then refined.  One iteration =
  slam.render_batch      every keyframe's forward + loss + backward on its own HIP stream (light variant, track_off=True:
                         no pose gradient), gradients summed into the parameters' .grad
  add_densification_stats 3DGS's per-view bookkeeping (screen-space gradient norm, view count, largest radius), one launch
  SparseAdam.step        fused Adam over the rows some keyframe saw, one launch per tensor
all on the GPU, no host synchronisation inside the loop.  --fused renders the keyframe batch through the batched entry points
instead (slam.render_batch_fused: one forward and one backward call for all keyframes, gradients summed in the kernels).

  python examples/mapping.py [--graph] [--fused] [--iters 100] [--keyframes 4] [--width 640 --height 480 --gaussians 100000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402


class MapModel:
    """3DGS's GaussianModel reduced to what a mapping step touches: raw leaves + the activations render() reads."""

    def __init__(self, s, dev, degrade=None):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        xyz, opac, scal, rot, shs = t(s.means), t(s.opac), t(s.scales), t(s.rots), t(s.shs)
        if degrade is not None:
            g = torch.Generator(device="cpu").manual_seed(degrade)
            xyz = xyz + 0.15 * scal.mean(1, keepdim=True) * torch.randn(xyz.shape, generator=g).to(dev)
            shs = shs * 0.5
            opac = (opac * 0.6).clamp(0.02, 0.98)
        opac = opac.clamp(1e-4, 1 - 1e-4)
        self._xyz = xyz.clone().requires_grad_()
        self._features = shs.clone().requires_grad_()
        self._opacity = torch.log(opac / (1 - opac)).requires_grad_()      # inverse sigmoid
        self._scaling = torch.log(scal).requires_grad_()
        self._rotation = rot.clone().requires_grad_()
        self.active_sh_degree = 3
        P = xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros(P, device=dev)

    get_xyz = property(lambda self: self._xyz)
    get_features = property(lambda self: self._features)
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))

    def groups(self):
        return [{"params": [self._xyz], "lr": 1.6e-4}, {"params": [self._features], "lr": 2.5e-3},
                {"params": [self._opacity], "lr": 5e-2}, {"params": [self._scaling], "lr": 5e-3},
                {"params": [self._rotation], "lr": 1e-3}]


def mapping_loop(dev, P, W, H, keyframes, iters, views_in_flight=3, log=None, graph=False, fused=False):
    """Returns (losses of the first and last iteration, model, seconds per iteration).  graph=True records the whole
    iteration (renders, losses, backward passes, statistics, Adam) into one hipGraph after three eager iterations."""
    from dgr_amd import light, slam
    from dgr_amd.optim import SparseAdam, add_densification_stats
    from dgr_amd.synth import make_scene

    scenes = [make_scene(P, W, H, 3, view_index=k) for k in range(keyframes)]
    s = scenes[0]
    bg, gt = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
    cams = [dict(viewmatrix=torch.from_numpy(sc.view).to(dev), fov=(sc.tanfovx, sc.tanfovy), HW=(H, W), gt_depth=gt)
            for sc in scenes]
    kw = dict(track_off=True, map_off=False)
    truth = MapModel(s, dev)
    with torch.no_grad():
        obs = [slam.render(None, truth, None, bg, viewmatrix=c["viewmatrix"], fov=c["fov"], HW=c["HW"], gt_depth=gt, **kw)
               for c in cams]
    obs = [(o["render"].detach(), o["depth"].detach()) for o in obs]

    if graph:
        views_in_flight = 1  # (branches of one graph do not overlap on this ROCm, and the leaves' gradient accumulation
                             #  belongs to the stream they were created on: keep the recorded iteration on one stream)
    pc = MapModel(s, dev, degrade=7)
    opt = SparseAdam(pc.groups(), eps=1e-15, capturable=graph)
    seen = torch.zeros(P, dtype=torch.int32, device=dev)
    outs = [None] * keyframes

    def loss_fn(out, k):
        outs[k] = out
        return slam.l1_loss(out["render"], out["depth"], obs[k][0], obs[k][1], 1.0, 0.5)  # one fused reduction

    obs_color, obs_depth = torch.stack([o[0] for o in obs]), torch.stack([o[1] for o in obs])

    def batch_loss_fn(out):
        # the keyframes' L1 losses summed, as ONE fused reduction over the stacked images: sum_k (mean_k |c - c_obs| + 0.5
        # mean_k |d - d_obs|) = V x the means over the whole stack (the keyframes share a size)
        V = float(out["render"].size(0))
        return slam.l1_loss(out["render"], out["depth"], obs_color, obs_depth, V * 1.0, V * 0.5)

    def iteration_fused():
        # the keyframe batch through ONE batched forward and ONE batched backward (dgr_amd.batch): the Gaussians' gradients
        # arrive summed over the keyframes, the screen-space gradients per view
        opt.zero_grad(set_to_none=True)
        losses, out = slam.render_batch_fused(cams, pc, None, bg, loss_fn, batch_loss_fn=batch_loss_fn, **kw)
        pts = out["viewspace_points"].grad
        for k in range(keyframes):
            add_densification_stats(pts[k], out["radii"][k], pc.xyz_gradient_accum, pc.denom, pc.max_radii2D)
        torch.amax(out["radii"], dim=0, out=seen)
        if not graph:  # lazy status mode: every outstanding forward reports before the step (an overflowed one raises here)
            light.check_async_errors()
        opt.step(visible=seen)
        return losses[0] / keyframes   # (the batch loss is the sum over the keyframes)

    def iteration():
        if fused:
            return iteration_fused()
        opt.zero_grad(set_to_none=True)
        if views_in_flight > 1:
            losses = slam.render_batch(cams, pc, None, bg, loss_fn, views_in_flight=views_in_flight, **kw)
        else:  # one keyframe after the other on the caller's stream
            losses = []
            for k, c in enumerate(cams):
                loss = loss_fn(slam.render(None, pc, None, bg, viewmatrix=c["viewmatrix"], fov=c["fov"], HW=c["HW"],
                                           gt_depth=gt, **kw), k)
                loss.backward()
                losses.append(loss.detach())
        seen.zero_()
        for out in outs:
            add_densification_stats(out["viewspace_points"].grad, out["radii"], pc.xyz_gradient_accum, pc.denom,
                                    pc.max_radii2D)
            torch.maximum(seen, out["radii"], out=seen)
        if not graph:
            light.check_async_errors()
        opt.step(visible=seen)
        return torch.stack(losses).mean()

    run = iteration
    if graph:  # (the three eager iterations run inside CapturedStep, on the stream the graph is recorded on)
        from dgr_amd.multiview import CapturedStep
        with torch.no_grad():
            first = float(torch.stack([slam.l1_loss(*(slam.render(None, pc, None, bg, viewmatrix=c["viewmatrix"], fov=c["fov"],
                                                                  HW=c["HW"], gt_depth=gt, **kw)[k_] for k_ in ("render", "depth")),
                                                    obs[i][0], obs[i][1], 1.0, 0.5) for i, c in enumerate(cams)]).mean())
        step = CapturedStep(iteration, warmup=3)
        run = step.replay
    else:
        first = float(iteration())
        for _ in range(2):
            iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters - 3):
        loss = run()
        if log and (i + 3) % 20 == 0:
            log(f"iteration {i + 3:4d}: loss {float(loss):.4e}")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(iters - 3, 1)
    if graph:
        step.check()
    return (first, float(loss)), pc, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--keyframes", type=int, default=4)
    ap.add_argument("--views-in-flight", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="record the iteration into a hipGraph and replay it")
    ap.add_argument("--fused", action="store_true",
                    help="the keyframe batch through one batched forward + backward (slam.render_batch_fused) instead of one "
                         "rasterizer call per keyframe")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--gaussians", type=int, default=100000)
    args = ap.parse_args()
    import torch as _torch
    _torch.autograd.set_multithreading_enabled(False)  # one device, one thread: no engine-thread hand-off per backward
    if args.graph:
        os.environ["DGR_SYNC_MODE"] = "lazy"  # a blocking status read cannot be captured
    dev = torch.device("cuda:0")
    (l0, l1), pc, dt = mapping_loop(dev, args.gaussians, args.width, args.height, args.keyframes, args.iters,
                                    args.views_in_flight, log=None if args.graph else print, graph=args.graph, fused=args.fused)
    n = float(pc.denom.sum())
    print(f"loss {l0:.4e} -> {l1:.4e}; {dt * 1e3:.3f} ms per mapping iteration over {args.keyframes} keyframes"
          f" ({dt / args.keyframes * 1e3:.3f} ms per keyframe); {int((pc.denom > 0).sum())} Gaussians seen,"
          f" {n:.0f} (Gaussian, view) statistics accumulated")


if __name__ == "__main__":
    main()
