#!/usr/bin/env python3
"""Camera tracking against a fixed Gaussian map (CG-SLAM's tracking loop in miniature).

A synthetic map is rendered from the true pose to get the "observed" frame; the pose estimate starts perturbed and is
refined by Adam on (quaternion, translation) through the rasterizer's analytic viewmatrix gradient (tracking mode:
map_off=True, no Gaussian gradients).  With --graph the whole iteration -- pose -> matrices -> render -> loss -> backward ->
Adam step -- is recorded once into a hipGraph and replayed (dgr_amd.multiview.CapturedStep), which removes the host
from the loop.

  python examples/tracking.py [--graph] [--fused] [--iters 150] [--width 640 --height 480 --gaussians 100000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--fused", action="store_true",
                    help="pose -> camera tensors, the L1 loss and the Adam step as single launches (slam.pose_to_camera, "
                         "slam.l1_loss, optim.SparseAdam)")
    ap.add_argument("--iters", type=int, default=150)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--gaussians", type=int, default=100000)
    args = ap.parse_args()
    if args.graph or args.fused:
        # a blocking status read cannot be captured, and a tracking loop has no use for num_rendered on the host: no host wait
        os.environ.setdefault("DGR_SYNC_MODE", "lazy") if not args.graph else os.environ.__setitem__("DGR_SYNC_MODE", "lazy")
    from dgr_amd import light, slam
    from dgr_amd.multiview import CapturedStep
    from dgr_amd.synth import camera, make_scene
    from test_slam_render import Model, rot_to_quat

    dev = torch.device("cuda:0")
    W, H = args.width, args.height
    s = make_scene(args.gaussians, W, H, 3)
    pc = Model(s, dev)
    tanfovx, tanfovy, Rm, t_true, *_ = camera(W, H, 0.05)
    bg, gt_depth = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
    kw = dict(fov=(tanfovx, tanfovy), HW=(H, W), gt_depth=gt_depth, track_off=False, map_off=True)

    # one device, one Python thread: the autograd engine's worker thread only adds a hand-off per backward (0.47 -> 0.32 ms per
    # iteration here); a switch of PyTorch, not of the rasterizer
    torch.autograd.set_multithreading_enabled(False)

    def pose(q, t):
        if args.fused:
            return slam.pose_to_camera(q, t, tanfovx, tanfovy)
        return slam.camera_tensors(slam.w2c_from_quat_trans(q, t), tanfovx, tanfovy)

    def render(cam):  # (fused: projmatrix / campos come from the pose kernel; otherwise render() forms them from the viewmatrix)
        return slam.render(None, pc, None, bg, viewmatrix=cam[0], pose_tensors=cam if args.fused else None, **kw)

    q_true = torch.tensor(rot_to_quat(Rm), dtype=torch.float32, device=dev)
    t_true = torch.tensor(t_true, dtype=torch.float32, device=dev)
    with torch.no_grad():
        obs = render(pose(q_true, t_true))
    obs_c, obs_d = obs["render"].detach(), obs["depth"].detach()

    q = (q_true + torch.tensor([0.0, 0.004, -0.006, 0.003], device=dev)).requires_grad_()
    t = (t_true + torch.tensor([0.012, -0.009, 0.015], device=dev)).requires_grad_()
    groups = [{"params": [q], "lr": 5e-4}, {"params": [t], "lr": 1.5e-3}]
    if args.fused:  # one launch per tensor, step count on the device when the iteration is recorded into a graph
        from dgr_amd.optim import SparseAdam
        opt = SparseAdam(groups, capturable=args.graph)
    else:
        opt = torch.optim.Adam(groups, capturable=args.graph)

    def iteration():
        opt.zero_grad(set_to_none=True)
        out = render(pose(q, t))
        if args.fused:
            loss = slam.l1_loss(out["render"], out["depth"], obs_c, obs_d, 1.0, 0.5)
        else:
            loss = (out["render"] - obs_c).abs().mean() + 0.5 * (out["depth"] - obs_d).abs().mean()
        loss.backward()
        if not args.graph:  # lazy status mode: every outstanding forward reports before the step (an overflowed one raises here)
            light.check_async_errors()
        opt.step()
        return loss.detach()

    def err():
        with torch.no_grad():
            return float((q / q.norm() - q_true).norm()), float((t - t_true).norm())

    print(f"start : rotation error {err()[0]:.2e}, translation error {err()[1]:.2e}")
    if args.graph:
        step = CapturedStep(iteration, warmup=3)  # (three eager iterations first)
        run = step.replay
    else:
        for _ in range(3):  # the same three iterations, so that one-time costs (kernel loading, optimiser set-up) are not timed
            iteration()
        run = iteration
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.iters):
        loss = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if args.graph:
        step.check()
    print(f"finish: rotation error {err()[0]:.2e}, translation error {err()[1]:.2e}, loss {float(loss):.3e}")
    print(f"{args.iters} iterations in {dt * 1e3:.1f} ms = {dt / args.iters * 1e3:.3f} ms per tracking iteration"
          f" ({'hipGraph replay' if args.graph else 'eager'}{', fused pose and loss' if args.fused else ''})")


if __name__ == "__main__":
    main()
