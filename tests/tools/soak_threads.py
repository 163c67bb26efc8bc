#!/usr/bin/env python3
"""Soak of the library under SEVERAL HOST THREADS (not part of the suite): a SLAM process runs a tracker thread and a mapper thread over
one map, each on its own stream.  Every thread renders its own sequence of frames (light and full variant, mapping and tracking steps,
strict and lazy status mode, shapes shared between the threads so that the per-shape hints -- capacity, tile schedule, segment size,
armed status slots -- are read and written from all of them) and compares every result with what the same frame gave on ONE thread
before the soak started: images bit for bit, gradients up to the float atomics' order.

  python tests/tools/soak_threads.py [--seconds 60] [--threads 4] [--seed 0]
"""
import argparse
import os
import sys
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from dgr_amd import light as L, full as F
    from dgr_amd.multiview import make_settings
    from dgr_amd.synth import cluster_scene, heavy_tail_scene, make_scene
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

    # the frames: a few shapes, several scenes per shape (different instance counts under one shape: the lazy capacity guess is exercised)
    frames = []
    for P, W, H in [(20000, 320, 200), (6000, 97, 61), (50000, 640, 480), (3000, 16, 64)]:
        for k in range(3):
            s = make_scene(P, W, H, 100 * k + P % 97)
            if k == 1:
                s = cluster_scene(s)
            if k == 2:
                s = heavy_tail_scene(s, frac=0.03, sigma_px=(8.0, 0.5 * max(W, H)), seed=k)
            for variant, tracking in (("light", False), ("light", True), ("full", False)):
                frames.append((s, variant, tracking, int(rng.integers(0, 4))))

    def render(frame):
        s, variant, tracking, deg = frame
        leaves = [T(a).requires_grad_(not tracking) for a in (s.means, s.shs, s.opac, s.scales, s.rots)]
        view = T(s.view).requires_grad_(True)
        m2 = torch.zeros((s.P, 3), device=dev, requires_grad=not tracking)
        if variant == "full":
            st = F.GaussianRasterizationSettings(image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=T(s.bg),
                                                 scale_modifier=1.0, viewmatrix=T(s.view), projmatrix=T(s.proj), sh_degree=deg,
                                                 campos=T(s.campos), prefiltered=False, perspec_matrix=T(s.persp))
            o = F.GaussianRasterizer(st)(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3],
                                         rotations=leaves[4], viewmatrix=view, gt_depth=T(s.gt))
            torch.autograd.backward([o[0], o[2]], [T(s.gC), T(s.gD[None])])
            imgs = [o[0], o[2]]
        else:
            o = L.GaussianRasterizer(make_settings(s, deg, dev, map_off=tracking))(
                means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                viewmatrix=view, gt_depth=T(s.gt))
            torch.autograd.backward([o[0], o[2], o[3], o[4]], [T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None])])
            imgs = [o[0], o[2], o[3], o[5]]
        grads = [view.grad] + ([] if tracking else [leaves[0].grad, leaves[1].grad, leaves[2].grad])
        return [x.detach().clone() for x in imgs], [g.detach().clone() for g in grads]

    os.environ["DGR_SYNC_MODE"] = "strict"
    refs = [render(f) for f in frames]
    torch.cuda.synchronize()

    stop = time.time() + args.seconds
    counts = [0] * args.threads
    problems = []
    lock = threading.Lock()

    def worker(t):
        r = np.random.default_rng(args.seed * 1000 + t)
        stream = torch.cuda.Stream(dev)
        try:
            with torch.cuda.stream(stream):
                while time.time() < stop and not problems:
                    i = int(r.integers(0, len(frames)))
                    # (the status mode is an environment variable read per call: all threads switch together, at random moments)
                    if t == 0 and r.random() < 0.1:
                        os.environ["DGR_SYNC_MODE"] = "lazy" if r.random() < 0.5 else "strict"
                    try:
                        imgs, grads = render(frames[i])
                        L.check_async_errors()
                        F.check_async_errors() if hasattr(F, "check_async_errors") else None
                    except RuntimeError as e:
                        if "overflow" in str(e):   # a lazy forward whose guess was too small says so: the contract, not a failure
                            counts[t] += 1
                            continue
                        raise
                    stream.synchronize()
                    ri, rg = refs[i]
                    for a, b in zip(imgs, ri):
                        if not torch.equal(a, b):
                            if torch.isnan(a).all():   # (an overflowed lazy frame whose error a later poll will bring)
                                break
                            with lock:
                                problems.append(f"thread {t} frame {i} {frames[i][1:]}: an image differs (max {float((a - b).abs().max()):.3e})")
                            break
                    else:
                        for a, b in zip(grads, rg):
                            scale = float(b.abs().max()) or 1.0
                            if float((a - b).abs().max()) > 2e-5 * scale:
                                with lock:
                                    problems.append(f"thread {t} frame {i} {frames[i][1:]}: a gradient differs by {float((a - b).abs().max()) / scale:.2e} of scale")
                                break
                    counts[t] += 1
        except Exception:
            with lock:
                problems.append(f"thread {t}: " + traceback.format_exc()[-1500:])

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(args.threads)]
    t0 = time.time()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    try:
        L.check_async_errors()
    except RuntimeError as e:
        if "overflow" not in str(e):
            problems.append("final check: " + str(e))
    print(f"soak_threads: {sum(counts)} frames on {args.threads} threads in {time.time() - t0:.0f} s (per thread {counts}), seed {args.seed} -- "
          + ("results identical to the one-thread run" if not problems else f"{len(problems)} PROBLEMS"))
    for p in problems[:10]:
        print("PROBLEM", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
