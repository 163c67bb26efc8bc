#!/usr/bin/env python3
"""Soak of round 7's host path (not part of the suite): random frames through the compiled autograd node in lazy mode, several
views in flight on several streams (resident scratch per stream, armed status slots, tile schedule by policy), against the
Python autograd.Function over the same `_C` on the same inputs, one view at a time, strict.  No oracle involved -- the suite pins
the kernels to it; this looks for anything the new plumbing could get wrong: a scratch that is not clean, a status word that
belongs to another forward, a slot that never completes, a policy decision that changes a result.

  python tests/tools/soak_node.py [--seconds 120] [--seed 0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--drop-inputs", action="store_true",
                    help="make every view's inputs on the caller's stream right before its call and drop them right after it, with the "
                         "views still queued on their side streams: what the compiled binding's stream record (csrc/torch_ext.cpp: "
                         "keep_until_read) makes safe")
    args = ap.parse_args()
    from dgr_amd import light as L, full as F
    from dgr_amd.multiview import ViewStreams, make_settings
    from dgr_amd.synth import cluster_scene, make_scene
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    torch.autograd.set_multithreading_enabled(bool(rng.integers(0, 2)))
    t_end = time.time() + args.seconds
    draws = worst = bad = 0
    shapes = [(int(rng.integers(1, 60000)), int(rng.choice([7, 33, 100, 250, 321, 640])), int(rng.choice([5, 47, 97, 200, 480])))
              for _ in range(10)]  # a few shapes, revisited: the capacity and schedule hints of a shape carry over
    while time.time() < t_end:
        P, W, H = shapes[int(rng.integers(0, len(shapes)))]
        deg = int(rng.integers(0, 4))
        variant = "full" if rng.random() < 0.3 else "light"
        tracking = variant == "light" and rng.random() < 0.3
        nviews = int(rng.integers(1, 6))
        scenes = [make_scene(P, W, H, int(rng.integers(0, 1 << 30)), view_index=v) for v in range(nviews)]
        if rng.random() < 0.3:
            scenes = [cluster_scene(s) for s in scenes]
        if rng.random() < 0.2:  # translucent: many below the alpha threshold; few instances: the next frame of the shape overflows nothing
            scenes = [s._replace(opac=(s.opac * 0.12).astype(np.float32)) for s in scenes]
        res = {}
        for mode in ("reference", "node"):
            os.environ["DGR_SYNC_MODE"] = "strict" if mode == "reference" else "lazy"
            L._USE_NODE = mode == "node"
            views = ViewStreams(int(rng.integers(2, 5)), dev) if (mode == "node" and nviews > 1) else None
            outs, keep = [], []
            # Default: every view's inputs are made BEFORE any view is issued and kept alive until the device has drained -- a
            # tensor allocated on the caller's stream and freed while a side stream still reads it would be handed out again
            # (PyTorch's rule for tensors used on another stream than the one they were allocated on).  --drop-inputs does the
            # opposite in node mode: the compiled binding records the side stream on its inputs, so it must make no difference.

            def make(s):
                leaves = [T(a).requires_grad_(not tracking) for a in (s.means, s.shs, s.opac, s.scales, s.rots)]
                view = T(s.view).requires_grad_(True)
                m2 = torch.zeros((P, 3), device=dev, requires_grad=not tracking)
                if variant == "full":
                    tt = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)  # noqa: E731
                    st = F.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=tt(s.bg),
                                                         scale_modifier=1.0, viewmatrix=tt(s.view), projmatrix=tt(s.proj), sh_degree=deg,
                                                         campos=tt(s.campos), prefiltered=False, perspec_matrix=tt(s.persp))
                    rast = F.GaussianRasterizer(st)
                else:
                    rast = L.GaussianRasterizer(make_settings(s, deg, dev, map_off=tracking))
                return (leaves, view, m2, rast, [T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None])], T(s.gt))

            def issue(leaves, view, m2, rast, g, gt):
                def one():
                    o = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                             viewmatrix=view, gt_depth=gt)
                    if variant == "full":
                        torch.autograd.backward([o[0], o[2], o[3]], [g[0], g[1], g[3]])
                    else:
                        torch.autograd.backward([o[0], o[2], o[3], o[4]], g)
                    return [x.detach() for x in o]
                if views is not None:
                    with views.next():
                        o = one()
                else:
                    o = one()
                return (o, [x.grad for x in leaves + [m2, view]])

            drop = args.drop_inputs and mode == "node"
            if not drop:
                keep = [make(s) for s in scenes]
                torch.cuda.synchronize()
                outs = [issue(*k) for k in keep]
            else:
                for s in scenes:
                    k = make(s)                       # on the caller's stream
                    torch.cuda.current_stream().synchronize()
                    outs.append(issue(*k))
                    del k                             # the view is still queued on its side stream
                    trash = [torch.full((n,), float("nan"), device=dev) for n in (16, W * H, 3 * W * H, 3 * P, 48 * P)]  # noqa: F841
                    del trash
            if views is not None:
                views.join()
            torch.cuda.synchronize()
            L.check_async_errors()
            res[mode] = [([x.detach().cpu().numpy() for x in o], [None if x is None else x.cpu().numpy() for x in gr]) for o, gr in outs]
            del keep, outs
        for v, ((o_a, g_a), (o_b, g_b)) in enumerate(zip(res["reference"], res["node"])):
            for i, (a, b) in enumerate(zip(o_a, o_b)):
                if variant == "light" and i == 6:
                    assert np.allclose(a, b, rtol=1e-5, atol=1e-7), (draws, v, "gau_uncertainty")
                else:
                    assert np.array_equal(a, b), (draws, v, "output", i, P, W, H, variant)
            for i, (a, b) in enumerate(zip(g_a, g_b)):
                assert (a is None) == (b is None), (draws, v, "grad present", i)
                if a is None:
                    continue
                scale = max(float(np.abs(a).max()), 1e-30)
                e = float(np.abs(a - b).max()) / scale
                worst = max(worst, e)
                # two runs of the same backward differ by the arrival order of float atomics, which computeCov2DCUDA's backward
                # amplifies on ill-conditioned Gaussians: 5e-3 on single rows is the reference's own spread
                if e > 5e-3:
                    print("MISMATCH", dict(draw=draws, view=v, grad=i, err=e, P=P, W=W, H=H, variant=variant, tracking=tracking, deg=deg,
                                           nviews=nviews, streams=None if nviews == 1 else "yes", mt=torch.autograd.is_multithreading_enabled()),
                          "ref", a.ravel()[:4], "node", b.ravel()[:4], flush=True)
                    bad += 1
        draws += 1
    L._USE_NODE = True
    assert bad == 0, f"{bad} mismatches"
    print(f"soak_node{' --drop-inputs' if args.drop_inputs else ''}: {draws} draws ({args.seconds:.0f} s, seed {args.seed}) -- outputs identical, worst gradient difference {worst:.2e} of scale")


if __name__ == "__main__":
    main()
