#!/usr/bin/env python3
"""bench.py -- Mviews/s forward+backward @1080p, 500k Gaussians (BASELINE.json `metric`).

One "step" = one view: GaussianRasterizer.forward + backward through the reference-shaped autograd
surface (light variant, SH degree 3, all four pixel-gradient images non-zero, track_off = map_off =
False) on the synth-v1 scene of BASELINE config 3, inputs resident in HBM before the timed region.
By default 21 independent views are in flight on 21 HIP streams (--views-in-flight; three in strict / graph / tracking mode
and with --gpus N; every view is
a complete forward + backward with its own state) and the forward checks its status word lazily
(--sync-mode); `config.ms_per_view_one_stream` is the strictly serial figure, `config.ms_per_view_strict_one_stream` the same in the
library's default status mode (strict), i.e. what a drop-in caller that renders one view at a time sees.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, rank r renders view r of the same Gaussians (weak scaling) and the
per-Gaussian gradients are all-reduced (RCCL, one fused buffer per group of --views-per-allreduce local
views) inside the timed region, as a mapping step over a keyframe batch needs; pose gradients stay per
view.  value = views of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line; `roofline` describes the dominant kernel, `cpu_baseline` the CPU oracle
timed on this host (oracle/ is used here only as the reported baseline, never inside the timed path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "diff-gaussian-rasterization_amd")
for p in (ROOT, PKG, os.path.join(PKG, "light")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (P, W, H, sh_degree)
    "config3": (500_000, 1920, 1080, 3),
    "config2": (100_000, 640, 480, 3),
    "config1": (10_000, 256, 256, 0),
    "config4": (2_000_000, 1920, 1080, 3),   # per-GPU view of BASELINE config 4
    "config5": (5_000_000, 3840, 2160, 3),   # per-GPU view of BASELINE config 5
}

WORKLOAD_NAMES = {
    "config1": "BASELINE config 1's size, on the GPU", "config2": "BASELINE config 2", "config3": "BASELINE config 3, the one the metric is quoted on",
    "config4": "BASELINE config 4: the per-GPU view of its 8 views over 8 GPUs", "config5": "BASELINE config 5: one view of its 32-view batch over 8 GPUs",
}


def algorithmic_bytes(stage, P, V, R, N, M):
    """Minimum HBM bytes of one launch of `stage`: SURVEY.md s8(d)'s per-view model (every stage reads its inputs once and
    writes its outputs once, tile batches staged in LDS, one gradient read-modify-write per instance), term by term as the
    survey lists them, assigned to the kernel that does that part here.  The stages add up to the survey's
    316 P + 566 V + 172 R + 72 N for SH degree 3 (V = visible Gaussians, R = tile instances, N = pixels)."""
    sh = 12 * M
    table = {
        "preprocess_fwd": 44 * P + 8 * P + (sh + 67) * V,   # means/scales/rot/opacity in, radii + tiles_touched out; SH + state per visible
        "scan_blocks": 8 * P,                                # the scan of tiles_touched
        "count_rank": 8 * P + 12 * V,                        # duplicateWithKeys: its reads
        "bin_segments": 8 * P + 12 * V + 12 * R,             # segment binning, first kernel: duplicateWithKeys' reads and key/value pairs
        "bin_tiles": 24 * R + 8 * R,                         # ... second kernel: the sort's read + write and the range detection
        "emit_instances": 12 * R,                            # ... its key/value pairs out
        "sort_tiles": 24 * R,                                # one read + one write of the 12-byte pairs
        "scan_tiles": 8 * R,                                 # range detection
        "render_fwd": 44 * R + 36 * N,                       # id, xy, conic+opacity, rgb, depth per instance; gt in, images + n_contrib out
        "tile_schedule": 0,                                  # (no counterpart: the reference's block -> tile map is its launch grid)
        "zero_scratch": 0,                                   # (no counterpart in the ideal model)
        "zero_counters": 0,                                  # (likewise: the padded tile counters of the fused count)
        "render_bwd": 44 * R + 40 * R + 36 * N,              # instance data again + 10 gradient floats RMW per instance; 9 images in
        "preprocess_bwd": (103 + sh) * V + (56 + sh) * P,    # 295 V + 248 P at degree 3: per-visible inputs, dense outputs
    }
    return float(table[stage])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="light", choices=["light", "full"],
                    help="light = the headline (config 3); full = the -full flavour (use with --workload config2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=5)
    ap.add_argument("--views-in-flight", type=int, default=0,
                    help="independent views (forward+backward each) issued round-robin on this many HIP streams: the "
                         "bandwidth- and latency-bound kernels of one view run under the VALU-bound blend kernels of another, and two "
                         "views' blend kernels fill each other's tails; 1 = strictly one view at a time.  Measured (profiles/r6/"
                         "views_in_flight.txt, ms per step at 20 / 100 steps): 3 views 0.470-0.487 / 0.431-0.438, 5: 0.458-0.483 / "
                         "0.426-0.440, 7: 0.443-0.468 / 0.420-0.428, 9-13 as 7; even counts (4, 8) measure worse than their neighbours.  "
                         "Round 7 (profiles/r7/views_in_flight.txt; the host issues a view in 0.1 ms now): the steady state is 0.415 for "
                         "3 .. 48 views alike, the ends of a short timed region are not -- 20 steps: 3 views 0.476, 7: 0.452-0.463, 11: "
                         "0.444-0.454, 15: 0.446, 21: 0.431-0.442, 32 / 48: 0.437 (views on one stream run one after the other; a stream "
                         "per view lets the GPU start every forward at once and pack the tail).  "
                         "Default (0): 21 for the eager lazy-status mapping step on one GPU, 3 otherwise (strict status, graph replay, tracking step, "
                         "--gpus N: measured worse with more, or not measurable here)")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: record one view per stream into a hipGraph (dgr_amd.multiview.CapturedStep) and replay the "
                         "graphs round-robin instead of issuing the views from Python; pays for host-bound sizes (config 2)")
    ap.add_argument("--tracking", action="store_true",
                    help="not the headline workload: a tracking step instead of a mapping step -- only the viewmatrix requires "
                         "a gradient (map_off), so the backward forms the pose gradient alone (light variant, N=1)")
    ap.add_argument("--tight-cull", action="store_true",
                    help="opt-in alpha-aware tile rectangles (same images and gradients, NOT the reference's integer "
                         "path: num_rendered and the tile lists shrink); off for the headline number")
    ap.add_argument("--views-per-allreduce", type=int, default=1,
                    help="N>1: local views whose gradient arenas are summed before ONE all-reduce (global batch = this "
                         "many views per GPU); 1 (default) = BASELINE config 4's pattern, one view per GPU and one fused "
                         "all-reduce after every view; 4 = config 5's (32 views over 8 GPUs)")
    ap.add_argument("--allreduce", default="blocking", choices=["overlap", "blocking"],
                    help="N>1: blocking (default) = the VIEW'S STREAM waits for its gradient all-reduce, the host does not: with "
                         "several views in flight (the default: three) the other streams keep rendering under the collective, i.e. "
                         "it is overlapped; overlap = additionally defer that stream's wait to the next view issued (what "
                         "--views-in-flight 1 needs to overlap anything)")
    ap.add_argument("--batch", type=int, default=0,
                    help="not the headline workload: a step renders this many camera views (2..8) of the same Gaussians through "
                         "the batched entry points (dgr_amd.batch: one per-Gaussian launch each way for the whole batch, the "
                         "Gaussians' gradients summed over the views in registers); value still counts views.  N>1: one fused "
                         "all-reduce per batch (BASELINE config 5's pattern)")
    ap.add_argument("--batch-streams", type=int, default=0, help="--batch: dgr_set_option('batch_streams') (0 = library default)")
    ap.add_argument("--batch-order", type=int, default=-1, help="--batch: dgr_set_option('batch_order') (-1 = library default)")
    ap.add_argument("--group", type=int, default=0,
                    help="the comparison for --batch: the same views one call at a time, .grad accumulating over this many "
                         "views (autograd's `+=`) before it is reset -- what a mapping iteration over a keyframe batch does "
                         "on the one-view surface; implies --views-in-flight 1")
    ap.add_argument("--blend-wgs-per-cu", type=int, default=-1,
                    help="dgr_set_option('blend_wgs_per_cu'): cap on the blend kernels' workgroups per CU (3..7; 0 = none; default "
                         "-1 = none on one GPU, 7 with --gpus N: one wave slot per SIMD, 64 registers per lane and 20 KB of LDS stay "
                         "free on every CU for the collective's kernels, which otherwise get in only as blend workgroups drain). They "
                         "hold every wave slot of a CU otherwise, and kernels of other streams -- RCCL's with --gpus N, other views' "
                         "front ends -- only get in as blend workgroups drain (one GPU, three views in flight: 0.448 ms per view "
                         "without, 0.445 at 7, 0.454 at 6; profiles/r5/blend_cap_summary.txt)")
    ap.add_argument("--scene", default="synth-v1", choices=["synth-v1", "clustered", "heavy_tail"],
                    help="not the headline workload: clustered = dgr_amd.synth.cluster_scene of the same Gaussians (60 %% of them "
                         "pulled into one region of the frame: tile lists of 51 .. 1135 entries instead of 203 +- 20 %%), the "
                         "case the blend kernels' heaviest-first tile schedule is for; heavy_tail = dgr_amd.synth.heavy_tail_scene (1 %% "
                         "of the Gaussians with an on-screen sigma of 20 .. 150 px: a few splats touching hundreds to thousands of "
                         "tiles each among many touching three), the case the front end's per-Gaussian rectangle walks are for")
    ap.add_argument("--lean-loss", action="store_true",
                    help="not the headline workload: the loss uses colour and depth only (no gradient image for the median depth and "
                         "the depth variance, as in CG-SLAM's losses): the compiled node hands the C ABI NULL for both and the blend "
                         "backward runs its leaner kernel (light variant)")
    ap.add_argument("--sync-mode", default="lazy", choices=["lazy", "strict"],
                    help="lazy: forward's status word is checked one step late (no host sync in the step); "
                         "strict: one blocking status read per forward, like the reference")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("DGR_BENCH_SPAWN") == "1"):
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL rendezvous on
        # 127.0.0.1) through torch.distributed.run, exactly as the driver's own launch line does
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU path")
    if os.environ.get("DGR_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # test hook: every rank on GPU 0 (with DGR_BENCH_BACKEND=gloo, to exercise the N>1 logic on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DGR_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("DGR_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; gloo only for the test hook above
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    os.environ["DGR_SYNC_MODE"] = args.sync_mode
    # One device, one Python thread: the autograd engine's per-device worker thread only adds a hand-off per backward (the
    # caller parks until the worker has run the graph: 20-70 us of futex latency per view, profiles/host_breakdown.py --
    # more than a 640x480 view's kernels leave room for).  With multithreading off the engine runs the graph on the calling
    # thread, on the same streams.  A switch of PyTorch, not of the rasterizer; DGR_BENCH_AUTOGRAD_THREADS=1 leaves it on.
    if os.environ.get("DGR_BENCH_AUTOGRAD_THREADS") != "1":
        torch.autograd.set_multithreading_enabled(False)
    from dgr_amd import _capi, light
    from dgr_amd.multiview import GradientArena, GroupedReduce, ViewStreams, make_settings
    from dgr_amd.synth import make_scene
    if args.variant == "full":
        from dgr_amd import full as V
    else:
        from dgr_amd import light as V
    GaussianRasterizer = V.GaussianRasterizer

    if args.tight_cull:
        _capi.set_option("tight_cull", 1)
    if args.blend_wgs_per_cu < 0:
        args.blend_wgs_per_cu = 7 if dist is not None else 0
    if args.blend_wgs_per_cu:
        _capi.set_option("blend_wgs_per_cu", args.blend_wgs_per_cu)
    if args.batch_streams:
        _capi.set_option("batch_streams", args.batch_streams)
    if args.batch_order >= 0:
        _capi.set_option("batch_order", args.batch_order)
    P, W, H, deg = WORKLOADS[args.workload]
    s = make_scene(P, W, H, seed=0, view_index=rank)  # rank r renders view r of the same Gaussians
    if args.scene == "clustered":
        from dgr_amd.synth import cluster_scene
        s = cluster_scene(s)
    elif args.scene == "heavy_tail":
        from dgr_amd.synth import heavy_tail_scene
        s = heavy_tail_scene(s)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    mapping = not args.tracking
    means3D = t(s.means).requires_grad_(mapping)
    shs = t(s.shs).requires_grad_(mapping)
    opac = t(s.opac).requires_grad_(mapping)
    scales = t(s.scales).requires_grad_(mapping)
    rots = t(s.rots).requires_grad_(mapping)
    means2D = torch.zeros((P, 3), device=dev, requires_grad=mapping)
    view = t(s.view).requires_grad_(True)
    gt = t(s.gt)
    gC, gD, gM, gV = t(s.gC), t(s.gD[None]), t(s.gM[None]), t(s.gV[None])
    if args.variant == "full":
        tt = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)  # noqa: E731
        settings = V.GaussianRasterizationSettings(
            image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=tt(s.bg), scale_modifier=1.0,
            viewmatrix=tt(s.view), projmatrix=tt(s.proj), sh_degree=deg, campos=tt(s.campos), prefiltered=False,
            perspec_matrix=tt(s.persp))
    else:
        settings = make_settings(s, deg, dev, map_off=args.tracking)
    rast = GaussianRasterizer(settings)
    params = [means3D, means2D, shs, opac, scales, rots]
    Vb = max(0, args.batch)
    if Vb:
        if args.variant != "light" or args.tracking:
            raise SystemExit("--batch: light variant, mapping step")
        from dgr_amd import batch as Bm
        from dgr_amd.synth import camera
        # rank r renders views r*Vb .. r*Vb + Vb - 1 of the same Gaussians (view k: the camera at angle 0.05 (k + 1))
        cams = [camera(W, H, 0.05 * (rank * Vb + k + 1)) for k in range(Vb)]
        views_b = t(np.stack([c_[4] for c_ in cams])).requires_grad_(True)
        projs_b, campos_b = t(np.stack([c_[5] for c_ in cams])), t(np.stack([c_[7] for c_ in cams]))
        gts_b = gt[None].expand(Vb, -1, -1).contiguous()
        gCb, gDb, gMb, gVb = (g_[None].expand(Vb, *g_.shape).contiguous() for g_ in (gC, gD, gM, gV))
        means2D = torch.zeros((Vb, P, 3), device=dev, requires_grad=True)
        params = [means3D, means2D, shs, opac, scales, rots]
        rast_b = Bm.GaussianRasterizerBatch(Bm.BatchRasterizationSettings(
            s.H, s.W, s.tanfovx, s.tanfovy, settings.bg, 1.0, views_b, projs_b, deg, campos_b, False, False,
            settings.perspec_matrix, False, False))
    # the fused all-reduce span is found through the gradients (the batch's per-view means2D gradient is not part of it)
    arena = GradientArena([means3D, shs, opac, scales, rots] if Vb else params) if dist is not None else None
    group_i = [0]

    pending = [None]
    G = max(1, args.views_per_allreduce)
    grouped = GroupedReduce(arena, dist, G) if (arena is not None and G > 1) else None

    def step_batch():
        for p_ in params + [views_b]:
            p_.grad = None
        color, radii, depth, median, var, alpha, unc, px = rast_b(
            means3D, means2D, opac, shs=shs, scales=scales, rotations=rots, viewmatrices=views_b, gt_depths=gts_b)
        torch.autograd.backward([color, depth, median, var], [gCb, gDb, gMb, gVb])
        if arena is not None:  # one fused RCCL all-reduce of the batch's summed per-Gaussian gradients
            arena.all_reduce(dist)
        if args.graph:  # (a replayed graph rewrites these very tensors: what the error figure reads after a replay)
            return radii[0], {"dL_dmeans3D": means3D.grad, "dL_dsh": shs.grad, "dL_dopacity": opac.grad, "dL_dscales": scales.grad,
                              "dL_drotations": rots.grad, "dL_dview": views_b.grad}
        return radii[0]

    def step():
        if Vb:
            return step_batch()
        if args.group <= 1 or group_i[0] % args.group == 0:
            for p_ in params + [view]:
                p_.grad = None
        group_i[0] += 1
        outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                    viewmatrix=view, gt_depth=gt)
        if args.variant == "full":
            color, radii, depth, unc = outs
            torch.autograd.backward([color, depth, unc], [gC, gD, gV])
        else:
            color, radii, depth, median, var, alpha, unc, px = outs
            if args.lean_loss:
                torch.autograd.backward([color, depth], [gC, gD])
            else:
                torch.autograd.backward([color, depth, median, var], [gC, gD, gM, gV])
        if grouped is not None:  # sum over the group's local views, then one fused RCCL all-reduce
            grouped.add_view()
        elif arena is not None:  # one fused RCCL all-reduce of the per-Gaussian gradients after every view
            if args.allreduce == "blocking":
                arena.all_reduce(dist)
            else:
                if pending[0] is not None:
                    pending[0].wait()
                pending[0] = arena.all_reduce(dist, async_op=True)
        return radii

    def drain():
        if grouped is not None:
            grouped.flush()
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    args.views_in_flight_requested = args.views_in_flight
    if args.views_in_flight <= 0:
        # twenty-one where the host issues every launch itself and never waits for a status word (the measurements above); three
        # where it waits once per forward (strict: 0.478 with three views against 0.563 with seven), where the views are replayed
        # from graphs (config 2: 0.106 against 0.118), for the tracking step (0.355 against 0.378) and with a collective per view
        args.views_in_flight = 21 if (dist is None and args.sync_mode == "lazy" and not args.graph and not args.tracking) else 3
    # (a batch spreads its views over streams itself: one batch at a time unless --views-in-flight asks for more)
    K = (max(1, args.views_in_flight_requested) if Vb else 1) if (Vb or args.group > 1) else max(1, args.views_in_flight)
    views = ViewStreams(K, dev) if K > 1 else None  # dgr_amd.multiview: independent views on K HIP streams

    captured = []

    def run(n):
        r = None
        if captured:
            for i in range(n):
                r = captured[i % len(captured)].replay()
            return r[0] if isinstance(r, tuple) else radii
        for i in range(n):
            if views is None:
                r = step()
            else:
                with views.next():
                    r = step()
        return first(r)

    def first(r):  # (a captured batch step returns (radii, gradients))
        return r[0] if isinstance(r, tuple) else r

    for _ in range(args.warmup):
        radii = first(step())
    # calibration pass (untimed): every stage bracketed, to find the dominant kernel
    _capi.set_option("profile_every", 1)
    _capi.profile_select("all")
    for _ in range(8):
        radii = first(step())
    drain()
    torch.cuda.synchronize(dev)
    stage_ms, stage_n = {}, {}
    for st_name in _capi.profile_stages():
        tot, n = _capi.profile_read(st_name)
        if n:
            stage_ms[st_name] = tot / n
            stage_n[st_name] = n
    dominant = max(stage_ms, key=stage_ms.get)
    # during the timed region only the dominant kernel is bracketed, on ~16 of its launches whatever --steps is: the two
    # events ride in the kernel's dispatch packet and cost a little of the overlap between streams when attached to
    # every launch
    _capi.profile_select(dominant)
    _capi.set_option("profile_every", int(os.environ.get("DGR_BENCH_PROFILE_EVERY", max(1, args.steps // 16))))

    if args.graph:
        if dist is not None:
            raise SystemExit("--graph is a single-GPU mode")
        from dgr_amd.multiview import CapturedStep
        # (no dispatch-packet events inside a capture: nothing can be bracketed live in a replayed graph, and events on
        #  EVERY launch of a captured stream -- `profile_every` = 1 below 32 steps -- crash this ROCm's capture)
        _capi.profile_select("")
        captured.extend(CapturedStep(step, stream=torch.cuda.Stream(device=dev)) for _ in range(K))
    # one view at a time, for reference (short, untimed by the contract)
    serial_ms = None
    strict_serial_ms = None
    if K > 1:
        drain()
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            radii = first(step())
        drain()
        barrier()
        serial_ms = (time.perf_counter() - t0) / 20 * 1e3
        # ... and as a drop-in sees it: the library's default status mode (strict: every forward waits for its own status word,
        # overflow retried inside the call), one view at a time, CG-SLAM's calling pattern
        if args.sync_mode == "lazy" and not args.graph and dist is None:
            os.environ["DGR_SYNC_MODE"] = "strict"
            for _ in range(3):
                first(step())
            drain()
            barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                radii = first(step())
            drain()
            barrier()
            strict_serial_ms = (time.perf_counter() - t0) / 20 * 1e3
            os.environ["DGR_SYNC_MODE"] = "lazy"
        run(2 * K)  # warm the side streams (allocator pools, status words)
    drain()
    barrier()
    _capi.profile_read(dominant)  # discard the events of the untimed passes
    t0 = time.perf_counter()
    radii = run(args.steps)
    t_issue = time.perf_counter() - t0
    light.check_async_errors()  # status words of every timed step (lazy mode): raises if any forward was invalid
    t_check = time.perf_counter() - t0
    drain()  # the last view's gradient sum completes inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("DGR_BENCH_TRACE") == "1" and rank == 0:
        print(f"[trace] host issue of {args.steps} steps done at {t_issue * 1e3:.3f} ms, status checks at {t_check * 1e3:.3f} ms, "
              f"GPU idle at {elapsed * 1e3:.3f} ms", file=sys.stderr)
    dom_tot, dom_n = _capi.profile_read(dominant)
    _capi.profile_select("")

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # The collective by itself (untimed, after the timed region): the fused span of one view's gradients reduced with nothing
    # else on the GPUs -- the figure the model below is checked against on the first multi-GPU run.
    allreduce_alone_ms, payload_bytes = None, None
    if dist is not None and arena is not None and mapping:
        step()
        drain()
        span = arena.fused_span()
        if span is not None:
            payload_bytes = span.numel() * span.element_size()
            buf = span.clone()
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            barrier()
            allreduce_alone_ms = (time.perf_counter() - t1) / 5 * 1e3

    rank_gpus = [torch.cuda.current_device()]
    if dist is not None:
        ids = [None] * world
        dist.all_gather_object(ids, (rank, torch.cuda.current_device(), torch.cuda.get_device_name(dev)))
        rank_gpus = ids
    if rank == 0:
        V = int((radii > 0).sum().item())
        R = int(_capi_last_num_rendered(P, H, W, dev))
        N = W * H
        # secondary figure of SURVEY.md s8(d): (pixel, Gaussian) pairs evaluated = 2 x sum of n_contrib per view (fwd + bwd)
        pair_evals, lane_lists = None, None
        if args.variant == "light":
            st_ = light._C.rasterize_gaussians(
                settings.bg, means3D.detach(), torch.empty(0, device=dev), opac.detach(), scales.detach(), rots.detach(),
                1.0, torch.empty(0, device=dev), settings.viewmatrix, gt, settings.projmatrix, settings.tanfovx,
                settings.tanfovy, H, W, shs.detach(), deg, settings.campos, False, False)
            nc = torch.zeros(N, dtype=torch.int32, device=dev)
            lib_ = _capi.load()
            cap_ = int(st_[0])  # n_contrib lives in the image buffer: any valid binning capacity will do for this export
            if lib_.dgr_state_export(_capi.stream_handle(), b"n_contrib", P, W, H, int(st_[0]), cap_, st_[7].data_ptr(),
                                     st_[8].data_ptr(), st_[9].data_ptr(), nc.data_ptr()) >= 0:
                pair_evals = 2 * int(nc.to(torch.int64).sum().item())
            # which lane lists this frame's blend kernels walked (option "lane_lists" = 2: decided per frame on the device)
            fl = torch.zeros(1, dtype=torch.int32, device=dev)
            if lib_.dgr_state_export(_capi.stream_handle(), b"sched_flag", P, W, H, int(st_[0]), cap_, st_[7].data_ptr(),
                                     st_[8].data_ptr(), st_[9].data_ptr(), fl.data_ptr()) >= 0:
                lane_lists = "quadrant" if (int(fl.item()) >> 2) & 1 else "half-wave / paired"
        views_per_s = world * args.steps * max(1, Vb) / elapsed
        dom_ms = dom_tot / max(dom_n, 1)
        live = dom_n > 0
        if not live:  # hipGraph replay: the launches are inside the graphs, nothing is bracketed live
            dom_ms = stage_ms[dominant]
        abytes = algorithmic_bytes(dominant, P, V, R, N, 16)
        iso_ms = stage_ms[dominant]  # the kernel with nothing else on the GPU (calibration pass, one view at a time):
        achieved = abytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0  # what a rocprofv3 kernel trace reproduces
        overlap = abytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0   # the same kernel sharing the GPU with the other streams
        # Counter figures come from the committed PMC passes (profiles/pmc_traffic.json; rocprofv3 cannot run inside this
        # process): HBM bytes per launch, and the vector instructions per launch of the two blend kernels.  They go stale
        # when a kernel changes, so the commit they were taken at is printed with them.
        traffic, pmc_commit, pmc, pmc_current = None, None, {}, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file):
            try:
                pmc = json.load(open(pmc_file))
                traffic = pmc.get(args.workload, {}).get(dominant)
                pmc_commit = pmc.get("commit")
                sys.path.insert(0, os.path.join(ROOT, "profiles"))
                from make_pmc_traffic_sha import kernel_sources_sha  # noqa: E402
                pmc_current = pmc.get("kernel_sources_sha16") == kernel_sources_sha()
            except Exception:
                traffic, pmc = None, {}
        # The blend kernels are bound by vector-instruction issue, not by HBM: SQ_INSTS_VALU per launch x 2
        # cycles (a wave64 instruction occupies a SIMD-32 for two cycles) against the cycles of the chip's 1024 SIMDs over the
        # kernel's isolated duration measured in THIS run.  The measured mix averages ~4 cycles per instruction (DPP adds,
        # compares into SGPR pairs, selects 4.2; double-pipe and transcendental operations 5-8), so ~0.5 is this bound's
        # practical ceiling.
        clock_ghz = 2.4  # MI355X nominal shader clock (MI355X_MICROARCH.md); the chip clocks lower under load, which makes `frac` a lower bound
        roofline_valu = None
        if args.variant == "light" and args.workload == "config3" and not args.tight_cull:
            roofline_valu = {}
            for k in ("render_fwd", "render_bwd"):
                n_valu = pmc.get("config3_insts", {}).get(k, {}).get("valu")
                t_ms = stage_ms.get(k)
                if n_valu and t_ms:
                    roofline_valu[k] = {"insts_valu_per_launch": n_valu, "issue_cycles_per_inst": 2, "simds": 1024,
                                        "clock_ghz": clock_ghz, "avg_ms": t_ms,
                                        "frac": n_valu * 2.0 / (1024 * clock_ghz * 1e9 * t_ms * 1e-3),
                                        "hbm_frac": algorithmic_bytes(k, P, V, R, N, 16) / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            roofline_valu["note"] = ("which roof binds the blend kernels: `frac` = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x 2.4 GHz "
                                     "nominal x this run's isolated kernel time), `hbm_frac` = the same kernel against 8 TB/s; counters "
                                     "from profiles/pmc_traffic.json, commit " + str(pmc_commit))
        line = {
            "metric": "Mviews/sec fwd+bwd @1080p, 500k Gaussians",
            "value": views_per_s / 1e6,
            "unit": "Mviews/s",
            "n_gpus": world,
            "views_in_flight": K,  # `value` is a throughput of K independent views in flight; config.ms_per_view_one_stream is the serial figure
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} ({WORKLOAD_NAMES[args.workload]}): synth-v1 seed 0{' CLUSTERED (not a BASELINE scene: dgr_amd.synth.cluster_scene)' if args.scene == 'clustered' else ' HEAVY-TAILED (not a BASELINE scene: dgr_amd.synth.heavy_tail_scene)' if args.scene == 'heavy_tail' else ''}, P={P}, {W}x{H}, SH degree {deg}, {args.variant} variant, "
                                   f"fwd+bwd incl. viewmatrix gradient, "
                                   + (f"one view per step, {K} independent views in flight per GPU" if not Vb else
                                      f"NOT the headline step: {Vb} camera views of the same Gaussians per step through the batched entry "
                                      f"points (gradients of the Gaussians summed over the batch)")
                                   + (f"; .grad accumulates over groups of {args.group} views" if args.group > 1 else "")
                                   + ("" if dist is None else f"; {world} GPUs, rank r renders view r of the same Gaussians (weak scaling, the "
                                      f"per-GPU view is the N=1 workload), exchange pattern of BASELINE config "
                                      f"{'4: one fused all-reduce of the Gaussian gradients after every view' if G == 1 else '5: one fused all-reduce per ' + str(G) + ' local views'}")
                                   + (" -- TRACKING step: pose gradient only (map_off), not the headline mapping step" if args.tracking else "")
                                   + (" -- LEAN LOSS: colour and depth only (no median / variance gradient images), not the headline" if args.lean_loss else ""), "visible": V,
                       "num_rendered": R, "views_per_s": views_per_s, "sync_mode": args.sync_mode,
                       "autograd_engine_thread": torch.autograd.is_multithreading_enabled(),
                       "tile_schedule": {0: "never", 1: "always", 2: "by the frame (skipped on even frames)"}.get(_capi.get_option("tile_schedule")),
                       "views_in_flight": K, "ms_per_view_one_stream": serial_ms,
                       "ms_per_view_strict_one_stream": strict_serial_ms,
                       # which binding ran (the per-step figures depend on it): the compiled autograd node with raw TensorImpl
                       # output views (1), with at::from_blob windows (0), or the ctypes binding (None)
                       "binding": {"compiled": light._C is light._CompiledC, "raw_views": (light._CompiledC.ext.raw_views() if light._CompiledC.ext is not None else None)}, "views_per_step": max(1, Vb), "lane_lists": lane_lists,
                       "ms_per_view": 1e3 * elapsed / args.steps / max(1, Vb), "hipgraph_replay": bool(args.graph),
                       "binning": "two-level segment binning (csrc/segment_binning.hip)" if _capi.get_option("lds_count") else "global tile counters (csrc/binning.hip)",
                       "pair_evals_per_view": pair_evals,
                       "pair_evals_per_s": None if pair_evals is None else pair_evals * views_per_s / world, "tight_cull": bool(args.tight_cull),
                       "view_hbm_frac_one_stream": None if not serial_ms else
                                                   (316 * P + 566 * V + 172 * R + 72 * N) / (serial_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                       "rccl_ranks": None if dist is None else dist.get_world_size(),
                       "rank_gpus": rank_gpus,
                       "gradient_allreduce": (None if dist is None else
                                              (f"one fused RCCL sum of 248 B/Gaussian per {G} local view(s)"
                                               + ("" if G > 1 else f" ({args.allreduce})")) if not Vb else
                                              f"one fused RCCL sum of 248 B/Gaussian per batched step of {Vb} local views (blocking)"),
                       "allreduce_model": None if dist is None else allreduce_model(payload_bytes, world, 1e3 * elapsed / args.steps * max(1, G if not Vb else 1),
                                                                                   allreduce_alone_ms, args.blend_wgs_per_cu),
                       "view_hbm_frac": (316 * P + 566 * V + 172 * R + 72 * N) * (views_per_s / world) / (HBM_PEAK_GBS * 1e9),
                       "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()}},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_from": None if traffic is None else
                         "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, 2 x FETCH_SIZE + WRITE_SIZE), taken at commit " + str(pmc_commit),
                         "traffic_taken_on_these_kernel_sources": pmc_current,
                         "algorithmic_bytes": abytes,
                         "avg_ms": iso_ms, "launches": stage_n[dominant],
                         "how": "HIP events in the kernel's dispatch packet, on the launching stream; `frac` is the kernel alone "
                                "on the GPU (untimed calibration pass, one view at a time: the figure profiles/*_kernel_stats.txt "
                                "reproduces); `*_under_overlap` is the same kernel over the timed region, where it shares the "
                                "CUs with the other streams' kernels",
                         "frac_under_overlap": overlap / HBM_PEAK_GBS, "avg_ms_under_overlap": dom_ms,
                         "launches_under_overlap": dom_n, "measured_in_timed_region": live},
            "roofline_valu": roofline_valu,
        }
        if not args.no_cpu_baseline and world == 1:  # (the CPU baseline: rank 0 at N = 1 only)
            # Beside the headline (eager: every step goes through the autograd surface on the host): the same K steps with
            # one captured view per stream replayed from hipGraphs -- what a fixed-shape SLAM loop can do.  Reported, never
            # `value`; measured in a child process so that nothing of it touches this run.
            if not args.graph and not Vb and args.group <= 1 and not os.environ.get("DGR_BENCH_NO_GRAPH_LINE"):
                line["config"]["ms_per_step_hipgraph_replay"] = graph_replay_line(args)
            # (a bounded sample: 5 views at configs 1-3, 2 at the 2 M / 5 M Gaussian views, whose oracle pass takes 3-8 s)
            runs = args.cpu_runs if P <= 500_000 else min(args.cpu_runs, 2)
            if Vb:
                # a batch: the oracle renders each of its views once (that is also the timed sample) and the Gaussians'
                # gradients are summed over them in double, as the batched backward sums them in registers
                line["cpu_baseline"], ref_grads = cpu_baseline_batch(s, cams, deg)
            else:
                line["cpu_baseline"], ref_grads = cpu_baseline(s, deg, runs, args.variant)
            # second half of BASELINE's metric: gradient max-abs-err against the CPU restatement of the reference, same
            # inputs and loss scaling (pixel-gradient images N(0,1)/(H W)); one extra untimed view on the default stream
            if captured:  # (after an eager step .grad no longer aliases the captured buffers: read what the graph returned)
                step_result = captured[-1].replay()
            else:
                step_result = step()
            torch.cuda.synchronize(dev)
            pairs = {"dL_dmeans3D": means3D, "dL_dsh": shs, "dL_dopacity": opac, "dL_dscales": scales,
                     "dL_drotations": rots, "dL_dview": views_b if Vb else view}
            if captured and isinstance(step_result, tuple) and len(step_result) == 2 and isinstance(step_result[1], dict):
                class _G:  # the gradient tensors recorded in the graph
                    def __init__(self, g): self.grad = g
                pairs = {k: _G(step_result[1][k]) for k in pairs if k in step_result[1]}
            pairs = {k: v for k, v in pairs.items() if v.grad is not None}  # (--tracking: the pose gradient only)
            errs = {k: float(np.abs(v.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
                                    - np.asarray(ref_grads[k], np.float64).reshape(-1)).max()) for k, v in pairs.items()}
            line["config"]["grad_max_abs_err"] = dict(
                errs, max=max(errs.values()), scale={k: float(np.abs(ref_grads[k]).max()) for k in pairs},
                note="end to end (HIP forward feeding HIP backward) vs the CPU oracle, BASELINE's loss scaling; north_star's "
                     "tolerance is 1e-5 abs.  The default alpha path evaluates alpha with the oracle's bits (one fp32-only "
                     "polynomial expf, <= 0.9 ulp, evaluated operation for operation by both: csrc/exact_math.h, "
                     "oracle/dgr_oracle.cpp: expf_p32): the alpha image IS the oracle's, so the light backward's "
                     "T_final = 1 - alpha_image amplifies nothing (alpha_mode 1, fast: 5.8e-5 on dL_dview; DESIGN.md s4.6)"
                     + ("" if args.variant == "light" else "; full variant: dL_dview follows the well-defined reading of ComputePG"))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is flushed at exit when stdout is a pipe: push it
        # out now so that the JSON line is the last line of this process's stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7  # MI355X: seven xGMI links per GPU, ~153 GB/s each way (MI355X_MICROARCH.md)


def allreduce_model(payload_bytes, world, ms_per_collective_interval, alone_ms, blend_cap):
    """What the gradient all-reduce costs on point-to-point xGMI, for the first multi-GPU run to be checked against
    (SURVEY.md s8(e)): a ring moves 2 (N-1)/N x payload through ONE link per GPU; the direct algorithm (reduce-scatter +
    all-gather with every peer at once) moves 2 x payload / N through each of the N-1 links.  `ceiling_*` = weak-scaling
    efficiency if the collective were the only loss: serial = nothing overlaps it, overlapped = it hides under the other
    views' kernels as long as it is shorter than the interval between collectives."""
    if not payload_bytes or world < 2:
        return {"payload_bytes": payload_bytes, "note": "one rank: nothing crosses a link"} if payload_bytes else None
    bw = XGMI_LINK_GBS * 1e9
    ring = 2.0 * (world - 1) / world * payload_bytes / bw * 1e3
    direct = 2.0 * payload_bytes / world / bw * 1e3
    t = ms_per_collective_interval
    return {"payload_bytes": payload_bytes, "link_GBps": XGMI_LINK_GBS, "links_per_gpu": XGMI_LINKS,
            "ring_ms": ring, "direct_ms": direct, "measured_alone_ms": alone_ms, "ms_between_collectives": t,
            "ceiling_serial": {"ring": t / (t + ring), "direct": t / (t + direct)},
            "ceiling_overlapped": {"ring": min(1.0, t / ring), "direct": min(1.0, t / direct)},
            "blend_wgs_per_cu": blend_cap,
            "note": "ms_between_collectives is THIS run's measured interval (it already contains whatever the collective cost here); "
                    "an efficiency below ceiling_overlapped.direct means the collective's kernels were starved or serialised"}


def graph_replay_line(args):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--graph",
           "--no-cpu-baseline", "--workload", args.workload, "--variant", args.variant, "--views-in-flight", str(args.views_in_flight_requested),
           "--scene", args.scene, "--sync-mode", "lazy"]
    # (every flag that shapes the workload goes to the child, so that the figure sits beside the line it belongs to)
    if args.tight_cull:
        cmd.append("--tight-cull")
    if args.blend_wgs_per_cu:
        cmd += ["--blend-wgs-per-cu", str(args.blend_wgs_per_cu)]
    if args.tracking:
        cmd.append("--tracking")
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        return float(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"])
    except Exception:
        return None


def spawn_ranks(n):
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("DGR_BENCH_SPAWN", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if "--gpus" not in " ".join(sys.argv[1:]):
        cmd += ["--gpus", str(n)]
    return subprocess.call(cmd, env=env)


def _capi_last_num_rendered(P, H, W, dev):
    from dgr_amd import light
    return light._capacity_cache.get((dev.index, P, H, W), 0)


def host_cpu():
    """(model name, sockets, physical cores, hardware threads) from /proc/cpuinfo."""
    model, cores, threads = "unknown", set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif k == "processor":
                threads += 1
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    sockets = len({p for p, _ in cores}) or 1
    ncores = len(cores) or (os.cpu_count() or 1)
    try:
        ncores = min(ncores, len(os.sched_getaffinity(0)))  # a container may expose fewer CPUs than the host lists
    except AttributeError:
        pass
    return model, sockets, ncores, threads or (os.cpu_count() or 1)


def cpu_baseline(s, deg, runs, variant="light"):
    """The CPU oracle (OpenMP restatement of the reference path) timed on this host as BASELINE.md s3 prescribes: built
    -O3 -march=native here, OMP threads = physical cores, 1 warm-up + `runs` repetitions of the full view, forward and
    backward timed separately.  A reported baseline, not the thing measured above."""
    from oracle import oracle as O
    model, sockets, cores, threads = host_cpu()
    O.build()
    O.use_native(True)
    O.set_threads(cores)
    # the restatement evaluates the expf of the library's alpha mode (0: the fp32 polynomial, both sides' default; 2: glibc's form)
    from dgr_amd import _capi as _c
    O.set_exp_mode(1 if _c.get_option("alpha_mode") == 2 else 0)
    grads = {}
    tf, tb = [], []

    def once(record):
        t0 = time.perf_counter()
        if variant == "full":
            st, out = O.full_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                     s.tanfovx, s.tanfovy, s.H, s.W, s.shs, deg, s.campos)
            t1 = time.perf_counter()
            g = O.full_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                                s.tanfovy, s.gC, s.gD, s.gV, s.shs, deg, s.campos, s.persp)
        else:
            st, out = O.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                      s.tanfovx, s.tanfovy, s.H, s.W, s.shs, deg, s.campos)
            t1 = time.perf_counter()
            g = O.light_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.proj, s.tanfovx, s.tanfovy,
                                 s.gC, s.gD, s.gM, s.gV, s.gt, s.shs, deg, s.campos, out["opacity_map"], s.persp)
        t2 = time.perf_counter()
        grads.update(g)
        if record:
            tf.append(t1 - t0)
            tb.append(t2 - t1)

    try:
        once(False)
        for _ in range(max(runs, 1)):
            once(True)
    finally:
        O.use_native(False)
    tot = [a + b for a, b in zip(tf, tb)]
    med = float(np.median(tot))
    return {"value": 1.0 / med / 1e6, "unit": "Mviews/s", "cores": cores, "kind": "port",
            "host": {"cpu": model, "sockets": sockets, "physical_cores": cores, "hardware_threads": threads,
                     "omp_threads": cores},
            "sample": f"{len(tot)} complete fwd+bwd views ({variant} variant) of the same workload after 1 warm-up: median {med:.3f} s (min "
                      f"{min(tot):.3f} s; forward median {float(np.median(tf)):.3f} s, backward {float(np.median(tb)):.3f} s), "
                      f"oracle built -O3 -march=native, {cores} OpenMP threads = physical cores of {sockets} x {model}"}, grads


def cpu_baseline_batch(s, cams, deg):
    """The oracle on every view of a batch (cams: dgr_amd.synth.camera tuples): each view's forward + backward timed once after
    one warm-up view; returns the baseline object and the reference gradients of the BATCH -- the Gaussians' gradients summed
    over the views in double, the pose gradients stacked per view."""
    from oracle import oracle as O
    from dgr_amd import _capi as _c
    model, sockets, cores, threads = host_cpu()
    O.build()
    O.use_native(True)
    O.set_threads(cores)
    O.set_exp_mode(1 if _c.get_option("alpha_mode") == 2 else 0)
    tot, sums, dviews = [], {}, []

    def one(cam, record):
        sv = s._replace(view=cam[4], proj=cam[5], persp=cam[6], campos=cam[7])
        t0 = time.perf_counter()
        st, out = O.light_forward(sv.bg, sv.means, None, sv.opac, sv.scales, sv.rots, 1.0, None, sv.view, sv.gt, sv.proj,
                                  sv.tanfovx, sv.tanfovy, sv.H, sv.W, sv.shs, deg, sv.campos)
        g = O.light_backward(st, sv.bg, sv.means, None, sv.scales, sv.rots, 1.0, None, sv.view, sv.proj, sv.tanfovx, sv.tanfovy,
                             sv.gC, sv.gD, sv.gM, sv.gV, sv.gt, sv.shs, deg, sv.campos, out["opacity_map"], sv.persp)
        dt = time.perf_counter() - t0
        if record:
            tot.append(dt)
            for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
                sums[k] = np.asarray(g[k], np.float64) + sums.get(k, 0.0)
            dviews.append(np.asarray(g["dL_dview"], np.float64).reshape(4, 4))

    try:
        one(cams[0], False)
        for cam in cams:
            one(cam, True)
    finally:
        O.use_native(False)
    sums["dL_dview"] = np.stack(dviews)
    med = float(np.median(tot))
    return {"value": 1.0 / med / 1e6, "unit": "Mviews/s", "cores": cores, "kind": "port",
            "host": {"cpu": model, "sockets": sockets, "physical_cores": cores, "hardware_threads": threads, "omp_threads": cores},
            "sample": f"the {len(tot)} views of one batch, each a complete fwd+bwd (light variant) after 1 warm-up view: median {med:.3f} s per "
                      f"view (min {min(tot):.3f} s), oracle built -O3 -march=native, {cores} OpenMP threads = physical cores of "
                      f"{sockets} x {model}"}, sums


if __name__ == "__main__":
    main()
