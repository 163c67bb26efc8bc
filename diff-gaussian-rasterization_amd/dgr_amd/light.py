"""Host-side mirror of diff-gaussian-rasterization-light/diff_gaussian_rasterization/__init__.py.

Same names, argument order, return arity and error behaviour as the reference module
(`GaussianRasterizationSettings`, `GaussianRasterizer`, `rasterize_gaussians`, `_RasterizeGaussians`),
with `_C.*` replaced by calls into the gfx950 C ABI (include/dgr_hip.h).  Differences, all invisible
to a caller of the autograd surface:
  * the per-pixel [H*W,4,4] pose-gradient buffer and its torch.sum (reference __init__.py:160-161)
    do not exist; the C ABI returns the reduced [4,4];
  * forward runs without a mid-pipeline host sync when a binning capacity is known from an earlier
    call of the same shape (one status read at the end instead); DGR_FORWARD_MODE=callback selects the
    strict mirror with allocation callbacks and the reference's blocking read;
  * the debug path's NameError (`deoth_var`, reference __init__.py:93) is not reproduced.
"""
import ctypes as C
import os
import weakref
from typing import NamedTuple

import threading

import torch
import torch.nn as nn

from . import _capi


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def _f32c(t, dev):
    """contiguous fp32 tensor on `dev` (L/rasterize_points.cu:101-125 calls .contiguous() on every input)."""
    if t.dtype is torch.float32 and t.is_contiguous() and t.device == dev:
        return t  # the common case, checked first
    if t.device != dev:
        t = t.to(dev)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# binning capacity learned per (device, P, H, W): largest num_rendered seen for that shape
_capacity_cache = {}

# DGR_SYNC_MODE=lazy: once a shape's num_rendered is known, forward performs NO host synchronisation.  The binning
# buffer is over-provisioned (1.5x the largest count seen); the device status word {num_rendered, overflow,
# prefiltered violation, -} is copied asynchronously to pinned host memory behind an event, and inspected when a
# later call starts (or by check_async_errors()), by which time the event has long fired.  An overflow or a
# `prefiltered` violation therefore raises one or two calls late.  Default ("strict"): one status read at the end of
# every forward, like the reference's blocking copy of num_rendered (L/cuda_rasterizer/rasterizer_impl.cu:287).
_pending_status = []   # [(ticket of dgr_status_post, key)]
# Status words left unread when a forward is issued.  1: view i is issued once view i-2's forward has reported -- with several
# views in flight on several streams (dgr_amd.multiview.ViewStreams) that starves a stream whose previous view has finished
# while the view whose report the host waits for is still in its blend kernels; ViewStreams raises it to its number of streams
# (three views in flight, config 3: 0.483 -> 0.473 ms per step over 20 steps, 0.434 -> 0.428 over 100: profiles/r6/lazy_depth.txt).
# The price: an overflow or a `prefiltered` violation is reported up to depth + 1 calls late instead of two.
_LAZY_DEPTH = max(1, int(os.environ.get("DGR_LAZY_DEPTH", "1")))


def lazy_depth():
    return _LAZY_DEPTH


def set_lazy_depth(n):
    """See _LAZY_DEPTH above; returns the previous value."""
    global _LAZY_DEPTH
    prev, _LAZY_DEPTH = _LAZY_DEPTH, max(1, int(n))
    return prev
_last_status = {}      # key -> the most recent status word read back for that shape


def _sync_mode():
    return os.environ.get("DGR_SYNC_MODE", "strict")


# Status words of forwards recorded into a hipGraph (torch.cuda.graph): nothing can be read back while capturing, so the
# device tensors are kept and inspected on request after a replay (check_captured_status()).
_captured_status = []     # weak references: a status word lives as long as the capture that owns it
_capture_keepalive = []   # strong references collected during ONE capture; CapturedStep takes them over


def _post_status(status, key):
    if torch.cuda.is_current_stream_capturing():
        _captured_status.append(weakref.ref(status))  # kept alive by the captured step's results (CapturedStep.keep)
        _capture_keepalive.append(status)
        return
    # the library copies the word to pinned host memory behind an event (include/dgr_hip.h: dgr_status_post)
    ticket = _capi.load().dgr_status_post(_capi.stream_handle(status.device.index), status.data_ptr())
    _check(ticket)
    _pending_status.append((ticket, key))


# Lazy mode is only as safe as its capacity guess (1.5 x the largest count seen for the shape): a count that GROWS -- the camera
# closing in, splats being scaled up, a map that densifies without changing P -- reaches it within a few frames, and frames rendered
# past it have empty tile lists.  Three guards (round 9):
#   * the blend kernels write NaN images for an overflowed forward (csrc/render_light.hip), never a plausible empty frame, and its
#     backward (empty lists) yields zero gradients: nothing wrong reaches an optimiser unnoticed;
#   * a shape whose count grew by more than 25 % between two status reads, came within 20 % of the capacity it was rendered with,
#     or overflowed, is UNSETTLED: its next forwards run strict (exact count, overflow retried inside the call) until three
#     reads in a row show less than 10 % growth;
#   * check_async_errors() before optimizer.step() reads every outstanding word (the forwards' words arrive while their backward
#     kernels are still queued: the wait costs the GPU nothing) and raises -- dgr_amd.slam's loops and examples/mapping.py do.
_unsettled = {}        # key -> strict forwards still to run
_GROWTH_STRICT, _SETTLED_READS = 1.25, 3


def _note_growth(key, prev, s, capacity_used=None):
    grew = prev > 0 and s[0] > _GROWTH_STRICT * prev
    near = capacity_used is not None and s[0] > 0.8 * capacity_used
    if s[1] or grew or near:
        _unsettled[key] = _SETTLED_READS
    elif key in _unsettled and prev > 0 and s[0] <= 1.1 * prev:
        _unsettled[key] -= 1
        if _unsettled[key] <= 0:
            del _unsettled[key]


def _strict_read(key, cap, rendered):
    """A strict forward returned its exact count (and retried an overflow inside the call)."""
    last = _last_status.get(key)
    if _sync_mode() == "lazy":
        _note_growth(key, last[0] if last else 0, (rendered, 0, 0, 0))
        _last_status[key] = [rendered, 0, 0, last[3] if last else 0]
    _capacity_cache[key] = max(cap, rendered)


def _check_oldest():
    ticket, key = _pending_status.pop(0)
    buf = (C.c_int * 4)()
    _check(_capi.load().dgr_status_poll(ticket, 1, buf))  # waits for that forward only
    s = list(buf)
    prev = _last_status.get(key, (0, 0, 0, 0))[0]
    _note_growth(key, prev, s, capacity_used=int(_capacity_cache.get(key, 0) * 1.5) + 4096)
    _capacity_cache[key] = max(_capacity_cache.get(key, 0), s[0])
    _last_status[key] = s
    if s[2]:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    if s[1]:
        raise RuntimeError(f"dgr_hip: binning buffer overflow in an earlier lazily-checked forward (needed {s[0]} "
                           f"instances); its outputs were invalid -- rerun that step")


def check_captured_status():
    """After replaying a graph that contains forwards: raises if one of them overflowed its binning buffer (the graph
    was captured with a smaller scene than it is replayed on) or hit the prefiltered trap.  Blocks on the device.
    Status words whose graph no longer exists (the tensor's storage was freed with the graph's pool) are dropped."""
    _captured_status[:] = [r for r in _captured_status if r() is not None]
    for ref in _captured_status:
        st = ref()
        if st is None:
            continue
        for s in st.reshape(-1, 4).tolist():  # (a batched forward keeps the [V,4] status words of its views in one tensor)
            if s[2]:
                raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
            if s[1]:
                raise RuntimeError(f"dgr_hip: binning buffer overflow in a graph-captured forward (needed {s[0]} instances): "
                                   f"re-capture after an eager warm-up on the larger scene")


def check_async_errors():
    """Raises if an earlier lazily-checked forward overflowed its binning buffer or hit the prefiltered trap."""
    while _pending_status:
        _check_oldest()


# dgr_amd.multiview.ViewStreams.before_backward(): an event the rasterizer backward THAT RUNS ON A GIVEN STREAM makes that
# stream wait for once its kernels are issued -- i.e. before autograd goes on to the activations' backward and to the
# accumulation into the leaves' .grad, the only part of a view's backward that touches state shared with the previous view.
# Keyed by the raw stream handle: the wait is set by the thread that calls loss.backward() and consumed by the autograd
# engine's thread (which runs the node on the forward's stream), and two ViewStreams objects -- or two threads -- never
# see each other's entries.  ViewStreams drops a view's entry when the view's block ends (a backward that never reached
# the rasterizer must not leave a stale wait behind).
_post_backward_waits = {}
_post_backward_lock = threading.Lock()


def _set_post_backward_wait(stream, event):
    with _post_backward_lock:
        _post_backward_waits[int(stream.cuda_stream)] = (stream, event)  # (also keeps both objects alive)
    if _CompiledC.ext is not None:  # the compiled autograd node consumes the wait inside the engine, without Python
        _CompiledC.ext.set_post_backward_wait(int(stream.cuda_stream), int(event.cuda_event))


def _drop_post_backward_wait(stream):
    # (the C++ table is dropped whether or not the Python table still has the entry: the Python Function's backward may have
    #  consumed it -- debug = True, DGR_AUTOGRAD=python -- and the raw event handle the extension holds must not outlive the Event)
    key = int(stream.cuda_stream)
    if _post_backward_waits:
        with _post_backward_lock:
            _post_backward_waits.pop(key, None)
    if _CompiledC.ext is not None:
        _CompiledC.ext.drop_post_backward_wait(key)


def _consume_post_backward_wait():
    if not _post_backward_waits:
        return
    key = int(torch.cuda.current_stream().cuda_stream)
    with _post_backward_lock:
        w = _post_backward_waits.pop(key, None)
    if w is not None:
        if _CompiledC.ext is not None:  # the extension's copy of the entry goes with it (it holds the event's raw handle)
            _CompiledC.ext.drop_post_backward_wait(key)
        w[0].wait_event(w[1])


# The flat gradient arena of a backward is found through the gradients themselves (`p.grad._base`, see
# dgr_amd.multiview.GradientArena): no module-level "last arena" is kept, so nothing outlives the gradients and threads
# cannot see each other's arenas.  SPAN_SEGMENTS names the leading segments that form the all-reduce payload.
SPAN_SEGMENTS = ("means3D", "means2D", "sh", "opacity", "scales", "rotations")


def _grad_arena(P, M, f32):
    """One flat arena holds every per-Gaussian gradient (returned as views), ordered so that the tensors a
    mapping step all-reduces across GPUs -- means3D, means2D, sh, opacity, scales, rotations -- are one
    contiguous span (dgr_amd.multiview.GradientArena).  Every row is written by the kernels (zeros for
    invisible Gaussians), so the arena is not zero-filled."""
    shapes = [("means3D", (P, 3)), ("means2D", (P, 3)), ("sh", (P, M, 3)), ("opacity", (P, 1)),
              ("scales", (P, 3)), ("rotations", (P, 4)), ("cov3D", (P, 6)), ("colors", (P, 3))]
    offs, o = {}, 0
    for name, shp in shapes:
        n = 1
        for d_ in shp:
            n *= d_
        offs[name] = (o, n, shp)
        o += (n + 63) // 64 * 64  # 256-byte aligned segments (vector stores in the kernels)
    arena = (torch.empty if P else torch.zeros)((max(o, 1),), **f32)
    return {name: arena[a:a + n].view(shp) for name, (a, n, shp) in offs.items()}


def _early_status(lib):
    """{num_rendered, -, prefiltered violation, -} of the presized forward just issued (include/dgr_hip.h: early status).
    The buffer is per call: ctypes releases the GIL during the wait, so a shared one could be read by another thread."""
    buf = (C.c_int * 4)()
    _check(lib.dgr_early_status_wait(buf))
    return list(buf)


def _check(rc):
    if rc >= 0:
        return rc
    msg = _capi.last_error()
    if rc == _capi.DGR_ERR_PREFILTERED:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    if rc == _capi.DGR_ERR_BAD_ARGUMENT:
        raise RuntimeError(f"dgr_hip: bad argument: {msg}")
    raise RuntimeError(f"dgr_hip: error {rc}: {msg}")


def set_tight_culling(on=True):
    """Opt in to alpha-aware tile rectangles (include/dgr_hip.h: dgr_set_option "tight_cull"): same images and gradients,
    ~40 % fewer tile instances; `num_rendered` and the opaque state buffers are then not the reference's.  Process-wide."""
    _capi.set_option("tight_cull", 1 if on else 0)


def _device_guarded(arg_index):
    """Runs a `_C` function with its tensors' device current (kernels, events and the stream handle all belong to the
    device of `means3D`, whichever device the caller had selected)."""
    def deco(fn):
        def wrapped(*a, **kw):
            with _capi.on_device(a[arg_index].device):
                return fn(*a, **kw)
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return staticmethod(wrapped)
    return deco


class _C:
    """Functions with the signatures of the reference's pybind11 module `_C` (L/ext.cpp:15-19)."""

    @_device_guarded(1)
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                            cov3D_precomp, viewmatrix, gt_depth, projmatrix, tan_fovx, tan_fovy,
                            image_height, image_width, sh, degree, campos, prefiltered, debug):
        # L/rasterize_points.cu:35-129
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        lib = _capi.load()
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)")
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        means3D = _f32c(means3D, dev)
        background, colors, opacity = _f32c(background, dev), _f32c(colors, dev), _f32c(opacity, dev)
        scales, rotations, cov3D_precomp = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3D_precomp, dev)
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, dev), _f32c(projmatrix, dev), _f32c(campos, dev)
        gt_depth, sh = _f32c(gt_depth, dev), _f32c(sh, dev)
        M = sh.size(1) if sh.numel() != 0 else 0

        out_color = torch.empty((3, H, W), **f32)
        out_depth = torch.empty((1, H, W), **f32)
        out_median = torch.empty((1, H, W), **f32)
        out_var = torch.empty((1, H, W), **f32)
        out_alpha = torch.empty((1, H, W), **f32)
        # radii is written for every Gaussian and the two median statistics are zeroed by the C ABI (stream memsets)
        mk = torch.empty if P else torch.zeros
        radii = mk((P,), **i32)
        gau_unc = mk((P, 1), **f32)
        gau_px = mk((P, 1), **i32)
        u8 = dict(dtype=torch.uint8, device=dev)
        st = _capi.stream_handle(dev.index)
        p = _capi.ptr

        common = (P, int(degree), M, p(background), W, H, p(means3D), p(sh), p(colors), p(opacity), p(scales),
                  float(scale_modifier), p(rotations), p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos),
                  float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), p(out_color), p(out_depth),
                  p(out_median), p(out_alpha), p(gt_depth), p(out_var), p(gau_unc), p(gau_px), p(radii))

        if os.environ.get("DGR_FORWARD_MODE", "presized") == "callback" or P == 0:
            bufs = {k: torch.empty((0,), **u8) for k in ("geom", "binning", "img")}

            def mk(name):
                def cb(nbytes, _user):
                    bufs[name] = torch.empty((max(int(nbytes), 1),), **u8)
                    return bufs[name].data_ptr()
                return _capi.ALLOC_FN(cb)
            cbs = [mk("geom"), mk("binning"), mk("img")]
            rendered = _check(lib.dgr_light_forward(st, cbs[0], cbs[1], cbs[2], None, *common, int(bool(debug))))
            geomBuffer, binningBuffer, imgBuffer = bufs["geom"], bufs["binning"], bufs["img"]
        else:
            geomBuffer = torch.empty((lib.dgr_geometry_bytes(P),), **u8)
            imgBuffer = torch.empty((lib.dgr_image_bytes(W, H),), **u8)
            status = torch.empty((4,), **i32)
            key = (dev.index, P, H, W)
            cap = _capacity_cache.get(key, 0)
            lazy = _sync_mode() == "lazy" and cap > 0
            if lazy:
                # status words of earlier calls have long completed: reading them does not stall the pipeline
                while len(_pending_status) > _LAZY_DEPTH and not torch.cuda.is_current_stream_capturing():
                    _check_oldest()
                cap = int(cap * 1.5) + 4096
                binningBuffer = torch.empty((lib.dgr_binning_bytes(cap, W, H),), **u8)
                _check(lib.dgr_light_forward_presized(st, p(geomBuffer), p(binningBuffer), cap, p(imgBuffer),
                                                      p(status), *common))
                _post_status(status, key)
                rendered = _capacity_cache[key]
                return (rendered, out_color, out_depth, out_median, out_var, out_alpha, radii, geomBuffer,
                        binningBuffer, imgBuffer, gau_unc, gau_px)
            cap = int(cap * 1.25) + 4096 if cap else 4 * P + 4096
            while True:
                binningBuffer = torch.empty((lib.dgr_binning_bytes(cap, W, H),), **u8)
                lib.dgr_early_status_arm()
                _check(lib.dgr_light_forward_presized(st, p(geomBuffer), p(binningBuffer), cap, p(imgBuffer),
                                                      p(status), *common))
                s = _early_status(lib)  # the one host wait of this forward: until num_rendered is known, a tenth of the
                if s[2]:                # way into the forward -- not until the forward has finished
                    raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
                rendered = s[0]
                _capacity_cache[key] = max(_capacity_cache.get(key, 0), rendered)
                if rendered <= cap:
                    break
                cap = int(rendered * 1.1) + 4096  # overflow: every tile list was left empty; run again
            if debug:
                torch.cuda.synchronize(dev)
        return (rendered, out_color, out_depth, out_median, out_var, out_alpha, radii, geomBuffer, binningBuffer,
                imgBuffer, gau_unc, gau_px)

    @_device_guarded(1)
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                     dL_dout_depth, dL_dout_median_depth, dL_dout_depth_var, gt_depth, sh, degree,
                                     campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug,
                                     perspec_matrix, track_off, map_off, need_gaussian_grads=True):
        # L/rasterize_points.cu:131-236.  `need_gaussian_grads=False` (an extension: the autograd Function passes it when
        # no Gaussian input requires a gradient, i.e. tracking) returns None for the eight per-Gaussian gradients and lets
        # the library skip their dense rows; the pose gradient is the same.
        lib = _capi.load()
        dev = means3D.device
        P = means3D.size(0)
        H, W = dL_dout_color.size(1), dL_dout_color.size(2)
        f32 = dict(dtype=torch.float32, device=dev)
        means3D = _f32c(means3D, dev)
        background, colors = _f32c(background, dev), _f32c(colors, dev)
        scales, rotations, cov3D_precomp = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3D_precomp, dev)
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, dev), _f32c(projmatrix, dev), _f32c(campos, dev)
        gt_depth, sh, alphas = _f32c(gt_depth, dev), _f32c(sh, dev), _f32c(alphas, dev)
        perspec_matrix = _f32c(perspec_matrix, dev)
        gC, gD = _f32c(dL_dout_color, dev), _f32c(dL_dout_depth, dev)
        gM, gV = _f32c(dL_dout_median_depth, dev), _f32c(dL_dout_depth_var, dev)
        M = sh.size(1) if sh.numel() != 0 else 0
        if need_gaussian_grads:
            seg = _grad_arena(P, M, f32)
            dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dopacity = seg["means3D"], seg["means2D"], seg["sh"], seg["opacity"]
            dL_dscales, dL_drotations, dL_dcov3D, dL_dcolors = seg["scales"], seg["rotations"], seg["cov3D"], seg["colors"]
        else:
            dL_dmeans3D = dL_dmeans2D = dL_dsh = dL_dopacity = dL_dscales = dL_drotations = dL_dcov3D = dL_dcolors = None
            map_off = True  # nobody reads the per-Gaussian sums: the blend kernel forms the three pose sums only
        # [1,4,4]: the reference binding returns the per-pixel [H*W,4,4] buffer and its __init__.py sums dim 0
        # (L/rasterize_points.cu:186,235, L/__init__.py:160-161); one already-reduced "pixel" keeps that code working
        dL_dview = torch.empty((1, 4, 4), **f32)
        # (option "deterministic_grads": + 64 bytes per tile instance; R = num_rendered, or the capacity of a lazy forward)
        scratch = torch.empty((max(lib.dgr_light_backward_scratch_bytes_r(P, W, H, int(R)), 1),), dtype=torch.uint8, device=dev)
        p = _capi.ptr
        q = lambda t: None if t is None else p(t)  # noqa: E731
        _check(lib.dgr_light_backward(
            _capi.stream_handle(dev.index), P, int(degree), M, int(R), p(background), W, H, p(means3D), p(sh), p(colors),
            p(alphas), p(scales), float(scale_modifier), p(rotations), p(cov3D_precomp), p(viewmatrix),
            p(projmatrix), p(campos), float(tan_fovx), float(tan_fovy), p(radii), p(geomBuffer), p(binningBuffer),
            p(imageBuffer), p(gC), p(gD), p(gM), p(gV), q(dL_dmeans2D), None, q(dL_dopacity), q(dL_dcolors), None,
            q(dL_dmeans3D), q(dL_dcov3D), q(dL_dsh), q(dL_dscales), q(dL_drotations), int(bool(debug)), None,
            p(perspec_matrix), p(dL_dview), None, p(gt_depth), int(bool(track_off)), int(bool(map_off)),
            p(scratch), scratch.numel()))
        return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
                dL_dview)

    @_device_guarded(0)
    def mark_visible(means3D, viewmatrix, projmatrix):
        # L/rasterize_points.cu:238-256
        lib = _capi.load()
        dev = means3D.device
        P = means3D.size(0)
        present = torch.zeros((P,), dtype=torch.bool, device=dev)
        if P != 0:
            means3D, viewmatrix, projmatrix = _f32c(means3D, dev), _f32c(viewmatrix, dev), _f32c(projmatrix, dev)
            _check(lib.dgr_mark_visible(_capi.stream_handle(dev.index), P, _capi.ptr(means3D), _capi.ptr(viewmatrix),
                                        _capi.ptr(projmatrix), present.data_ptr()))
        return present


class _CompiledC:
    """The same three functions over the compiled torch extension (csrc/torch_ext.cpp -> dgr_amd/_dgr_torch_ext.so), the
    counterpart of the reference's pybind11 `_C` (L/ext.cpp:15-19): tensor allocation, argument marshalling and the C-ABI
    call happen in C++; only the capacity / status policy above stays in Python.  Selected when the extension is built
    (DGR_BINDING=ctypes forces the ctypes class)."""

    ext = None

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, gt_depth, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                            campos, prefiltered, debug):
        ext = _CompiledC.ext
        P, H, W = means3D.size(0) if means3D.dim() else 0, int(image_height), int(image_width)
        key = (means3D.device.index, P, H, W)
        mode, use, cap = _binning_policy(key, P)
        (rendered, ticket, _, status, color, depth, median, var, alpha, radii, geom, binning, img, unc, px) = ext.light_forward(
            background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix,
            gt_depth, projmatrix, float(tan_fovx), float(tan_fovy), H, W, sh, int(degree), campos, bool(prefiltered),
            bool(debug), use, mode)
        if mode == 2:
            if ticket >= 0:
                _pending_status.append((ticket, key))
            else:  # recorded into a hipGraph: nothing can be read back now
                _captured_status.append(weakref.ref(status))
                _capture_keepalive.append(status)
            rendered = _capacity_cache[key]
        elif mode == 1:
            _strict_read(key, cap, rendered)
        return (rendered, color, depth, median, var, alpha, radii, geom, binning, img, unc, px)

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                     viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                     dL_dout_median_depth, dL_dout_depth_var, gt_depth, sh, degree, campos, geomBuffer, R,
                                     binningBuffer, imageBuffer, alphas, debug, perspec_matrix, track_off, map_off,
                                     need_gaussian_grads=True):
        g = _CompiledC.ext.light_backward(
            background, means3D, radii, colors, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix,
            projmatrix, float(tan_fovx), float(tan_fovy), dL_dout_color, dL_dout_depth, dL_dout_median_depth,
            dL_dout_depth_var, gt_depth, sh, int(degree), campos, geomBuffer, int(R), binningBuffer, imageBuffer, alphas,
            bool(debug), perspec_matrix, bool(track_off), bool(map_off), bool(need_gaussian_grads))
        return tuple(g)

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        return _CompiledC.ext.mark_visible(means3D, viewmatrix, projmatrix)


# DGR_AUTOGRAD=python keeps the Python autograd.Function over the compiled `_C` (the A/B for profiles/host_breakdown.py)
_USE_NODE = os.environ.get("DGR_AUTOGRAD", "compiled") != "python"
_CtypesC = _C
# (DGR_HIP_LIB selects another build of the C ABI for the ctypes loader; the extension is linked against the in-tree one)
if os.environ.get("DGR_BINDING", "compiled") != "ctypes" and not os.environ.get("DGR_HIP_LIB"):
    try:
        from . import _dgr_torch_ext as _ext
        _CompiledC.ext = _ext
        _C = _CompiledC
    except ImportError:  # the extension is optional (make -C diff-gaussian-rasterization_amd builds it); ctypes still binds the C ABI
        pass


def _binning_policy(key, P):
    """(mode, capacity) of the next forward of shape `key` = (device, P, H, W); see csrc/torch_ext.cpp: light_forward_core."""
    cap = _capacity_cache.get(key, 0)
    if os.environ.get("DGR_FORWARD_MODE", "presized") == "callback" or P == 0:
        return 0, 0, cap
    if _sync_mode() == "lazy" and cap > 0:
        while len(_pending_status) > _LAZY_DEPTH and not torch.cuda.is_current_stream_capturing():
            _check_oldest()  # status words of earlier calls have long completed: no stall
        if key not in _unsettled or torch.cuda.is_current_stream_capturing():
            return 2, int(cap * 1.5) + 4096, cap
        # an unsettled shape (above): strict forwards, which also keep the count history going
        return 1, int(cap * 1.5) + 4096, cap
    return 1, (int(cap * 1.25) + 4096 if cap else 4 * P + 4096), cap


def _rasterize_compiled(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                        gt_depth, rs):
    """`_RasterizeGaussians.apply` through the autograd node compiled into the extension (csrc/torch_ext.cpp: LightNode):
    one Python -> C++ crossing per forward, the backward runs inside the autograd engine without the interpreter.  Same
    inputs, outputs, saved state and gradients as the Python Function below, which stays for the debug path and the ctypes
    binding."""
    P, H, W = (means3D.size(0) if means3D.dim() == 2 else 0), rs.image_height, rs.image_width
    key = (means3D.device.index, P, H, W)
    mode, use, cap = _binning_policy(key, P)
    out, rendered, ticket, _, status = _CompiledC.ext.light_apply(
        means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix, gt_depth, rs.bg,
        rs.projmatrix, rs.campos, rs.perspec_matrix, rs.scale_modifier, rs.tanfovx, rs.tanfovy, H, W, rs.sh_degree,
        rs.prefiltered, rs.track_off, rs.map_off, use, mode)
    if mode == 2:
        if ticket >= 0:
            _pending_status.append((ticket, key))
        else:  # recorded into a hipGraph: nothing can be read back now
            _captured_status.append(weakref.ref(status))
            _capture_keepalive.append(status)
    elif mode == 1:
        _strict_read(key, cap, rendered)
    return tuple(out)


def rasterize_gaussians(
    means3D,
    means2D,
    sh,
    colors_precomp,
    opacities,
    scales,
    rotations,
    cov3Ds_precomp,
    viewmatrix,
    gt_depth,
    raster_settings,
):
    if _C is _CompiledC and not raster_settings.debug and _USE_NODE:
        return _rasterize_compiled(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                   viewmatrix, gt_depth, raster_settings)
    return _RasterizeGaussians.apply(
        means3D,
        means2D,
        sh,
        colors_precomp,
        opacities,
        scales,
        rotations,
        cov3Ds_precomp,
        viewmatrix,
        gt_depth,
        raster_settings,
    )


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                viewmatrix, gt_depth, raster_settings):
        # argument packing of L/diff_gaussian_rasterization/__init__.py:66-87
        args = (
            raster_settings.bg,
            means3D,
            colors_precomp,
            opacities,
            scales,
            rotations,
            raster_settings.scale_modifier,
            cov3Ds_precomp,
            viewmatrix,
            gt_depth,
            raster_settings.projmatrix,
            raster_settings.tanfovx,
            raster_settings.tanfovy,
            raster_settings.image_height,
            raster_settings.image_width,
            sh,
            raster_settings.sh_degree,
            raster_settings.campos,
            raster_settings.prefiltered,
            raster_settings.debug,
        )
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                out = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            out = _C.rasterize_gaussians(*args)
        (num_rendered, color, depth, depth_median, depth_var, opacity_map, radii, geomBuffer, binningBuffer,
         imgBuffer, gau_uncertainty, gau_related_pixels) = out

        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.dgr_options = _capi.load().dgr_thread_options_effective()  # the backward runs under the forward's options
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrix, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer, opacity_map, gt_depth)
        # Four of the eight outputs (radii, opacity_map, gau_uncertainty, gau_related_pixels) have no gradient input in
        # the C++ backward; autograd would still zero-fill a gradient tensor for each of them on every backward.
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, gau_related_pixels)
        return color, radii, depth, depth_median, depth_var, opacity_map, gau_uncertainty, gau_related_pixels

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_depth_median, grad_depth_var, grad_alpha,
                 grad_gau_uncertainty, grad_gau_realted_pixels):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrix, radii, sh, geomBuffer,
         binningBuffer, imgBuffer, opacity_map, gt_depth) = ctx.saved_tensors
        # an output that did not take part in the loss arrives as None (gradients are not materialised): zeros, as the
        # reference's autograd would have passed
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        zeros = lambda c: torch.zeros((c, H, W), dtype=torch.float32, device=means3D.device)  # noqa: E731
        grad_color = zeros(3) if grad_color is None else grad_color
        grad_depth = zeros(1) if grad_depth is None else grad_depth
        grad_depth_median = zeros(1) if grad_depth_median is None else grad_depth_median
        grad_depth_var = zeros(1) if grad_depth_var is None else grad_depth_var

        # argument packing of L/diff_gaussian_rasterization/__init__.py:116-146
        args = (raster_settings.bg,
                means3D,
                radii,
                colors_precomp,
                scales,
                rotations,
                raster_settings.scale_modifier,
                cov3Ds_precomp,
                viewmatrix,
                raster_settings.projmatrix,
                raster_settings.tanfovx,
                raster_settings.tanfovy,
                grad_color,
                grad_depth,
                grad_depth_median,
                grad_depth_var,
                gt_depth,
                sh,
                raster_settings.sh_degree,
                raster_settings.campos,
                geomBuffer,
                num_rendered,
                binningBuffer,
                imgBuffer,
                opacity_map,
                raster_settings.debug,
                raster_settings.perspec_matrix,
                raster_settings.track_off,
                raster_settings.map_off)

        with _capi.under_options(ctx.dgr_options):  # (the autograd engine may run this on a thread of its own)
            if raster_settings.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    out = _C.rasterize_gaussians_backward(*args)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp): tracking needs none
                out = _C.rasterize_gaussians_backward(*args, need_gaussian_grads=any(ctx.needs_input_grad[:8]))
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_viewmatrix) = out
        # reference: torch.sum(grad_viewmatrix, dim=0) over a [H*W,4,4] buffer (__init__.py:160-161);
        # here the buffer is [1,4,4], already reduced: a view instead of a reduction kernel.
        grad_viewmatrix = grad_viewmatrix.view(4, 4)
        _consume_post_backward_wait()

        grads = (
            grad_means3D,
            grad_means2D,
            grad_sh,
            grad_colors_precomp,
            grad_opacities,
            grad_scales,
            grad_rotations,
            grad_cov3Ds_precomp,
            grad_viewmatrix,
            None,
            None,
        )
        return grads


_EMPTY = torch.Tensor([])  # stands for "None" at the C++ boundary (L/__init__.py:223-232)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    perspec_matrix: torch.Tensor
    track_off: bool
    map_off: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(
                positions,
                raster_settings.viewmatrix,
                raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, viewmatrix=None, gt_depth=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        # (the reference builds a fresh `torch.Tensor([])` per missing input and call; one shared empty tensor says the same)
        if shs is None:
            shs = _EMPTY
        if colors_precomp is None:
            colors_precomp = _EMPTY
        if scales is None:
            scales = _EMPTY
        if rotations is None:
            rotations = _EMPTY
        if cov3D_precomp is None:
            cov3D_precomp = _EMPTY

        return rasterize_gaussians(
            means3D,
            means2D,
            shs,
            colors_precomp,
            opacities,
            scales,
            rotations,
            cov3D_precomp,
            viewmatrix,
            gt_depth,
            raster_settings,
        )
