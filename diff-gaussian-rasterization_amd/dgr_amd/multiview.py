"""One-view-per-GPU sharding helpers: views are independent, Gaussians are replicated,
and the only exchange step of a mapping iteration is a sum of the per-Gaussian gradients over GPUs.

`GradientArena.all_reduce` performs that sum with ONE collective: the backward writes all per-Gaussian
gradients into one flat arena (dgr_amd.light._C.rasterize_gaussians_backward) and autograd hands the
arena's views to `.grad` without copying, so the contiguous span holding
[dL_dmeans3D | dL_dmeans2D | dL_dsh | dL_dopacity | dL_dscales | dL_drot] (236 B + 12 B per Gaussian at
SH degree 3) is reduced in place.  Pose gradients (`viewmatrix.grad`) are per view and never reduced.
In tracking mode (map_off) there are no Gaussian gradients and no collective at all.
"""
import torch

from . import _capi, light


def make_settings(s, sh_degree, device, track_off=False, map_off=False, debug=False, prefiltered=False,
                  scale_modifier=1.0):
    """GaussianRasterizationSettings (light field order) for a dgr_amd.synth.Scene."""
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)  # noqa: E731
    return light.GaussianRasterizationSettings(
        image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=t(s.bg),
        scale_modifier=scale_modifier, viewmatrix=t(s.view), projmatrix=t(s.proj), sh_degree=sh_degree,
        campos=t(s.campos), prefiltered=prefiltered, debug=debug, perspec_matrix=t(s.persp),
        track_off=track_off, map_off=map_off)


class _SharedCov3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scales, rotations, scale_modifier):
        from . import _capi
        scales, rotations = scales.contiguous().float(), rotations.contiguous().float()
        cov = torch.empty((scales.shape[0], 6), dtype=torch.float32, device=scales.device)
        with _capi.on_device(scales.device):
            rc = _capi.load().dgr_cov3d_forward(_capi.stream_handle(scales.device.index), scales.shape[0], _capi.ptr(scales),
                                                _capi.ptr(rotations), float(scale_modifier), _capi.ptr(cov))
        if rc:
            raise RuntimeError(_capi.last_error())
        ctx.save_for_backward(scales, rotations)
        ctx.mod = float(scale_modifier)
        return cov

    @staticmethod
    def backward(ctx, dL_dcov):
        from . import _capi
        scales, rotations = ctx.saved_tensors
        dL_dcov = dL_dcov.contiguous().float()
        ds, dr = torch.empty_like(scales), torch.empty_like(rotations)
        with _capi.on_device(scales.device):
            rc = _capi.load().dgr_cov3d_backward(_capi.stream_handle(scales.device.index), scales.shape[0], _capi.ptr(scales),
                                                 _capi.ptr(rotations), ctx.mod, _capi.ptr(dL_dcov), _capi.ptr(ds), _capi.ptr(dr))
        if rc:
            raise RuntimeError(_capi.last_error())
        return ds, dr, None


def shared_cov3D(scales, rotations, scale_modifier=1.0):
    """The view-independent part of a keyframe batch, computed once: the 3D covariances of all Gaussians,
    bit-identical to what every view's forward would compute from `scales` / `rotations`.  Pass the result as
    `cov3D_precomp=` (and no scales / rotations) to the rasterizer of every view of the batch: the views then skip
    computeCov3D and its backward, autograd sums their `dL_dcov3D` ([P,6]) and ONE conversion -- the covariance backward is
    linear in dL_dcov3D -- produces `scales.grad` and `rotations.grad`.  Same images; gradients equal up to summation order."""
    return _SharedCov3D.apply(scales, rotations, scale_modifier)


class ViewStreams:
    """Independent views in flight on several HIP streams.

    About a quarter of a view's GPU time is spent in kernels that leave most of the chip idle (the atomic-bound
    instance ranking, single-workgroup scans, the latency-bound per-tile sort) while the blend kernels are bound by
    VALU issue.  Views of a batch do not depend on each other, so issuing them round-robin on a few streams lets the
    two kinds of kernels overlap (MI355X, config 3: 0.64 -> 0.52 ms per view with three streams).

        views = ViewStreams(3)
        for cam in batch:
            with views.next():                 # this view's forward AND backward run on one side stream
                out = rasterizer(...)
                loss(out).backward()           # autograd runs the backward on the forward's stream
        views.join()                           # the caller's stream waits for every view

    Every view keeps its own state buffers and gradient arena (they are allocated per call), so nothing is shared
    between views in flight except the read-only inputs -- and the leaves' `.grad`.  PyTorch runs a leaf's
    accumulation (`.grad += ...`) on the stream of the backward that reaches it, and nothing orders two views on two
    streams that accumulate into the same `.grad`.  Two safe patterns:
      * no accumulation across views: set `p.grad = None` before every view and collect the per-view gradients
        yourself (`GroupedReduce` sums the views' arenas on one stream) -- what bench.py does;
      * accumulation in `.grad`: call `views.before_backward()` between a view's forward and its `backward()`.  The
        view's stream then waits for the END of the previous view at the one point where it matters: right after the
        rasterizer's own backward kernels have been issued, before autograd goes on to the activations' backward and
        the accumulation into the leaves (`dgr_amd.light._set_post_backward_wait`, keyed by the view's stream).  The rasterizer kernels of consecutive
        views keep overlapping; only the short tail that touches shared `.grad` is ordered -- what
        `dgr_amd.slam.render_batch` does.  (If the loss reaches other shared leaves before the rasterizer node, order
        the whole backward instead: `views.before_backward(whole=True)`.)"""

    class _View:
        """The context manager next() hands out: makes one of the streams current for the block.  One object per stream,
        reused (a view costs two `set_stream` calls on the host, no Stream / Event / context objects)."""

        def __init__(self, owner, stream):
            self.owner, self.stream = owner, stream
            self.prev_stream = None   # the stream of the view issued before this one (before_backward orders after its end)
            self.outer = None         # the caller's stream, restored on exit
            self.outer_device = None  # ... and the caller's current device (set_stream switches to the stream's device)

        def __enter__(self):
            if self.outer is not None:
                raise RuntimeError("ViewStreams: a view's block is already open on this stream (the per-stream context is not re-entrant)")
            self.owner._current = self
            self.outer_device = torch.cuda.current_device()
            self.outer = torch.cuda.current_stream(self.owner.device)
            torch.cuda.set_stream(self.stream)
            return self.stream

        def __exit__(self, *exc):
            self.owner._last_stream = self.stream
            self.owner._current = None
            light._drop_post_backward_wait(self.stream)  # (a backward that never reached the rasterizer leaves nothing behind)
            torch.cuda.set_stream(self.outer)
            if self.outer_device != torch.cuda.current_device():  # ViewStreams(device=another GPU): give the caller its device back
                torch.cuda.set_device(self.outer_device)
            self.outer = None
            return False

    def __init__(self, n=3, device=None, count_with_atomics=None):
        if count_with_atomics is not None:  # (rounds 2-4 switched the binning path by the number of views in flight)
            import warnings
            warnings.warn("ViewStreams(count_with_atomics=...) is ignored: the segment binning serves every case",
                          DeprecationWarning, stacklevel=2)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(n)))]
        self._views = [ViewStreams._View(self, st) for st in self.streams]
        # Lazy status mode: while views are in flight the host may run as many forward passes ahead of the status words as
        # there are streams (light.py: _LAZY_DEPTH) -- raised by the first next(), put back by join(), so that code outside
        # the in-flight window sees a binning overflow one or two calls late as documented, not n + 1.
        self._saved_depth = None
        self._i = 0
        self._fresh = set()       # streams already ordered after the caller's stream since the last join()
        self._last_stream = None  # stream of the most recent view (nothing else is enqueued on it until its next turn)
        self._current = None

    def next(self):
        """Context manager: the next stream.  The first use of a stream after construction or join() is ordered after
        everything the caller's stream has issued (input preparation, an optimiser step); later uses are not, so that
        views keep overlapping -- call join() before changing the inputs."""
        if self._saved_depth is None:
            self._saved_depth = light.set_lazy_depth(max(light.lazy_depth(), len(self.streams)))
        k = self._i % len(self.streams)
        st = self.streams[k]
        self._i += 1
        if k not in self._fresh:
            st.wait_stream(torch.cuda.current_stream(self.device))
            self._fresh.add(k)
        v = self._views[k]
        v.prev_stream = self._last_stream
        return v

    def before_backward(self, whole=False):
        """Inside a `with views.next():` block, before `backward()`: orders the part of this view's backward that
        accumulates into shared `.grad` after the end of the previous view (see the class docstring)."""
        v = self._current
        if v is None or v.prev_stream is None or v.prev_stream is v.stream:
            return
        # the end of the previous view: its stream has had nothing enqueued since (streams take turns), so an event recorded
        # on it now marks that point -- and views that never call this pay for no event at all
        prev_done = torch.cuda.Event()
        prev_done.record(v.prev_stream)
        if whole:
            v.stream.wait_event(prev_done)
        else:
            light._set_post_backward_wait(v.stream, prev_done)

    def join(self):
        """The caller's stream waits for every view issued so far."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)
        self._fresh.clear()
        self._last_stream = None
        if self._saved_depth is not None:
            light.set_lazy_depth(self._saved_depth)
            self._saved_depth = None


class CapturedStep:
    """hipGraph capture of a fixed-shape step -- a forward + backward, a whole tracking iteration including the pose
    optimiser (use `capturable=True` optimisers), ... -- so that replaying it costs one launch on the host.

    The rasterizer's lazy forward and its backward perform no host synchronisation, launch kernels only (no memset
    nodes: those were seen to re-execute with corrupted parameters from a captured graph on this ROCm) and allocate
    only through torch's caching allocator, so `torch.cuda.graph` records them as they are.  Worth it when the step is
    host-bound: config 2 (100 k Gaussians, 640x480) goes from 0.28-0.37 ms per view eager to 0.20 ms replayed; a
    1080p / 500 k view is GPU-bound and gains nothing (and views on several streams inside ONE graph do not overlap
    on this ROCm -- use ViewStreams eagerly for that).

        step = CapturedStep(fn)      # runs fn() a few times eagerly first (sizes the binning buffer), then records it
        out = step.replay()          # whatever fn returned at capture time: the same tensors, rewritten by every replay
        step.check()                 # raises if a replayed forward overflowed its binning buffer

    Several steps captured on their own streams (`stream=`) and replayed round-robin DO overlap -- graphs on different
    streams, unlike branches of one graph: config 2 reaches 0.10 ms per view (light) / 0.11 ms (full) with three of them
    (`profiles/graph_experiment.py`), against 0.19-0.21 ms for one graph and 0.25-0.37 ms eager.

    `fn` must read its inputs from fixed tensors (update them in place between replays) and should return every tensor
    the caller wants to read afterwards (e.g. the leaves' `.grad`): after an eager step `.grad` no longer aliases the
    captured buffers."""

    def __init__(self, fn, warmup=3, device=None, stream=None):
        import os
        if os.environ.get("DGR_SYNC_MODE", "strict") != "lazy":
            raise RuntimeError("CapturedStep needs DGR_SYNC_MODE=lazy (a blocking status read cannot be captured)")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        light.check_async_errors()
        self.graph = torch.cuda.CUDAGraph()
        self.stream = stream  # replay on this stream (None: the caller's current stream)
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream(dev))
        del light._capture_keepalive[:]
        with torch.cuda.graph(self.graph, stream=stream):
            self.result = fn()
        # the status words of the captured forwards live exactly as long as this object (light.check_captured_status
        # holds weak references only)
        self.keep = list(light._capture_keepalive)
        del light._capture_keepalive[:]

    def replay(self):
        if self.stream is None:
            self.graph.replay()
        else:
            with torch.cuda.stream(self.stream):
                self.graph.replay()
        return self.result

    @staticmethod
    def check():
        light.check_captured_status()


class GradientArena:
    """Sums the gradients of the Gaussian parameters over all ranks after a backward."""

    def __init__(self, params):
        self.params = [p for p in params if p is not None]

    def fused_span(self):
        """The flat span covering every parameter's .grad when they are all contiguous pieces of ONE storage (the arena
        of the backward that produced them), or None if autograd copied or accumulated instead of aliasing.  Found
        through the gradients themselves: no module-level "last arena", nothing to go stale across threads."""
        store, lo, hi = None, None, None
        for p in self.params:
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype is not torch.float32:
                return None
            st = g.untyped_storage()
            if store is None:
                store = st
            elif st.data_ptr() != store.data_ptr():
                return None
            a = g.storage_offset()
            lo = a if lo is None else min(lo, a)
            hi = a + g.numel() if hi is None else max(hi, a + g.numel())
        if store is None:
            return None
        g = self.params[0].grad
        return torch.empty((0,), dtype=g.dtype, device=g.device).set_(store, lo, (hi - lo,))

    def grad_offsets(self, span):
        """[(parameter, first float of its gradient inside `span`)] -- how a reduced span maps back to the parameters."""
        base = span.data_ptr()
        return [(p, (p.grad.data_ptr() - base) // 4) for p in self.params]

    def all_reduce(self, dist, group=None, async_op=False):
        """Sums the gradients over the ranks.  Returns the number of collectives issued, or with `async_op` a
        `PendingReduce` whose wait() orders the current stream after them: the collective runs on the backend's own
        stream, so the next view's forward/backward (which writes a fresh arena) overlaps it."""
        span = self.fused_span()
        tensors = [span] if span is not None else [p.grad for p in self.params if p.grad is not None]
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op) for t in tensors]
        if async_op:
            return PendingReduce(works, tensors)
        return len(tensors)


class PendingReduce:
    """Handles of an in-flight gradient sum; keeps the reduced buffers alive until wait()."""

    def __init__(self, works, tensors):
        self.works, self.tensors = works, tensors

    def __len__(self):
        return len(self.works)

    def wait(self):
        for w in self.works:
            w.wait()
        return self.tensors


class GroupedReduce:
    """One all-reduce per GROUP of local views instead of one per view (fewer, larger collectives: the global batch is
    `group_size` views per GPU).  Protocol per view: set the parameters' `.grad` to None, run the view's backward, call
    `add_view()` on that view's stream.  When the group is full its gradient arenas are summed into the first one (on
    the current stream, after the other views' streams), that sum is all-reduced, and every parameter's `.grad` is
    pointed at its slice of the reduced buffer, so an optimiser step after `flush()` sees the batch gradient.
    `flush()` reduces an incomplete group.

    A view whose gradients are not one fused arena (autograd accumulated into an existing `.grad`, or copied) cannot
    join a group: summing or reducing an accumulating `.grad` again would count earlier views twice.  `add_view()`
    raises in that case instead of guessing."""

    def __init__(self, arena, dist, group_size, group=None):
        self.arena, self.dist, self.n, self.group = arena, dist, max(1, int(group_size)), group
        self.pending = []   # [(span, event recorded on the producing stream)]
        self.layout = None  # [(parameter, offset)] of the group's first view
        self.collectives = 0
        self.last_total = None

    def add_view(self):
        span = self.arena.fused_span()
        if span is None:
            raise RuntimeError(
                "GroupedReduce.add_view: the parameters' .grad are not views of one gradient arena (autograd accumulated "
                "into an existing .grad or copied): set p.grad = None before every view's backward")
        layout = self.arena.grad_offsets(span)
        if not self.pending:
            self.layout = layout
        elif [o for _, o in layout] != [o for _, o in self.layout]:
            raise RuntimeError("GroupedReduce.add_view: views of one group must share the arena layout (same P and SH degree)")
        ev = torch.cuda.Event() if span.is_cuda else None
        if ev is not None:
            ev.record()
        self.pending.append((span, ev))
        if len(self.pending) >= self.n:
            self.flush()

    def flush(self):
        if not self.pending:
            return None
        total = self.pending[0][0]
        if total.is_cuda:
            cur = torch.cuda.current_stream(total.device)
            for span, ev in self.pending:
                cur.wait_event(ev)
                span.record_stream(cur)
        for span, _ in self.pending[1:]:
            total.add_(span)
        self.dist.all_reduce(total, op=self.dist.ReduceOp.SUM, group=self.group)
        self.collectives += 1
        for p, off in self.layout:  # the batch gradient, where an optimiser looks for it
            p.grad = total[off:off + p.numel()].view(p.shape)
        self.pending = []
        self.last_total = total
        return total
