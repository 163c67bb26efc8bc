"""Round 7: the launches that left the one-view path, each against the path it replaces.

  * status word through pinned host memory written by the binning kernel (dgr_status_arm) instead of a copy behind an event;
  * tile schedule by policy: a frame with even tile lists skips tile_schedule_kernel and its blend kernels walk the static
    XCD band map -- same results, bit for bit where the path is bit-exact;
  * resident backward scratch (dgr_backward_scratch_clean_arm): no clearing launch, the per-Gaussian kernel clears what it reads.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from dgr_amd import _capi
from dgr_amd.synth import cluster_scene
from util import make_scene
import hip_helpers as hh

pytestmark = pytest.mark.gpu
T, E = hh.T, hh.E


def presized_forward(s, deg, cap, arm=False, P=None, reuse=None):
    """dgr_light_forward_presized through ctypes with every buffer allocated here (or those of an earlier call: `reuse`); returns
    (ticket, dict of tensors)."""
    lib = _capi.load()
    P = s.P if P is None else P
    dev = hh.dev()
    if reuse is not None:
        b, inp, p = reuse, reuse["inputs"], _capi.ptr
        ticket = lib.dgr_status_arm() if arm else -1
        b["rc"] = lib.dgr_light_forward_presized(
            _capi.stream_handle(), p(b["geom"]), p(b["binning"]), cap, p(b["img"]), p(b["status"]), P, deg, 16, p(inp[0]), s.W, s.H,
            p(inp[1]), p(inp[2]), None, p(inp[3]), p(inp[4]), 1.0, p(inp[5]), None, p(inp[6]), p(inp[7]), p(inp[8]), s.tanfovx,
            s.tanfovy, 0, p(b["color"]), p(b["depth"]), p(b["median"]), p(b["alpha"]), p(inp[9]), p(b["var"]), p(b["unc"]),
            p(b["px"]), p(b["radii"]))
        return ticket, b
    u8 = dict(dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    b = dict(geom=torch.empty((max(lib.dgr_geometry_bytes(P), 1),), **u8), img=torch.empty((lib.dgr_image_bytes(s.W, s.H),), **u8),
             binning=torch.empty((lib.dgr_binning_bytes(cap, s.W, s.H),), **u8), status=torch.full((4,), -7, **i32),
             color=torch.empty((3, s.H, s.W), **f32), depth=torch.empty((1, s.H, s.W), **f32),
             median=torch.empty((1, s.H, s.W), **f32), var=torch.empty((1, s.H, s.W), **f32), alpha=torch.empty((1, s.H, s.W), **f32),
             radii=torch.empty((max(P, 1),), **i32), unc=torch.empty((max(P, 1), 1), **f32), px=torch.empty((max(P, 1), 1), **i32))
    inp = [T(a) for a in (s.bg, s.means[:P], s.shs[:P], s.opac[:P], s.scales[:P], s.rots[:P], s.view, s.proj, s.campos, s.gt)]
    b["inputs"] = inp
    p = _capi.ptr
    ticket = lib.dgr_status_arm() if arm else -1
    rc = lib.dgr_light_forward_presized(
        _capi.stream_handle(), p(b["geom"]), p(b["binning"]), cap, p(b["img"]), p(b["status"]), P, deg, 16, p(inp[0]), s.W, s.H,
        p(inp[1]), p(inp[2]), None, p(inp[3]), p(inp[4]), 1.0, p(inp[5]), None, p(inp[6]), p(inp[7]), p(inp[8]), s.tanfovx,
        s.tanfovy, 0, p(b["color"]), p(b["depth"]), p(b["median"]), p(b["alpha"]), p(inp[9]), p(b["var"]), p(b["unc"]),
        p(b["px"]), p(b["radii"]))
    b["rc"] = rc
    return ticket, b


def poll(ticket, wait=1):
    buf = (C.c_int * 4)()
    rc = _capi.load().dgr_status_poll(ticket, wait, buf)
    return rc, list(buf)


@pytest.mark.parametrize("case", [(3000, 96, 64, 3, 1), (100000, 640, 480, 3, 0), (20000, 4000, 2300, 1, 2)])
def test_armed_status_word_is_the_device_word(case):
    """(the last shape is too large for the segment tables: the global-counter path's scan_tiles reports)"""
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    ticket, b = presized_forward(s, deg, 8 * P + 4096, arm=True)
    assert b["rc"] >= 0 and ticket >= 0
    rc, word = poll(ticket)
    torch.cuda.synchronize()
    dev_word = b["status"].tolist()
    assert rc == 1 and word[:3] == dev_word[:3] and word[0] > 0 and word[1] == 0
    assert _capi.load().dgr_status_poll(ticket, 0, (C.c_int * 4)()) < 0  # the ticket was released by the read


def test_armed_status_reports_an_overflow_and_the_slot_is_reusable():
    s = make_scene(20000, 320, 200, 5)
    t1, b = presized_forward(s, 3, 64, arm=True)  # far too small
    rc, word = poll(t1)
    torch.cuda.synchronize()
    assert rc == 1 and word[1] == 1 and word[0] == b["status"].tolist()[0] > 64
    t2, b2 = presized_forward(s, 3, word[0] + 10, arm=True)
    rc, word2 = poll(t2)
    assert rc == 1 and word2[1] == 0 and word2[0] == word[0]
    # many in flight before any is read: distinct slots, each with its own forward's word
    tickets = [presized_forward(make_scene(1000 + 500 * i, 64, 48, i), 1, 40000, arm=True) for i in range(6)]
    torch.cuda.synchronize()
    for t, bb in tickets:
        rc, w = poll(t, wait=0)
        assert rc == 1 and w[:3] == bb["status"].tolist()[:3]


def test_armed_slot_is_completed_by_the_library_when_no_kernel_runs():
    s = make_scene(500, 64, 48, 1)
    t0, b = presized_forward(s, 3, 100, arm=True, P=0)   # nothing to render: no binning kernel
    assert b["rc"] == 0 and poll(t0, wait=0) == (1, [0, 0, 0, 0])
    lib = _capi.load()
    t1 = lib.dgr_status_arm()
    rc = lib.dgr_light_forward_presized(_capi.stream_handle(), None, None, 10, None, None, 5, 0, 0, None, 64, 48, *([None] * 5), 1.0,
                                        *([None] * 5), 0.5, 0.5, 0, *([None] * 9))
    assert rc == _capi.DGR_ERR_BAD_ARGUMENT and poll(t1, wait=0) == (1, [0, 0, 0, 0])
    t2 = lib.dgr_status_arm()
    assert lib.dgr_status_poll(t2, 0, (C.c_int * 4)()) < 0   # armed, not yet taken by a forward: not pollable
    t3, b3 = presized_forward(s, 3, 4000, arm=False)          # (this forward takes t2's arm)
    assert poll(t2)[0] == 1


def test_a_blocking_poll_ends_when_the_word_can_never_arrive():
    """The armed word comes from one workgroup of one kernel.  A forward issued into a stream that is recording a hipGraph enqueues
    nothing that runs: the wait used to spin for ever.  It now asks the stream -- idle, no word -- and returns DGR_ERR_HIP; the
    ticket is released; the next armed forward works."""
    import time
    lib = _capi.load()
    s = make_scene(3000, 96, 64, 1)
    _, b0 = presized_forward(s, 3, 40000)  # (kernels loaded; the buffers and inputs of the captured call)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            ticket, b = presized_forward(s, 3, 40000, arm=True, reuse=b0)
    assert b["rc"] >= 0 and ticket >= 0
    t0 = time.time()
    rc, _ = poll(ticket, wait=1)
    assert rc == _capi.DGR_ERR_HIP and time.time() - t0 < 5.0, (rc, time.time() - t0)
    assert "never arrived" in _capi.last_error()
    assert lib.dgr_status_poll(ticket, 0, (C.c_int * 4)()) < 0  # released
    t2, b2 = presized_forward(s, 3, 40000, arm=True)
    rc, word = poll(t2)
    assert rc == 1 and word[0] > 0


def test_strict_mode_refuses_a_capturing_stream(monkeypatch):
    """One host wait per forward cannot be recorded into a hipGraph: the compiled binding says so instead of arming a status slot
    whose word would never come."""
    from dgr_amd import light
    from dgr_amd.multiview import make_settings
    if light._C is not light._CompiledC:
        pytest.skip("compiled binding not in use")
    monkeypatch.setenv("DGR_SYNC_MODE", "strict")
    s = make_scene(3000, 96, 64, 1)
    dev = hh.dev()
    rast = light.GaussianRasterizer(make_settings(s, 3, dev))
    args = dict(means3D=T(s.means), means2D=torch.zeros((s.P, 3), device=dev), opacities=T(s.opac), shs=T(s.shs), scales=T(s.scales),
                rotations=T(s.rots), viewmatrix=T(s.view), gt_depth=T(s.gt))
    rast(**args)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="captured"):
            with torch.cuda.graph(g, stream=side):
                rast(**args)
    torch.cuda.synchronize()
    rast(**args)  # and the rasterizer is usable afterwards


@pytest.mark.parametrize("variant", ["light", "full"])
def test_callback_entry_points_refuse_a_capturing_stream(monkeypatch, variant):
    """The resize-callback forwards (the reference's own interface: DGR_FORWARD_MODE=callback) block the host to size the binning
    buffer.  On a capturing stream that synchronisation fails AND invalidates the capture -- before round 9 every later call of
    the process on that stream failed with it; now the entry point refuses before it touches the stream, and everything is usable
    afterwards."""
    from dgr_amd import full, light
    from dgr_amd.multiview import make_settings
    monkeypatch.setenv("DGR_FORWARD_MODE", "callback")
    s = make_scene(3000, 96, 64, 1)
    dev = hh.dev()
    if variant == "light":
        rast = light.GaussianRasterizer(make_settings(s, 3, dev))
    else:
        rast = full.GaussianRasterizer(full.GaussianRasterizationSettings(
            image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=T(s.bg), scale_modifier=1.0, viewmatrix=T(s.view),
            projmatrix=T(s.proj), sh_degree=3, campos=T(s.campos), prefiltered=False, perspec_matrix=T(s.persp)))
    args = dict(means3D=T(s.means), means2D=torch.zeros((s.P, 3), device=dev), opacities=T(s.opac), shs=T(s.shs), scales=T(s.scales),
                rotations=T(s.rots), viewmatrix=T(s.view), gt_depth=T(s.gt))
    ref = rast(**args)[0].clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="captured"):
            with torch.cuda.graph(g, stream=side):
                rast(**args)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):                       # the stream, and the process, are not left in a failed capture
        again = rast(**args)[0]
    torch.cuda.synchronize()
    assert torch.equal(again, ref)


def forward_under(option, s, deg):
    _capi.set_option("tile_schedule", option)
    try:
        return hh.hip_forward(s, deg)
    finally:
        _capi.set_option("tile_schedule", 2)


@pytest.mark.parametrize("case", [(3000, 96, 64, 3, 7, False), (100000, 640, 480, 3, 0, False), (100000, 640, 480, 3, 0, True),
                                  (500000, 1920, 1080, 3, 0, False)])
def test_static_band_map_and_tile_schedule_give_the_same_frame(oracle, case):
    """tile_schedule = 0 (blend workgroups find their tile through the static XCD band map) against 1 (through the schedule):
    images, n_contrib, contribution tags and the median statistics identical; the backward of either state within the float
    atomics' noise of the other and at the 1e-5 bar against the oracle."""
    from test_hip_light_parity import check_backward
    P, W, H, deg, seed, clustered = case
    s = make_scene(P, W, H, seed)
    if clustered:
        s = cluster_scene(s)
    out1, d1 = forward_under(1, s, deg)
    out0, d0 = forward_under(0, s, deg)
    assert hh.hip_state("sched_flag", s, d1)[0] & 1 == 1 and hh.hip_state("sched_flag", s, d0)[0] & 1 == 0   # (bit 0 of the frame's blend flags)
    for k in ("color", "depth", "depth_median", "opacity_map", "radii", "gau_related_pixels"):
        assert np.array_equal(d0[k], d1[k]), k
    for name in ("n_contrib", "point_list", "contribution_tags", "ranges"):
        assert np.array_equal(hh.hip_state(name, s, d0), hh.hip_state(name, s, d1)), name
    g0, g1 = hh.hip_backward(s, deg, out0), hh.hip_backward(s, deg, out1)
    for k in g0:
        scale = max(np.abs(g1[k]).max(), 1e-30)
        assert np.abs(g0[k] - g1[k]).max() <= 2e-5 * scale, k
    if P <= 100000:
        _capi.set_option("tile_schedule", 0)
        try:
            check_backward(oracle, s, deg)
        finally:
            _capi.set_option("tile_schedule", 2)


def test_schedule_policy_follows_the_frame():
    """tile_schedule = 2: a forward that reports through an armed slot tells the library its longest list, and the next forward
    of that shape drops the schedule on an even frame and keeps it on a clustered one; forwards that report nothing keep it."""
    s = make_scene(60000, 480, 320, 3)
    c = cluster_scene(s)
    cap = 12 * s.P
    assert _capi.get_option("tile_schedule") == 2

    def flag(b):
        return hh.hip_state("sched_flag", s, {"num_rendered": 0, "geom": b["geom"], "binning": b["binning"], "img": b["img"]}, capacity=cap)[0] & 1

    for scene, want in ((s, 0), (c, 1), (s, 0)):
        t, b = presized_forward(scene, 3, cap, arm=True)
        assert poll(t)[0] == 1                 # the report of THIS frame decides the NEXT forward of the shape
        t, b = presized_forward(scene, 3, cap, arm=True)
        assert poll(t)[0] == 1
        assert flag(b) == want, (want,)
    _, b = presized_forward(s, 3, cap, arm=False)
    assert flag(b) == 1                         # no report asked for: the schedule stays


def test_full_variant_walks_the_static_map_too(oracle):
    s = make_scene(30000, 320, 240, 2)
    _capi.set_option("tile_schedule", 0)
    try:
        out0, d0 = hh.hip_full_forward(s, 3)
        g0 = hh.hip_full_backward(s, 3, out0)
    finally:
        _capi.set_option("tile_schedule", 2)
    _capi.set_option("tile_schedule", 1)
    try:
        out1, d1 = hh.hip_full_forward(s, 3)
        g1 = hh.hip_full_backward(s, 3, out1)
    finally:
        _capi.set_option("tile_schedule", 2)
    assert hh.hip_state("sched_flag", s, d0)[0] & 1 == 0 and hh.hip_state("sched_flag", s, d1)[0] & 1 == 1
    for k in ("color", "depth", "uncertainty", "radii"):
        assert np.array_equal(d0[k], d1[k]), k
    assert d0["num_related"] == d1["num_related"]
    for k in g0:
        assert np.abs(g0[k] - g1[k]).max() <= 2e-5 * max(np.abs(g1[k]).max(), 1e-30), k


@pytest.mark.parametrize("mode", [(False, False), (True, False), (False, True)])
def test_resident_scratch_backward_equals_the_cleared_one_and_leaves_the_scratch_zero(mode):
    lib = _capi.load()
    s = make_scene(40000, 400, 300, 9)
    out, d = hh.hip_forward(s, 3)
    ref = hh.hip_backward(s, 3, out, track_off=mode[0], map_off=mode[1])
    (R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
    dev = hh.dev()
    P = s.P
    n = lib.dgr_light_backward_scratch_bytes(P, s.W, s.H)
    scratch = torch.zeros((n,), dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    p = _capi.ptr
    inp = [T(a) for a in (s.bg, s.means, s.shs, s.scales, s.rots, s.view, s.proj, s.campos, s.gt, s.persp, s.gC, s.gD[None], s.gM[None], s.gV[None])]
    for rep in range(3):  # the second and third call run on what the call before left behind
        g = dict(m2=torch.empty((P, 3), **f32), op=torch.empty((P, 1), **f32), col=torch.empty((P, 3), **f32), m3=torch.empty((P, 3), **f32),
                 cov=torch.empty((P, 6), **f32), sh=torch.empty((P, 16, 3), **f32), sc=torch.empty((P, 3), **f32), rot=torch.empty((P, 4), **f32),
                 view=torch.empty((16,), **f32))
        assert lib.dgr_backward_scratch_clean_arm() == 0
        rc = lib.dgr_light_backward(
            _capi.stream_handle(), P, 3, 16, int(R), p(inp[0]), s.W, s.H, p(inp[1]), p(inp[2]), None, p(alpha), p(inp[3]), 1.0, p(inp[4]),
            None, p(inp[5]), p(inp[6]), p(inp[7]), s.tanfovx, s.tanfovy, p(radii), p(geom), p(binning), p(img), p(inp[10]), p(inp[11]),
            p(inp[12]), p(inp[13]), p(g["m2"]), None, p(g["op"]), p(g["col"]), None, p(g["m3"]), p(g["cov"]), p(g["sh"]), p(g["sc"]),
            p(g["rot"]), 0, None, p(inp[9]), p(g["view"]), None, p(inp[8]), int(mode[0]), int(mode[1]), p(scratch), n)
        assert rc == 0, _capi.last_error()
        torch.cuda.synchronize()
        assert int(scratch.count_nonzero()) == 0, f"call {rep}: the scratch is not all zero again"
        got = {"dL_dmeans2D": g["m2"], "dL_dopacity": g["op"], "dL_dmeans3D": g["m3"], "dL_dcov3D": g["cov"], "dL_dsh": g["sh"],
               "dL_dscales": g["sc"], "dL_drotations": g["rot"], "dL_dview": g["view"].view(4, 4)}
        for k, v in got.items():
            a, b_ = v.cpu().numpy(), ref[k]
            assert np.abs(a - b_).max() <= 2e-5 * max(np.abs(b_).max(), 1e-30), (rep, k)


def test_compiled_node_keeps_its_scratch_and_agrees_with_the_python_function():
    """The autograd node of the compiled extension (resident scratch, armed status, arena outputs) against the Python
    autograd.Function over `_C` -- the reference-shaped path -- on the same inputs: identical images, gradients within the
    float atomics' noise; repeated, so that the resident scratch is reused."""
    from dgr_amd import light as L
    from dgr_amd.multiview import make_settings
    if L._C is not L._CompiledC:
        pytest.skip("ctypes binding selected")
    s = make_scene(30000, 320, 240, 4)
    dev = hh.dev()
    rast = L.GaussianRasterizer(make_settings(s, 3, dev))

    def run(use_node):
        leaves = [T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
        m2 = torch.zeros((s.P, 3), device=dev, requires_grad=True)
        old = L._USE_NODE
        L._USE_NODE = use_node
        try:
            o = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                     viewmatrix=leaves[5], gt_depth=T(s.gt))
        finally:
            L._USE_NODE = old
        torch.autograd.backward([o[0], o[2], o[3], o[4]], [T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None])])
        torch.cuda.synchronize()
        return [x.detach().cpu().numpy() for x in o], [x.grad.cpu().numpy() for x in leaves + [m2]]

    o_ref, g_ref = run(False)
    for rep in range(3):
        o, g = run(True)
        for i, (a, b) in enumerate(zip(o, o_ref)):
            if i == 6:  # gau_uncertainty: a sum of float atomics, equal up to their arrival order
                assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
            else:
                assert np.array_equal(a, b), i
        for a, b in zip(g, g_ref):
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30), rep
    L.check_async_errors()


def test_inputs_made_on_the_callers_stream_and_dropped_after_the_call_stay_valid(monkeypatch):
    """Views in flight run on side streams; PyTorch hands a freed block out again on its allocation stream without
    waiting for other streams' readers unless the reader was recorded on it.  The compiled forward (and the backward, for
    the gradient images of a graph root) records its inputs when it is issued on a side stream (csrc/torch_ext.cpp:
    keep_until_read): a per-view viewmatrix / gt_depth / gradient image made on the caller's stream and dropped right
    after the call must not be handed out again while the view's kernels are still to run.  The side stream is kept
    busy (a 30 ms spin) so that they certainly are; the next allocations of those sizes must then come from other
    blocks (with DGR_RECORD_INPUT_STREAMS=0 they are the same blocks: that is the hazard), and the view's results are
    those of a run that kept its inputs."""
    from dgr_amd import light as L
    from dgr_amd.multiview import ViewStreams, make_settings
    if L._C is not L._CompiledC:
        pytest.skip("ctypes binding selected: the documented rule (keep the inputs alive until join()) applies there")
    s = make_scene(20000, 256, 192, 3)
    dev = hh.dev()
    rast = L.GaussianRasterizer(make_settings(s, 3, dev))
    leaves = [T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots)]
    m2 = torch.zeros((s.P, 3), device=dev, requires_grad=True)

    def one(view, gt, grads, rast=rast):
        for p_ in leaves + [m2, view]:
            p_.grad = None
        o = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                 viewmatrix=view, gt_depth=gt)
        torch.autograd.backward([o[0], o[2], o[3], o[4]], grads)
        return [x.detach() for x in (o[0], o[2], o[3], o[5], view.grad, leaves[0].grad, leaves[1].grad)]  # (the graph dies here)

    def fresh():
        return (T(s.view).requires_grad_(), T(s.gt), [T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None])])

    ref = [x.clone() for x in one(*fresh())]      # (strict: this call also learns the binning capacity of the shape)
    torch.cuda.synchronize()
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")   # no host wait inside the forward: the calls return with everything still queued
    views = ViewStreams(2)
    kept, handed_out_again = [], []
    for rep in range(4):
        view, gt, grads = fresh()                     # on the caller's stream
        settings = make_settings(s, 3, dev)           # ... and the camera's tensors (bg, projmatrix, campos, perspec_matrix)
        cam = [settings.bg, settings.projmatrix, settings.campos, settings.perspec_matrix]  # (settings.viewmatrix: markVisible only)
        torch.cuda.current_stream().synchronize()
        with views.next():
            torch.cuda._sleep(60_000_000)              # ~30 ms: everything issued below is still to run when the inputs go
            kept.append(one(view, gt, grads, L.GaussianRasterizer(settings)))
        shapes = [t.shape for t in [view, gt] + grads + cam]
        blocks = {t.data_ptr(): n for t, n in zip([view, gt] + grads + cam, "view gt gC gD gM gV bg proj campos perspec".split())}
        del view, gt, grads, settings, cam
        again = [torch.full(shape, float("nan"), device=dev) for shape in shapes]   # what the caller's stream allocates next
        handed_out_again += [blocks[t.data_ptr()] for t in again if t.data_ptr() in blocks]
        del again
    views.join()
    torch.cuda.synchronize()
    L.check_async_errors()
    assert not handed_out_again, f"input blocks reused while a side stream still had to read them: {handed_out_again}"
    for rep, got in enumerate(kept):
        for i, (a, b) in enumerate(zip(got, ref)):
            a, b = a.cpu().numpy(), b.cpu().numpy()
            assert np.isfinite(a).all(), (rep, i)
            if i < 4:
                assert np.array_equal(a, b), (rep, i)
            else:
                assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-30), (rep, i)


@pytest.mark.parametrize("variant", ["light", "full"])
@pytest.mark.parametrize("cap", [7, 5])
def test_capped_blend_kernels_give_the_same_view(variant, cap):
    """dgr_set_option("blend_wgs_per_cu", n): the blend kernels claim enough dynamic LDS that only n of their workgroups fit a
    CU (room for the other streams' kernels and RCCL's; csrc/api.hip: blend_pad_bytes).  A scheduling choice: images and lists
    bit-identical, gradients equal up to the order of the float atomics."""
    s = make_scene(30000, 320, 240, 6)
    fwd, bwd = (hh.hip_forward, hh.hip_backward) if variant == "light" else (hh.hip_full_forward, hh.hip_full_backward)
    out0, d0 = fwd(s, 3)
    g0 = bwd(s, 3, out0)
    keep = _capi.get_option("blend_wgs_per_cu")
    _capi.set_option("blend_wgs_per_cu", cap)
    try:
        assert _capi.get_option("blend_wgs_per_cu") == cap
        out1, d1 = fwd(s, 3)
        g1 = bwd(s, 3, out1)
    finally:
        _capi.set_option("blend_wgs_per_cu", keep)
    for k, v in d0.items():
        if isinstance(v, np.ndarray) and k not in ("geom", "binning", "img", "gau_uncertainty"):
            assert np.array_equal(v, d1[k]), k
    for k in g0:
        assert np.abs(g1[k] - g0[k]).max() <= 2e-5 * max(np.abs(g0[k]).max(), 1e-30), k


@pytest.mark.parametrize("variant", ["light", "full"])
def test_a_transposed_perspec_matrix_is_read_in_place(variant):
    """perspec_matrix usually reaches the rasterizer as `projection.transpose(0, 1)`: a non-contiguous tensor.  The kernels
    read its entries 0 and 5 only (L/cuda_rasterizer/backward.cu:725-739) -- the diagonal, which keeps its place -- so the
    compiled binding passes such a tensor as it is instead of launching a copy per backward; same pose gradient, bit for bit."""
    from dgr_amd import light as L, full as F
    if L._C is not L._CompiledC:
        pytest.skip("ctypes binding selected")
    s = make_scene(8000, 160, 120, 2)
    dev = hh.dev()
    persp_c = T(s.persp)                                  # contiguous
    persp_t = persp_c.t().contiguous().t()                # same values, column-major strides
    assert not persp_t.is_contiguous() and torch.equal(persp_c, persp_t)
    got = []
    for persp in (persp_c, persp_t):
        leaves = [T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
        m2 = torch.zeros((s.P, 3), device=dev, requires_grad=True)
        if variant == "light":
            from dgr_amd.multiview import make_settings
            rast = L.GaussianRasterizer(make_settings(s, 3, dev)._replace(perspec_matrix=persp))
        else:
            rast = F.GaussianRasterizer(F.GaussianRasterizationSettings(
                image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=T(s.bg), scale_modifier=1.0,
                viewmatrix=T(s.view), projmatrix=T(s.proj), sh_degree=3, campos=T(s.campos), prefiltered=False, perspec_matrix=persp))
        o = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4],
                 viewmatrix=leaves[5], gt_depth=T(s.gt))
        if variant == "light":
            torch.autograd.backward([o[0], o[2], o[3], o[4]], [T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None])])
        else:
            torch.autograd.backward([o[0], o[2], o[3]], [T(s.gC), T(s.gD[None]), T(s.gV[None])])
        got.append(leaves[5].grad.cpu().numpy())
    assert np.abs(got[0]).max() > 0
    assert np.allclose(got[0], got[1], rtol=2e-5, atol=1e-7 * np.abs(got[0]).max())  # (double atomics into 64 buckets: order only)
