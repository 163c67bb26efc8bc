"""Edge cases of the reference's API on the HIP path: optional inputs, empty input, the prefiltered trap, markVisible,
oversize tiles (global-memory sort path, many staging batches), binning-buffer overflow retry, lazy status checks."""
import numpy as np
import pytest
import torch

from util import assert_grad_close, assert_image_close, make_scene, mask_flipped_pixels
import hip_helpers as hh
from test_hip_light_parity import assert_images_carry_the_references_bits

pytestmark = pytest.mark.gpu


def test_precomputed_colors_and_covariances(oracle):
    """colors_precomp skips SH (dL_dcolors is then the returned gradient); cov3D_precomp skips scale/rotation
    (L/cuda_rasterizer/rasterizer_impl.cu:327,429,469)."""
    s = make_scene(4000, 96, 80, 3)
    st0, ref0 = hh.oracle_forward(oracle, s, 3)
    colors = st0.get("rgb").reshape(-1, 3).copy()
    cov3D = st0.get("cov3D").reshape(-1, 6).copy()
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    for kw in (dict(colors_precomp=colors), dict(cov3D_precomp=cov3D), dict(colors_precomp=colors, cov3D_precomp=cov3D)):
        out, d = hh.hip_forward(s, 3, **kw)
        st, ref = hh.oracle_forward(oracle, s, 3, **kw)
        assert np.array_equal(d["radii"], ref["radii"])
        assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
        assert_images_carry_the_references_bits(d, st, ref, s)
        g = hh.hip_backward(s, 3, out, grads=grads, **kw)  # end to end
        gr = hh.oracle_backward(oracle, st, s, 3, ref["opacity_map"], grads=grads, **kw)
        names = ["dL_dmeans3D", "dL_dopacity", "dL_dview"]
        names += ["dL_dcolors"] if "colors_precomp" in kw else ["dL_dsh"]
        names += ["dL_dcov3D"] if "cov3D_precomp" in kw else ["dL_dscales", "dL_drotations"]
        for k in names:
            assert_grad_close(g[k], gr[k], k, rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-3)
        if "cov3D_precomp" in kw:
            assert not g["dL_dscales"].any() and not g["dL_drotations"].any()
        if "colors_precomp" in kw:
            assert g["dL_dsh"].size == 0


def test_empty_input_returns_zeros():
    """P == 0: nothing runs, outputs are the zero fills (L/rasterize_points.cu:88,188) -- not the background."""
    from dgr_amd import light as L
    dev = hh.dev()
    s = make_scene(10, 48, 32, 0)
    E = lambda *shape: torch.empty(shape, device=dev)  # noqa: E731
    out = L._C.rasterize_gaussians(hh.T(s.bg), E(0, 3), E(0), E(0, 1), E(0, 3), E(0, 4), 1.0, E(0), hh.T(s.view), hh.T(s.gt),
                                   hh.T(s.proj), s.tanfovx, s.tanfovy, s.H, s.W, E(0, 16, 3), 3, hh.T(s.campos), False, False)
    assert out[0] == 0
    for t in out[1:6]:
        assert float(t.abs().sum()) == 0.0
    g = L._C.rasterize_gaussians_backward(hh.T(s.bg), E(0, 3), out[6], E(0), E(0, 3), E(0, 4), 1.0, E(0), hh.T(s.view),
                                          hh.T(s.proj), s.tanfovx, s.tanfovy, hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gM[None]),
                                          hh.T(s.gV[None]), hh.T(s.gt), E(0, 16, 3), 3, hh.T(s.campos), out[7], 0, out[8],
                                          out[9], out[5], False, hh.T(s.persp), False, False)
    assert g[8].shape == (1, 4, 4) and float(g[8].abs().sum()) == 0.0 and g[3].shape == (0, 3)


def test_bad_means_shape_raises_like_the_reference():
    from dgr_amd import light as L
    s = make_scene(10, 48, 32, 0)
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        L._C.rasterize_gaussians(hh.T(s.bg), hh.T(s.means.reshape(-1)), hh.E(), hh.T(s.opac), hh.T(s.scales), hh.T(s.rots),
                                 1.0, hh.E(), hh.T(s.view), hh.T(s.gt), hh.T(s.proj), s.tanfovx, s.tanfovy, s.H, s.W,
                                 hh.T(s.shs), 3, hh.T(s.campos), False, False)


def test_prefiltered_violation_is_reported():
    """A culled point with prefiltered=True traps in the reference (cuda_rasterizer/auxiliary.h:154-161)."""
    s = make_scene(500, 48, 32, 0)
    means = s.means.copy()
    means[7] = (-np.linalg.inv(s.view.T)[:3, :3] @ np.array([0, 0, 5.0]) + s.campos).astype(np.float32)  # behind the camera
    s2 = s._replace(means=means)
    with pytest.raises(RuntimeError, match="prefiltered"):
        hh.hip_forward(s2, 3, prefiltered=True)
    hh.hip_forward(s2, 3, prefiltered=False)  # same scene without the flag renders


def test_mark_visible(oracle):
    from dgr_amd import light as L
    s = make_scene(5000, 64, 64, 4)
    means = s.means.copy()
    means[::7] *= -1.0  # push some behind the camera
    got = L._C.mark_visible(hh.T(means), hh.T(s.view), hh.T(s.proj)).cpu().numpy()
    want = oracle.mark_visible(means, s.view, s.proj)
    assert got.dtype == bool and np.array_equal(got, want) and 0 < want.sum() < len(want)


@pytest.mark.parametrize("P", [9000, 24000])
def test_oversize_tiles_and_many_batches(oracle, P):
    """~2300 / ~6000 instances per tile: the per-tile sort leaves LDS (n > 2048) and the blend kernels stage many batches."""
    s = make_scene(P, 32, 32, 9)
    out, d = hh.hip_forward(s, 1)
    st, ref = hh.oracle_forward(oracle, s, 1)
    rg = st.get("ranges").reshape(-1, 2)
    assert (rg[:, 1] - rg[:, 0]).max() > 2048
    assert np.array_equal(hh.hip_state("ranges", s, d), st.get("ranges"))
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert_images_carry_the_references_bits(d, st, ref, s)
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    grads, _ = mask_flipped_pixels(grads, hh.hip_state("n_contrib", s, d), st.get("n_contrib"), s.W, s.H, "oversize tiles",
                                   median_margin=oracle.light_median_margin(st, ref["opacity_map"]))
    gr = hh.oracle_backward(oracle, st, s, 1, ref["opacity_map"], grads=grads)
    for alphas in (ref["opacity_map"], None):  # stage-isolated, end to end
        g = hh.hip_backward(s, 1, out, grads=grads, alphas=alphas)
        for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dview"):
            assert_grad_close(g[k], gr[k], k, rel_to_max=1e-5, elem_rtol=2e-3, elem_frac=2e-3)


def test_frame_too_wide_for_the_segment_tables(oracle):
    """250 x 144 tiles = 9 072 four-tile segments: above what bin_segments' LDS tables hold (8 192), so this frame NEEDS the
    global-counter binning (count_rank / scan_tiles / emit_instances / sort_tiles) rather than being sent there by an option.
    Lists, images and gradients against the oracle like every other frame."""
    s = make_scene(20000, 4000, 2300, 2)
    out, d = hh.hip_forward(s, 1)
    st, ref = hh.oracle_forward(oracle, s, 1)
    assert d["num_rendered"] == st.num_rendered > s.P
    assert np.array_equal(hh.hip_state("ranges", s, d), st.get("ranges"))
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert_images_carry_the_references_bits(d, st, ref, s)
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    gr = hh.oracle_backward(oracle, st, s, 1, ref["opacity_map"], grads=grads)
    g = hh.hip_backward(s, 1, out, grads=grads)
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dview"):
        assert_grad_close(g[k], gr[k], k, rel_to_max=1e-5, elem_rtol=2e-3, elem_frac=2e-3)


def test_binning_overflow_is_retried(oracle):
    """A too-small learned capacity must not change the result (presized path re-runs with a larger buffer)."""
    from dgr_amd import light as L
    s = make_scene(6000, 128, 96, 6)
    _, d0 = hh.hip_forward(s, 2)
    L._capacity_cache[(hh.dev().index, s.P, s.H, s.W)] = 16  # pretend the last frame was nearly empty
    _, d1 = hh.hip_forward(s, 2)
    assert d0["num_rendered"] == d1["num_rendered"] > 4096 + 20
    for k in ("color", "depth", "opacity_map", "radii"):
        assert np.array_equal(d0[k], d1[k]), k


@pytest.mark.parametrize("variant", ["light", "full"])
@pytest.mark.parametrize("P,W,H,seed,sm,cap", [
    (6000, 128, 96, 6, 1.0, 4116),      # every 256-Gaussian block stages its ranks in LDS; most runs start past the capacity
    (9734, 64, 97, 1107, 2.5, 43032),   # the draw of tests/tools/soak_parity.py that faulted: R = 81 804 > capacity
    (9734, 64, 97, 1107, 2.5, 1000),    # capacity inside the first block
    (3000, 48, 48, 5, 6.0, 20000),      # blocks above the LDS stage (direct stores), capacity in the middle of a block
])
def test_overflow_writes_stay_inside_the_state_buffers(variant, P, W, H, seed, sm, cap):
    """A presized forward whose binning buffer is too small must report the overflow (status[0] > capacity, status[1] = 1)
    and must not write one byte outside the three state buffers: they sit between 1 MiB guard regions here.
    (Round 2: a wave-uniform saturating subtraction was compiled without the saturation and every block whose ranks
    started past the capacity wrote them behind the buffer -- csrc/count_rank.h.)"""
    from dgr_amd import _capi
    lib = _capi.load(); dev = hh.dev(); G = 1 << 20
    s = make_scene(P, W, H, seed)
    s = s._replace(opac=(s.opac * 0.12).astype(np.float32))

    def guarded(n):
        whole = torch.full((n + 2 * G,), 0xAB, dtype=torch.uint8, device=dev)
        return whole, whole[G:G + n]
    gb, geom = guarded(lib.dgr_geometry_bytes(P)); bb, binning = guarded(lib.dgr_binning_bytes(cap, W, H))
    ib, img = guarded(lib.dgr_image_bytes(W, H))
    f = lambda *sh: torch.empty(sh, device=dev)  # noqa: E731
    color, depth, median, var, alpha = f(3, H, W), f(1, H, W), f(1, H, W), f(1, H, W), f(1, H, W)
    # (light: gau_uncertainty, one float per Gaussian; full: the uncertainty IMAGE, one per pixel -- a [P, 1] buffer here let
    #  the full blend write H W floats into P of them whenever P < H W, over whatever the allocator had put behind it)
    unc = f(P, 1) if variant == "light" else f(1, H, W)
    radii = torch.empty(P, dtype=torch.int32, device=dev); px = torch.empty((P, 1), dtype=torch.int32, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    k = dict(bg=hh.T(s.bg), means=hh.T(s.means), opac=hh.T(s.opac), scales=hh.T(s.scales), rots=hh.T(s.rots), view=hh.T(s.view),
             proj=hh.T(s.proj), campos=hh.T(s.campos), gt=hh.T(s.gt), colors=torch.rand((P, 3), device=dev))
    p = _capi.ptr
    front = (_capi.stream_handle(), p(geom), p(binning), cap, p(img), p(status), P, 1, 0, p(k["bg"]), W, H, p(k["means"]), None,
             p(k["colors"]), p(k["opac"]), p(k["scales"]), sm, p(k["rots"]), None, p(k["view"]), p(k["proj"]), p(k["campos"]),
             s.tanfovx, s.tanfovy, 0)
    if variant == "light":
        rc = lib.dgr_light_forward_presized(*front, p(color), p(depth), p(median), p(alpha), p(k["gt"]), p(var), p(unc), p(px),
                                            p(radii))
    else:
        rc = lib.dgr_full_forward_presized(*front, p(color), p(depth), p(k["gt"]), p(unc), p(radii))
    torch.cuda.synchronize()
    assert rc == 0
    st = status.tolist()
    assert st[0] > cap and st[1] == 1, st
    for name, whole, n in (("geometry", gb, geom.numel()), ("binning", bb, binning.numel()), ("image", ib, img.numel())):
        lo, hi = whole[:G], whole[G + n:]
        assert int((lo != 0xAB).sum()) == 0, f"{name}: bytes written below the buffer"
        assert int((hi != 0xAB).sum()) == 0, f"{name}: bytes written above the buffer"
    # and the same call with room for everything renders (the caller's retry)
    cap2 = st[0]
    bb2, binning2 = guarded(lib.dgr_binning_bytes(cap2, W, H))
    front = front[:2] + (p(binning2), cap2) + front[4:]
    if variant == "light":
        rc = lib.dgr_light_forward_presized(*front, p(color), p(depth), p(median), p(alpha), p(k["gt"]), p(var), p(unc), p(px),
                                            p(radii))
    else:
        rc = lib.dgr_full_forward_presized(*front, p(color), p(depth), p(k["gt"]), p(unc), p(radii))
    torch.cuda.synchronize()
    st2 = status.tolist()
    assert rc == 0 and st2[0] == st[0] and st2[1] == 0, st2
    assert int((bb2[:G] != 0xAB).sum()) == 0 and int((bb2[G + binning2.numel():] != 0xAB).sum()) == 0
    assert float(depth.max()) > 0.0


def test_lazy_status_mode_matches_strict(monkeypatch):
    """DGR_SYNC_MODE=lazy: no host read in forward once the shape is known; results identical, errors raised late."""
    from dgr_amd import light as L
    s = make_scene(5000, 96, 64, 8)
    _, d_strict = hh.hip_forward(s, 3)          # strict call teaches the capacity
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")
    _, d_lazy = hh.hip_forward(s, 3)
    assert len(L._pending_status) >= 1
    L.check_async_errors()
    assert not L._pending_status
    assert d_lazy["num_rendered"] == d_strict["num_rendered"]
    for k in ("color", "depth", "depth_median", "opacity_map", "radii"):
        assert np.array_equal(d_lazy[k], d_strict[k]), k
    # an overflow in lazy mode surfaces at the next check
    L._capacity_cache[(hh.dev().index, s.P, s.H, s.W)] = 1
    hh.hip_forward(s, 3)
    with pytest.raises(RuntimeError, match="overflow"):
        L.check_async_errors()
    L._capacity_cache.pop((hh.dev().index, s.P, s.H, s.W), None)


def test_light_autograd_surface_and_gradient_arena():
    """The drop-in module: 8 outputs, gradient order, None for gt_depth/settings, gradients alias one flat arena."""
    import diff_gaussian_rasterization as D
    from dgr_amd import light as L
    from dgr_amd.multiview import GradientArena, make_settings
    s = make_scene(3000, 96, 64, 7)
    dev = hh.dev()
    assert D.GaussianRasterizationSettings._fields[11:] == ("debug", "perspec_matrix", "track_off", "map_off")
    rast = D.GaussianRasterizer(make_settings(s, 3, dev))
    means3D, shs, opac = hh.T(s.means).requires_grad_(), hh.T(s.shs).requires_grad_(), hh.T(s.opac).requires_grad_()
    scales, rots, view = hh.T(s.scales).requires_grad_(), hh.T(s.rots).requires_grad_(), hh.T(s.view).requires_grad_()
    means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
    gt = hh.T(s.gt).requires_grad_()
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=means3D, means2D=means2D, opacities=opac, scales=scales, rotations=rots, viewmatrix=view, gt_depth=gt)
    outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                viewmatrix=view, gt_depth=gt)
    assert len(outs) == 8
    color, radii, depth, median, var, alpha, unc, px = outs
    assert color.shape == (3, s.H, s.W) and median.shape == (1, s.H, s.W) and unc.shape == (s.P, 1)
    assert px.dtype == torch.int32 and radii.dtype == torch.int32 and float(var.detach().abs().sum()) == 0.0
    torch.autograd.backward([color, depth, median, var], [hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gM[None]), hh.T(s.gV[None])])
    assert gt.grad is None and view.grad.shape == (4, 4) and means2D.grad.shape == (s.P, 3)
    assert float(means2D.grad[:, 2].abs().sum()) == 0.0
    arena = GradientArena([means3D, means2D, shs, opac, scales, rots])
    span = arena.fused_span()
    assert span is not None and span.numel() >= s.P * (3 + 3 + 48 + 1 + 3 + 4)  # one contiguous all-reduce payload
    vis = rast.markVisible(means3D.detach())
    assert vis.dtype == torch.bool and vis.shape == (s.P,)


def test_loss_on_a_subset_of_outputs():
    """Outputs left out of the loss reach the backward as None (gradients are not materialised): the result must be
    what explicit zero gradients give."""
    from dgr_amd import light as D
    from dgr_amd.multiview import make_settings
    s = make_scene(3000, 96, 64, 11)
    dev = hh.dev()
    rast = D.GaussianRasterizer(make_settings(s, 3, dev))

    def run(explicit_zeros):
        means3D, shs, opac = hh.T(s.means).requires_grad_(), hh.T(s.shs).requires_grad_(), hh.T(s.opac).requires_grad_()
        scales, rots, view = hh.T(s.scales).requires_grad_(), hh.T(s.rots).requires_grad_(), hh.T(s.view).requires_grad_()
        means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
        color, radii, depth, median, var, alpha, unc, px = rast(
            means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view,
            gt_depth=hh.T(s.gt))
        if explicit_zeros:
            z = torch.zeros_like(depth)
            torch.autograd.backward([color, depth, median, var], [hh.T(s.gC), z, z, z])
        else:
            torch.autograd.backward([color], [hh.T(s.gC)])
        return [t.grad.cpu().numpy() for t in (means3D, shs, opac, scales, rots, view, means2D)]

    for a, b in zip(run(False), run(True)):
        assert_grad_close(a, b, "subset", rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-3)


def test_views_in_flight_match_serial_views():
    """dgr_amd.multiview.ViewStreams: views issued round-robin on three streams give the results of the same views
    rendered one after the other (per-view outputs and pose gradients; Gaussian gradients to atomic-order noise)."""
    from dgr_amd import light as D
    from dgr_amd.multiview import ViewStreams, make_settings
    dev = hh.dev()
    scenes = [make_scene(4000, 128, 96, 5, view_index=k) for k in range(4)]
    s0 = scenes[0]
    rasts = [D.GaussianRasterizer(make_settings(sc, 3, dev)) for sc in scenes]
    gC, gD, gM, gV = hh.T(s0.gC), hh.T(s0.gD[None]), hh.T(s0.gM[None]), hh.T(s0.gV[None])
    gt = hh.T(s0.gt)

    def render(k, res):
        means3D, shs, opac = hh.T(s0.means).requires_grad_(), hh.T(s0.shs).requires_grad_(), hh.T(s0.opac).requires_grad_()
        scales, rots = hh.T(s0.scales).requires_grad_(), hh.T(s0.rots).requires_grad_()
        view = hh.T(scenes[k].view).requires_grad_()
        means2D = torch.zeros((s0.P, 3), device=dev, requires_grad=True)
        return (means3D, shs, opac, scales, rots, view, means2D), res

    def run(streams):
        leaves = [render(k, None)[0] for k in range(4)]
        torch.cuda.synchronize()
        outs = []
        for rep in range(3):  # several rounds so that views really overlap
            for k in range(4):
                means3D, shs, opac, scales, rots, view, means2D = leaves[k]
                for t in leaves[k]:
                    t.grad = None
                ctx = streams.next() if streams is not None else torch.cuda.stream(torch.cuda.current_stream())
                with ctx:
                    color, radii, depth, median, var, alpha, unc, px = rasts[k](
                        means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                        viewmatrix=view, gt_depth=gt)
                    torch.autograd.backward([color, depth, median, var], [gC, gD, gM, gV])
                if rep == 2:
                    outs.append((color, depth, alpha))
        if streams is not None:
            streams.join()
        torch.cuda.synchronize()
        return [([o.detach().cpu().numpy() for o in outs[k]], [t.grad.cpu().numpy() for t in leaves[k]]) for k in range(4)]

    serial = run(None)
    piped = run(ViewStreams(3, dev))
    for (o_s, g_s), (o_p, g_p) in zip(serial, piped):
        for a, b in zip(o_s, o_p):
            assert np.array_equal(a, b)  # forward is deterministic
        for a, b in zip(g_s, g_p):
            assert_grad_close(b, a, "views in flight", rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-3)


def test_tight_culling_changes_the_lists_but_not_the_results():
    """dgr_set_option("tight_cull", 1): alpha-aware tile rectangles.  Every dropped instance is one no
    pixel blends, so images are bit-identical and gradients equal up to atomic-order noise; radii (the caller's
    visibility filter) are untouched; num_rendered shrinks."""
    from dgr_amd import _capi
    s = make_scene(20000, 320, 240, 9)
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    out0, d0 = hh.hip_forward(s, 3)
    g0 = hh.hip_backward(s, 3, out0, grads=grads)
    _capi.set_option("tight_cull", 1)
    try:
        assert _capi.get_option("tight_cull") == 1
        out1, d1 = hh.hip_forward(s, 3)
        g1 = hh.hip_backward(s, 3, out1, grads=grads)
    finally:
        _capi.set_option("tight_cull", 0)
    assert d1["num_rendered"] < 0.8 * d0["num_rendered"]
    assert np.array_equal(d0["radii"], d1["radii"])
    for k in ("color", "depth", "depth_median", "opacity_map", "gau_related_pixels"):
        assert np.array_equal(d0[k], d1[k]), k
    assert_grad_close(d1["gau_uncertainty"], d0["gau_uncertainty"], "gau_uncertainty", rel_to_max=1e-6)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dview"):
        assert_grad_close(g1[k], g0[k], k, rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)



@pytest.mark.parametrize("deg,seed", [(3, 9), (1, 21)])
def test_tight_culling_against_the_oracle(oracle, deg, seed):
    """The opt-in tile culling checked against the ORACLE (which has no such option), not against the default HIP path:
    images inside the usual bars, stage-isolated gradients at 1e-5 of scale, every tile list a subsequence (same
    order) of the reference's, and every instance the reference blends somewhere still present."""
    from dgr_amd import _capi
    s = make_scene(20000, 320, 240, seed)
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    st, ref = hh.oracle_forward(oracle, s, deg)
    _capi.set_option("tight_cull", 1)
    try:
        out, d = hh.hip_forward(s, deg)
        pl = hh.hip_state("point_list", s, d)
        rg = hh.hip_state("ranges", s, d).reshape(-1, 2)
        images = ("color", "depth", "depth_median", "opacity_map")
        # (tile lists differ from the reference's, the blended pairs and their order do not: the threshold-carrying images
        #  keep the reference's bits; n_contrib is a position in a different list)
        for k in ("depth_median", "opacity_map"):
            assert np.array_equal(d[k], ref[k]), k
        for k in ("color", "depth"):
            a_, b_ = d[k].astype(np.float64), ref[k].astype(np.float64)
            assert np.all(np.abs(a_ - b_) <= 1e-6 * np.maximum(1.0, np.abs(b_))), k
        same = np.zeros(s.W * s.H, np.uint32)  # n_contrib is a position in a different list here: mask by images only
        grads, _ = mask_flipped_pixels(grads, same, same, s.W, s.H, "tight cull", images=[(d[k], ref[k]) for k in images],
                                       median_margin=oracle.light_median_margin(st, ref["opacity_map"]))
        g = hh.hip_backward(s, deg, out, grads=grads, alphas=ref["opacity_map"])
    finally:
        _capi.set_option("tight_cull", 0)
    assert d["num_rendered"] < 0.8 * ref["num_rendered"]
    assert np.array_equal(d["radii"], ref["radii"])
    assert np.mean(d["gau_related_pixels"] != ref["gau_related_pixels"]) <= 1e-3
    # lists: per tile a subsequence of the reference's sorted list
    rpl, rrg = st.get("point_list"), st.get("ranges").reshape(-1, 2)
    for t in range(len(rrg)):
        mine, theirs = pl[rg[t, 0]:rg[t, 1]], rpl[rrg[t, 0]:rrg[t, 1]]
        pos = {int(v): i for i, v in enumerate(theirs)}
        idx = [pos.get(int(v), -1) for v in mine]
        assert all(i >= 0 for i in idx) and all(a < b for a, b in zip(idx, idx[1:])), f"tile {t}"
    gr = hh.oracle_backward(oracle, st, s, deg, ref["opacity_map"], grads=grads)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dview"):
        assert_grad_close(g[k], gr[k], k, rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-3, outlier_rows=0)


@pytest.mark.parametrize("variant", ["light", "full"])
def test_captured_step_replays_forward_and_backward(monkeypatch, variant):
    """dgr_amd.multiview.CapturedStep: one view recorded into a hipGraph; repeated replays reproduce the eager results,
    also after the inputs were changed in place."""
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")
    from dgr_amd import full as F, light as D
    from dgr_amd.multiview import CapturedStep, make_settings
    dev = hh.dev()
    s = make_scene(5000, 160, 120, 4)
    if variant == "light":
        rast = D.GaussianRasterizer(make_settings(s, 3, dev))
    else:
        rast = F.GaussianRasterizer(F.GaussianRasterizationSettings(
            image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=hh.T(s.bg), scale_modifier=1.0,
            viewmatrix=hh.T(s.view), projmatrix=hh.T(s.proj), sh_degree=3, campos=hh.T(s.campos), prefiltered=False,
            perspec_matrix=hh.T(s.persp)))
    means3D, shs, opac = hh.T(s.means).requires_grad_(), hh.T(s.shs).requires_grad_(), hh.T(s.opac).requires_grad_()
    scales, rots, view = hh.T(s.scales).requires_grad_(), hh.T(s.rots).requires_grad_(), hh.T(s.view).requires_grad_()
    means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
    gt, gC, gD = hh.T(s.gt), hh.T(s.gC), hh.T(s.gD[None])
    leaves = [means3D, shs, opac, scales, rots, view]

    def step():
        for t in leaves + [means2D]:
            t.grad = None
        outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                    viewmatrix=view, gt_depth=gt)
        torch.autograd.backward([outs[0], outs[2]], [gC, gD])
        return [outs[0].detach()] + [t.grad for t in leaves]  # (the captured step's outputs live in these tensors)

    def snapshot(tensors):
        torch.cuda.synchronize()
        return [t.cpu().numpy().copy() for t in tensors]

    cap = CapturedStep(step)
    for scale in (1.0, 0.7, 1.0):  # opacities changed in place between replays; every setting replayed twice
        with torch.no_grad():
            opac.copy_(hh.T(s.opac) * scale)
        want = snapshot(step())
        for _ in range(2):
            got = snapshot(cap.replay())
            cap.check()
            assert np.array_equal(got[0], want[0])
            for a, b in zip(got[1:], want[1:]):
                assert_grad_close(a, b, "replay", rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)
    D.check_async_errors()


@pytest.mark.parametrize("variant", ["light", "full"])
def test_results_do_not_depend_on_stale_memory(variant):
    """Every buffer the kernels read is written first: the same view rendered after filling the caching allocator's free
    blocks with NaN bit patterns, with 0xFF bytes and with zeros gives identical images and gradients up to atomic order."""
    from dgr_amd import full as F, light as D
    from dgr_amd.multiview import make_settings
    dev = hh.dev()
    s = make_scene(6000, 200, 136, 12)
    if variant == "light":
        rast = D.GaussianRasterizer(make_settings(s, 3, dev))
    else:
        rast = F.GaussianRasterizer(F.GaussianRasterizationSettings(
            image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=hh.T(s.bg), scale_modifier=1.0,
            viewmatrix=hh.T(s.view), projmatrix=hh.T(s.proj), sh_degree=3, campos=hh.T(s.campos), prefiltered=False,
            perspec_matrix=hh.T(s.persp)))
    gt, gC, gD = hh.T(s.gt), hh.T(s.gC), hh.T(s.gD[None])

    def poison(kind):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        blocks = []
        for nbytes in (64 << 20, 16 << 20, 4 << 20, 1 << 20, 256 << 10, 64 << 10, 4 << 10, 512):
            for _ in range(6):
                t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                if kind == "nan":
                    t.view(torch.float32).fill_(float("nan"))
                else:
                    t.fill_(0xFF if kind == "ff" else 0)
                blocks.append(t)
        torch.cuda.synchronize()
        del blocks  # back to the allocator's free lists, contents intact

    def run():
        leaves = [hh.T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
        means3D, shs, opac, scales, rots, view = leaves
        means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
        outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                    viewmatrix=view, gt_depth=gt)
        torch.autograd.backward([outs[0], outs[2]], [gC, gD])
        torch.cuda.synchronize()
        return [o.detach().cpu().numpy() for o in outs if o.dtype == torch.float32], [t.grad.cpu().numpy() for t in leaves]

    results = []
    for kind in ("zero", "nan", "ff"):
        poison(kind)
        results.append(run())
    for outs, grads in results[1:]:
        for a, b in zip(outs, results[0][0]):
            if a.ndim == 3:
                assert np.array_equal(a, b)  # images are deterministic
            else:
                assert_grad_close(a, b, "gau_uncertainty", rel_to_max=2e-6)  # float atomics: order only
        for a, b in zip(grads, results[0][1]):
            assert np.isfinite(a).all()
            assert_grad_close(a, b, "stale memory", rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)


def test_inputs_are_converted_like_the_reference_binding():
    """The reference's binding calls .contiguous() on every input (L/rasterize_points.cu:101-125) and CG-SLAM may hand it
    double tensors or views: float64, non-contiguous and fp32-contiguous inputs must render the same frame."""
    from dgr_amd import light as D
    from dgr_amd.multiview import make_settings
    dev = hh.dev()
    s = make_scene(3000, 96, 64, 13)
    rast = D.GaussianRasterizer(make_settings(s, 3, dev))
    gt = hh.T(s.gt)

    def run(conv):
        ts = [conv(hh.T(a)) for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
        for t in ts:
            t.requires_grad_()
        means3D, shs, opac, scales, rots, view = ts
        means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
        outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                    viewmatrix=view, gt_depth=gt)
        torch.autograd.backward([outs[0], outs[2]], [hh.T(s.gC), hh.T(s.gD[None])])
        torch.cuda.synchronize()
        return outs[0].detach().cpu().numpy(), [t.grad.float().cpu().numpy() for t in ts]

    def strided(t):  # same values, memory with a gap after every row / element
        if t.dim() == 1:
            return torch.stack([t, t], 1)[:, 0]
        big = torch.zeros(t.shape[:-1] + (t.shape[-1] + 1,), device=t.device, dtype=t.dtype)
        big[..., :-1] = t
        return big[..., :-1]

    base_c, base_g = run(lambda t: t.clone())
    for conv in (lambda t: t.double(), strided):
        c, g = run(conv)
        assert np.array_equal(c, base_c)
        for a, b in zip(g, base_g):
            assert a.shape == b.shape
            assert_grad_close(a, b, "converted input", rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)


def test_debug_mode_dumps_a_snapshot_on_failure(tmp_path, monkeypatch):
    """`debug=True` (L/diff_gaussian_rasterization/__init__.py:75-84): a failing forward leaves snapshot_fw.dump with the
    CPU copies of its arguments, and a passing one synchronises and leaves nothing."""
    from dgr_amd import light as D
    from dgr_amd.multiview import make_settings
    monkeypatch.chdir(tmp_path)
    dev = hh.dev()
    s = make_scene(500, 64, 48, 2)
    far = s.means.copy()
    far[:, 2] = -50.0  # behind the camera: culled
    bad = s._replace(means=np.concatenate([s.means[:-5], far[-5:]]).astype(np.float32))

    def call(scene, prefiltered):
        rast = D.GaussianRasterizer(make_settings(scene, 3, dev, debug=True, prefiltered=prefiltered))
        return rast(means3D=hh.T(scene.means), means2D=torch.zeros((scene.P, 3), device=dev), opacities=hh.T(scene.opac),
                    shs=hh.T(scene.shs), scales=hh.T(scene.scales), rotations=hh.T(scene.rots), viewmatrix=hh.T(scene.view),
                    gt_depth=hh.T(scene.gt))

    call(s, False)
    assert not (tmp_path / "snapshot_fw.dump").exists()
    with pytest.raises(RuntimeError, match="prefiltered"):
        call(bad, True)
    dump = torch.load(tmp_path / "snapshot_fw.dump", weights_only=False)
    assert isinstance(dump, (tuple, list)) and len(dump) == 20  # the 20 arguments of rasterize_gaussians
    assert all(not (isinstance(t, torch.Tensor) and t.is_cuda) for t in dump)


def test_captured_steps_on_their_own_streams(monkeypatch):
    """Three views, each recorded into its own hipGraph on its own stream and replayed round-robin: every replay gives
    that view's eager result (the graphs overlap on the GPU)."""
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")
    from dgr_amd import light as D
    from dgr_amd.multiview import CapturedStep, make_settings
    dev = hh.dev()
    scenes = [make_scene(5000, 160, 120, 4, view_index=k) for k in range(3)]
    s0 = scenes[0]
    gt, gC, gD = hh.T(s0.gt), hh.T(s0.gC), hh.T(s0.gD[None])
    shared = [hh.T(a) for a in (s0.means, s0.shs, s0.opac, s0.scales, s0.rots)]

    def make_step(k):
        rast = D.GaussianRasterizer(make_settings(scenes[k], 3, dev))
        leaves = [t.clone().requires_grad_() for t in shared] + [hh.T(scenes[k].view).requires_grad_()]
        means3D, shs, opac, scales, rots, view = leaves
        means2D = torch.zeros((s0.P, 3), device=dev, requires_grad=True)

        def step():
            for t in leaves + [means2D]:
                t.grad = None
            outs = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots,
                        viewmatrix=view, gt_depth=gt)
            torch.autograd.backward([outs[0], outs[2]], [gC, gD])
            return [outs[0].detach()] + [t.grad for t in leaves]
        return step

    steps = [make_step(k) for k in range(3)]
    want = []
    for st in steps:
        res = st()
        torch.cuda.synchronize()
        want.append([t.cpu().numpy().copy() for t in res])
    caps = [CapturedStep(st, stream=torch.cuda.Stream()) for st in steps]
    for rnd in range(4):
        for c in caps:
            c.replay()
    torch.cuda.synchronize()
    for c, w in zip(caps, want):
        got = [t.cpu().numpy() for t in c.result]
        assert np.array_equal(got[0], w[0])
        for a, b in zip(got[1:], w[1:]):
            assert_grad_close(a, b, "replay on a stream", rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)
    caps[0].check()


def test_compiled_and_ctypes_bindings_agree():
    """`_C` exists twice: the compiled torch extension (csrc/torch_ext.cpp, the counterpart of L/ext.cpp) and the ctypes
    class over the same C ABI.  Same 12-tuple / 9-tuple shapes, bit-identical forward, gradients equal up to atomic order."""
    from dgr_amd import light as L
    assert L._C is L._CompiledC, "the compiled extension was not built / not importable"
    s = make_scene(4000, 112, 80, 12)
    a = (hh.T(s.bg), hh.T(s.means), hh.E(), hh.T(s.opac), hh.T(s.scales), hh.T(s.rots), 1.0, hh.E(), hh.T(s.view), hh.T(s.gt),
         hh.T(s.proj), s.tanfovx, s.tanfovy, s.H, s.W, hh.T(s.shs), 3, hh.T(s.campos), False, False)
    outs = [C.rasterize_gaussians(*a) for C in (L._CompiledC, L._CtypesC)]
    assert len(outs[0]) == len(outs[1]) == 12 and outs[0][0] == outs[1][0] > 0
    for i in (1, 2, 3, 4, 5, 6, 11):
        assert torch.equal(outs[0][i], outs[1][i]), i
    assert torch.allclose(outs[0][10], outs[1][10], rtol=1e-5, atol=1e-7)  # gau_uncertainty: float atomics
    grads = []
    for C, o in zip((L._CompiledC, L._CtypesC), outs):
        (R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = o
        grads.append(C.rasterize_gaussians_backward(
            hh.T(s.bg), hh.T(s.means), radii, hh.E(), hh.T(s.scales), hh.T(s.rots), 1.0, hh.E(), hh.T(s.view), hh.T(s.proj),
            s.tanfovx, s.tanfovy, hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gM[None]), hh.T(s.gV[None]), hh.T(s.gt), hh.T(s.shs), 3,
            hh.T(s.campos), geom, R, binning, img, alpha, False, hh.T(s.persp), False, False))
    assert len(grads[0]) == len(grads[1]) == 9 and tuple(grads[0][8].shape) == (1, 4, 4)
    for ga, gb in zip(*grads):
        assert ga.shape == gb.shape
        assert_grad_close(ga.cpu().numpy(), gb.cpu().numpy(), "binding", rel_to_max=2e-6, elem_rtol=1e-3, elem_frac=1e-3)
    # tracking form: no per-Gaussian gradients
    (R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = outs[0]
    g = L._CompiledC.rasterize_gaussians_backward(
        hh.T(s.bg), hh.T(s.means), radii, hh.E(), hh.T(s.scales), hh.T(s.rots), 1.0, hh.E(), hh.T(s.view), hh.T(s.proj),
        s.tanfovx, s.tanfovy, hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gM[None]), hh.T(s.gV[None]), hh.T(s.gt), hh.T(s.shs), 3,
        hh.T(s.campos), geom, R, binning, img, alpha, False, hh.T(s.persp), False, True, need_gaussian_grads=False)
    assert all(x is None for x in g[:8]) and g[8].abs().sum() > 0
    present = L._CompiledC.mark_visible(hh.T(s.means), hh.T(s.view), hh.T(s.proj))
    assert torch.equal(present, L._CtypesC.mark_visible(hh.T(s.means), hh.T(s.view), hh.T(s.proj)))


def test_shared_cov3D_across_the_views_of_a_batch():
    """dgr_amd.multiview.shared_cov3D: the covariance of every Gaussian computed once for a batch of views
    and handed to each view as cov3D_precomp; autograd sums the views' dL_dcov3D and ONE conversion gives the scale /
    rotation gradients.  Against the per-view scale / rotation path: cov3D bit-identical, images bit-identical, every
    gradient equal to summation order."""
    from dgr_amd import light as L
    from dgr_amd.multiview import make_settings, shared_cov3D
    dev = hh.dev()
    scenes = [make_scene(6000, 128, 96, 33, view_index=k) for k in range(3)]
    s0 = scenes[0]

    def leaves():
        return [hh.T(a).requires_grad_() for a in (s0.means, s0.shs, s0.opac, s0.scales, s0.rots)]

    results = []
    for shared in (False, True):
        means3D, shs, opac, scales, rots = leaves()
        cov = shared_cov3D(scales, rots, 1.0) if shared else None
        imgs = []
        for s in scenes:
            rast = L.GaussianRasterizer(make_settings(s, 3, dev))
            view = hh.T(s.view).requires_grad_()
            kw = dict(cov3D_precomp=cov) if shared else dict(scales=scales, rotations=rots)
            color, radii, depth, median, var, alpha, unc, px = rast(
                means3D=means3D, means2D=torch.zeros((s.P, 3), device=dev, requires_grad=True), opacities=opac, shs=shs,
                viewmatrix=view, gt_depth=hh.T(s.gt), **kw)
            torch.autograd.backward([color, depth, median], [hh.T(s.gC) * 1e3, hh.T(s.gD[None]) * 1e3, hh.T(s.gM[None]) * 1e3],
                                    retain_graph=shared)
            imgs.append((color.detach(), depth.detach(), view.grad.clone()))
        results.append((imgs, [t.grad.clone() for t in (means3D, shs, opac, scales, rots)], cov))
    (img_a, g_a, _), (img_b, g_b, cov) = results
    # (that the shared covariance is what a view's forward computes shows below: the images are identical)
    for (ca, da, va), (cb, db, vb) in zip(img_a, img_b):
        assert torch.equal(ca, cb) and torch.equal(da, db)
        assert_grad_close(vb.cpu().numpy(), va.cpu().numpy(), "dL_dview", rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=0.1)
    for name, a, b in zip(("means3D", "sh", "opacity", "scales", "rotations"), g_a, g_b):
        assert a.abs().max() > 0
        assert_grad_close(b.cpu().numpy(), a.cpu().numpy(), name, rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-3)


@pytest.mark.parametrize("binding", ["compiled", "ctypes"])
def test_callback_entry_points_match_the_presized_path(monkeypatch, oracle, binding):
    """dgr_light_forward / dgr_full_forward -- the literal mirror of CudaRasterizer::Rasterizer::forward with its three
    allocation callbacks and the reference's blocking read of num_rendered (L/cr/rasterizer.h:40-70) -- against the
    presized entry points the bindings use by default (which count inside preprocess_fwd): same num_rendered, same
    lists, bit-identical images, and a backward that runs on the callback-allocated state."""
    from dgr_amd import _capi, light as L, full as F
    LC = L._CompiledC if binding == "compiled" else L._CtypesC
    FC = F._CompiledC if binding == "compiled" else F._CtypesC
    s = make_scene(5000, 144, 100, 21)
    a = (hh.T(s.bg), hh.T(s.means), hh.E(), hh.T(s.opac), hh.T(s.scales), hh.T(s.rots), 1.0, hh.E(), hh.T(s.view), hh.T(s.gt),
         hh.T(s.proj), s.tanfovx, s.tanfovy, s.H, s.W, hh.T(s.shs), 3, hh.T(s.campos), False)
    st, ref = hh.oracle_forward(oracle, s, 3)
    outs = {}
    for mode in ("presized", "callback"):
        monkeypatch.setenv("DGR_FORWARD_MODE", mode)
        outs[mode] = (LC.rasterize_gaussians(*a, False), FC.rasterize_gaussians(*a))
    for v in (0, 1):
        p, c = outs["presized"][v], outs["callback"][v]
        assert p[0] == c[0] == ref["num_rendered"]
        for x, y in zip(p[1:], c[1:]):
            if isinstance(x, torch.Tensor) and x.dtype != torch.uint8 and x.dtype == torch.float32 and x.dim() == 3:
                assert torch.equal(x, y)  # images
    lt = outs["callback"][0]
    names = ["num_rendered", "color", "depth", "depth_median", "depth_var", "opacity_map", "radii", "geom", "binning", "img"]
    d = dict(zip(names, lt))
    assert lt[8].numel() == _capi.load().dgr_binning_bytes(lt[0], s.W, s.H)  # sized from num_rendered, as the reference does
    assert np.array_equal(hh.hip_state("point_list", s, d, capacity=lt[0]), st.get("point_list"))
    assert np.array_equal(hh.hip_state("ranges", s, d, capacity=lt[0]), st.get("ranges"))
    g = LC.rasterize_gaussians_backward(
        hh.T(s.bg), hh.T(s.means), lt[6], hh.E(), hh.T(s.scales), hh.T(s.rots), 1.0, hh.E(), hh.T(s.view), hh.T(s.proj),
        s.tanfovx, s.tanfovy, hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gM[None]), hh.T(s.gV[None]), hh.T(s.gt), hh.T(s.shs), 3,
        hh.T(s.campos), lt[7], lt[0], lt[8], lt[9], hh.T(ref["opacity_map"]), False, hh.T(s.persp), False, False)
    gr = hh.oracle_backward(oracle, st, s, 3, ref["opacity_map"])
    for i, k in ((3, "dL_dmeans3D"), (5, "dL_dsh"), (6, "dL_dscales"), (7, "dL_drotations")):
        assert_grad_close(g[i].cpu().numpy(), gr[k], k, rel_to_max=2e-5, elem_rtol=2e-3, elem_frac=2e-3, outlier_rows=2)


@pytest.mark.parametrize("variant", ["light", "full"])
@pytest.mark.parametrize("P,W,H,sm", [(20000, 320, 240, 1.0), (3000, 97, 61, 4.0), (70000, 640, 480, 1.0), (1500, 16, 16, 6.0)])
def test_lds_count_and_global_atomic_count_agree(variant, P, W, H, sm):
    """The forward bins tile instances with the two-level segment binning (csrc/segment_binning.hip: pairs per row segment,
    tile lists built and sorted in LDS); dgr_set_option("lds_count", 0) selects round 2's returning global atomics (inside
    preprocess_fwd on the presized path; csrc/binning.hip), which also serve frames whose segment tables do not fit LDS.
    Both end in a sort of every tile's list on unique keys, so num_rendered, ranges, point_list, keys and every output must be
    identical bit for bit.  Shapes: 20 / 69 bin_segments workgroups, a ragged frame with large splats (one Gaussian covering
    many rows and segments), a single tile."""
    from dgr_amd import _capi
    s = make_scene(P, W, H, 5)
    res = {}
    for mode in (2, 0):  # (2: accepted as a synonym of the default 1)
        _capi.set_option("lds_count", mode)
        try:
            assert _capi.get_option("lds_count") == mode
            out, d = hh.hip_forward(s, 3, scale_modifier=sm) if variant == "light" else hh.hip_full_forward(s, 3)
            res[mode] = (d, {k: hh.hip_state(k, s, d) for k in ("point_list", "ranges", "keys")})
        finally:
            _capi.set_option("lds_count", 1)
    (d1, st1), (d0, st0) = res[2], res[0]
    assert d1["num_rendered"] == d0["num_rendered"] and d1["num_rendered"] > 0
    for k in st1:
        assert np.array_equal(st1[k], st0[k]), k
    for k in d1:
        if not isinstance(d1[k], np.ndarray):
            continue
        if k == "gau_uncertainty":  # a sum of float atomics: its order varies from run to run of the SAME path
            assert_grad_close(d1[k], d0[k], k, rel_to_max=1e-6)
        else:
            assert np.array_equal(d1[k], d0[k]), k


@pytest.mark.parametrize("P,W,H,sm", [(40000, 320, 240, 1.0), (70000, 320, 240, 1.0), (9000, 97, 61, 2.0)])
def test_tile_lists_with_equal_and_nearly_equal_depths(oracle, P, W, H, sm):
    """The per-tile sort orders a list by ONE 32-bit word per entry -- the depth's upper 22 bits and the entry's slot -- and
    settles entries that agree in those bits with a fix-up on the full (depth, id) keys (csrc/tile_sort.h).  Here a third of
    the Gaussians lie on one plane of constant camera depth (identical depth bits: the id decides), a third within 2^-17
    relative of another (equal upper bits, different full keys), the rest anywhere.  Lists of ~450 entries (one register pass),
    ~800 (four parts merged by rank) and ~2000-2800 (the workgroup's LDS sort), runs of equal upper bits hundreds long.
    point_list and ranges must be the oracle's bit for bit."""
    from dgr_amd.synth import camera
    s = make_scene(P, W, H, 9)
    rng = np.random.default_rng(3)
    _, _, Rm, t, *_ = camera(W, H, 0.05)
    cam = (s.means.astype(np.float64) @ Rm.T) + t          # camera-space positions of the scene's Gaussians
    third = P // 3
    for lo, hi, z in ((0, third, np.full(third, 4.0)), (third, 2 * third, 6.0 * (1.0 + rng.integers(0, 64, third) * 2.0 ** -23))):
        k = z / cam[lo:hi, 2]
        cam[lo:hi] *= k[:, None]                           # same pixel, new depth
    s = s._replace(means=((cam - t) @ Rm).astype(np.float32))
    _, d = hh.hip_forward(s, 3, scale_modifier=sm)
    st, ref = hh.oracle_forward(oracle, s, 3, scale_modifier=sm)
    depths = hh.hip_state("depths", s, d)
    vis = ref["radii"] > 0
    assert np.unique(depths[:third][vis[:third]]).size <= 16         # (one plane up to the rounding of the re-projection)
    assert d["num_rendered"] == ref["num_rendered"] and np.array_equal(d["radii"], ref["radii"])
    rg = hh.hip_state("ranges", s, d).reshape(-1, 2)
    assert (rg[:, 1] - rg[:, 0]).max() > 400
    assert np.array_equal(rg.reshape(-1), st.get("ranges"))
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    assert np.array_equal(hh.hip_state("keys", s, d), st.get("keys"))



def test_backward_rejects_missing_scale_rotation_and_state():
    """The forward keeps no 3D covariance: a backward without cov3D_precomp needs the forward's scales and rotations, and all
    three state buffers (a NULL used to be a GPU fault)."""
    from dgr_amd import _capi
    lib = _capi.load()
    s = make_scene(300, 32, 32, 2)
    out, _ = hh.hip_forward(s, 0)
    (R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
    dev = hh.dev()
    P = s.P
    p = _capi.ptr
    t = {k: hh.T(v) for k, v in dict(bg=s.bg, m=s.means, sh=s.shs, sc=s.scales, ro=s.rots, vw=s.view, pj=s.proj, cp=s.campos, gt=s.gt,
                                     ps=s.persp, gC=s.gC, gD=s.gD[None], gM=s.gM[None], gV=s.gV[None]).items()}
    scratch = torch.empty((lib.dgr_light_backward_scratch_bytes(P, s.W, s.H),), dtype=torch.uint8, device=dev)
    dview = torch.empty(16, device=dev)

    def call(scales, rots, geom_):
        return lib.dgr_light_backward(
            _capi.stream_handle(), P, 0, 16, int(R), p(t["bg"]), s.W, s.H, p(t["m"]), p(t["sh"]), None, p(alpha), scales, 1.0, rots, None,
            p(t["vw"]), p(t["pj"]), p(t["cp"]), s.tanfovx, s.tanfovy, p(radii), geom_, p(binning), p(img), p(t["gC"]), p(t["gD"]),
            p(t["gM"]), p(t["gV"]), *([None] * 10), 0, None, p(t["ps"]), p(dview), None, p(t["gt"]), 0, 1, p(scratch), scratch.numel())

    assert call(None, p(t["ro"]), p(geom)) == _capi.DGR_ERR_BAD_ARGUMENT
    assert call(p(t["sc"]), None, p(geom)) == _capi.DGR_ERR_BAD_ARGUMENT
    assert call(p(t["sc"]), p(t["ro"]), None) == _capi.DGR_ERR_BAD_ARGUMENT
    assert call(p(t["sc"]), p(t["ro"]), p(geom)) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("field", ["means", "scales", "rots", "opac", "shs"])
def test_poisoned_rows_do_not_fault(field):
    """NaN / Inf / 1e30 / denormal values in one Gaussian out of a hundred (a diverged optimisation): the reference validates
    nothing and answers with garbage for what such a Gaussian touches, not with a fault.  Same here, both variants: forward and
    backward complete, the lists stay inside their buffers (the library's own status word says so), and where the poison
    culls its own Gaussian (scales, rotations, opacities) every other row keeps a finite gradient and the image stays finite.
    (tests/tools/nan_inputs.py runs more shapes.)"""
    rng = np.random.default_rng(5)
    bad_values = [np.nan, np.inf, -np.inf, 1e30, -1e30, 0.0, 1e-38, -0.0]
    for case, (P, W, H) in enumerate([(12000, 250, 97), (30000, 64, 480)]):
        s = make_scene(P, W, H, 300 + case)
        a = getattr(s, field).copy()
        rows = rng.choice(P, size=P // 100, replace=False)
        flat = a.reshape(P, -1)
        for r in rows:
            flat[r, rng.integers(0, flat.shape[1])] = rng.choice(bad_values)
        s = s._replace(**{field: a})
        clean = np.ones(P, bool)
        clean[rows] = False
        for variant in ("light", "full"):
            if variant == "light":
                out, d = hh.hip_forward(s, 3)
                g = hh.hip_backward(s, 3, out)
            else:
                out, d = hh.hip_full_forward(s, 3)
                g = hh.hip_full_backward(s, 3, out)
            torch.cuda.synchronize()
            assert 0 < d["num_rendered"] < 64 * P
            if field in ("scales", "rots", "opac"):
                assert np.isfinite(d["color"]).all() and np.isfinite(d["depth"]).all(), (variant, case)
                for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
                    assert np.isfinite(g[k].reshape(P, -1)[clean]).all(), (variant, case, k)
