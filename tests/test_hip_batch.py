"""The batched multi-view entry points (dgr_light_forward_batch / dgr_light_backward_batch,
dgr_amd.batch) -- V cameras over one set of Gaussians per call.

What is pinned:
  * every view's outputs and state buffers are BIT-IDENTICAL to a one-view call (which the other parity tests hold
    against the oracle), and the views are checked against the oracle directly as well;
  * the gradients of the Gaussians are the sum over the views: against the sum of the ORACLE's per-view backward passes
    (stage-isolated, flipped pixels masked, the bars of tests/test_hip_light_parity.py), and against the one-view HIP
    backward accumulated in view order to 1e-5 of each tensor's scale (measured <= 2e-6) (the per-Gaussian stage adds the views' terms in that
    order with the one-view kernel's operations; what is left is the order of the blend backward's float atomics, which
    differs between any two runs);
  * pose gradients and dL_dmeans2D stay per view.
"""
import numpy as np
import pytest
import torch

from util import assert_grad_close, assert_image_close, make_scene, mask_flipped_pixels
import hip_helpers as hh
from dgr_amd import batch as B
from dgr_amd import light as L

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["compiled", "ctypes"])
def binding(request, monkeypatch):
    """every test runs over the compiled torch extension (csrc/torch_ext.cpp: light_forward_batch / light_backward_batch) and
    over the ctypes binding of the same C ABI"""
    if request.param == "ctypes":
        monkeypatch.setattr(L, "_C", L._CtypesC)
    elif L._C is not L._CompiledC:
        pytest.skip("compiled extension not built")
    assert (B._ext() is not None) == (request.param == "compiled")

IMAGES = ("color", "depth", "depth_median", "opacity_map")
T, E = hh.T, hh.E


def close(a, b, tol=1e-5):
    """Two runs of the blend backward add their float atomics in different orders, so two backward passes agree to
    rounding of the sums, not bit for bit: max |a - b| <= tol * max |b|.  (Measured on these scenes: <= 2e-6; the bar leaves
    room for the heavy tail tests/tools/soak_batch.py found on ill-conditioned Gaussians, where the one-view backward
    differs from ITSELF by more.)"""
    a, b = (x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x) for x in (a, b))
    scale = float(np.abs(b).max())
    return float(np.abs(a.astype(np.float64) - b).max()) <= tol * scale + 1e-30


def scenes(P, W, H, V, seed=0):
    return [make_scene(P, W, H, seed, view_index=v) for v in range(V)]


def batch_forward(ss, deg, colors_precomp=None, cov3D_precomp=None):
    s = ss[0]
    use_sh, use_sr = colors_precomp is None, cov3D_precomp is None
    views = T(np.stack([x.view for x in ss]))
    projs = T(np.stack([x.proj for x in ss]))
    campos = T(np.stack([x.campos for x in ss]))
    gts = T(np.stack([x.gt for x in ss]))
    out = B._forward_batch(T(s.bg), T(s.means), E() if use_sh else T(colors_precomp), T(s.opac), T(s.scales) if use_sr else E(),
                           T(s.rots) if use_sr else E(), 1.0, E() if use_sr else T(cov3D_precomp), views, gts, projs,
                           s.tanfovx, s.tanfovy, s.H, s.W, T(s.shs) if use_sh else E(), deg, campos, False)
    return out, (views, projs, campos, gts)


def batch_backward(ss, deg, out, cams, grads, alphas=None, colors_precomp=None, cov3D_precomp=None, track_off=False,
                   map_off=False, need_gaussian_grads=True):
    s = ss[0]
    use_sh, use_sr = colors_precomp is None, cov3D_precomp is None
    (R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
    views, projs, campos, gts = cams
    gC = T(np.stack([g[0] for g in grads]))
    gD, gM, gV = (T(np.stack([g[i][None] for g in grads])) for i in (1, 2, 3))
    if alphas is not None:
        alpha = T(np.stack(alphas))
    g = B._backward_batch(T(s.bg), T(s.means), radii, E() if use_sh else T(colors_precomp), T(s.scales) if use_sr else E(),
                          T(s.rots) if use_sr else E(), 1.0, E() if use_sr else T(cov3D_precomp), views, projs, s.tanfovx,
                          s.tanfovy, gC, gD, gM, gV, gts, T(s.shs) if use_sh else E(), deg, campos, geom, binning, img, alpha,
                          T(s.persp), track_off, map_off, need_gaussian_grads, True, num_rendered=R)
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations",
             "dL_dview"]
    return {n: (None if v is None else v.cpu().numpy()) for n, v in zip(names, g)}


def one_view_dict(out, v):
    """the batch's outputs of view v in the shape hip_helpers' one-view functions take"""
    (R, color, depth, median, var, alpha, radii, geom, binning, img, unc, px) = out
    return (R[v], color[v], depth[v], median[v], var[v], alpha[v], radii[v], geom[v], binning[v], img[v], unc[v], px[v])


@pytest.mark.parametrize("case", [(2000, 70, 45, 3, 1, 3), (20000, 320, 200, 3, 0, 4), (20000, 320, 200, 1, 2, 8),
                                  (3000, 64, 48, 0, 3, 1)])
def test_every_view_of_a_batch_is_bit_identical_to_a_one_view_call(case):
    P, W, H, deg, seed, V = case
    ss = scenes(P, W, H, V, seed)
    out, _ = batch_forward(ss, deg)
    names = ["color", "depth", "depth_median", "depth_var", "opacity_map", "radii", None, None, None, "gau_uncertainty",
             "gau_related_pixels"]
    for v, s in enumerate(ss):
        one, d1 = hh.hip_forward(s, deg)
        ov = one_view_dict(out, v)
        assert ov[0] == one[0]
        for k, name in enumerate(names):
            if name == "gau_uncertainty":  # (a sum of float atomics: its order differs between any two runs)
                assert close(ov[1 + k], one[1 + k]), (v, name)
            elif name is not None:
                assert torch.equal(ov[1 + k], one[1 + k]), (v, name)
        dv = {"num_rendered": ov[0], "geom": ov[7], "binning": ov[8], "img": ov[9]}
        for name in ("ranges", "point_list", "keys", "n_contrib", "tiles_touched", "contribution_tags"):
            a, b = hh.hip_state(name, s, dv), hh.hip_state(name, s, d1)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (v, name)
        vis = d1["radii"] > 0  # (per-Gaussian state is written for visible Gaussians; the other rows are never read)
        for name, w in (("means2D", 2), ("conic_opacity", 4), ("rgb", 3), ("clamped", 3), ("depths", 1)):
            a, b = hh.hip_state(name, s, dv).reshape(P, w)[vis], hh.hip_state(name, s, d1).reshape(P, w)[vis]
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), (v, name)


def test_batch_with_precomputed_colours_and_covariances_and_the_global_atomic_count():
    P, W, H, deg, V = 5000, 160, 96, 3, 3
    ss = scenes(P, W, H, V, 5)
    rng = np.random.default_rng(7)
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    cov = torch.empty((P, 6), device=hh.dev())
    L._capi.load().dgr_cov3d_forward(L._capi.stream_handle(), P, T(ss[0].scales).data_ptr(), T(ss[0].rots).data_ptr(), 1.0,
                                     cov.data_ptr())
    cov = cov.cpu().numpy()
    for lds in (1, 0):  # 0: the batch falls back to the one-view front end per view (fused global-atomic count)
        L._capi.set_option("lds_count", lds)
        try:
            out, _ = batch_forward(ss, deg, colors_precomp=colors, cov3D_precomp=cov)
            for v, s in enumerate(ss):
                one, _ = hh.hip_forward(s, deg, colors_precomp=colors, cov3D_precomp=cov)
                ov = one_view_dict(out, v)
                assert ov[0] == one[0]
                for k in (1, 2, 3, 5, 6, 11):
                    assert torch.equal(ov[k], one[k]), (lds, v, k)
                assert close(ov[10], one[10]), (lds, v)  # gau_uncertainty: a sum of float atomics
        finally:
            L._capi.set_option("lds_count", 1)


# (the last case is BASELINE config 3's size, two views: ~5 s of oracle time on the GPU box's cores)
@pytest.mark.parametrize("case", [(20000, 320, 200, 3, 0, 4), (100000, 640, 480, 3, 0, 3), (500000, 1920, 1080, 3, 0, 2)])
def test_batch_against_the_oracle(oracle, case):
    """Forward images per view and the batch's summed gradients against the oracle's per-view passes (stage-isolated:
    the oracle's alpha image feeds both backward passes; pixels on which the forward passes decided a hard threshold
    differently get zero incoming gradient on both sides, tests/util.py)."""
    P, W, H, deg, seed, V = case
    ss = scenes(P, W, H, V, seed)
    out, cams = batch_forward(ss, deg)
    grads, alphas, ref_sum, ref_view, ref_m2d = [], [], None, [], []
    for v, s in enumerate(ss):
        st, ref = hh.oracle_forward(oracle, s, deg)
        ov = one_view_dict(out, v)
        assert ov[0] == ref["num_rendered"] and np.array_equal(ov[6].cpu().numpy(), ref["radii"])
        d = {"color": ov[1].cpu().numpy(), "depth": ov[2].cpu().numpy(), "depth_median": ov[3].cpu().numpy(),
             "opacity_map": ov[5].cpu().numpy()}
        # (default alpha path: the alpha image and the median depth carry the oracle's bits)
        assert np.array_equal(d["opacity_map"], ref["opacity_map"]) and np.array_equal(d["depth_median"], ref["depth_median"])
        for k in IMAGES:
            assert_image_close(d[k], ref[k], k, tol=1e-6, max_outliers=0.0)
        dv = {"num_rendered": ov[0], "geom": ov[7], "binning": ov[8], "img": ov[9]}
        assert np.array_equal(hh.hip_state("point_list", s, dv), st.get("point_list"))
        g = tuple(x * (W * H) ** 0.5 for x in (s.gC, s.gD, s.gM, s.gV))
        g, _ = mask_flipped_pixels(g, hh.hip_state("n_contrib", s, dv), st.get("n_contrib"), W, H, f"batch view {v}",
                                   images=[(d[k], ref[k]) for k in IMAGES],
                                   median_margin=oracle.light_median_margin(st, ref["opacity_map"]))
        gr = hh.oracle_backward(oracle, st, s, deg, ref["opacity_map"], grads=g)
        grads.append(g)
        alphas.append(ref["opacity_map"])
        ref_view.append(gr["dL_dview"])
        ref_m2d.append(gr["dL_dmeans2D"])
        if ref_sum is None:
            ref_sum = {k: gr[k].astype(np.float64) for k in gr}
        else:
            for k in gr:
                ref_sum[k] += gr[k]
    g = batch_backward(ss, deg, out, cams, grads, alphas=alphas)
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert_grad_close(g[k], ref_sum[k].reshape(g[k].shape), k, rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-4)
    for v in range(V):
        assert_grad_close(g["dL_dmeans2D"][v], ref_m2d[v], f"dL_dmeans2D[{v}]", rel_to_max=1e-5, elem_rtol=1e-3, elem_frac=1e-4)
        assert_grad_close(g["dL_dview"][v], ref_view[v], f"dL_dview[{v}]", rel_to_max=1e-4, elem_rtol=1e-2, elem_frac=0.25)


@pytest.mark.parametrize("case", [(2000, 70, 45, 3, 1, 3), (20000, 320, 200, 3, 0, 4), (20000, 320, 200, 2, 4, 8)])
def test_batch_backward_is_the_one_view_backward_accumulated_in_view_order(case):
    P, W, H, deg, seed, V = case
    ss = scenes(P, W, H, V, seed)
    out, cams = batch_forward(ss, deg)
    grads = [tuple(x * (W * H) ** 0.5 for x in (s.gC, s.gD, s.gM, s.gV)) for s in ss]
    g = batch_backward(ss, deg, out, cams, grads)
    acc = None
    for v, s in enumerate(ss):
        # (the one-view backward on the state buffers the BATCHED forward left: they are interchangeable)
        g1 = hh.hip_backward(s, deg, one_view_dict(out, v), grads=grads[v])
        assert close(g["dL_dmeans2D"][v], g1["dL_dmeans2D"]), v
        assert close(g["dL_dview"][v], g1["dL_dview"], 1e-5), v
        if acc is None:
            acc = {k: g1[k].copy() for k in g1}
        else:
            for k in g1:
                acc[k] = acc[k] + g1[k]  # float32, view order: what autograd's `.grad +=` does
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
        assert close(g[k], acc[k]), k


def test_batch_with_precomputed_inputs_backward_and_tracking_mode():
    P, W, H, deg, V = 5000, 160, 96, 3, 3
    ss = scenes(P, W, H, V, 5)
    rng = np.random.default_rng(7)
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    cov = torch.empty((P, 6), device=hh.dev())
    L._capi.load().dgr_cov3d_forward(L._capi.stream_handle(), P, T(ss[0].scales).data_ptr(), T(ss[0].rots).data_ptr(), 1.0,
                                     cov.data_ptr())
    cov = cov.cpu().numpy()
    out, cams = batch_forward(ss, deg, colors_precomp=colors, cov3D_precomp=cov)
    grads = [tuple(x * (W * H) ** 0.5 for x in (s.gC, s.gD, s.gM, s.gV)) for s in ss]
    g = batch_backward(ss, deg, out, cams, grads, colors_precomp=colors, cov3D_precomp=cov)
    acc = None
    for v, s in enumerate(ss):
        g1 = hh.hip_backward(s, deg, one_view_dict(out, v), grads=grads[v], colors_precomp=colors, cov3D_precomp=cov)
        assert close(g["dL_dview"][v], g1["dL_dview"], 1e-5)
        acc = {k: g1[k].copy() for k in g1} if acc is None else {k: acc[k] + g1[k] for k in g1}
    for k in ("dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dcov3D"):
        assert close(g[k], acc[k]), k
    # tracking: no per-Gaussian gradient is asked for; the pose gradients are those of the one-view tracking backward
    gt = batch_backward(ss, deg, out, cams, grads, colors_precomp=colors, cov3D_precomp=cov, need_gaussian_grads=False)
    assert gt["dL_dmeans3D"] is None and gt["dL_dsh"] is None
    for v, s in enumerate(ss):
        g1 = hh.hip_backward(s, deg, one_view_dict(out, v), grads=grads[v], colors_precomp=colors, cov3D_precomp=cov, map_off=True)
        assert close(gt["dL_dview"][v], g1["dL_dview"], 1e-5)


def test_autograd_surface_of_the_batch_equals_the_loop_over_views():
    P, W, H, deg, V = 20000, 320, 200, 3, 4
    ss = scenes(P, W, H, V, 0)
    s = ss[0]
    f32 = dict(dtype=torch.float32, device=hh.dev())

    def leaves():
        return [T(a).clone().requires_grad_(True) for a in (s.means, s.shs, s.opac, s.scales, s.rots)]

    views = T(np.stack([x.view for x in ss]))
    projs, campos, gts = T(np.stack([x.proj for x in ss])), T(np.stack([x.campos for x in ss])), T(np.stack([x.gt for x in ss]))
    w = [torch.randn((V, c, H, W), **f32) / (H * W) ** 0.5 for c in (3, 1, 1)]

    # the loop: V one-view calls, autograd accumulates
    m1, sh1, o1, sc1, r1 = leaves()
    pose1 = views.clone().requires_grad_(True)
    m2d_1 = []
    for v in range(V):
        rs = L.GaussianRasterizationSettings(H, W, s.tanfovx, s.tanfovy, T(s.bg), 1.0, pose1[v], projs[v], deg, campos[v], False,
                                             False, T(s.persp), False, False)
        p2 = torch.zeros((P, 3), **f32, requires_grad=True)
        color, radii, depth, median, var, alpha, unc, px = L.GaussianRasterizer(rs)(
            m1, p2, o1, shs=sh1, scales=sc1, rotations=r1, viewmatrix=pose1[v], gt_depth=gts[v])
        ((color * w[0][v]).sum() + (depth * w[1][v]).sum() + (median * w[2][v]).sum()).backward()
        m2d_1.append(p2.grad)
    # the batch
    m2, sh2, o2, sc2, r2 = leaves()
    pose2 = views.clone().requires_grad_(True)
    rs = B.BatchRasterizationSettings(H, W, s.tanfovx, s.tanfovy, T(s.bg), 1.0, pose2, projs, deg, campos, False, False,
                                      T(s.persp), False, False)
    p2 = torch.zeros((V, P, 3), **f32, requires_grad=True)
    color, radii, depth, median, var, alpha, unc, px = B.GaussianRasterizerBatch(rs)(
        m2, p2, o2, shs=sh2, scales=sc2, rotations=r2, viewmatrices=pose2, gt_depths=gts)
    assert color.shape == (V, 3, H, W) and radii.shape == (V, P)
    ((color * w[0]).sum() + (depth * w[1]).sum() + (median * w[2]).sum()).backward()
    for a, b in ((m2.grad, m1.grad), (sh2.grad, sh1.grad), (o2.grad, o1.grad), (sc2.grad, sc1.grad), (r2.grad, r1.grad)):
        assert close(a, b)
    assert close(pose2.grad, pose1.grad, 1e-5)
    for v in range(V):
        assert close(p2.grad[v], m2d_1[v])
    # the summed gradients are views of one arena whose leading segments are the multi-GPU all-reduce span
    from dgr_amd.multiview import GradientArena
    span = GradientArena([m2, sh2, o2, sc2, r2]).fused_span()
    assert span is not None and span.numel() >= P * (3 + 3 + 48 + 1 + 3 + 4)


def test_empty_batch_and_bad_arguments():
    ss = scenes(0, 64, 48, 2, 0)
    out, cams = batch_forward(ss, 3)
    assert out[0] == [0, 0] and float(out[1].abs().max()) == 0.0
    ss = scenes(100, 64, 48, 9, 0)
    with pytest.raises(RuntimeError):
        batch_forward(ss, 3)


def test_a_batch_recorded_into_a_hipgraph_replays():
    P, W, H, deg, V = 20000, 320, 200, 3, 4
    ss = scenes(P, W, H, V, 0)
    s = ss[0]
    views = T(np.stack([x.view for x in ss]))
    projs, campos, gts = T(np.stack([x.proj for x in ss])), T(np.stack([x.campos for x in ss])), T(np.stack([x.gt for x in ss]))
    means, shs, opac, scales, rots = (T(a) for a in (s.means, s.shs, s.opac, s.scales, s.rots))
    gC = T(np.stack([x.gC for x in ss])) * (W * H) ** 0.5
    gD, gM, gV = (T(np.stack([getattr(x, n)[None] for x in ss])) * (W * H) ** 0.5 for n in ("gD", "gM", "gV"))
    import os
    old = os.environ.get("DGR_SYNC_MODE")
    os.environ["DGR_SYNC_MODE"] = "lazy"
    try:
        bg, persp, e0 = T(s.bg), T(s.persp), E()

        def step():
            out = B._forward_batch(bg, means, e0, opac, scales, rots, 1.0, e0, views, gts, projs, s.tanfovx, s.tanfovy, H, W,
                                   shs, deg, campos, False)
            g = B._backward_batch(bg, means, out[6], e0, scales, rots, 1.0, e0, views, projs, s.tanfovx, s.tanfovy, gC, gD,
                                  gM, gV, gts, shs, deg, campos, out[7], out[8], out[9], out[5], persp, False, False, True, True)
            return out[1], g[3], g[8]
        eager = [t.clone() for t in step()]  # (warm-up: learns the capacity, creates the internal streams)
        L.check_async_errors()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                res = step()
        for _ in range(2):
            for t in res:
                t.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(res[0], eager[0]) and close(res[2], eager[2], 1e-5) and close(res[1], eager[1])
        L.check_captured_status()
    finally:
        if old is None:
            os.environ.pop("DGR_SYNC_MODE", None)
        else:
            os.environ["DGR_SYNC_MODE"] = old


def test_batch_at_baseline_config3_size():
    """BASELINE config 3's Gaussians (500 k, 1920x1080, SH degree 3) from three cameras in one batch: every view bit-identical
    to its one-view call (which tests/test_hip_light_parity.py holds against the oracle at this size), the summed gradients
    equal to the one-view backward passes accumulated in view order."""
    P, W, H, deg, V = 500000, 1920, 1080, 3, 3
    ss = scenes(P, W, H, V, 0)
    out, cams = batch_forward(ss, deg)
    grads = [(s.gC, s.gD, s.gM, s.gV) for s in ss]
    g = batch_backward(ss, deg, out, cams, grads)
    acc = None
    for v, s in enumerate(ss):
        one, d1 = hh.hip_forward(s, deg)
        ov = one_view_dict(out, v)
        assert ov[0] == one[0] == d1["num_rendered"]
        for k in (1, 2, 3, 5, 6, 11):
            assert torch.equal(ov[k], one[k]), (v, k)
        dv = {"num_rendered": ov[0], "geom": ov[7], "binning": ov[8], "img": ov[9]}
        for name in ("ranges", "point_list", "n_contrib"):
            assert np.array_equal(hh.hip_state(name, s, dv), hh.hip_state(name, s, d1)), (v, name)
        g1 = hh.hip_backward(s, deg, one, grads=grads[v])
        assert close(g["dL_dview"][v], g1["dL_dview"], 1e-5) and close(g["dL_dmeans2D"][v], g1["dL_dmeans2D"])
        acc = {k: g1[k].copy() for k in g1} if acc is None else {k: acc[k] + g1[k] for k in g1}
    if ss[0].view is not None:
        assert out[0][0] == 1654310  # num_rendered of config 3's view 0
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
        assert close(g[k], acc[k]), k


@pytest.mark.parametrize("option", [("batch_order", 1), ("batch_streams", 1), ("batch_streams", 4)])
def test_stream_layout_options_do_not_change_a_batch(option):
    """dgr_set_option("batch_order" / "batch_streams"): how a batch's per-view stages are spread over the internal streams
    (round robin over K streams, or all binning on one stream and all blending on another) is a scheduling choice -- the
    forward stays bit-identical, the gradients agree to the order of the float atomics."""
    P, W, H, deg, V = 20000, 320, 200, 3, 5
    ss = scenes(P, W, H, V, 0)
    grads = [tuple(x * (W * H) ** 0.5 for x in (s.gC, s.gD, s.gM, s.gV)) for s in ss]
    out0, cams = batch_forward(ss, deg)
    g0 = batch_backward(ss, deg, out0, cams, grads)
    keep = L._capi.get_option(option[0])
    L._capi.set_option(*option)
    try:
        out1, cams1 = batch_forward(ss, deg)
        g1 = batch_backward(ss, deg, out1, cams1, grads)
    finally:
        L._capi.set_option(option[0], keep)
    for k in (1, 2, 3, 5, 6, 11):
        assert torch.equal(out0[k], out1[k]), k
    assert out0[0] == out1[0]
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
        assert close(g1[k], g0[k]), k
    assert close(g1["dL_dview"], g0["dL_dview"], 1e-5)
