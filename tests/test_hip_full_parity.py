"""Parity of the gfx950 -full path against the CPU oracle, through the C ABI (BASELINE config 2 is the -full
variant at 100k Gaussians, 640x480, SH degree 3, forward+backward incl. the viewmatrix gradient).

The pose gradient is compared against the oracle's WELL-DEFINED ComputePG (every recorded pair consumed); the
reference's own result is undefined for tiles holding a pixel without valid contributors -- see
tests/test_oracle_known_answers.py::test_appendix_c_full_variant and DESIGN.md."""
import numpy as np
import pytest

from util import assert_grad_close, assert_image_close, make_scene
import hip_helpers as hh

pytestmark = pytest.mark.gpu

CASES = [(2000, 64, 48, 0, 1), (2000, 70, 45, 3, 2), (10000, 256, 256, 0, 0), (10000, 256, 256, 3, 0),
         (100000, 640, 480, 3, 0)]


@pytest.mark.parametrize("case", CASES)
def test_full_forward(oracle, case):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    _, d = hh.hip_full_forward(s, deg)
    st, ref, _ = hh.oracle_full(oracle, s, deg, backward=False)
    assert np.array_equal(d["radii"], ref["radii"]) and d["num_rendered"] == ref["num_rendered"]
    assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
    # the default alpha path carries the host's bits (csrc/exact_math.h): every threshold decision, the final transmittance
    # and the "uncertainty" image (the sum of alpha T) are the restatement's; colour and depth are summed with fused
    # multiply-adds and stay within a few ulp
    for k in ("color", "depth", "uncertainty"):
        assert d[k].shape == ref[k].shape
    assert np.array_equal(d["uncertainty"], ref["uncertainty"])
    for k in ("color", "depth"):
        a, b = d[k].astype(np.float64), ref[k].astype(np.float64)
        assert np.all(np.abs(a - b) <= 1e-6 * np.maximum(1.0, np.abs(b))), (k, float(np.abs(a - b).max()))
    assert np.array_equal(hh.hip_state("n_contrib", s, d), st.get("n_contrib"))
    assert np.array_equal(hh.hip_state("n_valid", s, d), st.get("n_valid_contrib"))
    assert d["num_related"] == ref["num_related"]
    assert np.array_equal(hh.hip_state("final_T", s, d).view(np.float32), st.get("final_T"))


@pytest.mark.parametrize("case", CASES)
def test_full_backward(oracle, case):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    grads = tuple(g * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gV))
    out, d = hh.hip_full_forward(s, deg)
    g = hh.hip_full_backward(s, deg, out, grads=grads)
    st, ref, gr = hh.oracle_full(oracle, s, deg, grads=grads)
    # both forward passes walk the same lists (no threshold decision differs: the default alpha path carries the host's
    # bits), so one unconditional bar
    assert np.array_equal(hh.hip_state("n_contrib", s, d), st.get("n_contrib"))
    tol = dict(rel_to_max=1e-5, elem_rtol=2e-3, elem_frac=1e-3)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert g[k].shape == gr[k].shape, k
        assert_grad_close(g[k], gr[k], k, **tol)
    assert g["dL_dview"].shape == (4, 4)
    assert not g["dL_dview"].reshape(-1)[[3, 7, 11, 15]].any()
    assert_grad_close(g["dL_dview"], gr["dL_dview"], "dL_dview", rel_to_max=tol["rel_to_max"] * 5, elem_rtol=5e-3,
                      elem_frac=0.1)


def test_full_autograd_surface():
    """`diff_gaussian_rasterization` of the full flavour: field order, return arity, gradient order."""
    import torch
    from dgr_amd import full as F
    s = make_scene(3000, 96, 64, 7)
    dev = hh.dev()
    settings = F.GaussianRasterizationSettings(
        image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=hh.T(s.bg), scale_modifier=1.0,
        viewmatrix=hh.T(s.view), projmatrix=hh.T(s.proj), sh_degree=3, campos=hh.T(s.campos), prefiltered=False,
        perspec_matrix=hh.T(s.persp))
    assert F.GaussianRasterizationSettings._fields[-1] == "perspec_matrix"
    means3D, shs, opac = hh.T(s.means).requires_grad_(), hh.T(s.shs).requires_grad_(), hh.T(s.opac).requires_grad_()
    scales, rots, view = hh.T(s.scales).requires_grad_(), hh.T(s.rots).requires_grad_(), hh.T(s.view).requires_grad_()
    means2D = torch.zeros((s.P, 3), device=dev, requires_grad=True)
    color, radii, depth, unc = F.GaussianRasterizer(settings)(
        means3D=means3D, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=view,
        gt_depth=hh.T(s.gt))
    assert color.shape == (3, s.H, s.W) and depth.shape == (1, s.H, s.W) and unc.shape == (1, s.H, s.W)
    assert radii.dtype == torch.int32 and radii.shape == (s.P,)
    torch.autograd.backward([color, depth, unc], [hh.T(s.gC), hh.T(s.gD[None]), hh.T(s.gV[None])])
    assert view.grad.shape == (4, 4) and means2D.grad.shape == (s.P, 3) and shs.grad.shape == (s.P, 16, 3)
    assert float(view.grad.abs().sum()) > 0 and float(means3D.grad.abs().sum()) > 0


def test_full_precomputed_colors_and_covariances(oracle):
    """colors_precomp skips SH (dL_dcolors is then returned and the colour -> campos part of the pose gradient vanishes);
    cov3D_precomp skips scale/rotation (F/cuda_rasterizer/rasterizer_impl.cu)."""
    s = make_scene(4000, 96, 80, 3)
    st0, ref0, _ = hh.oracle_full(oracle, s, 3, backward=False)
    colors = st0.get("rgb").reshape(-1, 3).copy()
    cov3D = st0.get("cov3D").reshape(-1, 6).copy()
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gV))
    for kw in (dict(colors_precomp=colors), dict(cov3D_precomp=cov3D), dict(colors_precomp=colors, cov3D_precomp=cov3D)):
        out, d = hh.hip_full_forward(s, 3, **kw)
        st, ref, gr = hh.oracle_full(oracle, s, 3, grads=grads, **kw)
        assert np.array_equal(d["radii"], ref["radii"])
        assert np.array_equal(hh.hip_state("point_list", s, d), st.get("point_list"))
        assert np.array_equal(d["uncertainty"], ref["uncertainty"]) and d["num_related"] == ref["num_related"]
        assert np.array_equal(hh.hip_state("n_contrib", s, d), st.get("n_contrib"))
        for k in ("color", "depth"):
            a_, b_ = d[k].astype(np.float64), ref[k].astype(np.float64)
            assert np.all(np.abs(a_ - b_) <= 1e-6 * np.maximum(1.0, np.abs(b_))), k
        g = hh.hip_full_backward(s, 3, out, grads=grads, **kw)
        names = ["dL_dmeans3D", "dL_dopacity", "dL_dview"]
        names += ["dL_dcolors"] if "colors_precomp" in kw else ["dL_dsh"]
        names += ["dL_dcov3D"] if "cov3D_precomp" in kw else ["dL_dscales", "dL_drotations"]
        for k in names:
            assert_grad_close(g[k], gr[k], k, rel_to_max=1e-5, elem_rtol=2e-3, elem_frac=0.1 if k == "dL_dview" else 2e-3)


def test_full_backward_without_an_uncertainty_gradient_is_the_lean_kernel_and_equals_a_zero_image():
    """A loss on colour and depth only hands the compiled node no gradient for the uncertainty output: NULL at the C ABI, the
    lean blend backward (csrc/render_full.hip: LEAN) -- the same bits as an explicit all-zero image, up to the order of float
    atomics (F/cuda_rasterizer/backward.cu:701-708 is the term that drops out)."""
    import torch
    from dgr_amd import full as F
    s = make_scene(10000, 256, 256, 0)
    T = hh.T
    got = []
    for explicit_zero in (False, True):
        leaves = [T(a).requires_grad_() for a in (s.means, s.shs, s.opac, s.scales, s.rots, s.view)]
        m2 = torch.zeros((s.P, 3), device=hh.dev(), requires_grad=True)
        rast = F.GaussianRasterizer(F.GaussianRasterizationSettings(
            image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=T(s.bg), scale_modifier=1.0,
            viewmatrix=T(s.view), projmatrix=T(s.proj), sh_degree=3, campos=T(s.campos), prefiltered=False, perspec_matrix=T(s.persp)))
        color, radii, depth, unc = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], shs=leaves[1], scales=leaves[3],
                                        rotations=leaves[4], viewmatrix=leaves[5], gt_depth=T(s.gt))
        outs, grads = [color, depth], [T(s.gC), T(s.gD[None])]
        if explicit_zero:
            outs.append(unc)
            grads.append(torch.zeros_like(unc))
        torch.autograd.backward(outs, grads)
        got.append([x.grad.cpu().numpy() for x in leaves])
    for a, b in zip(*got):
        assert np.abs(b).max() > 0
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
