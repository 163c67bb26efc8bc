"""An INDEPENDENT check of the oracle's backward math: a float64 PyTorch formulation of the light
forward, written from SURVEY.md Appendix A's formulas (not from oracle/dgr_oracle.cpp), differentiated by autograd and
compared with the oracle's analytic gradients on tiny scenes.  A transcription error in one component of, say,
dL_drotations that the oracle and the kernels share would pass every oracle-vs-kernel test; it cannot pass this one.

Hard decisions (visibility, the per-pair alpha / power tests, the last contributor) are evaluated in float64 from the
same formulas and the test first asserts that the resulting images equal the oracle's, i.e. that the decisions agree.
Where the reference's analytic backward is deliberately NOT the derivative of its forward, the forward below is
shaped so that autograd produces the reference's quantity (each is a documented quirk):
  * alpha = min(0.99, o G) but the backward never masks the clamp (L/cr/backward.cu:627): straight-through clamp;
  * depth_var is 0 in the forward yet its gradient is consumed as d/d sum (d - gt)^2 alpha T (:600-608): the loss below
    contains that sum;
  * the median-depth gradient goes to the deepest valid Gaussian with T > 0.5 after the division (:656-663);
  * the pose gradient covers only mean2D (through proj = view x perspec) and the depth sum (:633-651): the view matrix
    enters the forward below as three tensors (ndc path, depth path, everything else) and only the first two are
    differentiated;
  * the clamp of t.x / t.z treats the clamped coordinate as independent of t.z (:175-176,262-264).
"""
import numpy as np
import pytest
import torch

from util import make_scene

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, d):
    """SURVEY A-P 9 / A-G: basis order and signs of */cr/forward.cu:30-59; sh [V,16,3], d [V,3] unit directions."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
        if deg > 2:
            r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
                 + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                 + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
                 + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp(r + 0.5, min=0.0)


def torch_light(s, deg, vis, point_list, ranges, n_contrib, grads, colors_precomp=None, cov3D_precomp=None):
    """Returns (loss, leaves dict, images dict).  `vis`, `point_list`, `ranges`, `n_contrib` come from the oracle's
    integer path (pinned separately by SURVEY Appendix C); everything float is recomputed here in float64.
    `colors_precomp` [P, 3] / `cov3D_precomp` [P, 6] (xx, xy, xz, yy, yz, zz) replace the SH evaluation / R diag(s^2) R^T as in the
    reference (L/cuda_rasterizer/forward.cu:208-218, 242-247); their gradients are then leaves `colors` / `cov3D`."""
    f = lambda a: torch.tensor(np.asarray(a, np.float64))  # noqa: E731
    W, H = s.W, s.H
    leaves = dict(means3D=f(s.means), scales=f(s.scales), rotations=f(s.rots), opacities=f(s.opac), shs=f(s.shs),
                  view_ndc=f(s.view), view_depth=f(s.view))
    if colors_precomp is not None:
        leaves["colors"] = f(colors_precomp)
    if cov3D_precomp is not None:
        leaves["cov3D"] = f(cov3D_precomp)
    for v in leaves.values():
        v.requires_grad_(True)
    view_o, persp, campos, bg, gt = f(s.view), f(s.persp), f(s.campos), f(s.bg), f(s.gt)
    idx = torch.tensor(np.nonzero(vis)[0])
    m = leaves["means3D"][idx]
    mh = torch.cat([m, torch.ones(len(idx), 1, dtype=torch.float64)], 1)
    # A-P 2, 7: p_hom = proj m, p_w = 1 / (w + 1e-7), pixel = ((ndc + 1) S - 1) / 2          (pose path 1)
    p_hom = mh @ (leaves["view_ndc"] @ persp)
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    pix = torch.stack([((p_hom[:, 0] * p_w + 1.0) * W - 1.0) * 0.5, ((p_hom[:, 1] * p_w + 1.0) * H - 1.0) * 0.5], 1)
    pix.retain_grad()   # (dL_dmeans2D in pixels, for tests/tools/arbitrate_fp64.py: returned as `_pix`, rows `_idx`)
    z_depth = (mh @ leaves["view_depth"])[:, 2]                                              # (pose path 2)
    t = (mh @ view_o)[:, :3]                                                                  # (no pose gradient)
    z_cam = t[:, 2]
    # A-P 3 / A-G: Sigma = R diag(s^2) R^T with the UNNORMALISED quaternion (r, x, y, z)
    q = leaves["rotations"][idx]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    sc = leaves["scales"][idx]
    Sigma = R @ torch.diag_embed(sc * sc) @ R.transpose(1, 2)
    if cov3D_precomp is not None:  # (an off-diagonal value stands at two places of Sigma: its gradient is their sum, backward.cu:281-283)
        c6 = leaves["cov3D"][idx]
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]], 1).reshape(-1, 3, 3)
    # A-P 4 / A-G: cov2D = A Sigma A^T + 0.3 I, A = Ju Rcam, t.x/t.z clamped to +-1.3 tanfov
    fx, fy = W / (2.0 * s.tanfovx), H / (2.0 * s.tanfovy)
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    rx, ry = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    tx = torch.where(rx.abs() > limx, (torch.clamp(rx, -limx, limx) * t[:, 2]).detach(), t[:, 0])
    ty = torch.where(ry.abs() > limy, (torch.clamp(ry, -limy, limy) * t[:, 2]).detach(), t[:, 1])
    tz = t[:, 2]
    zero = torch.zeros_like(tz)
    Ju = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    A = Ju @ view_o[:3, :3].t()
    cov = A @ Sigma @ A.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    con_a, con_b, con_c = c / det, -b / det, a / det
    # A-P 9
    dirs = m - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    rgb = sh_to_rgb(deg, leaves["shs"][idx], dirs) if colors_precomp is None else leaves["colors"][idx]
    opac = leaves["opacities"][idx, 0]
    slot = np.full(s.P, -1, np.int64)
    slot[np.nonzero(vis)[0]] = np.arange(len(idx))

    color = torch.zeros(3, H, W, dtype=torch.float64)
    depth = torch.zeros(H, W, dtype=torch.float64)
    alpha_img = torch.zeros(H, W, dtype=torch.float64)
    var = torch.zeros(H, W, dtype=torch.float64)
    median = torch.zeros(H, W, dtype=torch.float64)
    gx = (W + 15) // 16
    nc = torch.tensor(np.asarray(n_contrib, np.int64).reshape(H, W))
    for tile, (lo, hi) in enumerate(np.asarray(ranges).reshape(-1, 2)):
        if hi <= lo:
            continue
        x0, y0 = (tile % gx) * 16, (tile // gx) * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 16, H)
        ids = torch.tensor(slot[np.asarray(point_list[lo:hi], np.int64)])
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxs, pys = xs.reshape(-1).double(), ys.reshape(-1).double()
        dx = pix[ids, 0:1] - pxs[None]          # A-R: d = xy - pixel
        dy = pix[ids, 1:2] - pys[None]
        power = -0.5 * (con_a[ids, None] * dx * dx + con_c[ids, None] * dy * dy) - con_b[ids, None] * dx * dy
        oG = opac[ids, None] * torch.exp(power)
        alpha = oG + (torch.clamp(oG, max=0.99) - oG).detach()   # straight-through clamp (backward.cu:627)
        pos = torch.arange(hi - lo)[:, None]
        ncp = nc[y0:y1, x0:x1].reshape(-1)[None]
        valid = (power <= 0) & (alpha >= 15.0 / 255.0) & (pos < ncp)
        av = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tincl = torch.cumprod(1.0 - av, 0)
        Texcl = torch.cat([torch.ones(1, av.shape[1], dtype=torch.float64), Tincl[:-1]], 0)
        w = av * Texcl
        T_final = Tincl[-1]
        sel = (slice(None), slice(y0, y1), slice(x0, x1))
        color[sel] = ((w[:, :, None] * rgb[ids][:, None, :]).sum(0) + T_final[:, None] * bg[None]).t().reshape(3, y1 - y0, x1 - x0)
        depth[sel[1:]] = (w * z_depth[ids, None]).sum(0).reshape(y1 - y0, x1 - x0)
        alpha_img[sel[1:]] = w.sum(0).reshape(y1 - y0, x1 - x0)
        e = z_cam[ids, None] - gt[y0:y1, x0:x1].reshape(-1)[None]
        var[sel[1:]] = (w * e * e).sum(0).reshape(y1 - y0, x1 - x0)
        # deepest valid Gaussian whose transmittance before it exceeds 0.5 (the backward's criterion, :656-663)
        cand = valid & (Texcl > 0.5)
        last = (cand * (pos + 1)).max(0).values - 1
        has = last >= 0
        zm = z_cam[ids][last.clamp(min=0)]
        median[sel[1:]] = torch.where(has, zm, torch.zeros_like(zm)).reshape(y1 - y0, x1 - x0)
    gC, gD, gM, gV = (f(g) for g in grads)
    loss = (gC * color).sum() + (gD * depth).sum() + (gM * median).sum() + (gV * var).sum()
    return loss, leaves, dict(color=color.detach().numpy(), depth=depth.detach().numpy(),
                              opacity_map=alpha_img.detach().numpy(), _pix=pix, _idx=idx.numpy())


CASES = [(400, 64, 48, 3, 11), (300, 40, 40, 0, 12), (500, 70, 45, 2, 13), (300, 40, 40, 1, 15)]


@pytest.mark.parametrize("case", CASES)
def test_oracle_backward_equals_fp64_autograd(oracle, case):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    grads = tuple(np.asarray(g, np.float64) * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    st, ref = oracle.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                   s.tanfovx, s.tanfovy, H, W, s.shs, deg, s.campos)
    loss, leaves, img = torch_light(s, deg, ref["radii"] > 0, st.get("point_list"), st.get("ranges"), st.get("n_contrib"),
                                    grads)
    # same decisions: the float64 forward reproduces the oracle's float32 images to rounding
    for k, tol in (("color", 2e-6), ("depth", 1e-5), ("opacity_map", 2e-6)):
        d = np.abs(img[k].reshape(-1) - ref[k].astype(np.float64).reshape(-1))
        assert d.max() <= tol, f"{k}: float64 forward differs from the oracle by {d.max():.2e}"
    loss.backward()
    g = oracle.light_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.proj, s.tanfovx, s.tanfovy,
                              *(np.asarray(x, np.float32) for x in grads), s.gt, s.shs, deg, s.campos,
                              img["opacity_map"].astype(np.float32)[None], s.persp)
    pairs = dict(dL_dmeans3D=leaves["means3D"].grad, dL_dscales=leaves["scales"].grad, dL_drotations=leaves["rotations"].grad,
                 dL_dopacity=leaves["opacities"].grad, dL_dsh=leaves["shs"].grad,
                 dL_dview=leaves["view_ndc"].grad + leaves["view_depth"].grad)
    for k, t in pairs.items():
        a, b = np.asarray(g[k], np.float64).reshape(-1), t.numpy().reshape(-1)
        if k == "dL_dview":
            b = b.copy()
            b[[3, 7, 11, 15]] = 0.0  # never written by the reference (L/rasterize_points.cu:235)
        scale = np.abs(b).max()
        assert scale > 0, k
        err = np.abs(a - b).max() / scale
        # the oracle works in float32 (and forms T_final = 1 - alpha, a cancellation): measured 1e-6 .. 3e-5 of scale on
        # these scenes; a wrong term or sign in any component shows up at >= 1e-3
        assert err <= 5e-5, f"{k}: oracle vs float64 autograd differ by {err:.2e} of the tensor's scale"


def torch_full(s, deg, vis, point_list, ranges, n_contrib, grads):
    """The -full variant, written from SURVEY.md Appendix A (A-R, A-F) and the structure of ComputePG
    (F/cuda_rasterizer/backward.cu:990-1072, 1246-1289, 1316-1338) -- not from oracle/dgr_oracle.cpp.  Differences to the
    light formulation above, each shaped so that autograd yields the reference's quantity:
      * the terminating Gaussian IS blended (n_contrib includes it); no median; the third image is U = sum alpha T in the
        forward but its gradient is consumed as d/d sum (d - gt)^2 alpha T (quirk F2): the loss contains that sum;
      * the pose gradient is part 1 + part 2-1 of ComputePG only (part 2-2 is computed and never summed, F3):
          part 1   colour -> campos -> view, with campos = -(v0 v12 + v1 v13 + v2 v14, v4 v12 + .., v8 v12 + ..) and the colour's
                   derivative NOT masked by the 0-clamp (dgc_dCampos, F:159-166): an unclamped copy of the colour carries it;
          part 2-1 alpha -> ndc -> view for the COLOUR channels of every valid pair, without the background's share;
          depth    dL_depth * dd_dv is ASSIGNED per pair, not accumulated (F4): only the pixel's front-most valid Gaussian
                   contributes, through its own depth (v2, v6, v10, v14) and through its alpha's ndc path;
          the uncertainty channel does not enter the pose gradient (F7).
        Hence three copies of alpha per pair -- for the colour, the depth and the uncertainty channel -- equal in value and
        different in what they let a gradient reach."""
    f = lambda a: torch.tensor(np.asarray(a, np.float64))  # noqa: E731
    W, H = s.W, s.H
    leaves = dict(means3D=f(s.means), scales=f(s.scales), rotations=f(s.rots), opacities=f(s.opac), shs=f(s.shs),
                  view_ndc=f(s.view), view_depth=f(s.view), view_campos=f(s.view))
    for v in leaves.values():
        v.requires_grad_(True)
    view_o, persp, campos, bg, gt = f(s.view), f(s.persp), f(s.campos), f(s.bg), f(s.gt)
    idx = torch.tensor(np.nonzero(vis)[0])
    m = leaves["means3D"][idx]
    one = torch.ones(len(idx), 1, dtype=torch.float64)
    mh, mh_c = torch.cat([m, one], 1), torch.cat([m.detach(), one], 1)

    def pixels(mh_, proj_):
        p_hom = mh_ @ proj_
        p_w = 1.0 / (p_hom[:, 3] + 1e-7)
        return torch.stack([((p_hom[:, 0] * p_w + 1.0) * W - 1.0) * 0.5, ((p_hom[:, 1] * p_w + 1.0) * H - 1.0) * 0.5], 1)

    pix = pixels(mh, view_o @ persp)                     # gradient -> means
    pix_pose = pixels(mh_c, leaves["view_ndc"] @ persp)  # gradient -> view (ndc path), nothing else
    t = (mh @ view_o)[:, :3]
    z = t[:, 2]                                          # gradient -> means
    z_pose = (mh_c @ leaves["view_depth"])[:, 2]         # gradient -> view (depth path)
    q = leaves["rotations"][idx]
    r, x, y, zq = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + zq * zq), 2 * (x * y - r * zq), 2 * (x * zq + r * y),
                     2 * (x * y + r * zq), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - r * x),
                     2 * (x * zq - r * y), 2 * (y * zq + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    sc = leaves["scales"][idx]
    Sigma = R @ torch.diag_embed(sc * sc) @ R.transpose(1, 2)
    fx, fy = W / (2.0 * s.tanfovx), H / (2.0 * s.tanfovy)
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    rx, ry = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    tx = torch.where(rx.abs() > limx, (torch.clamp(rx, -limx, limx) * t[:, 2]).detach(), t[:, 0])
    ty = torch.where(ry.abs() > limy, (torch.clamp(ry, -limy, limy) * t[:, 2]).detach(), t[:, 1])
    tz = t[:, 2]
    zero = torch.zeros_like(tz)
    Ju = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    A = Ju @ view_o[:3, :3].t()
    cov = A @ Sigma @ A.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    con_a, con_b, con_c = c / det, -b / det, a / det
    dirs = m - campos
    rgb = sh_to_rgb(deg, leaves["shs"][idx], dirs / dirs.norm(dim=1, keepdim=True))
    # part 1: campos as the reference differentiates it, reached through an UNCLAMPED copy of the colour
    vc = leaves["view_campos"].reshape(-1)
    cam_v = -torch.stack([vc[0] * vc[12] + vc[1] * vc[13] + vc[2] * vc[14], vc[4] * vc[12] + vc[5] * vc[13] + vc[6] * vc[14],
                          vc[8] * vc[12] + vc[9] * vc[13] + vc[10] * vc[14]])
    assert float((cam_v.detach() - campos).abs().max()) < 1e-5  # (the scene's campos is that of its view matrix)
    dirs_c = m.detach() - cam_v
    rgb_c = sh_to_rgb_unclamped(deg, leaves["shs"][idx].detach(), dirs_c / dirs_c.norm(dim=1, keepdim=True))
    rgb = rgb + (rgb_c - rgb_c.detach())
    opac = leaves["opacities"][idx, 0]
    slot = np.full(s.P, -1, np.int64)
    slot[np.nonzero(vis)[0]] = np.arange(len(idx))

    color = torch.zeros(3, H, W, dtype=torch.float64)
    depth = torch.zeros(H, W, dtype=torch.float64)
    unc = torch.zeros(H, W, dtype=torch.float64)
    var = torch.zeros(H, W, dtype=torch.float64)
    gx = (W + 15) // 16
    nc = torch.tensor(np.asarray(n_contrib, np.int64).reshape(H, W))

    def blend_weights(av):
        Tincl = torch.cumprod(1.0 - av, 0)
        Texcl = torch.cat([torch.ones(1, av.shape[1], dtype=torch.float64), Tincl[:-1]], 0)
        return av * Texcl, Tincl[-1]

    for tile, (lo, hi) in enumerate(np.asarray(ranges).reshape(-1, 2)):
        if hi <= lo:
            continue
        x0, y0 = (tile % gx) * 16, (tile // gx) * 16
        x1, y1 = min(x0 + 16, W), min(y0 + 16, H)
        ids = torch.tensor(slot[np.asarray(point_list[lo:hi], np.int64)])
        ys, xs = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxs, pys = xs.reshape(-1).double(), ys.reshape(-1).double()

        def alpha_of(pix_, ca, cb, cc, o):
            dx = pix_[ids, 0:1] - pxs[None]
            dy = pix_[ids, 1:2] - pys[None]
            power = -0.5 * (ca[ids, None] * dx * dx + cc[ids, None] * dy * dy) - cb[ids, None] * dx * dy
            oG = o[ids, None] * torch.exp(power)
            return power, oG + (torch.clamp(oG, max=0.99) - oG).detach()   # straight-through clamp

        power, alpha = alpha_of(pix, con_a, con_b, con_c, opac)
        _, alpha_p = alpha_of(pix_pose, con_a.detach(), con_b.detach(), con_c.detach(), opac.detach())
        pos = torch.arange(hi - lo)[:, None]
        ncp = nc[y0:y1, x0:x1].reshape(-1)[None]
        valid = (power <= 0) & (alpha >= 15.0 / 255.0) & (pos < ncp)
        first = valid & (torch.cumsum(valid.long(), 0) == 1)   # the pixel's front-most valid Gaussian
        zeros = torch.zeros_like(alpha)
        pose_term = alpha_p - alpha_p.detach()                  # value 0, gradient -> view (ndc path)
        a_col = torch.where(valid, alpha + pose_term, zeros)
        a_dep = torch.where(valid, alpha + torch.where(first, pose_term, zeros), zeros)
        a_unc = torch.where(valid, alpha, zeros)
        w_col, _ = blend_weights(a_col)
        w_dep, _ = blend_weights(a_dep)
        # (dpixel_dalpha = T (c - accum_rec), F/cuda_rasterizer/backward.cu:692: the background's share of dL/dalpha (:733)
        #  is NOT in the pose gradient -- the background is weighted with the transmittance that carries no pose term)
        w_unc, T_final = blend_weights(a_unc)
        zd = z[ids, None] + torch.where(first, (z_pose - z_pose.detach())[ids, None], zeros)
        sel = (slice(None), slice(y0, y1), slice(x0, x1))
        color[sel] = ((w_col[:, :, None] * rgb[ids][:, None, :]).sum(0) + T_final[:, None] * bg[None]).t().reshape(3, y1 - y0, x1 - x0)
        depth[sel[1:]] = (w_dep * zd).sum(0).reshape(y1 - y0, x1 - x0)
        unc[sel[1:]] = w_unc.sum(0).reshape(y1 - y0, x1 - x0)
        e = z[ids, None] - gt[y0:y1, x0:x1].reshape(-1)[None]
        var[sel[1:]] = (w_unc * e * e).sum(0).reshape(y1 - y0, x1 - x0)
    gC, gD, gU = (f(g) for g in grads)
    loss = (gC * color).sum() + (gD * depth).sum() + (gU * var).sum()
    return loss, leaves, dict(color=color.detach().numpy(), depth=depth.detach().numpy(), uncertainty=unc.detach().numpy())


def sh_to_rgb_unclamped(deg, sh, d):
    """sh_to_rgb without the final max(., 0): what dgc_dCampos differentiates (F/cuda_rasterizer/backward.cu:159-166)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
        if deg > 2:
            r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
                 + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                 + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
                 + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r + 0.5


@pytest.mark.parametrize("case", CASES)
def test_oracle_full_backward_equals_fp64_autograd(oracle, case):
    """a12 / a16: the full variant's per-Gaussian gradients (uncertainty consumed as a variance) and its pose gradient
    (ComputePG part 1 + part 2-1, depth terms of the front-most valid Gaussian only) against the formulation above."""
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    grads = tuple(np.asarray(g, np.float64) * (W * H) ** 0.5 for g in (s.gC, s.gD, s.gV))
    st, ref = oracle.full_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                  s.tanfovx, s.tanfovy, H, W, s.shs, deg, s.campos)
    loss, leaves, img = torch_full(s, deg, ref["radii"] > 0, st.get("point_list"), st.get("ranges"), st.get("n_contrib"), grads)
    for k, tol in (("color", 2e-6), ("depth", 1e-5), ("uncertainty", 2e-6)):
        d = np.abs(img[k].reshape(-1) - ref[k].astype(np.float64).reshape(-1))
        assert d.max() <= tol, f"{k}: float64 forward differs from the oracle by {d.max():.2e}"
    loss.backward()
    g = oracle.full_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj, s.tanfovx,
                             s.tanfovy, *(np.asarray(x, np.float32) for x in grads), s.shs, deg, s.campos, s.persp)
    pairs = dict(dL_dmeans3D=leaves["means3D"].grad, dL_dscales=leaves["scales"].grad, dL_drotations=leaves["rotations"].grad,
                 dL_dopacity=leaves["opacities"].grad, dL_dsh=leaves["shs"].grad,
                 dL_dview=sum(leaves[k].grad for k in ("view_ndc", "view_depth", "view_campos") if leaves[k].grad is not None))
    # (SH degree 0: the colour does not depend on the view direction, so nothing reaches view_campos)
    for k, t in pairs.items():
        a, b = np.asarray(g[k], np.float64).reshape(-1), t.numpy().reshape(-1)
        if k == "dL_dview":
            b = b.copy()
            b[[3, 7, 11, 15]] = 0.0  # never written by the reference (quirk F7)
        scale = np.abs(b).max()
        assert scale > 0, k
        err = np.abs(a - b).max() / scale
        assert err <= 5e-5, f"{k}: oracle vs float64 autograd differ by {err:.2e} of the tensor's scale"


def test_rigid_camera_identity(oracle):
    """for a rigid camera dL/dt = R sum_g dL_dmeans3D[g]; the light pose gradient omits the cov2D
    branch, so the identity holds to a few per cent in x and y (a convention check of view / proj / perspec)."""
    s = make_scene(3000, 96, 64, 7)
    N = s.W * s.H
    z = np.zeros((s.H, s.W), np.float32)
    st, ref = oracle.light_forward(s.bg, s.means, None, s.opac, s.scales, s.rots, 1.0, None, s.view, s.gt, s.proj,
                                   s.tanfovx, s.tanfovy, s.H, s.W, s.shs, 0, s.campos)
    g = oracle.light_backward(st, s.bg, s.means, None, s.scales, s.rots, 1.0, None, s.view, s.proj, s.tanfovx, s.tanfovy,
                              s.gC * N, s.gD * N, z, z, s.gt, s.shs, 0, s.campos, ref["opacity_map"], s.persp)
    Rm = s.view[:3, :3].T.astype(np.float64)  # view = W2C^T
    lhs = Rm @ g["dL_dmeans3D"].astype(np.float64).sum(0)
    rhs = g["dL_dview"].reshape(-1)[[12, 13, 14]].astype(np.float64)
    assert np.all(np.abs(lhs[:2] - rhs[:2]) <= 0.05 * np.abs(rhs[:2])), (lhs, rhs)
    assert np.sign(lhs[2]) == np.sign(rhs[2])
