"""dgr_amd.slam: CG-SLAM's `render()` call (reference README.md:33,71) and the pose helpers.

CPU: the camera tensors equal synth-v1's matrices.  GPU: a tracking step works end to end -- starting from a perturbed
camera pose, gradient descent on (quaternion, translation) through the rasterizer's analytic viewmatrix gradient recovers
the true pose.  This checks the pose gradient functionally, with no oracle involved."""
import numpy as np
import pytest
import torch

from dgr_amd import slam
from dgr_amd.synth import camera, make_scene


def rot_to_quat(R):
    # (r, x, y, z) of a rotation matrix close to identity (trace > 0)
    r = np.sqrt(1.0 + np.trace(R)) / 2.0
    return np.array([r, (R[2, 1] - R[1, 2]) / (4 * r), (R[0, 2] - R[2, 0]) / (4 * r), (R[1, 0] - R[0, 1]) / (4 * r)])


def test_camera_tensors_match_synth_v1():
    W, H = 640, 480
    tanfovx, tanfovy, Rm, t, view, proj, persp, campos = camera(W, H, 0.05)
    q = torch.tensor(rot_to_quat(Rm), dtype=torch.float64)
    w2c = slam.w2c_from_quat_trans(q, torch.tensor(t, dtype=torch.float64))
    v, p, ps, c = slam.camera_tensors(w2c, tanfovx, tanfovy)
    np.testing.assert_allclose(v.numpy(), view, atol=2e-7)
    np.testing.assert_allclose(p.numpy(), proj, atol=2e-6)
    np.testing.assert_allclose(ps.numpy(), persp, atol=2e-7)
    np.testing.assert_allclose(c.numpy(), campos, atol=2e-7)


class Model:
    """The accessors of 3DGS's GaussianModel that render() uses."""

    def __init__(self, s, dev):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.get_xyz, self.get_opacity, self.get_scaling = t(s.means), t(s.opac), t(s.scales)
        self.get_rotation, self.get_features = t(s.rots), t(s.shs)
        self.active_sh_degree = 3


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["light", "full"])
def test_tracking_recovers_a_perturbed_pose(variant):
    dev = torch.device("cuda:0")
    W, H = 256, 192
    s = make_scene(20000, W, H, 3)
    pc = Model(s, dev)
    tanfovx, tanfovy, Rm, t_true, *_ = camera(W, H, 0.05)
    bg = torch.from_numpy(s.bg).to(dev)
    gt_depth = torch.from_numpy(s.gt).to(dev)
    kw = dict(fov=(tanfovx, tanfovy), HW=(H, W), gt_depth=gt_depth, variant=variant)
    if variant == "light":
        kw.update(track_off=False, map_off=True)  # tracking: pose gradient only (README.md:71)

    def pose(q, t):
        return slam.camera_tensors(slam.w2c_from_quat_trans(q, t), tanfovx, tanfovy)[0]

    q_true = torch.tensor(rot_to_quat(Rm), dtype=torch.float32, device=dev)
    t_true = torch.tensor(t_true, dtype=torch.float32, device=dev)
    with torch.no_grad():
        target = slam.render(None, pc, None, bg, viewmatrix=pose(q_true, t_true), **kw)
    assert set(target) >= {"render", "depth", "opacity_map", "viewspace_points", "visibility_filter", "radii"}
    if variant == "light":
        assert set(target) >= {"depth_median", "depth_var", "gau_uncertainty", "num_related_pixels"}
    tgt_c, tgt_d = target["render"].detach(), target["depth"].detach()

    q = (q_true + torch.tensor([0.0, 0.004, -0.006, 0.003], device=dev)).requires_grad_()
    t = (t_true + torch.tensor([0.012, -0.009, 0.015], device=dev)).requires_grad_()
    opt = torch.optim.Adam([{"params": [q], "lr": 5e-4}, {"params": [t], "lr": 1.5e-3}])

    def errors():
        with torch.no_grad():
            return float((q / q.norm() - q_true).norm()), float((t - t_true).norm())

    e0 = errors()
    losses = []
    for it in range(150):
        opt.zero_grad()
        out = slam.render(None, pc, None, bg, viewmatrix=pose(q, t), **kw)
        loss = (out["render"] - tgt_c).abs().mean() + 0.5 * (out["depth"] - tgt_d).abs().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    e1 = errors()
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])
    assert e1[0] < 0.3 * e0[0] and e1[1] < 0.3 * e0[1], (e0, e1)


@pytest.mark.gpu
def test_render_batch_accumulates_like_a_serial_loop():
    """slam.render_batch: four keyframes on three streams give the gradient sum of rendering them one after the other."""
    dev = torch.device("cuda:0")
    W, H = 160, 120
    scenes = [make_scene(6000, W, H, 3, view_index=k) for k in range(4)]
    s = scenes[0]
    bg = torch.from_numpy(s.bg).to(dev)
    gt = torch.from_numpy(s.gt).to(dev)
    targets = [torch.rand((3, H, W), device=dev) for _ in scenes]

    def model():
        m = Model(s, dev)
        for name in ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features"):
            setattr(m, name, getattr(m, name).clone().requires_grad_())
        return m

    def cams():
        return [dict(viewmatrix=torch.from_numpy(sc.view).to(dev), fov=(sc.tanfovx, sc.tanfovy), HW=(H, W), gt_depth=gt)
                for sc in scenes]

    def loss_fn(out, k):
        return (out["render"] - targets[k]).abs().mean() + 0.1 * out["depth"].mean()

    a = model()
    la = slam.render_batch(cams(), a, None, bg, loss_fn, views_in_flight=3)
    b = model()
    lb = []
    for k, cam in enumerate(cams()):
        out = slam.render(None, b, None, bg, viewmatrix=cam["viewmatrix"], fov=cam["fov"], HW=cam["HW"], gt_depth=gt)
        loss = loss_fn(out, k)
        loss.backward()
        lb.append(loss.detach())
    torch.cuda.synchronize()
    for x, y in zip(la, lb):
        assert abs(float(x) - float(y)) <= 1e-6 * abs(float(y))
    for name in ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features"):
        ga, gb = getattr(a, name).grad.cpu().numpy(), getattr(b, name).grad.cpu().numpy()
        scale = np.abs(gb).max()
        assert np.abs(ga - gb).max() <= 2e-5 * scale, name


@pytest.mark.gpu
def test_render_batch_fused_equals_the_serial_loop():
    """slam.render_batch_fused (one batched forward + one batched backward, dgr_amd.batch) against rendering the keyframes one
    after the other: losses, the summed gradients of the Gaussians, every pose gradient, the per-view screen-space gradients."""
    dev = torch.device("cuda:0")
    W, H = 160, 120
    scenes = [make_scene(6000, W, H, 3, view_index=k) for k in range(4)]
    s = scenes[0]
    bg = torch.from_numpy(s.bg).to(dev)
    gt = torch.from_numpy(s.gt).to(dev)
    targets = [torch.rand((3, H, W), device=dev) for _ in scenes]

    def model():
        m = Model(s, dev)
        for name in ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features"):
            setattr(m, name, getattr(m, name).clone().requires_grad_())
        return m

    def cams():
        return [dict(viewmatrix=torch.from_numpy(sc.view).to(dev).requires_grad_(), fov=(sc.tanfovx, sc.tanfovy), HW=(H, W),
                     gt_depth=gt) for sc in scenes]

    def loss_fn(out, k):
        return (out["render"] - targets[k]).abs().mean() + 0.1 * out["depth"].mean() + 0.05 * out["depth_median"].mean()

    a, ca = model(), cams()
    la, out = slam.render_batch_fused(ca, a, None, bg, loss_fn)
    assert out["render"].shape == (4, 3, H, W) and out["viewspace_points"].grad.shape == (4, 6000, 3)
    b, cb = model(), cams()
    lb, pts = [], []
    for k, cam in enumerate(cb):
        o = slam.render(None, b, None, bg, viewmatrix=cam["viewmatrix"], fov=cam["fov"], HW=cam["HW"], gt_depth=gt)
        assert torch.equal(o["render"], out["render"][k]) and torch.equal(o["radii"], out["radii"][k])
        loss = loss_fn(o, k)
        loss.backward()
        lb.append(loss.detach())
        pts.append(o["viewspace_points"].grad)
    torch.cuda.synchronize()
    for x, y in zip(la, lb):
        assert abs(float(x) - float(y)) <= 1e-6 * abs(float(y))

    def close(x, y, tol, what):
        x, y = x.detach().cpu().numpy(), y.detach().cpu().numpy()
        assert np.abs(x - y).max() <= tol * np.abs(y).max(), what

    for name in ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features"):
        close(getattr(a, name).grad, getattr(b, name).grad, 2e-5, name)
    for k in range(4):
        close(ca[k]["viewmatrix"].grad, cb[k]["viewmatrix"].grad, 2e-5, f"pose {k}")
        close(out["viewspace_points"].grad[k], pts[k], 2e-5, f"viewspace_points {k}")


@pytest.mark.gpu
def test_one_loss_over_the_stacked_views_equals_the_per_view_losses():
    """render_batch_fused(batch_loss_fn=...): the keyframes' L1 losses as ONE reduction over the stacked outputs (what
    examples/mapping.py --fused does) against V per-view losses summed: same total, same gradients."""
    dev = torch.device("cuda:0")
    W, H, V = 160, 120, 3
    scenes = [make_scene(5000, W, H, 3, view_index=k) for k in range(V)]
    s = scenes[0]
    bg, gt = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
    obs_c, obs_d = torch.rand((V, 3, H, W), device=dev), 1.0 + 4.0 * torch.rand((V, 1, H, W), device=dev)
    names = ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features")

    def model():
        m = Model(s, dev)
        for name in names:
            setattr(m, name, getattr(m, name).clone().requires_grad_())
        return m

    def cams():
        return [dict(viewmatrix=torch.from_numpy(sc.view).to(dev).requires_grad_(), fov=(sc.tanfovx, sc.tanfovy), HW=(H, W),
                     gt_depth=gt) for sc in scenes]

    a, ca = model(), cams()
    la, _ = slam.render_batch_fused(ca, a, None, bg, lambda o, k: slam.l1_loss(o["render"], o["depth"], obs_c[k], obs_d[k], 1.0, 0.5))
    b, cb = model(), cams()
    lb, out = slam.render_batch_fused(cb, b, None, bg, None,
                                      batch_loss_fn=lambda o: slam.l1_loss(o["render"], o["depth"], obs_c, obs_d, V * 1.0, V * 0.5))
    torch.cuda.synchronize()
    assert len(lb) == 1 and abs(float(lb[0]) - float(torch.stack(la).sum())) <= 2e-6 * abs(float(lb[0]))
    for name in names:
        ga, gb = getattr(a, name).grad.cpu().numpy(), getattr(b, name).grad.cpu().numpy()
        assert np.abs(ga - gb).max() <= 2e-5 * np.abs(ga).max(), name
    for k in range(V):
        ga, gb = ca[k]["viewmatrix"].grad.cpu().numpy(), cb[k]["viewmatrix"].grad.cpu().numpy()
        assert np.abs(ga - gb).max() <= 2e-5 * np.abs(ga).max(), k


@pytest.mark.gpu
def test_render_views_keeps_the_camera_tensors_of_unchanged_keyframes():
    """slam.render_views derives projmatrices / campos / the depth stack from the keyframes' poses once and keeps them while
    the pose and depth tensors are the same objects at the same version; an in-place pose update (an optimiser step) or a new
    tensor must be seen.  Every call is compared with the one-view path after ITS cache (same rule, single poses) was emptied."""
    dev = torch.device("cuda:0")
    W, H = 160, 120
    scenes = [make_scene(5000, W, H, 3, view_index=k) for k in range(3)]
    s = scenes[0]
    bg, gt = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
    m = Model(s, dev)
    cams = [dict(viewmatrix=torch.from_numpy(sc.view).to(dev), fov=(sc.tanfovx, sc.tanfovy), HW=(H, W), gt_depth=gt) for sc in scenes]

    def check(what):
        with torch.no_grad():
            out = slam.render_views(cams, m, None, bg)
            one = lambda c: slam.render(None, m, None, bg, viewmatrix=c["viewmatrix"], fov=c["fov"], HW=c["HW"],  # noqa: E731
                                        gt_depth=c["gt_depth"])
            kept_one = [one(c) for c in cams]      # render() keeps a fixed pose's tensors by the same rule: whatever it holds now
            slam._VIEW_CACHE.clear()
            fresh = [one(c) for c in cams]         # ... and these calls derive them afresh (and fill the cache for the next check)
            for k, (o, ko) in enumerate(zip(fresh, kept_one)):
                assert torch.equal(o["render"], out["render"][k]) and torch.equal(o["depth"], out["depth"][k]), (what, k)
                assert torch.equal(o["render"], ko["render"]) and torch.equal(o["depth"], ko["depth"]), (what, k)
        return out

    slam._VIEWS_CACHE.clear()
    check("first call")
    assert len(slam._VIEWS_CACHE) == 1
    kept = next(iter(slam._VIEWS_CACHE.values()))
    check("second call")
    assert len(slam._VIEWS_CACHE) == 1 and next(iter(slam._VIEWS_CACHE.values())) is kept   # served from the kept tensors
    with torch.no_grad():                                   # an optimiser step on pose 1: same tensor, next version
        cams[1]["viewmatrix"][3, 0] += 0.05
    check("pose updated in place")
    assert next(reversed(slam._VIEWS_CACHE.values())) is not kept
    cams[2]["viewmatrix"] = torch.from_numpy(scenes[0].view).to(dev)      # another tensor
    check("pose replaced")
    cams[0]["gt_depth"] = gt * 1.5                                          # another depth image
    out = check("depth replaced")
    assert len(slam._VIEWS_CACHE) <= 4
    # the per-view mapping of render_batch_fused slices on demand
    v = slam._ViewOf(out, 1)
    assert set(v) == set(out) and torch.equal(v["render"], out["render"][1]) and v["viewspace_points"] is out["viewspace_points"]
    assert len(v._got) == 2 and v.get("nothing") is None and "radii" in v


@pytest.mark.gpu
def test_tracking_iteration_replayed_from_a_hipgraph(monkeypatch):
    """The whole tracking iteration -- pose -> camera matrices -> render -> loss -> backward -> Adam step -- recorded once
    (dgr_amd.multiview.CapturedStep) and replayed: the pose converges as in the eager loop (examples/tracking.py)."""
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")
    from dgr_amd.multiview import CapturedStep
    dev = torch.device("cuda:0")
    W, H = 256, 192
    s = make_scene(20000, W, H, 3)
    pc = Model(s, dev)
    tanfovx, tanfovy, Rm, t_true, *_ = camera(W, H, 0.05)
    bg, gt_depth = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
    kw = dict(fov=(tanfovx, tanfovy), HW=(H, W), gt_depth=gt_depth, track_off=False, map_off=True)

    def pose(q, t):
        return slam.camera_tensors(slam.w2c_from_quat_trans(q, t), tanfovx, tanfovy)[0]

    q_true = torch.tensor(rot_to_quat(Rm), dtype=torch.float32, device=dev)
    t_true = torch.tensor(t_true, dtype=torch.float32, device=dev)
    with torch.no_grad():
        obs = slam.render(None, pc, None, bg, viewmatrix=pose(q_true, t_true), **kw)
    obs_c, obs_d = obs["render"].detach(), obs["depth"].detach()
    q = (q_true + torch.tensor([0.0, 0.004, -0.006, 0.003], device=dev)).requires_grad_()
    t = (t_true + torch.tensor([0.012, -0.009, 0.015], device=dev)).requires_grad_()
    opt = torch.optim.Adam([{"params": [q], "lr": 5e-4}, {"params": [t], "lr": 1.5e-3}], capturable=True)

    def iteration():
        opt.zero_grad(set_to_none=True)
        out = slam.render(None, pc, None, bg, viewmatrix=pose(q, t), **kw)
        loss = (out["render"] - obs_c).abs().mean() + 0.5 * (out["depth"] - obs_d).abs().mean()
        loss.backward()
        opt.step()
        return loss.detach()

    def errors():
        with torch.no_grad():
            return float((q / q.norm() - q_true).norm()), float((t - t_true).norm())

    e0 = errors()
    step = CapturedStep(iteration, warmup=3)
    first = float(step.replay())
    for _ in range(150):
        last = step.replay()
    step.check()
    e1 = errors()
    assert float(last) < 0.3 * first
    assert e1[0] < 0.3 * e0[0] and e1[1] < 0.3 * e0[1], (e0, e1)


@pytest.mark.gpu
def test_fused_mapping_loop_follows_the_per_view_loop(monkeypatch):
    """examples/mapping.py --fused (the keyframe batch through the batched entry points), eager and replayed from a hipGraph:
    same loss trajectory and statistics as one rasterizer call per keyframe."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from mapping import mapping_loop
    iters, keyframes = 40, 3
    dev = torch.device("cuda:0")
    (a0, a1), pa, _ = mapping_loop(dev, 8000, 192, 144, keyframes, iters, views_in_flight=1)
    (b0, b1), pb, _ = mapping_loop(dev, 8000, 192, 144, keyframes, iters, fused=True)
    assert abs(a0 - b0) <= 1e-5 * a0 and abs(a1 - b1) <= 2e-3 * a1, ((a0, a1), (b0, b1))
    # (40 Adam steps amplify the rounding noise of the blend backward's float atomics -- two runs of the SAME loop differ by as
    #  much: a Gaussian's integer radius ceil(3 sigma) may then land one pixel apart, and with it whether a keyframe saw it)
    dr = (pa.max_radii2D - pb.max_radii2D).abs()
    assert float(dr.max()) <= 1.0 and int((dr > 0).sum()) <= 4, (float(dr.max()), int((dr > 0).sum()))
    assert int((pa.denom != pb.denom).sum()) <= 4
    assert float((pa.xyz_gradient_accum - pb.xyz_gradient_accum).abs().max()) <= 2e-2 * float(pa.xyz_gradient_accum.abs().max())
    monkeypatch.setenv("DGR_SYNC_MODE", "lazy")
    (c0, c1), pc_, _ = mapping_loop(dev, 8000, 192, 144, keyframes, iters, fused=True, graph=True)
    assert abs(c1 - a1) <= 2e-3 * a1 and float(pc_.denom.max()) == iters * keyframes


@pytest.mark.gpu
def test_mapping_loop_reduces_the_loss():
    """examples/mapping.py: keyframe batch on several streams + densification statistics + fused sparse Adam refine a
    degraded map; the statistics count each (Gaussian, view) a keyframe saw."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from mapping import mapping_loop
    iters, keyframes = 40, 3
    (l0, l1), pc, _ = mapping_loop(torch.device("cuda:0"), 8000, 192, 144, keyframes, iters)
    assert l1 < 0.6 * l0, (l0, l1)
    denom = pc.denom.cpu().numpy()
    assert denom.max() == iters * keyframes and (denom > 0).mean() > 0.5
    assert float(pc.xyz_gradient_accum.sum()) > 0 and float(pc.max_radii2D.max()) >= 1
    assert np.all((pc.xyz_gradient_accum.cpu().numpy() > 0) <= (denom > 0))


@pytest.mark.gpu
def test_fused_pose_and_loss_match_the_torch_ops():
    """slam.pose_to_camera and slam.l1_loss (one launch each way) against the differentiable torch formulation."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    tanfovx, tanfovy = 0.6, 0.45
    for trial in range(4):
        q0 = (torch.randn(4, generator=g) * (0.3 if trial else 1.0) + torch.tensor([1.0, 0, 0, 0])).to(dev)
        t0 = torch.randn(3, generator=g).to(dev)
        qa, ta = q0.clone().requires_grad_(), t0.clone().requires_grad_()
        qb, tb = q0.clone().requires_grad_(), t0.clone().requires_grad_()
        va, pa, psa, ca = slam.pose_to_camera(qa, ta, tanfovx, tanfovy)
        vb, pb, psb, cb = slam.camera_tensors(slam.w2c_from_quat_trans(qb, tb), tanfovx, tanfovy)
        for x, y in ((va, vb), (pa, pb), (psa, psb), (ca, cb)):
            np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=2e-6)
        assert va.requires_grad and not pa.requires_grad and not ca.requires_grad
        w = torch.randn((4, 4), generator=g).to(dev)
        (va * w).sum().backward()
        (vb * w).sum().backward()
        np.testing.assert_allclose(qa.grad.cpu().numpy(), qb.grad.cpu().numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(ta.grad.cpu().numpy(), tb.grad.cpu().numpy(), rtol=1e-6, atol=1e-7)
    H, W = 37, 53
    c = torch.rand((3, H, W), generator=g).to(dev).requires_grad_()
    d = torch.rand((1, H, W), generator=g).to(dev).requires_grad_()
    co, do = torch.rand((3, H, W), generator=g).to(dev), torch.rand((1, H, W), generator=g).to(dev)
    co[0, 0, :5] = c.detach()[0, 0, :5]  # exact ties: sign(0) = 0 on both sides
    la = slam.l1_loss(c, d, co, do, 1.0, 0.5)
    (3.0 * la).backward()
    ga_c, ga_d = c.grad.clone(), d.grad.clone()
    c.grad = d.grad = None
    lb = (c - co).abs().mean() + 0.5 * (d - do).abs().mean()
    (3.0 * lb).backward()
    assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb))
    np.testing.assert_allclose(ga_c.cpu().numpy(), c.grad.cpu().numpy(), rtol=1e-6, atol=0)
    np.testing.assert_allclose(ga_d.cpu().numpy(), d.grad.cpu().numpy(), rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_pose_only_backward_equals_the_full_backward():
    """When no Gaussian input requires a gradient (tracking) the backward skips the dense per-Gaussian rows; the pose
    gradient is the one the mapping-mode backward returns."""
    from dgr_amd import light
    dev = torch.device("cuda:0")
    W, H = 200, 150
    s = make_scene(15000, W, H, 3)
    pc = Model(s, dev)
    bg, gt = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
    gC, gD = torch.from_numpy(s.gC).to(dev), torch.from_numpy(s.gD).to(dev)
    views = []
    for need in (False, True):
        xyz = pc.get_xyz.clone().requires_grad_(need)
        vm = torch.from_numpy(s.view).to(dev).requires_grad_()
        st = light.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=vm.detach(),
            projmatrix=torch.from_numpy(s.proj).to(dev), sh_degree=3, campos=torch.from_numpy(s.campos).to(dev), prefiltered=False,
            debug=False, perspec_matrix=torch.from_numpy(s.persp).to(dev), track_off=False, map_off=False)
        out = light.GaussianRasterizer(st)(means3D=xyz, means2D=torch.zeros_like(xyz), opacities=pc.get_opacity,
                                           shs=pc.get_features, scales=pc.get_scaling, rotations=pc.get_rotation, viewmatrix=vm,
                                           gt_depth=gt)
        torch.autograd.backward([out[0], out[2]], [(gC * (W * H) ** 0.5).reshape(out[0].shape),
                                                   (gD * (W * H) ** 0.5).reshape(out[2].shape)])
        views.append(vm.grad.cpu().numpy())
        assert (xyz.grad is not None) == need
    assert np.abs(views[0]).max() > 0
    np.testing.assert_allclose(views[0], views[1], rtol=0, atol=2e-6 * np.abs(views[1]).max())
