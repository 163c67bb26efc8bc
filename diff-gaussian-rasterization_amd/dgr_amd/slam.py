"""The caller's side of the rasterizer: CG-SLAM's `render()` and the camera tensors a tracking or
mapping step feeds it.

The reference repository documents this call but does not contain it (README.md:33-47, 71-96: `render(viewpoint_cam,
gaussians, pipe, background, viewmatrix=w2cT, fov=(tanfovx, tanfovy), HW=(H, W), gt_depth=..., track_off=...,
map_off=...)` returning a dict); it lives in CG-SLAM and follows the 3DGS `gaussian_renderer.render`.  This module
restates it over `dgr_amd.light` / `dgr_amd.full`, duck-typed on the 3DGS `GaussianModel` accessors, and adds the
differentiable pose -> matrices helpers, so that a whole tracking iteration (pose parameters -> viewmatrix -> render ->
loss -> backward -> pose update) can run without leaving the GPU.

Matrix convention (cuda_rasterizer/auxiliary.h:58-77 reads 16 floats column-major): the rasterizer takes W2C^T,
(Proj W2C)^T and Proj^T.  The analytic pose gradient is returned w.r.t. the `viewmatrix` tensor only; `projmatrix` and
`campos` are recomputed from the same pose but enter as constants, exactly as in the reference (its backward adds their
dependence inside the kernels: L/cuda_rasterizer/backward.cu:633-651, 683-751).
"""
from collections.abc import Mapping as _Mapping

import torch

from . import full as _full
from . import light as _light


_proj_cache = {}


def projection_matrix(tanfovx, tanfovy, znear=0.01, zfar=100.0, device=None, dtype=torch.float32):
    """Proj (row-major math, z forward, as 3DGS `getProjectionMatrix` with symmetric frustum).  Cached per argument set
    (treat the result as read-only): building it copies host scalars to the device, which a hipGraph capture forbids,
    so a captured step finds the matrix its eager warm-up made."""
    key = (float(tanfovx), float(tanfovy), float(znear), float(zfar), str(device), dtype)
    P = _proj_cache.get(key)
    if P is None:
        P = torch.zeros((4, 4), dtype=dtype)
        P[0, 0] = 1.0 / tanfovx
        P[1, 1] = 1.0 / tanfovy
        P[2, 2] = zfar / (zfar - znear)
        P[2, 3] = -(zfar * znear) / (zfar - znear)
        P[3, 2] = 1.0
        P = _proj_cache[key] = P.to(device) if device is not None else P
    return P


_const_cache = {}


def _constants(device, dtype):
    """(identity, Levi-Civita tensor, bottom row) on `device`, built once (host -> device copies are not capturable)."""
    key = (str(device), dtype)
    c = _const_cache.get(key)
    if c is None:
        eps = torch.zeros((3, 3, 3), dtype=dtype)
        for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            eps[i, j, k], eps[i, k, j] = 1.0, -1.0
        c = _const_cache[key] = (torch.eye(3, dtype=dtype).to(device), eps.to(device),
                                 torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=dtype).to(device))
    return c


def quat_to_rotmat(q):
    """Unit quaternion (r, x, y, z) -> 3x3 rotation, differentiable (q is normalised here).
    R = (r^2 - |v|^2) I + 2 v v^T + 2 r [v]x in a dozen tensor operations (a tracking loop is launch-bound)."""
    eye, eps, _ = _constants(q.device, q.dtype)
    q = q / q.norm()
    r, v = q[0], q[1:]
    cross = -(eps * v).sum(-1)  # [v]x: cross[i][j] = -eps[i][j][k] v[k]
    return (r * r - (v * v).sum()) * eye + 2.0 * v.unsqueeze(1) * v.unsqueeze(0) + (2.0 * r) * cross


def w2c_from_quat_trans(q, t):
    """World-to-camera 4x4 from a quaternion and a translation (both differentiable leaves of a tracking step)."""
    top = torch.cat([quat_to_rotmat(q), t.reshape(3, 1)], dim=1)
    return torch.cat([top, _constants(q.device, q.dtype)[2]], dim=0)


def _small_matmul(a, b):
    """a @ b for 3x3 / 4x4 operands as one broadcast multiply and one sum: a BLAS call for sixteen numbers costs more
    host time on ROCm (≈1.5 ms: handle and heuristics) than the whole render."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


def camera_tensors(w2c, tanfovx, tanfovy, znear=0.01, zfar=100.0):
    """(viewmatrix, projmatrix, perspec_matrix, campos) for the rasterizer from a world-to-camera matrix.
    `viewmatrix` stays attached to `w2c`'s graph; the other three are detached (see the module docstring)."""
    P = projection_matrix(tanfovx, tanfovy, znear, zfar, device=w2c.device, dtype=w2c.dtype)
    viewmatrix = w2c.transpose(0, 1).contiguous()
    with torch.no_grad():
        perspec = P.transpose(0, 1).contiguous()
        projmatrix = _small_matmul(w2c.transpose(0, 1), P.transpose(0, 1)).contiguous()
        campos = (-(w2c[:3, :3] * w2c[:3, 3:4]).sum(0)).contiguous()  # -(R^T t)
    return viewmatrix, projmatrix, perspec, campos


class _PoseToCamera(torch.autograd.Function):
    """(quaternion, translation) -> (viewmatrix, projmatrix, campos) in one launch, dL/dviewmatrix -> (dL/dq, dL/dt) in
    another (C ABI: dgr_pose_forward / dgr_pose_backward).  Same function as
    `camera_tensors(w2c_from_quat_trans(q, t), ...)`: only `viewmatrix` carries a gradient."""

    @staticmethod
    def forward(ctx, q, t, perspec):
        from . import _capi
        lib = _capi.load()
        q, t, perspec = q.contiguous(), t.contiguous(), perspec.contiguous()
        out = torch.empty((35,), dtype=torch.float32, device=q.device)
        view, proj, campos = out[:16].view(4, 4), out[16:32].view(4, 4), out[32:35]
        if lib.dgr_pose_forward(_capi.stream_handle(), q.data_ptr(), t.data_ptr(), perspec.data_ptr(), view.data_ptr(),
                                proj.data_ptr(), campos.data_ptr()):
            raise RuntimeError(_capi.last_error())
        ctx.save_for_backward(q)
        ctx.mark_non_differentiable(proj, campos)
        return view, proj, campos

    @staticmethod
    def backward(ctx, dview, _dproj, _dcampos):
        from . import _capi
        lib = _capi.load()
        (q,) = ctx.saved_tensors
        grads = torch.empty((7,), dtype=torch.float32, device=q.device)
        dview = dview.contiguous()
        if lib.dgr_pose_backward(_capi.stream_handle(), q.data_ptr(), dview.data_ptr(), grads.data_ptr(),
                                 grads[4:].data_ptr()):
            raise RuntimeError(_capi.last_error())
        return grads[:4], grads[4:], None


# ---- tensors kept between calls (the caches below) are made by kernels of whichever stream was current at their first use.  A
# later call on ANOTHER stream (dgr_amd.multiview.ViewStreams: a stream per view) must not read them before those kernels have
# finished: every cache entry carries a stamp -- the creating stream and an event recorded behind the kernels that wrote the
# tensors -- and a consumer on a different stream waits for that event once (per stream).  While a hipGraph is being recorded
# nothing is cached.  The keys are (id, _version) of the source tensors: an in-place `viewmatrix.data = ...` swap changes
# neither and is NOT supported for cached poses -- assign a new tensor, or update it in place with copy_().
class _Stamp:
    __slots__ = ("stream", "event", "seen")

    def __init__(self, device):
        st = torch.cuda.current_stream(device)
        self.stream = int(st.cuda_stream)
        self.event = torch.cuda.Event()
        self.event.record(st)
        self.seen = {self.stream}

    def order(self, device):
        # (the raw handle first: building a Stream object per call costs more than the tensors the cache saves)
        h = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
        if h not in self.seen:
            torch.cuda.current_stream(device).wait_event(self.event)
            if len(self.seen) < 64:
                self.seen.add(h)


def _stamp(device):
    dev = torch.device(device)
    return _Stamp(dev) if dev.type == "cuda" else None


_PERSPEC = {}  # (tanfovx, tanfovy, znear, zfar, device) -> (Proj^T, stamp): a constant of the sensor, built once


def _perspec_cached(tanfovx, tanfovy, znear, zfar, device):
    key = (float(tanfovx), float(tanfovy), float(znear), float(zfar), device)
    hit = _PERSPEC.get(key)
    if hit is None:
        if len(_PERSPEC) > 32:
            _PERSPEC.clear()
        m = projection_matrix(tanfovx, tanfovy, znear, zfar, device=device).transpose(0, 1).contiguous()
        if m.is_cuda and torch.cuda.is_current_stream_capturing():
            return m
        hit = _PERSPEC[key] = (m, _stamp(m.device))
    if hit[1] is not None:
        hit[1].order(hit[0].device)
    return hit[0]


def pose_to_camera(q, t, tanfovx, tanfovy, znear=0.01, zfar=100.0):
    """(viewmatrix, projmatrix, perspec_matrix, campos) from a quaternion (r, x, y, z) and a translation, float32 GPU
    tensors: the fused form of `camera_tensors(w2c_from_quat_trans(q, t), tanfovx, tanfovy)` (2 launches for forward +
    backward instead of ~40 elementwise torch kernels)."""
    perspec = _perspec_cached(tanfovx, tanfovy, znear, zfar, q.device)
    view, proj, campos = _PoseToCamera.apply(q, t, perspec)
    return view, proj, perspec, campos


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, depth, color_obs, depth_obs, w_color, w_depth):
        from . import _capi
        lib = _capi.load()
        color, depth, color_obs, depth_obs = (x.contiguous() for x in (color, depth, color_obs, depth_obs))
        buf = torch.empty((lib.dgr_l1_loss_scratch_floats() + 1,), dtype=torch.float32, device=color.device)
        if lib.dgr_l1_loss_forward(_capi.stream_handle(), color.numel(), color.data_ptr(), color_obs.data_ptr(), depth.numel(),
                                   depth.data_ptr(), depth_obs.data_ptr(), float(w_color), float(w_depth), buf[1:].data_ptr(),
                                   buf.data_ptr()):
            raise RuntimeError(_capi.last_error())
        ctx.save_for_backward(color, depth, color_obs, depth_obs)
        ctx.weights = (float(w_color), float(w_depth))
        return buf[0]

    @staticmethod
    def backward(ctx, upstream):
        from . import _capi
        lib = _capi.load()
        color, depth, color_obs, depth_obs = ctx.saved_tensors
        dc, dd = torch.empty_like(color), torch.empty_like(depth)
        upstream = upstream.contiguous()
        if lib.dgr_l1_loss_backward(_capi.stream_handle(), color.numel(), color.data_ptr(), color_obs.data_ptr(), depth.numel(),
                                    depth.data_ptr(), depth_obs.data_ptr(), ctx.weights[0], ctx.weights[1],
                                    upstream.data_ptr(), dc.data_ptr(), dd.data_ptr()):
            raise RuntimeError(_capi.last_error())
        return dc, dd, None, None, None, None


def l1_loss(color, depth, color_obs, depth_obs, w_color=1.0, w_depth=0.5):
    """w_color * mean|color - color_obs| + w_depth * mean|depth - depth_obs| as one fused reduction, with both gradient
    images written by one launch in the backward (float32 GPU tensors; the observations carry no gradient)."""
    return _L1Loss.apply(color, depth, color_obs, depth_obs, w_color, w_depth)


def _get(obj, name, default=None):
    v = getattr(obj, name, default)
    return v() if callable(v) and not isinstance(v, torch.Tensor) else v


def _matmul_fixed_order(a, b):
    """a @ b for [..., n, 4] @ [4, m] with the four products of every entry added in index order by elementwise kernels
    (see _campos: the result must not depend on whether `a` is one matrix or a slice of a stack).  Four launches."""
    r = a[..., :, 0:1] * b[0]
    for k in (1, 2, 3):
        r = torch.addcmul(r, a[..., :, k:k + 1], b[k])
    return r


def _campos(vm):
    """Camera centre -(R^T t) from viewmatrix = W2C^T ([4,4] or [V,4,4]), with the three products added in a fixed order
    by elementwise kernels: a `sum()` reduction picks its summation order from the tensor's layout and alignment, so the
    same pose gave cameras one ulp apart as a [4,4] tensor and as a slice of a [V,4,4] one.  Three launches."""
    nt = -vm[..., 3, 0:3]
    r = vm[..., :3, 0] * nt[..., 0:1]
    r = torch.addcmul(r, vm[..., :3, 1], nt[..., 1:2])
    return torch.addcmul(r, vm[..., :3, 2], nt[..., 2:3]).contiguous()


_VIEWS_CACHE = {}   # render_views: camera tensors of the last few keyframe batches (see there)
_VIEW_CACHE = {}    # render: the same for single fixed poses
_ZERO_POINTS = {}  # (device, P) -> a [P, 3] zero tensor for calls whose screen-space gradient nobody reads (tracking)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, viewmatrix=None, fov=None,
           HW=None, gt_depth=None, track_off=False, map_off=False, variant="light", pose_tensors=None):
    """CG-SLAM's `render()` (reference README.md:33,71).

    `pc`: anything with the 3DGS GaussianModel accessors `get_xyz`, `get_opacity`, `get_scaling`, `get_rotation`,
    `get_features` ([P, M, 3]) and `active_sh_degree`.  `viewmatrix` is W2C^T (differentiable for tracking); `fov` the
    two half-angle tangents; `HW` = (H, W).  `viewpoint_camera` may carry `projection_matrix` (Proj^T), `znear`, `zfar`;
    without it a symmetric frustum with znear 0.01 / zfar 100 is used.  `pipe.debug` is honoured.
    `pose_tensors` = what `pose_to_camera(q, t, ...)` returned for this `viewmatrix` -- (viewmatrix, projmatrix, perspec_matrix,
    campos): the projection, the full projection and the camera centre are then taken from there instead of being formed again
    from `viewmatrix` with a dozen small torch kernels (a 640x480 tracking iteration is bound by exactly those).
    Returns the reference's dict (light: render, depth, depth_median, opacity_map, depth_var, gau_uncertainty,
    num_related_pixels; full: render, depth, opacity_map) plus the 3DGS bookkeeping entries viewspace_points,
    visibility_filter, radii."""
    if viewmatrix is None or fov is None or HW is None:
        raise ValueError("render() needs viewmatrix=W2C^T, fov=(tanfovx, tanfovy) and HW=(H, W)")
    mod = _light if variant == "light" else _full
    H, W = int(HW[0]), int(HW[1])
    tanfovx, tanfovy = float(fov[0]), float(fov[1])
    dev = viewmatrix.device
    znear = float(_get(viewpoint_camera, "znear", 0.01)) if viewpoint_camera is not None else 0.01
    zfar = float(_get(viewpoint_camera, "zfar", 100.0)) if viewpoint_camera is not None else 100.0
    if pose_tensors is not None:
        vm = viewmatrix.detach()
        projmatrix, perspec, campos = pose_tensors[1].detach(), pose_tensors[2], pose_tensors[3].detach()
    else:
        perspec_src = _get(viewpoint_camera, "projection_matrix") if viewpoint_camera is not None else None
        # a fixed keyframe pose (mapping) is rendered again and again: its derived tensors are kept while `viewmatrix` is the
        # same object at the same version (render_views does the same for a batch); a pose under optimisation is a new
        # tensor every iteration and is not kept; never while a hipGraph is being recorded (a replay re-derives them)
        keep = (not viewmatrix.requires_grad and viewmatrix.is_cuda and not torch.cuda.is_current_stream_capturing())
        key = (id(viewmatrix), viewmatrix._version, id(perspec_src), getattr(perspec_src, "_version", 0), tanfovx, tanfovy,
               znear, zfar) if keep else None
        hit = _VIEW_CACHE.get(key) if keep else None
        if hit is None:
            with torch.no_grad():
                perspec = perspec_src
                if perspec is None:
                    perspec = _perspec_cached(tanfovx, tanfovy, znear, zfar, dev)
                perspec = perspec.to(dev, torch.float32)
                vm = viewmatrix.detach()
                projmatrix = _matmul_fixed_order(vm, perspec).contiguous()
                campos = _campos(vm)
            # (the sources too: ids are unique among live objects; the stamp: see _Stamp)
            hit = (perspec, vm, projmatrix, campos, viewmatrix, perspec_src, _stamp(dev) if keep else None)
            if keep:
                if len(_VIEW_CACHE) >= 16:
                    _VIEW_CACHE.pop(next(iter(_VIEW_CACHE)))
                _VIEW_CACHE[key] = hit
        elif hit[6] is not None:
            hit[6].order(dev)
        perspec, vm, projmatrix, campos = hit[:4]

    means3D = pc.get_xyz
    # 3DGS keeps a zero tensor whose .grad receives the screen-space gradient (densification statistics).  A tracking
    # step (map_off, or a map that carries no gradients) has no use for it, and without it the backward can skip every
    # per-Gaussian gradient row (dgr_amd.light: need_gaussian_grads).
    shs_or_colors = override_color if override_color is not None else pc.get_features
    opacity, scaling, rotation = pc.get_opacity, pc.get_scaling, pc.get_rotation
    mapping = (variant != "light" or not map_off) and any(
        t.requires_grad for t in (means3D, shs_or_colors, opacity, scaling, rotation))
    if mapping:
        screenspace_points = torch.zeros_like(means3D, requires_grad=True)
    else:  # never read, never written: one shared zero tensor per size instead of a fill per call
        key = (means3D.device, means3D.shape[0])
        zp = _ZERO_POINTS.get(key)
        if zp is None:
            if len(_ZERO_POINTS) > 16:
                _ZERO_POINTS.clear()
            z = torch.zeros_like(means3D)
            zp = (z, _stamp(z.device))
            if not (z.is_cuda and torch.cuda.is_current_stream_capturing()):
                _ZERO_POINTS[key] = zp
        elif zp[1] is not None:
            zp[1].order(zp[0].device)
        screenspace_points = zp[0]
    debug = bool(getattr(pipe, "debug", False)) if pipe is not None else False
    common = dict(image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
                  scale_modifier=scaling_modifier, viewmatrix=vm, projmatrix=projmatrix,
                  sh_degree=int(pc.active_sh_degree), campos=campos, prefiltered=False)
    if variant == "light":
        settings = mod.GaussianRasterizationSettings(**common, debug=debug, perspec_matrix=perspec, track_off=track_off,
                                                     map_off=map_off)
    else:
        settings = mod.GaussianRasterizationSettings(**common, perspec_matrix=perspec)
    rasterizer = mod.GaussianRasterizer(raster_settings=settings)
    shs, colors = (None, override_color) if override_color is not None else (shs_or_colors, None)
    out = rasterizer(means3D=means3D, means2D=screenspace_points, opacities=opacity, shs=shs,
                     colors_precomp=colors, scales=scaling, rotations=rotation, cov3D_precomp=None,
                     viewmatrix=viewmatrix, gt_depth=gt_depth)
    if variant == "light":
        color, radii, depth, depth_median, depth_var, opacity_map, gau_uncertainty, gau_related_pixels = out
        res = {"render": color, "depth": depth, "depth_median": depth_median, "opacity_map": opacity_map,
               "depth_var": depth_var, "gau_uncertainty": gau_uncertainty, "num_related_pixels": gau_related_pixels}
    else:
        color, radii, depth, uncertainty = out
        res = {"render": color, "depth": depth, "opacity_map": uncertainty}
    res.update(viewspace_points=screenspace_points, visibility_filter=radii > 0, radii=radii)
    return res


def render_views(cameras, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, track_off=False, map_off=False):
    """`render()` for the V cameras of a keyframe batch in ONE call of the batched entry points (`dgr_amd.batch`, SURVEY.md
    s8(f) item 2; light variant): the cameras share `fov` and `HW` (one sensor, V poses), every per-view entry of `render()`'s
    dict comes back with a leading view dimension, and one backward through it yields the Gaussians' gradients already summed
    over the views, the pose gradient per `viewmatrix` and `viewspace_points.grad` ([V,P,3]) per view.
    `cameras`: sequence of dicts with `viewmatrix` (W2C^T), `fov`, `HW` and optionally `gt_depth`, `viewpoint_camera`."""
    from . import batch as _batch
    cameras = list(cameras)
    if not 1 <= len(cameras) <= _batch.MAX_VIEWS:
        raise ValueError(f"1 .. {_batch.MAX_VIEWS} cameras per call")
    c0 = cameras[0]
    H, W = int(c0["HW"][0]), int(c0["HW"][1])
    tanfovx, tanfovy = float(c0["fov"][0]), float(c0["fov"][1])
    for c in cameras[1:]:
        if (int(c["HW"][0]), int(c["HW"][1])) != (H, W) or (float(c["fov"][0]), float(c["fov"][1])) != (tanfovx, tanfovy):
            raise ValueError("the cameras of a batch share fov and HW")
    cam0 = c0.get("viewpoint_camera")
    znear = float(_get(cam0, "znear", 0.01)) if cam0 is not None else 0.01
    zfar = float(_get(cam0, "zfar", 100.0)) if cam0 is not None else 100.0
    gts = [c.get("gt_depth") for c in cameras]
    if any(g is None for g in gts):
        raise ValueError("the light variant needs gt_depth for every camera")
    vms = [c["viewmatrix"] for c in cameras]
    perspec_src = _get(cam0, "projection_matrix") if cam0 is not None else None
    # A mapping loop renders the same keyframes iteration after iteration: the camera tensors derived from the poses (a dozen
    # small launches) are kept while the tensors they were made from are the same objects at the same version -- an optimiser
    # step on a pose, or a new depth image, changes the key.
    key = (H, W, tanfovx, tanfovy, znear, zfar, id(perspec_src), getattr(perspec_src, "_version", 0),
           tuple((id(v), v._version, id(g), g._version) for v, g in zip(vms, gts)))
    # (never while a hipGraph is being recorded: a replay must re-derive the cameras from whatever the poses hold then)
    capturing = vms[0].is_cuda and torch.cuda.is_current_stream_capturing()
    hit = None if capturing else _VIEWS_CACHE.get(key)
    if hit is None:
        dev = vms[0].device
        with torch.no_grad():
            perspec = perspec_src
            if perspec is None:
                perspec = _perspec_cached(tanfovx, tanfovy, znear, zfar, dev)
            perspec = perspec.to(dev, torch.float32)
            vm = torch.stack([v.detach() for v in vms])
            # (the operations of render(), in a fixed order: the cameras -- hence the images -- are those of the one-view
            #  path bit for bit)
            projmatrices = _matmul_fixed_order(vm, perspec).contiguous()
            campos = _campos(vm)
            gt_depths = torch.stack([g.reshape(H, W) for g in gts])
        # (the sources are held too: an id() is only unique among live objects)
        hit = (perspec, vm, projmatrices, campos, gt_depths, list(vms), list(gts), perspec_src, None if capturing else _stamp(dev))
        if not capturing:
            if len(_VIEWS_CACHE) >= 4:
                _VIEWS_CACHE.pop(next(iter(_VIEWS_CACHE)))
            _VIEWS_CACHE[key] = hit
    elif hit[8] is not None:
        hit[8].order(hit[1].device)
    perspec, vm, projmatrices, campos, gt_depths = hit[:5]
    dev = vm.device
    # (differentiable where a pose is a leaf: every pose keeps its gradient)
    viewmatrices = torch.stack(vms) if any(v.requires_grad for v in vms) else vm
    means3D = pc.get_xyz
    shs_or_colors = override_color if override_color is not None else pc.get_features
    opacity, scaling, rotation = pc.get_opacity, pc.get_scaling, pc.get_rotation
    mapping = not map_off and any(t.requires_grad for t in (means3D, shs_or_colors, opacity, scaling, rotation))
    screenspace_points = torch.zeros((len(cameras),) + tuple(means3D.shape), dtype=means3D.dtype, device=dev, requires_grad=mapping)
    debug = bool(getattr(pipe, "debug", False)) if pipe is not None else False
    settings = _batch.BatchRasterizationSettings(
        image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrices=vm, projmatrices=projmatrices, sh_degree=int(pc.active_sh_degree), campos=campos, prefiltered=False,
        debug=debug, perspec_matrix=perspec, track_off=track_off, map_off=map_off)
    shs, colors = (None, override_color) if override_color is not None else (shs_or_colors, None)
    color, radii, depth, depth_median, depth_var, opacity_map, gau_uncertainty, gau_related_pixels = \
        _batch.GaussianRasterizerBatch(settings)(means3D, screenspace_points, opacity, shs=shs, colors_precomp=colors,
                                                 scales=scaling, rotations=rotation, viewmatrices=viewmatrices,
                                                 gt_depths=gt_depths)
    return {"render": color, "depth": depth, "depth_median": depth_median, "opacity_map": opacity_map, "depth_var": depth_var,
            "gau_uncertainty": gau_uncertainty, "num_related_pixels": gau_related_pixels,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


class _ViewOf(_Mapping):
    """View k of `render_views`' dict: entry n is `out[n][k]`, sliced when it is first asked for (a loss reads two or three
    of the ten; every slice is an autograd node and a launch-free but not cost-free call) -- except `viewspace_points`,
    which stays the whole [V,P,3] tensor: one backward fills every view's slice."""
    _NAMES = ("render", "depth", "depth_median", "opacity_map", "depth_var", "gau_uncertainty", "num_related_pixels",
              "visibility_filter", "radii", "viewspace_points")

    def __init__(self, out, k):
        self._out, self._k, self._got = out, k, {}

    def __getitem__(self, n):
        v = self._got.get(n)
        if v is None:
            if n not in self._NAMES:
                raise KeyError(n)
            v = self._got[n] = self._out[n] if n == "viewspace_points" else self._out[n][self._k]
        return v

    def __iter__(self):
        return iter(self._NAMES)

    def __len__(self):
        return len(self._NAMES)


def render_batch_fused(cameras, pc, pipe, bg_color, loss_fn, batch_loss_fn=None, **render_kwargs):
    """`render_batch` through ONE batched forward and ONE batched backward (`render_views`): `loss_fn(out_k, k)` sees the
    dict of view k (slices of the batched outputs), the losses are summed and back-propagated once.  Same gradients as
    `render_batch` -- the sum over the keyframes in the Gaussians' `.grad`, one pose gradient per `viewmatrix` -- without V - 1
    accumulation passes over the dense gradient rows and with the camera-independent per-Gaussian work done once.
    `batch_loss_fn(out)`, if given, replaces the V calls of `loss_fn`: it sees the batched dict and returns the SUM of the
    views' losses as one scalar (then the returned list holds that one value)."""
    out = render_views(cameras, pc, pipe, bg_color, **render_kwargs)
    if batch_loss_fn is not None:
        # ONE loss over the stacked outputs ([V,3,H,W], [V,1,H,W], ...): no per-view slice in the graph -- each is a node whose
        # backward zero-fills a tensor of the whole stack's size -- and one reduction instead of V
        total = batch_loss_fn(out)
        total.backward()
        return [total.detach()], out
    losses = [loss_fn(_ViewOf(out, k), k) for k in range(out["render"].size(0))]
    torch.stack(losses).sum().backward()
    return [l_.detach() for l_ in losses], out


def render_batch(cameras, pc, pipe, bg_color, loss_fn, views_in_flight=3, **render_kwargs):
    """One mapping step over a batch of keyframes: every camera is rendered, `loss_fn(out, k)`
    is evaluated on its output dict and back-propagated, each view's forward + loss + backward on its own HIP stream
    (`dgr_amd.multiview.ViewStreams`) so that the views overlap on the GPU; gradients accumulate in the `.grad` of the
    Gaussian parameters as usual.  Accumulation into a shared `.grad` from several streams needs an explicit order
    (PyTorch gives none): every view's backward waits for the end of the previous view (`views.before_backward()`), its
    forward and loss still overlap the previous view's backward.  `cameras`: sequence of dicts with `viewmatrix`
    (W2C^T), `fov`, `HW` and optionally `gt_depth`, `viewpoint_camera`.  Returns the list of detached loss values
    (device tensors); the caller's stream is ordered after all views on return."""
    from .multiview import ViewStreams
    cameras = list(cameras)
    if not cameras:
        return []
    views = ViewStreams(min(max(1, views_in_flight), len(cameras)), cameras[0]["viewmatrix"].device)

    def one_view(k, cam):  # (a function: no autograd graph of one view is kept alive into the next)
        out = render(cam.get("viewpoint_camera"), pc, pipe, bg_color, viewmatrix=cam["viewmatrix"], fov=cam["fov"],
                     HW=cam["HW"], gt_depth=cam.get("gt_depth"), **render_kwargs)
        loss = loss_fn(out, k)
        views.before_backward()
        loss.backward()
        return loss.detach()

    losses = []
    for k, cam in enumerate(cameras):
        with views.next():
            losses.append(one_view(k, cam))
    views.join()
    return losses
