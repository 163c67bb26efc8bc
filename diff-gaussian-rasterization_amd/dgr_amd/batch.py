"""Batched multi-view surface of the light variant (BASELINE configs 4 and 5): V cameras over ONE set
of Gaussians per call.

The reference renders a keyframe batch as V calls of `GaussianRasterizer.forward` + `.backward`
(L/diff_gaussian_rasterization/__init__.py:36-176) and lets autograd add the V dense gradient sets.  `rasterize_gaussians_batch`
keeps that function's argument meaning with a leading view dimension on everything that belongs to a camera and goes through
the C ABI's batched entry points (include/dgr_hip.h: dgr_light_forward_batch / dgr_light_backward_batch):

  * per view the outputs are bit-identical to the one-view surface (`dgr_amd.light`);
  * the gradients of the Gaussians come back SUMMED over the views -- what autograd's accumulation of V one-view backward
    passes yields (same operations in the same order in the per-Gaussian stage; two runs differ only by the order of the
    blend backward's float atomics, as two one-view runs do) -- formed in registers and written once; `means2D` (the
    screen-space points 3DGS reads its densification statistics from) and `viewmatrices` keep their per-view gradients;
  * they are views of one flat arena (`dgr_amd.light._grad_arena`), so `dgr_amd.multiview.GradientArena` finds the fused
    all-reduce span of a multi-GPU mapping step in them as it does for a one-view backward.

torch supplies device memory and the current stream; every compute call goes through the C ABI.  There is no CPU fallback.
"""
from typing import NamedTuple

import torch

from . import _capi
from . import light as _light

MAX_VIEWS = _capi.MAX_BATCH_VIEWS
_View, _ViewGrad = _capi.LightView, _capi.LightViewGrad
_lib = _capi.load


class BatchRasterizationSettings(NamedTuple):
    """`GaussianRasterizationSettings` (L/diff_gaussian_rasterization/__init__.py:180-195) for V cameras that share the
    image size, the field of view and the background: the three camera tensors carry a leading view dimension."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrices: torch.Tensor   # [V,4,4]
    projmatrices: torch.Tensor   # [V,4,4]
    sh_degree: int
    campos: torch.Tensor         # [V,3]
    prefiltered: bool
    debug: bool
    perspec_matrix: torch.Tensor  # [4,4] (one projection for the batch)
    track_off: bool
    map_off: bool


def _row(t, v):
    """device pointer of view v's slice of a contiguous [V, ...] tensor (NULL for None / empty)"""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr() + v * t.stride(0) * t.element_size()


def _ext():
    """the compiled torch extension (csrc/torch_ext.cpp: light_forward_batch / light_backward_batch -- allocation and
    marshalling in C++) when dgr_amd.light selected it, else None: the ctypes code below binds the same C ABI"""
    return _light._CompiledC.ext if _light._C is _light._CompiledC else None


def _forward_batch_compiled(ext, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices,
                            gt_depths, projmatrices, tanfovx, tanfovy, H, W, sh, degree, campos, prefiltered, key, V):
    cap = _light._capacity_cache.get(key, 0)
    lazy = _light._sync_mode() == "lazy" and cap > 0
    cap = (int(cap * 1.5) + 4096) if lazy else (int(cap * 1.25) + 4096 if cap else 4 * key[1] + 4096)
    capturing = torch.cuda.is_current_stream_capturing()
    while True:
        if lazy and not capturing:
            while len(_light._pending_status) > V:
                _light._check_oldest()  # status words of earlier calls have long completed: no stall
        out, tickets = ext.light_forward_batch(bg, means3D, colors, opacity, scales, rotations, float(scale_modifier),
                                               cov3D_precomp, viewmatrices, gt_depths, projmatrices, float(tanfovx),
                                               float(tanfovy), int(H), int(W), sh, int(degree), campos, bool(prefiltered), cap,
                                               bool(lazy))
        status = out[0]
        if key[1] == 0:
            rendered = [0] * V
            break
        if lazy or capturing:
            for t in tickets:
                _light._pending_status.append((t, key))
            if capturing:  # recorded into a hipGraph: nothing can be read back now (dgr_amd.light.check_captured_status)
                import weakref
                _light._captured_status.append(weakref.ref(status))
                _light._capture_keepalive.append(status)
            rendered = [_light._capacity_cache.get(key, 0)] * V
            break
        s = status.tolist()  # the one host wait of a strict batch
        if any(r[2] for r in s):
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
        rendered = [r[0] for r in s]
        _light._capacity_cache[key] = max(_light._capacity_cache.get(key, 0), max(rendered))
        if max(rendered) <= cap:
            break
        cap = int(max(rendered) * 1.1) + 4096  # overflow: those views' tile lists were left empty; run again
    return (rendered,) + tuple(out[1:])


def _forward_batch(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices, gt_depths,
                   projmatrices, tanfovx, tanfovy, H, W, sh, degree, campos, prefiltered):
    lib = _lib()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)")
    V = viewmatrices.size(0)
    if not 1 <= V <= MAX_VIEWS:
        raise RuntimeError(f"1 .. {MAX_VIEWS} views per batch")
    P = means3D.size(0)
    ext = _ext()
    if ext is not None:
        return _forward_batch_compiled(ext, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                       viewmatrices, gt_depths, projmatrices, tanfovx, tanfovy, H, W, sh, degree, campos,
                                       prefiltered, (dev.index, P, H, W), V)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    u8 = dict(dtype=torch.uint8, device=dev)
    c = _light._f32c
    means3D, bg, colors, opacity = c(means3D, dev), c(bg, dev), c(colors, dev), c(opacity, dev)
    scales, rotations, cov3D_precomp, sh = c(scales, dev), c(rotations, dev), c(cov3D_precomp, dev), c(sh, dev)
    viewmatrices, projmatrices, campos, gt_depths = c(viewmatrices, dev), c(projmatrices, dev), c(campos, dev), c(gt_depths, dev)
    M = sh.size(1) if sh.numel() != 0 else 0
    color = torch.empty((V, 3, H, W), **f32)
    depth, median, var, alpha = (torch.empty((V, 1, H, W), **f32) for _ in range(4))
    mk = torch.empty if P else torch.zeros
    radii = mk((V, P), **i32)
    unc = mk((V, P, 1), **f32)
    px = mk((V, P, 1), **i32)
    geom = torch.empty((V, max(lib.dgr_geometry_bytes(P), 1)), **u8)
    img = torch.empty((V, max(lib.dgr_image_bytes(W, H), 1)), **u8)
    status = torch.zeros((V, 4), **i32)
    st = _capi.stream_handle(dev.index)
    p = _capi.ptr
    key = (dev.index, P, H, W)
    cap = _light._capacity_cache.get(key, 0)
    lazy = _light._sync_mode() == "lazy" and cap > 0
    cap = (int(cap * 1.5) + 4096) if lazy else (int(cap * 1.25) + 4096 if cap else 4 * P + 4096)
    capturing = torch.cuda.is_current_stream_capturing()
    while True:
        binning = torch.empty((V, max(lib.dgr_binning_bytes(cap, W, H), 1)), **u8)
        views = (_View * V)()
        for v in range(V):
            w = views[v]
            w.geometry_buffer, w.binning_buffer, w.binning_capacity, w.image_buffer = _row(geom, v), _row(binning, v), cap, _row(img, v)
            w.status, w.viewmatrix, w.projmatrix, w.cam_pos = _row(status, v), _row(viewmatrices, v), _row(projmatrices, v), _row(campos, v)
            w.out_color, w.out_depth, w.out_median_depth, w.out_alpha = _row(color, v), _row(depth, v), _row(median, v), _row(alpha, v)
            w.gt_depth, w.out_depth_var = _row(gt_depths, v), _row(var, v)
            w.gau_uncertainty, w.gau_related_pixels, w.radii = _row(unc, v), _row(px, v), _row(radii, v)
        _light._check(lib.dgr_light_forward_batch(st, V, views, P, int(degree), M, p(bg), W, H, p(means3D), p(sh), p(colors),
                                                  p(opacity), p(scales), float(scale_modifier), p(rotations), p(cov3D_precomp),
                                                  float(tanfovx), float(tanfovy), int(bool(prefiltered))))
        if P == 0:
            rendered = [0] * V
            break
        if lazy or capturing:
            # no host synchronisation: the status words are looked at one or two calls later (dgr_amd.light.check_async_errors)
            for v in range(V):
                _light._post_status(status[v], key)
            rendered = [_light._capacity_cache.get(key, 0)] * V
            break
        s = status.tolist()  # the one host wait of a strict batch
        if any(r[2] for r in s):
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
        rendered = [r[0] for r in s]
        _light._capacity_cache[key] = max(_light._capacity_cache.get(key, 0), max(rendered))
        if max(rendered) <= cap:
            break
        cap = int(max(rendered) * 1.1) + 4096  # overflow: those views' tile lists were left empty; run again
    return rendered, color, depth, median, var, alpha, radii, geom, binning, img, unc, px


def _backward_batch(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices, projmatrices,
                    tanfovx, tanfovy, gC, gD, gM, gV, gt_depths, sh, degree, campos, geom, binning, img, alphas,
                    perspec_matrix, track_off, map_off, need_gaussian_grads, need_means2D, num_rendered=None):
    """`num_rendered`: per view, what the one-view backward takes as R (>= the view's instance count); read by the library only
    under deterministic_grads, where it sizes the views' row buffers."""
    V_ = viewmatrices.size(0)
    num_rendered = [int(r) for r in (num_rendered if num_rendered is not None else [0] * V_)]
    ext = _ext()
    if ext is not None:
        g = ext.light_backward_batch(bg, means3D, radii, colors, scales, rotations, float(scale_modifier), cov3D_precomp,
                                     viewmatrices, projmatrices, float(tanfovx), float(tanfovy), gC, gD, gM, gV, gt_depths, sh,
                                     int(degree), campos, geom, binning, img, alphas, perspec_matrix, bool(track_off),
                                     bool(map_off), bool(need_gaussian_grads), bool(need_means2D), num_rendered)
        return tuple(g)
    lib = _lib()
    dev = means3D.device
    V, P = viewmatrices.size(0), means3D.size(0)
    H, W = gC.size(2), gC.size(3)
    f32 = dict(dtype=torch.float32, device=dev)
    c = _light._f32c
    means3D, bg, colors = c(means3D, dev), c(bg, dev), c(colors, dev)
    scales, rotations, cov3D_precomp, sh = c(scales, dev), c(rotations, dev), c(cov3D_precomp, dev), c(sh, dev)
    viewmatrices, projmatrices, campos = c(viewmatrices, dev), c(projmatrices, dev), c(campos, dev)
    gt_depths, alphas, perspec_matrix = c(gt_depths, dev), c(alphas, dev), c(perspec_matrix, dev)
    gC, gD, gM, gV = c(gC, dev), c(gD, dev), c(gM, dev), c(gV, dev)
    M = sh.size(1) if sh.numel() != 0 else 0
    if need_gaussian_grads:
        seg = _light._grad_arena(P, M, f32)
        seg["means2D"].zero_()  # the arena's one-view slot: the batch returns means2D gradients per view, beside the arena
        d3, dsh, dop, dsc, drot, dcov, dcol = (seg[k] for k in ("means3D", "sh", "opacity", "scales", "rotations", "cov3D", "colors"))
        d2 = torch.empty((V, P, 3), **f32) if need_means2D else None
    else:
        d3 = dsh = dop = dsc = drot = dcov = dcol = d2 = None
        map_off = True  # nobody reads the per-Gaussian sums: the blend kernels form the three pose sums only
    dview = torch.empty((V, 4, 4), **f32)
    nscr = (max(lib.dgr_light_backward_scratch_bytes_r(P, W, H, max(num_rendered + [0])), 1) + 255) // 256 * 256
    scratch = torch.empty((V, nscr), dtype=torch.uint8, device=dev)
    views = (_ViewGrad * V)()
    pp = _capi.ptr(perspec_matrix)
    for v in range(V):
        w = views[v]
        w.geometry_buffer, w.binning_buffer, w.image_buffer = _row(geom, v), _row(binning, v), _row(img, v)
        w.viewmatrix, w.projmatrix, w.cam_pos, w.perspec_matrix = _row(viewmatrices, v), _row(projmatrices, v), _row(campos, v), pp
        w.alphas, w.gt_depth, w.radii = _row(alphas, v), _row(gt_depths, v), _row(radii, v)
        w.dL_dpix, w.dL_dpix_depth, w.dL_dpix_median_depth, w.dL_dpix_depth_var = _row(gC, v), _row(gD, v), _row(gM, v), _row(gV, v)
        w.dL_dmean2D, w.dL_dview, w.scratch, w.scratch_bytes = _row(d2, v), _row(dview, v), _row(scratch, v), nscr
        w.num_rendered = num_rendered[v]
    p = _capi.ptr
    q = lambda t: None if t is None else p(t)  # noqa: E731
    _light._check(lib.dgr_light_backward_batch(
        _capi.stream_handle(dev.index), V, views, P, int(degree), M, p(bg), W, H, p(means3D), p(sh), p(colors), p(scales),
        float(scale_modifier), p(rotations), p(cov3D_precomp), float(tanfovx), float(tanfovy), q(dop), q(dcol), q(d3), q(dcov),
        q(dsh), q(dsc), q(drot), int(bool(track_off)), int(bool(map_off))))
    return d2, dcol, dop, d3, dcov, dsh, dsc, drot, dview


class _RasterizeGaussiansBatch(torch.autograd.Function):
    """`_RasterizeGaussians` (L/diff_gaussian_rasterization/__init__.py:48-176) over V cameras."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrices,
                gt_depths, raster_settings):
        rs = raster_settings
        with _capi.on_device(means3D.device):
            out = _forward_batch(rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                                 viewmatrices, gt_depths, rs.projmatrices, rs.tanfovx, rs.tanfovy, int(rs.image_height),
                                 int(rs.image_width), sh, rs.sh_degree, rs.campos, rs.prefiltered)
        (num_rendered, color, depth, depth_median, depth_var, opacity_map, radii, geom, binning, img, unc, px) = out
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.dgr_options = _capi.load().dgr_thread_options_effective()  # the backward runs under the forward's options
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrices, radii, sh, geom, binning,
                              img, opacity_map, gt_depths)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, px)
        return color, radii, depth, depth_median, depth_var, opacity_map, unc, px

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_depth_median, grad_depth_var, grad_alpha, grad_unc, grad_px):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrices, radii, sh, geom, binning, img, opacity_map,
         gt_depths) = ctx.saved_tensors
        V, H, W = viewmatrices.size(0), int(rs.image_height), int(rs.image_width)
        zeros = lambda ch: torch.zeros((V, ch, H, W), dtype=torch.float32, device=means3D.device)  # noqa: E731
        grad_color = zeros(3) if grad_color is None else grad_color
        grad_depth = zeros(1) if grad_depth is None else grad_depth
        grad_depth_median = zeros(1) if grad_depth_median is None else grad_depth_median
        grad_depth_var = zeros(1) if grad_depth_var is None else grad_depth_var
        need = ctx.needs_input_grad
        with _capi.on_device(means3D.device), _capi.under_options(ctx.dgr_options):
            (g2, gcol, gop, g3, gcov, gsh, gsc, grot, gview) = _backward_batch(
                rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, viewmatrices,
                rs.projmatrices, rs.tanfovx, rs.tanfovy, grad_color, grad_depth, grad_depth_median, grad_depth_var, gt_depths, sh,
                rs.sh_degree, rs.campos, geom, binning, img, opacity_map, rs.perspec_matrix, rs.track_off, rs.map_off,
                need_gaussian_grads=any(need[:8]), need_means2D=bool(need[1]), num_rendered=ctx.num_rendered)
        _light._consume_post_backward_wait()
        return g3, g2, gsh, gcol, gop, gsc, grot, gcov, gview, None, None


def rasterize_gaussians_batch(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrices,
                              gt_depths, raster_settings):
    """`rasterize_gaussians` (L/diff_gaussian_rasterization/__init__.py:22-46) for V cameras: `means2D` is [V,P,3] (or a
    tensor that requires no gradient), `viewmatrices` [V,4,4], `gt_depths` [V,H,W]; the eight outputs carry a leading view
    dimension."""
    return _RasterizeGaussiansBatch.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                          viewmatrices, gt_depths, raster_settings)


class GaussianRasterizerBatch(torch.nn.Module):
    """`GaussianRasterizer` (L/diff_gaussian_rasterization/__init__.py:197-258) over the V cameras of its settings."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, viewmatrices=None, gt_depths=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = torch.Tensor([])
        shs = e if shs is None else shs
        colors_precomp = e if colors_precomp is None else colors_precomp
        scales = e if scales is None else scales
        rotations = e if rotations is None else rotations
        cov3D_precomp = e if cov3D_precomp is None else cov3D_precomp
        if viewmatrices is None:
            viewmatrices = self.raster_settings.viewmatrices
        return rasterize_gaussians_batch(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                         viewmatrices, gt_depths, self.raster_settings)
