"""Host-side mirror of diff-gaussian-rasterization-full/diff_gaussian_rasterization/__init__.py.

Same names, argument order and return arity as the reference module of the -full variant
(`GaussianRasterizationSettings` without debug / track_off / map_off, forward returning
`(color, radii, depth, uncertainty)`), with `_C.*` replaced by the gfx950 C ABI (include/dgr_hip.h).
`num_related_gaussians` (the reference's NG, which sizes its pair lists) is still produced and threaded
through the autograd context, but nothing here is sized by it.
"""
import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _capi
from . import light as _light
from .light import _capacity_cache, _check, _f32c, _grad_arena


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
    """Functions with the signatures of the full variant's pybind11 module (F/ext.cpp:15-19)."""

    @_device_guarded(1)
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                            cov3D_precomp, viewmatrix, gt_depth, projmatrix, tan_fovx, tan_fovy,
                            image_height, image_width, sh, degree, campos, prefiltered):
        # F/rasterize_points.cu:35-120
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        lib = _capi.load()
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("dgr_hip runs on the GPU only (no CPU path exists, as in the reference)")
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        means3D = _f32c(means3D, dev)
        background, colors, opacity = _f32c(background, dev), _f32c(colors, dev), _f32c(opacity, dev)
        scales, rotations, cov3D_precomp = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3D_precomp, dev)
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, dev), _f32c(projmatrix, dev), _f32c(campos, dev)
        gt_depth, sh = _f32c(gt_depth, dev), _f32c(sh, dev)
        M = sh.size(1) if sh.numel() != 0 else 0
        out_color = torch.empty((3, H, W), **f32)
        out_depth = torch.empty((1, H, W), **f32)
        out_unc = torch.empty((1, H, W), **f32)
        radii = torch.zeros((P,), **i32)
        st = _capi.stream_handle(dev.index)
        p = _capi.ptr
        common = (P, int(degree), M, p(background), W, H, p(means3D), p(sh), p(colors), p(opacity), p(scales),
                  float(scale_modifier), p(rotations), p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos),
                  float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), p(out_color), p(out_depth), p(gt_depth),
                  p(out_unc), p(radii))
        if os.environ.get("DGR_FORWARD_MODE", "presized") == "callback" or P == 0:
            import ctypes as C
            bufs = {k: torch.empty((0,), **u8) for k in ("geom", "binning", "img")}

            def mk(name):
                def cb(nbytes, _user):
                    bufs[name] = torch.empty((max(int(nbytes), 1),), **u8)
                    return bufs[name].data_ptr()
                return _capi.ALLOC_FN(cb)
            cbs = [mk("geom"), mk("binning"), mk("img")]
            ng = C.c_int(0)
            rendered = _check(lib.dgr_full_forward(st, cbs[0], cbs[1], cbs[2], None, *common, C.byref(ng)))
            related = ng.value
            geomBuffer, binningBuffer, imgBuffer = bufs["geom"], bufs["binning"], bufs["img"]
        else:
            geomBuffer = torch.empty((lib.dgr_geometry_bytes(P),), **u8)
            imgBuffer = torch.empty((lib.dgr_image_bytes(W, H),), **u8)
            status = torch.empty((4,), **i32)
            key = (dev.index, P, H, W)
            cap = _capacity_cache.get(key, 0)
            if _light._sync_mode() == "lazy" and cap > 0:
                # no host synchronisation (dgr_amd/light.py): the status word is checked one call late; the tuple's
                # num_rendered / num_related members are the latest values read back for this shape
                while len(_light._pending_status) > _light.lazy_depth() and not torch.cuda.is_current_stream_capturing():
                    _light._check_oldest()
                cap = int(cap * 1.5) + 4096
                binningBuffer = torch.empty((lib.dgr_binning_bytes(cap, W, H),), **u8)
                _check(lib.dgr_full_forward_presized(st, p(geomBuffer), p(binningBuffer), cap, p(imgBuffer), p(status),
                                                     *common))
                _light._post_status(status, key)
                related = _light._last_status.get(key, (0, 0, 0, 0))[3]
                return (_capacity_cache[key], related, out_color, out_depth, out_unc, radii, geomBuffer, binningBuffer,
                        imgBuffer)
            cap = int(cap * 1.25) + 4096 if cap else 4 * P + 4096
            while True:
                binningBuffer = torch.empty((lib.dgr_binning_bytes(cap, W, H),), **u8)
                lib.dgr_early_status_arm()
                _check(lib.dgr_full_forward_presized(st, p(geomBuffer), p(binningBuffer), cap, p(imgBuffer), p(status),
                                                     *common))
                s = _light._early_status(lib)  # waits until num_rendered is known, not for the whole forward
                if s[2]:
                    raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
                rendered = s[0]
                _capacity_cache[key] = max(_capacity_cache.get(key, 0), rendered)
                if rendered <= cap:
                    # num_related (the reference's NG) is produced by the forward blend: the second blocking read of
                    # the reference (F/cuda_rasterizer/rasterizer_impl.cu:498); lazy mode reports it one call late instead
                    s = status.tolist()
                    related = s[3]
                    _light._last_status[key] = s
                    break
                cap = int(rendered * 1.1) + 4096
        return rendered, related, out_color, out_depth, out_unc, radii, geomBuffer, binningBuffer, imgBuffer

    @_device_guarded(1)
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, gt_depth, projmatrix, tan_fovx, tan_fovy,
                                     dL_dout_color, dL_dout_depth, dL_dout_uncertainty, sh, degree, campos, geomBuffer,
                                     R, binningBuffer, imageBuffer, NG, perspec_matrix, need_gaussian_grads=True):
        # F/rasterize_points.cu:122-239.  need_gaussian_grads=False (tracking: no Gaussian input requires a gradient) returns
        # None for the eight per-Gaussian gradients and skips their dense rows; the pose gradient is the same.
        lib = _capi.load()
        dev = means3D.device
        P = means3D.size(0)
        H, W = dL_dout_color.size(1), dL_dout_color.size(2)
        f32 = dict(dtype=torch.float32, device=dev)
        means3D = _f32c(means3D, dev)
        background, colors = _f32c(background, dev), _f32c(colors, dev)
        scales, rotations, cov3D_precomp = _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3D_precomp, dev)
        viewmatrix, projmatrix, campos = _f32c(viewmatrix, dev), _f32c(projmatrix, dev), _f32c(campos, dev)
        gt_depth, sh, perspec_matrix = _f32c(gt_depth, dev), _f32c(sh, dev), _f32c(perspec_matrix, dev)
        gC, gD, gU = _f32c(dL_dout_color, dev), _f32c(dL_dout_depth, dev), _f32c(dL_dout_uncertainty, dev)
        M = sh.size(1) if sh.numel() != 0 else 0
        names = ("means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations")
        seg = _grad_arena(P, M, f32) if need_gaussian_grads else dict.fromkeys(names)
        dL_dview = torch.empty((4, 4), **f32)
        scratch = torch.empty((max(lib.dgr_light_backward_scratch_bytes_r(P, W, H, int(R)), 1),), dtype=torch.uint8, device=dev)
        p = lambda t: None if t is None else _capi.ptr(t)  # noqa: E731
        _check(lib.dgr_full_backward(
            _capi.stream_handle(dev.index), P, int(degree), M, int(R), p(background), W, H, p(means3D), p(sh), p(colors),
            p(scales), float(scale_modifier), p(rotations), p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos),
            float(tan_fovx), float(tan_fovy), p(radii), p(geomBuffer), p(binningBuffer), p(imageBuffer), p(gC), p(gD),
            p(seg["means2D"]), None, p(seg["opacity"]), p(seg["colors"]), p(seg["means3D"]), p(seg["cov3D"]),
            p(seg["sh"]), p(seg["scales"]), p(seg["rotations"]), None, None, None, None, None, p(perspec_matrix), None,
            None, None, p(dL_dview), None, None, None, p(gt_depth), p(gU), p(scratch), scratch.numel()))
        return (seg["means2D"], seg["colors"], seg["opacity"], seg["means3D"], seg["cov3D"], seg["sh"], seg["scales"],
                seg["rotations"], dL_dview)

    @_device_guarded(0)
    def mark_visible(means3D, viewmatrix, projmatrix):
        from .light import _C as _LC
        return _LC.mark_visible(means3D, viewmatrix, projmatrix)


class _CompiledC:
    """The same functions over the compiled torch extension (csrc/torch_ext.cpp), the counterpart of F/ext.cpp:15-19;
    see dgr_amd.light._CompiledC."""

    ext = None

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                            gt_depth, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered):
        P, H, W = means3D.size(0) if means3D.dim() else 0, int(image_height), int(image_width)
        key = (means3D.device.index, P, H, W)
        mode, use, cap = _light._binning_policy(key, P)
        (rendered, related, ticket, _, status, color, depth, unc, radii, geom, binning, img) = _CompiledC.ext.full_forward(
            background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix,
            gt_depth, projmatrix, float(tan_fovx), float(tan_fovy), H, W, sh, int(degree), campos, bool(prefiltered), use, mode)
        if mode == 2:
            if ticket >= 0:
                _light._pending_status.append((ticket, key))
            else:
                import weakref
                _light._captured_status.append(weakref.ref(status))
                _light._capture_keepalive.append(status)
            rendered, related = _capacity_cache[key], _light._last_status.get(key, (0, 0, 0, 0))[3]
        elif mode == 1:
            _capacity_cache[key] = max(cap, rendered)
            _light._last_status[key] = [rendered, 0, 0, related]
        return rendered, related, color, depth, unc, radii, geom, binning, img

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                     viewmatrix, gt_depth, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                     dL_dout_uncertainty, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, NG,
                                     perspec_matrix, need_gaussian_grads=True):
        return tuple(_CompiledC.ext.full_backward(
            background, means3D, radii, colors, scales, rotations, float(scale_modifier), cov3D_precomp, viewmatrix, gt_depth,
            projmatrix, float(tan_fovx), float(tan_fovy), dL_dout_color, dL_dout_depth, dL_dout_uncertainty, sh, int(degree),
            campos, geomBuffer, int(R), binningBuffer, imageBuffer, int(NG), perspec_matrix, bool(need_gaussian_grads)))

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        return _CompiledC.ext.mark_visible(means3D, viewmatrix, projmatrix)


_CtypesC = _C
if _light._C is getattr(_light, "_CompiledC", None):  # the light module decided (DGR_BINDING, DGR_HIP_LIB, extension built)
    _CompiledC.ext = _light._CompiledC.ext
    _C = _CompiledC


def _rasterize_compiled(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                        gt_depth, rs):
    """`_RasterizeGaussians.apply` through the autograd node compiled into the extension (csrc/torch_ext.cpp: FullNode); see
    dgr_amd.light._rasterize_compiled."""
    P, H, W = (means3D.size(0) if means3D.dim() == 2 else 0), rs.image_height, rs.image_width
    key = (means3D.device.index, P, H, W)
    mode, use, cap = _light._binning_policy(key, P)
    out, rendered, related, ticket, _, status = _CompiledC.ext.full_apply(
        means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix, gt_depth, rs.bg,
        rs.projmatrix, rs.campos, rs.perspec_matrix, rs.scale_modifier, rs.tanfovx, rs.tanfovy, H, W, rs.sh_degree,
        rs.prefiltered, use, mode)
    if mode == 2:
        if ticket >= 0:
            _light._pending_status.append((ticket, key))
        else:
            import weakref
            _light._captured_status.append(weakref.ref(status))
            _light._capture_keepalive.append(status)
    elif mode == 1:
        _capacity_cache[key] = max(cap, rendered)
        # (the strict node does not wait for num_related -- csrc/torch_ext.cpp: full_forward_core -- and reports -1: keep the last one read)
        _light._last_status[key] = [rendered, 0, 0, related if related >= 0 else _light._last_status.get(key, (0, 0, 0, 0))[3]]
    return tuple(out)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        viewmatrix, gt_depth, raster_settings):
    if _C is _CompiledC and _light._USE_NODE:
        return _rasterize_compiled(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                   viewmatrix, gt_depth, raster_settings)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, viewmatrix, gt_depth, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                gt_depth, raster_settings):
        # argument packing of F/diff_gaussian_rasterization/__init__.py:62-83
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
        )
        (num_rendered, num_related_gaussians, color, depth, uncertainty, radii, geomBuffer, binningBuffer,
         imgBuffer) = _C.rasterize_gaussians(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.dgr_options = _capi.load().dgr_thread_options_effective()  # the backward runs under the forward's options
        ctx.num_related_gaussians = num_related_gaussians
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrix, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer, gt_depth)
        ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for radii on every backward
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, uncertainty

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_out_depth, grad_out_uncertainty):
        num_rendered = ctx.num_rendered
        num_related_gaussians = ctx.num_related_gaussians
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, viewmatrix, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, gt_depth) = ctx.saved_tensors
        # outputs that did not take part in the loss arrive as None: zeros, as the reference's autograd would have passed
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        zeros = lambda c: torch.zeros((c, H, W), dtype=torch.float32, device=means3D.device)  # noqa: E731
        grad_out_color = zeros(3) if grad_out_color is None else grad_out_color
        grad_out_depth = zeros(1) if grad_out_depth is None else grad_out_depth
        grad_out_uncertainty = zeros(1) if grad_out_uncertainty is None else grad_out_uncertainty
        # argument packing of F/diff_gaussian_rasterization/__init__.py:104-131
        args = (raster_settings.bg,
                means3D,
                radii,
                colors_precomp,
                scales,
                rotations,
                raster_settings.scale_modifier,
                cov3Ds_precomp,
                viewmatrix,
                gt_depth,
                raster_settings.projmatrix,
                raster_settings.tanfovx,
                raster_settings.tanfovy,
                grad_out_color,
                grad_out_depth,
                grad_out_uncertainty,
                sh,
                raster_settings.sh_degree,
                raster_settings.campos,
                geomBuffer,
                num_rendered,
                binningBuffer,
                imgBuffer,
                num_related_gaussians,
                raster_settings.perspec_matrix)
        with _capi.under_options(ctx.dgr_options):  # (the autograd engine may run this on a thread of its own)
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
             grad_rotations, grad_viewmatrix) = _C.rasterize_gaussians_backward(*args, need_gaussian_grads=any(ctx.needs_input_grad[:8]))
        _light._consume_post_backward_wait()  # (dgr_amd.multiview.ViewStreams.before_backward)
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
    perspec_matrix: torch.Tensor


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
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

        if shs is None:
            shs = _light._EMPTY
        if colors_precomp is None:
            colors_precomp = _light._EMPTY
        if scales is None:
            scales = _light._EMPTY
        if rotations is None:
            rotations = _light._EMPTY
        if cov3D_precomp is None:
            cov3D_precomp = _light._EMPTY

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   viewmatrix, gt_depth, raster_settings)
