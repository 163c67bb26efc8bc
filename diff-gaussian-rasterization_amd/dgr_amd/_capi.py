"""ctypes binding of lib/libdgr_hip.so (C ABI declared in include/dgr_hip.h).

torch is used for device memory and the current HIP stream only; every compute call goes through
the C ABI.  `import torch` must precede loading the library so that it binds to the HIP runtime
torch already mapped (same SONAME, libamdhip64.so.7).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported before the HIP library is mapped)

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# DGR_HIP_LIB points at another build of the same ABI (kernel experiments); the default is the in-tree library
LIB_PATH = os.environ.get("DGR_HIP_LIB") or os.path.join(_PKG, "lib", "libdgr_hip.so")

DGR_OK = 0
DGR_ERR_BAD_ARGUMENT = -1
DGR_ERR_PREFILTERED = -2
DGR_ERR_BINNING_OVERFLOW = -3
DGR_ERR_HIP = -4
DGR_ERR_ALLOC = -5

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t



class LightView(C.Structure):  # dgr_light_view (include/dgr_hip.h): the per-camera arguments of a batched forward
    _fields_ = [("geometry_buffer", _vp), ("binning_buffer", _vp), ("binning_capacity", _i), ("image_buffer", _vp),
                ("status", _vp), ("viewmatrix", _vp), ("projmatrix", _vp), ("cam_pos", _vp), ("out_color", _vp),
                ("out_depth", _vp), ("out_median_depth", _vp), ("out_alpha", _vp), ("gt_depth", _vp),
                ("out_depth_var", _vp), ("gau_uncertainty", _vp), ("gau_related_pixels", _vp), ("radii", _vp)]


class LightViewGrad(C.Structure):  # dgr_light_view_grad: the per-camera arguments of a batched backward
    _fields_ = [("geometry_buffer", _vp), ("binning_buffer", _vp), ("image_buffer", _vp), ("viewmatrix", _vp),
                ("projmatrix", _vp), ("cam_pos", _vp), ("perspec_matrix", _vp), ("alphas", _vp), ("gt_depth", _vp),
                ("radii", _vp), ("dL_dpix", _vp), ("dL_dpix_depth", _vp), ("dL_dpix_median_depth", _vp),
                ("dL_dpix_depth_var", _vp), ("dL_dmean2D", _vp), ("dL_dview", _vp), ("scratch", _vp),
                ("scratch_bytes", _sz), ("num_rendered", _i)]


MAX_BATCH_VIEWS = 8  # DGR_MAX_BATCH_VIEWS

# argument lists follow include/dgr_hip.h one to one
_SIGS = {
    "dgr_light_forward_batch": (_i, [_vp, _i, C.POINTER(LightView), _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp,
                                     _vp, _f, _f, _i]),
    "dgr_light_backward_batch": (_i, [_vp, _i, C.POINTER(LightViewGrad), _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp,
                                      _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "dgr_last_error": (C.c_char_p, []),
    "dgr_status_post": (C.c_long, [_vp, _vp]),
    "dgr_status_poll": (_i, [C.c_long, _i, _vp]),
    "dgr_status_arm": (C.c_long, []),
    "dgr_stream_is_capturing": (_i, [_vp]),
    "dgr_early_status_arm": (_i, []),
    "dgr_backward_scratch_clean_arm": (_i, []),
    "dgr_early_status_wait": (_i, [_vp]),
    "dgr_pose_forward": (_i, [_vp] * 7),
    "dgr_pose_backward": (_i, [_vp] * 5),
    "dgr_l1_loss_scratch_floats": (_i, []),
    "dgr_l1_loss_forward": (_i, [_vp, C.c_long, _vp, _vp, C.c_long, _vp, _vp, _f, _f, _vp, _vp]),
    "dgr_l1_loss_backward": (_i, [_vp, C.c_long, _vp, _vp, C.c_long, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "dgr_densification_stats": (_i, [_vp, C.c_long, _vp, _vp, _vp, _vp, _vp]),
    "dgr_sparse_adam": (_i, [_vp, C.c_long, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i]),
    "dgr_sparse_adam_capturable": (_i, [_vp, C.c_long, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp]),
    "dgr_set_option": (_i, [C.c_char_p, _i]),
    "dgr_get_option": (_i, [C.c_char_p]),
    "dgr_set_thread_option": (_i, [C.c_char_p, _i]),
    "dgr_get_thread_option": (_i, [C.c_char_p]),
    "dgr_thread_options_effective": (_i, []),
    "dgr_thread_options_swap": (_i, [_i]),
    "dgr_version": (C.c_char_p, []),
    "dgr_geometry_bytes": (_sz, [_i]),
    "dgr_image_bytes": (_sz, [_i, _i]),
    "dgr_binning_bytes": (_sz, [_i, _i, _i]),
    "dgr_light_backward_scratch_bytes": (_sz, [_i, _i, _i]),
    "dgr_light_backward_scratch_bytes_r": (_sz, [_i, _i, _i, _i]),
    "dgr_mark_visible": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "dgr_light_forward": (_i, [_vp, ALLOC_FN, ALLOC_FN, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i,
                               _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dgr_light_forward_presized": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i,
                                        _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgr_light_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                                _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                _vp, _i, _i, _vp, _sz]),
    "dgr_full_forward": (_i, [_vp, ALLOC_FN, ALLOC_FN, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i,
                              _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i,
                              _vp, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "dgr_full_forward_presized": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i,
                                       _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f, _i,
                                       _vp, _vp, _vp, _vp, _vp]),
    "dgr_full_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "dgr_state_export": (C.c_long, [_vp, C.c_char_p, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dgr_cov3d_forward": (_i, [_vp, _i, _vp, _vp, _f, _vp]),
    "dgr_cov3d_backward": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "dgr_debug_wave_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgr_debug_exact_math": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "dgr_debug_half_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgr_debug_half_reduce16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dgr_debug_lane_lists": (_i, [_vp, _vp, _vp, _vp]),
    "dgr_debug_bin_tiles_trace": (_i, [_vp]),
    "dgr_profile_select": (_i, [C.c_char_p]),
    "dgr_profile_stage_count": (_i, []),
    "dgr_profile_stage_name": (C.c_char_p, [_i]),
    "dgr_profile_read": (_i, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i)]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Maps the HIP library; raises (never falls back) when it is missing or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C {_PKG}` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the rasterizer.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().dgr_last_error().decode()


def ptr(t):
    """Device pointer of a tensor; NULL for None / empty tensors (the reference's nullptr convention)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def stream_handle(device_index=None):
    """Raw handle of torch's current HIP stream ON THE GIVEN DEVICE -- pass the index of the device the tensors live on
    (the private torch binding is ~20x cheaper than building a Stream object)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if device_index is None else device_index)
    except AttributeError:  # pragma: no cover -- other torch builds
        return torch.cuda.current_stream(device_index).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """Context manager that makes `dev` the current HIP device for the C-ABI calls inside it (kernels are launched on
    a stream of `dev` against pointers of `dev`; the library's events and workspaces are created on the current
    device).  Free when `dev` already is current."""
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def set_option(name, value):
    """Process-wide library option (include/dgr_hip.h: dgr_set_option), e.g. set_option("tight_cull", 1)."""
    if load().dgr_set_option(name.encode(), int(value)):
        raise ValueError(last_error())


def get_option(name):
    return load().dgr_get_option(name.encode())


class thread_options:
    """`with thread_options(alpha_mode=1, tight_cull=1): ...` -- the calling THREAD's rasterizer calls inside the block use these
    values of the per-call options (include/dgr_hip.h: dgr_set_thread_option) whatever the process-wide ones are; other threads
    are not affected, and a backward runs under its forward's options wherever autograd runs it.  Names: alpha_mode (fast_alpha),
    tight_cull, deterministic_grads."""

    def __init__(self, **options):
        self.options = options
        self.prev = None

    def __enter__(self):
        lib = load()
        self.prev = lib.dgr_thread_options_swap(-1)
        for k, v in self.options.items():
            if lib.dgr_set_thread_option(k.encode(), int(v)):
                lib.dgr_thread_options_swap(self.prev)
                raise ValueError(last_error())
        return self

    def __exit__(self, *exc):
        load().dgr_thread_options_swap(self.prev)
        return False


class under_options:
    """Runs a block under a word of dgr_thread_options_effective() (a backward under its forward's options)."""

    def __init__(self, word):
        self.word = word

    def __enter__(self):
        self.prev = load().dgr_thread_options_swap(self.word)

    def __exit__(self, *exc):
        load().dgr_thread_options_swap(self.prev)
        return False


def profile_select(stage=""):
    rc = load().dgr_profile_select(stage.encode())
    if rc:
        raise ValueError(f"unknown stage {stage!r}")


def profile_stages():
    lib = load()
    return [lib.dgr_profile_stage_name(i).decode() for i in range(lib.dgr_profile_stage_count())]


def profile_read(stage):
    """(total milliseconds, launches) recorded for `stage` since the last read."""
    tot, n = C.c_double(0), C.c_int(0)
    rc = load().dgr_profile_read(stage.encode(), C.byref(tot), C.byref(n))
    if rc:
        raise ValueError(f"unknown stage {stage!r}")
    return tot.value, n.value
