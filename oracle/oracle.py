"""ctypes front-end of the CPU oracle (oracle/dgr_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
# Two builds of the same source (see Makefile):
#   float math (default)  exp/sqrt/ceil bind to the float overloads, as under nvcc -- the checker;
#   C math                they bind to the C double functions, as in the survey's CPU run of the
#                         reference -- reproduces SURVEY.md Appendix C digit for digit.
# use_cmath(True) / DGR_ORACLE_CMATH=1 selects the second one for subsequently created states.
_CMATH = os.environ.get("DGR_ORACLE_CMATH") == "1"
_NATIVE = False
_EXP_MODE = 0


def use_cmath(flag):
    global _CMATH
    _CMATH = bool(flag)


def use_native(flag):
    """cpu_baseline leg of bench.py only: the same source built `-O3 -march=native` ON THE MACHINE THAT RUNS IT
    (BASELINE.md s3.1); never the checker."""
    global _NATIVE
    _NATIVE = bool(flag)


def build_native():
    import platform
    import tempfile
    src = os.path.join(_HERE, "dgr_oracle.cpp")
    out = os.path.join(tempfile.gettempdir(), f"libdgr_oracle_native_{platform.node()}_{int(os.path.getmtime(src))}.so")
    if not os.path.exists(out):
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC",
                               "-Wno-unknown-pragmas", "-shared", "-o", out + ".tmp", src])
        os.replace(out + ".tmp", out)
    return out


def set_threads(n):
    """OpenMP threads of the oracle's runtime (libgomp is already initialised once torch or the library is loaded, so
    the environment variable alone would come too late)."""
    C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


def build(force=False):
    src = os.path.join(_HERE, "dgr_oracle.cpp")
    sos = [os.path.join(_HERE, n) for n in ("libdgr_oracle.so", "libdgr_oracle_cmath.so")]
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in sos):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return sos


def lib(cmath=None):
    cmath = _CMATH if cmath is None else bool(cmath)
    if _NATIVE:
        cmath = "native"
    if cmath not in _LIBS:
        l = C.CDLL(build_native() if cmath == "native" else build()[1 if cmath else 0])
        l.dgro_state_new.restype = C.c_void_p
        l.dgro_state_free.argtypes = [C.c_void_p]
        l.dgro_state_get.restype = C.c_long
        l.dgro_state_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        l.dgro_state_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
        l.dgro_state_set_dims.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.dgro_state_num_rendered.argtypes = [C.c_void_p]
        l.dgro_set_exp_mode(C.c_int(_EXP_MODE))
        _LIBS[cmath] = l
    return _LIBS[cmath]


def expf_restated(x):
    """dgr_oracle.cpp: expf_restated -- glibc's expf algorithm in IEEE double operations (machine-independent bits)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().dgro_expf_restated(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_long(x.size))
    return y


def expf_p32(x):
    """dgr_oracle.cpp: expf_p32 -- the fp32-only expf (fused multiply-adds and an exponent-field add; machine-independent bits),
    the default exponential of the float build since round 8."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().dgro_expf_p32(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_long(x.size))
    return y


def set_exp_mode(mode):
    """Which restated expf the float build's blend loops call: 0 = expf_p32 (default; the HIP kernels' alpha_mode 0),
    1 = expf_restated = glibc's algorithm (alpha_mode 2).  Returns the previous mode."""
    global _EXP_MODE
    old, _EXP_MODE = _EXP_MODE, (1 if mode else 0)
    for l in _LIBS.values():  # every build loaded so far (the C-math build ignores it); later loads pick it up in lib()
        l.dgro_set_exp_mode(C.c_int(_EXP_MODE))
    return old


def expf_p32_scan(lo_bits=0x80000000, hi_bits=0xC2D00000):
    """expf_p32 against exp() in double on every float with bits in [lo_bits, hi_bits]: dict(max_ulp_normal, max_ulp_denormal,
    not_correctly_rounded, scanned, at_normal, at_denormal)."""
    out = (C.c_double * 6)()
    lib(False).dgro_expf_p32_scan(C.c_uint32(lo_bits), C.c_uint32(hi_bits), out)
    return dict(max_ulp_normal=out[0], max_ulp_denormal=out[1], not_correctly_rounded=int(out[2]), scanned=int(out[3]),
                at_normal=int(out[4]), at_denormal=int(out[5]))


def exp_as_the_oracle_calls_it(x):
    """exp of a float32 array through the very function the blend loops of dgr_oracle.cpp call (expf_p32 -- or expf_restated
    after set_exp_mode(1) -- in the default build, the C double exp in the cmath build)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().dgro_exp(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_long(x.size))
    return y


def _p(a):
    """float32/int32 numpy array (or None) -> void pointer (NULL for None / empty)."""
    if a is None or a.size == 0:
        return C.c_void_p(None)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


_DTYPES = {
    "depths": np.float32, "means2D": np.float32, "cov3D": np.float32, "conic_opacity": np.float32,
    "rgb": np.float32, "clamped": np.uint8, "radii": np.int32, "tiles_touched": np.uint32,
    "point_offsets": np.uint32, "keys_unsorted": np.uint64, "keys": np.uint64,
    "point_list_unsorted": np.uint32, "point_list": np.uint32, "ranges": np.uint32,
    "n_contrib": np.uint32, "n_valid_contrib": np.uint32, "final_T": np.float32,
    "dgndcs_dview": np.float32, "dg_camd": np.float32,
}


class OracleState:
    """Owns the Geometry/Binning/Image state of one forward call (opaque byte buffers in the reference)."""

    def __init__(self, W=0, H=0):
        self._l = lib()
        self._h = C.c_void_p(self._l.dgro_state_new())
        self._W, self._H = W, H

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.dgro_state_free(self._h)
            self._h = None

    def get(self, name):
        ptr, elem = C.c_void_p(), C.c_int()
        n = self._l.dgro_state_get(self._h, name.encode(), C.byref(ptr), C.byref(elem))
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_DTYPES[name])
        assert dt.itemsize == elem.value
        if n == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=_DTYPES[name])
        if self._l.dgro_state_set(self._h, name.encode(), C.c_void_p(arr.ctypes.data), arr.size):
            raise KeyError(name)

    def set_dims(self, P, W, H):
        self._W, self._H = W, H
        self._l.dgro_state_set_dims(self._h, P, W, H)

    @property
    def num_rendered(self):
        return self._l.dgro_state_num_rendered(self._h)


def mark_visible(means, view, proj):
    means, view, proj = _f(means), _f(view), _f(proj)
    out = np.zeros(means.shape[0], np.uint8)
    lib().dgro_mark_visible(C.c_int(means.shape[0]), _p(means), _p(view), _p(proj), C.c_void_p(out.ctypes.data))
    return out.astype(bool)


def _common(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos):
    means3D = _f(means3D)
    shs, colors_precomp, cov3D_precomp = _f(shs), _f(colors_precomp), _f(cov3D_precomp)
    opacities, scales, rotations = _f(opacities), _f(scales), _f(rotations)
    P = means3D.shape[0]
    M = shs.shape[1] if shs is not None and shs.size else 0
    return (P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
            _f(viewmatrix), _f(projmatrix), _f(campos))


def preprocess(st, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
               viewmatrix, projmatrix, campos, tanfovx, tanfovy, sh_degree, prefiltered=False):
    (P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
     campos) = _common(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                       projmatrix, campos)
    st._W, st._H = W, H
    return st._l.dgro_preprocess(st._h, P, sh_degree, M, W, H, _p(means3D), _p(shs), _p(colors_precomp),
                                 _p(opacities), _p(scales), C.c_float(scale_modifier), _p(rotations),
                                 _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), C.c_float(tanfovx),
                                 C.c_float(tanfovy), int(prefiltered))


def binning(st):
    return st._l.dgro_binning(st._h)


def light_render_forward(st, bg, colors_precomp, gt_depth):
    W, H, P = _dims(st)
    bg, colors_precomp, gt_depth = _f(bg), _f(colors_precomp), _f(gt_depth)
    out = dict(color=np.zeros((3, H, W), np.float32), depth=np.zeros((1, H, W), np.float32),
               depth_median=np.zeros((1, H, W), np.float32), depth_var=np.zeros((1, H, W), np.float32),
               opacity_map=np.zeros((1, H, W), np.float32), gau_uncertainty=np.zeros((P, 1), np.float32),
               gau_related_pixels=np.zeros((P, 1), np.int32))
    st._l.dgro_light_render_forward(st._h, _p(bg), _p(colors_precomp), _p(gt_depth), _p(out["color"]), _p(out["depth"]),
                                    _p(out["depth_median"]), _p(out["opacity_map"]), _p(out["depth_var"]),
                                    _p(out["gau_uncertainty"]), _p(out["gau_related_pixels"]))
    return out


def light_median_margin(st, alphas=None):
    """[H, W] min_k |T_k - 0.5| over the blended Gaussians of each pixel (dgr_oracle.cpp: dgro_light_median_margin): how
    close the median-depth decision of L/cr/forward.cu:353 / backward.cu:1545 came to going the other way."""
    W, H, _ = _dims(st)
    out = np.zeros((H, W), np.float32)
    st._l.dgro_light_median_margin.restype = None
    alphas = None if alphas is None else _f(alphas)
    st._l.dgro_light_median_margin(st._h, _p(alphas), _p(out))
    return out


def _dims(st):
    ptr, elem = C.c_void_p(), C.c_int()
    P = st._l.dgro_state_get(st._h, b"radii", C.byref(ptr), C.byref(elem))
    return st._W, st._H, P


def light_forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                  viewmatrix, gt_depth, projmatrix, tanfovx, tanfovy, H, W, shs, sh_degree, campos,
                  prefiltered=False):
    """Argument order of `_C.rasterize_gaussians` (L/rasterize_points.h:18-39).  Returns (state, dict)."""
    (P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
     campos) = _common(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                       projmatrix, campos)
    bg, gt_depth = _f(bg), _f(gt_depth)
    st = OracleState(W, H)
    out = dict(color=np.zeros((3, H, W), np.float32), depth=np.zeros((1, H, W), np.float32),
               depth_median=np.zeros((1, H, W), np.float32), depth_var=np.zeros((1, H, W), np.float32),
               opacity_map=np.zeros((1, H, W), np.float32), radii=np.zeros(P, np.int32),
               gau_uncertainty=np.zeros((P, 1), np.float32), gau_related_pixels=np.zeros((P, 1), np.int32))
    R = 0
    if P:
        R = st._l.dgro_light_forward(
            st._h, P, sh_degree, M, _p(bg), W, H, _p(means3D), _p(shs), _p(colors_precomp), _p(opacities),
            _p(scales), C.c_float(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix),
            _p(campos), C.c_float(tanfovx), C.c_float(tanfovy), int(prefiltered), _p(out["color"]), _p(out["depth"]),
            _p(out["depth_median"]), _p(out["opacity_map"]), _p(gt_depth), _p(out["depth_var"]),
            _p(out["gau_uncertainty"]), _p(out["gau_related_pixels"]), _p(out["radii"]))
        if R < 0:
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    out["num_rendered"] = R
    return st, out


def light_backward(st, bg, means3D, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                   projmatrix, tanfovx, tanfovy, dL_dcolor, dL_ddepth, dL_dmedian, dL_dvar, gt_depth, shs,
                   sh_degree, campos, alphas, perspec_matrix, track_off=False, map_off=False, per_pixel_pose=False):
    """Argument meaning of `_C.rasterize_gaussians_backward` (L/rasterize_points.h:41-72); the saved
    byte buffers are the OracleState.  Returns the 9 gradients with grad_viewmatrix already [4,4]."""
    (P, M, means3D, shs, colors_precomp, _, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
     campos) = _common(means3D, shs, colors_precomp, None, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                       campos)
    W, H = st._W, st._H
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
             dL_ddepths=np.zeros((P, 1), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
             dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
             dL_dview=np.zeros((4, 4), np.float32))
    pix = np.zeros((H * W, 4, 4), np.float32) if per_pixel_pose else None
    if P:
        st._l.dgro_light_backward(
            st._h, P, sh_degree, M, _p(_f(bg)), _p(means3D), _p(shs), _p(colors_precomp), _p(_f(alphas)), _p(scales),
            C.c_float(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
            C.c_float(tanfovx), C.c_float(tanfovy), _p(_f(dL_dcolor)), _p(_f(dL_ddepth)), _p(_f(dL_dmedian)),
            _p(_f(dL_dvar)), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]),
            _p(g["dL_ddepths"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]),
            _p(g["dL_drotations"]), _p(_f(perspec_matrix)), _p(pix), _p(g["dL_dview"]), _p(_f(gt_depth)),
            int(track_off), int(map_off))
    if per_pixel_pose:
        g["dL_dview_pix"] = pix
    return g


# ------------------------------------------------------------------------------------------ full variant
def full_forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                 viewmatrix, gt_depth, projmatrix, tanfovx, tanfovy, H, W, shs, sh_degree, campos, prefiltered=False):
    """Argument order of the full `_C.rasterize_gaussians` (F/rasterize_points.h:18-38).  Returns (state, dict)."""
    (P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
     campos) = _common(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                       projmatrix, campos)
    bg, gt_depth = _f(bg), _f(gt_depth)
    st = OracleState(W, H)
    out = dict(color=np.zeros((3, H, W), np.float32), depth=np.zeros((1, H, W), np.float32),
               uncertainty=np.zeros((1, H, W), np.float32), radii=np.zeros(P, np.int32))
    R, ng = 0, C.c_int(0)
    if P:
        R = st._l.dgro_full_forward(
            st._h, P, sh_degree, M, _p(bg), W, H, _p(means3D), _p(shs), _p(colors_precomp), _p(opacities),
            _p(scales), C.c_float(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix),
            _p(campos), C.c_float(tanfovx), C.c_float(tanfovy), int(prefiltered), _p(out["color"]), _p(out["depth"]),
            _p(gt_depth), _p(out["uncertainty"]), _p(out["radii"]), C.byref(ng))
        if R < 0:
            raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    out["num_rendered"] = R
    out["num_related"] = ng.value
    return st, out


def full_backward(st, bg, means3D, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                  gt_depth, projmatrix, tanfovx, tanfovy, dL_dcolor, dL_ddepth, dL_duncertainty, shs, sh_degree, campos,
                  perspec_matrix, emulate_dropout=False):
    """Argument meaning of the full `_C.rasterize_gaussians_backward` (F/rasterize_points.h:40-67)."""
    (P, M, means3D, shs, colors_precomp, _, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
     campos) = _common(means3D, shs, colors_precomp, None, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                       campos)
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
             dL_dgau_depths=np.zeros((P, 1), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
             dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
             dL_dview=np.zeros((4, 4), np.float32))
    if P:
        st._l.dgro_full_backward(
            st._h, P, sh_degree, M, _p(_f(bg)), _p(means3D), _p(shs), _p(colors_precomp), _p(scales),
            C.c_float(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
            C.c_float(tanfovx), C.c_float(tanfovy), _p(_f(dL_dcolor)), _p(_f(dL_ddepth)), _p(g["dL_dmeans2D"]),
            _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]),
            _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]), _p(_f(perspec_matrix)), _p(g["dL_dview"]),
            _p(g["dL_dgau_depths"]), _p(_f(gt_depth)), _p(_f(dL_duncertainty)), int(emulate_dropout))
    return g
