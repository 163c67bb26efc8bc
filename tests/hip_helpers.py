"""Helpers that drive the HIP path through the C ABI / the reference-shaped Python surface."""
import ctypes as C

import numpy as np
import torch

from dgr_amd import _capi
from dgr_amd import light as L


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def E():
    return torch.empty(0, device=dev())


def hip_forward(s, deg, colors_precomp=None, cov3D_precomp=None, prefiltered=False, debug=False, scale_modifier=1.0):
    """Calls the `_C.rasterize_gaussians` mirror; returns (torch outputs tuple, dict of numpy arrays)."""
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    out = L._C.rasterize_gaussians(
        T(s.bg), T(s.means), E() if use_sh else T(colors_precomp), T(s.opac), T(s.scales) if use_sr else E(),
        T(s.rots) if use_sr else E(), scale_modifier, E() if use_sr else T(cov3D_precomp), T(s.view), T(s.gt),
        T(s.proj), s.tanfovx, s.tanfovy, s.H, s.W, T(s.shs) if use_sh else E(), deg, T(s.campos), prefiltered, debug)
    names = ["num_rendered", "color", "depth", "depth_median", "depth_var", "opacity_map", "radii", "geom", "binning",
             "img", "gau_uncertainty", "gau_related_pixels"]
    d = {}
    for n, v in zip(names, out):
        if n in ("geom", "binning", "img") or not isinstance(v, torch.Tensor):
            d[n] = v
        else:
            d[n] = v.cpu().numpy()
    return out, d


_EXPORT = {  # name -> (numpy dtype, elements per unit, unit)
    "depths": (np.float32, "P"), "radii": (np.int32, "P"), "means2D": (np.float32, "2P"),
    "conic_opacity": (np.float32, "4P"), "rgb": (np.float32, "3P"), "clamped": (np.uint8, "3P"),
    "tiles_touched": (np.uint32, "P"), "point_list": (np.uint32, "R"), "keys": (np.uint64, "R"),
    "contribution_tags": (np.uint8, "R1"),
    "ranges": (np.uint32, "2T"), "tile_sched": (np.uint32, "4T"), "sched_flag": (np.uint32, "1"), "n_contrib": (np.uint32, "N"), "n_valid": (np.uint32, "N"), "final_T": (np.float32, "N"),
}


def hip_state(name, s, d, capacity=None):
    """Exports one array of the opaque state buffers in the reference's element layout."""
    lib = _capi.load()
    P, W, H, R = s.P, s.W, s.H, d["num_rendered"]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    dt, unit = _EXPORT[name]
    n = {"P": P, "6P": 6 * P, "2P": 2 * P, "4P": 4 * P, "3P": 3 * P, "R": R, "R1": R, "1": 1, "2T": 2 * tiles, "4T": 4 * tiles, "N": W * H}[unit]
    torch_dt = {np.float32: torch.float32, np.int32: torch.int32, np.uint8: torch.uint8, np.uint32: torch.int32,
                np.uint64: torch.int64, np.uint16: torch.int16}[dt]
    dst = torch.zeros(max(n, 1), dtype=torch_dt, device=dev())
    cap = capacity if capacity is not None else binning_capacity(d, W, H)
    got = lib.dgr_state_export(_capi.stream_handle(), name.encode(), P, W, H, R, cap, _capi.ptr(d["geom"]),
                               _capi.ptr(d["binning"]), _capi.ptr(d["img"]), dst.data_ptr())
    assert got >= 0, _capi.last_error()
    torch.cuda.synchronize()
    return dst[:n].cpu().numpy().view(dt)


def binning_capacity(d, W, H):
    """Capacity the binning buffer was carved with, recovered from its size.  Byte size is monotone in the
    capacity and equal sizes imply equal sub-array offsets, so any capacity that reproduces the size will do."""
    lib = _capi.load()
    nbytes = d["binning"].numel()
    lo, hi = 0, nbytes // 16 + 2
    while lo < hi:
        mid = (lo + hi) // 2
        if lib.dgr_binning_bytes(mid, W, H) >= nbytes:
            hi = mid
        else:
            lo = mid + 1
    assert lib.dgr_binning_bytes(lo, W, H) == nbytes or nbytes <= 1, (lo, nbytes)
    # alignment padding can make several capacities share one size (and hence one layout): never report less than R
    R = d["num_rendered"]
    return lo if lo >= R else R


def hip_cov3D(s, scale_modifier=1.0):
    """computeCov3D as the forward and the backward evaluate it (the geometry state does not keep it: csrc/preprocess.hip's
    compute_cov3d, through dgr_cov3d_forward -- the same device function)."""
    from dgr_amd.multiview import shared_cov3D
    return shared_cov3D(T(s.scales), T(s.rots), scale_modifier).detach().cpu().numpy()


def hip_backward(s, deg, out, colors_precomp=None, cov3D_precomp=None, track_off=False, map_off=False,
                 scale_modifier=1.0, grads=None, alphas=None):
    """`alphas` overrides the forward's own opacity map (stage isolation: the light backward derives
    T_final = 1 - alpha, which amplifies one-ulp forward differences on nearly opaque pixels)."""
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    (R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
    gC, gD, gM, gV = grads if grads is not None else (s.gC, s.gD, s.gM, s.gV)
    if alphas is not None:
        alpha = T(alphas)
    g = L._C.rasterize_gaussians_backward(
        T(s.bg), T(s.means), radii, E() if use_sh else T(colors_precomp), T(s.scales) if use_sr else E(),
        T(s.rots) if use_sr else E(), scale_modifier, E() if use_sr else T(cov3D_precomp), T(s.view), T(s.proj),
        s.tanfovx, s.tanfovy, T(gC), T(gD[None]), T(gM[None]), T(gV[None]), T(s.gt), T(s.shs) if use_sh else E(), deg,
        T(s.campos), geom, R, binning, img, alpha, False, T(s.persp), track_off, map_off)
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations", "dL_dview"]
    assert tuple(g[8].shape) == (1, 4, 4)  # what the reference's own __init__.py sums over dim 0
    g = list(g[:8]) + [torch.sum(g[8], dim=0)]
    return {n: v.cpu().numpy() for n, v in zip(names, g)}


def oracle_forward(O, s, deg, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0):
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    return O.light_forward(s.bg, s.means, colors_precomp, s.opac, s.scales if use_sr else None,
                           s.rots if use_sr else None, scale_modifier, cov3D_precomp, s.view, s.gt, s.proj, s.tanfovx,
                           s.tanfovy, s.H, s.W, s.shs if use_sh else None, deg, s.campos)


def oracle_backward(O, st, s, deg, alphas, colors_precomp=None, cov3D_precomp=None, track_off=False, map_off=False,
                    scale_modifier=1.0, grads=None):
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    gC, gD, gM, gV = grads if grads is not None else (s.gC, s.gD, s.gM, s.gV)
    return O.light_backward(st, s.bg, s.means, colors_precomp, s.scales if use_sr else None,
                            s.rots if use_sr else None, scale_modifier, cov3D_precomp, s.view, s.proj, s.tanfovx,
                            s.tanfovy, gC, gD, gM, gV, s.gt, s.shs if use_sh else None, deg, s.campos, alphas, s.persp,
                            track_off=track_off, map_off=map_off)


# ------------------------------------------------------------------------------------------ full variant
def hip_full_forward(s, deg, colors_precomp=None, cov3D_precomp=None):
    from dgr_amd import full as F
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    out = F._C.rasterize_gaussians(
        T(s.bg), T(s.means), E() if use_sh else T(colors_precomp), T(s.opac), T(s.scales) if use_sr else E(),
        T(s.rots) if use_sr else E(), 1.0, E() if use_sr else T(cov3D_precomp), T(s.view), T(s.gt), T(s.proj), s.tanfovx,
        s.tanfovy, s.H, s.W, T(s.shs) if use_sh else E(), deg, T(s.campos), False)
    names = ["num_rendered", "num_related", "color", "depth", "uncertainty", "radii", "geom", "binning", "img"]
    d = {n: (v if n in ("geom", "binning", "img") or not isinstance(v, torch.Tensor) else v.cpu().numpy())
         for n, v in zip(names, out)}
    return out, d


def hip_full_backward(s, deg, out, colors_precomp=None, cov3D_precomp=None, grads=None):
    from dgr_amd import full as F
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    (R, NG, color, depth, unc, radii, geom, binning, img) = out
    gC, gD, gU = grads if grads is not None else (s.gC, s.gD, s.gV)
    g = F._C.rasterize_gaussians_backward(
        T(s.bg), T(s.means), radii, E() if use_sh else T(colors_precomp), T(s.scales) if use_sr else E(),
        T(s.rots) if use_sr else E(), 1.0, E() if use_sr else T(cov3D_precomp), T(s.view), T(s.gt), T(s.proj), s.tanfovx,
        s.tanfovy, T(gC), T(gD[None]), T(gU[None]), T(s.shs) if use_sh else E(), deg, T(s.campos), geom, R, binning, img,
        NG, T(s.persp))
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations", "dL_dview"]
    return {n: v.cpu().numpy() for n, v in zip(names, g)}


def oracle_full(O, s, deg, colors_precomp=None, cov3D_precomp=None, grads=None, backward=True):
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    st, out = O.full_forward(s.bg, s.means, colors_precomp, s.opac, s.scales if use_sr else None,
                             s.rots if use_sr else None, 1.0, cov3D_precomp, s.view, s.gt, s.proj, s.tanfovx, s.tanfovy,
                             s.H, s.W, s.shs if use_sh else None, deg, s.campos)
    if not backward:
        return st, out, None
    return st, out, oracle_full_backward(O, st, s, deg, colors_precomp, cov3D_precomp, grads)


def oracle_full_backward(O, st, s, deg, colors_precomp=None, cov3D_precomp=None, grads=None):
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    gC, gD, gU = grads if grads is not None else (s.gC, s.gD, s.gV)
    return O.full_backward(st, s.bg, s.means, colors_precomp, s.scales if use_sr else None, s.rots if use_sr else None,
                           1.0, cov3D_precomp, s.view, s.gt, s.proj, s.tanfovx, s.tanfovy, gC, gD, gU,
                           s.shs if use_sh else None, deg, s.campos, s.persp)
