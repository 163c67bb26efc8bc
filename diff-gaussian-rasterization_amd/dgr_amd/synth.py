"""synth-v1: the seeded synthetic scene generator every test and bench.py shares.

Definition: SURVEY.md Appendix C (numpy PCG64, draw order z, xc, yc, sig_px, q, opac,
shs, gt, gC, gD, gM, gV).  Matrix conventions follow the reference's column-major reads
(`cuda_rasterizer/auxiliary.h:58-77`): the tensors hold W2C^T, (Proj*W2C)^T and Proj^T.
"""
import hashlib
from typing import NamedTuple

import numpy as np


class Scene(NamedTuple):
    P: int
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    view: np.ndarray      # [4,4] f32  viewmatrix  (W2C^T)
    proj: np.ndarray      # [4,4] f32  projmatrix  ((Proj W2C)^T)
    persp: np.ndarray     # [4,4] f32  perspec_matrix (Proj^T)
    campos: np.ndarray    # [3]
    means: np.ndarray     # [P,3]
    scales: np.ndarray    # [P,3]
    rots: np.ndarray      # [P,4] (r,x,y,z)
    opac: np.ndarray      # [P,1]
    shs: np.ndarray       # [P,16,3]
    gt: np.ndarray        # [H,W]
    bg: np.ndarray        # [3]
    gC: np.ndarray        # [3,H,W] dL/dcolor
    gD: np.ndarray        # [H,W]   dL/ddepth
    gM: np.ndarray        # [H,W]   dL/dmedian depth (light)
    gV: np.ndarray        # [H,W]   dL/ddepth_var (light) / dL/duncertainty (full)


def camera(W, H, angle=0.05, tanfovx=0.6):
    tanfovy = tanfovx * H / W
    znear, zfar = 0.01, 100.0
    axis = np.array([0.2, 1.0, 0.1])
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rm = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K
    t = np.array([0.05, -0.02, 0.10])
    W2C = np.eye(4)
    W2C[:3, :3] = Rm
    W2C[:3, 3] = t
    Pm = np.zeros((4, 4))
    Pm[0, 0] = 1 / tanfovx
    Pm[1, 1] = 1 / tanfovy
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    Pm[3, 2] = 1
    view = W2C.T.astype(np.float32)
    persp = Pm.T.astype(np.float32)
    proj = (W2C.T @ Pm.T).astype(np.float32)
    campos = (-Rm.T @ t).astype(np.float32)
    return tanfovx, tanfovy, Rm, t, view, proj, persp, campos


def make_scene(P, W, H, seed=0, view_index=0):
    """view_index k>0 re-poses the camera at angle 0.05*(k+1) over the same Gaussians."""
    rng = np.random.default_rng(seed)
    tanfovx, tanfovy, Rm, t, view, proj, persp, campos = camera(W, H, 0.05)
    z = rng.uniform(1.0, 6.0, P)
    xc = rng.uniform(-1.1, 1.1, P) * tanfovx * z
    yc = rng.uniform(-1.1, 1.1, P) * tanfovy * z
    means = ((np.stack([xc, yc, z], 1) - t) @ Rm).astype(np.float32)
    sig_px = np.exp(rng.uniform(np.log(0.7), np.log(4.0), (P, 3)))
    scales = (sig_px * (2 * tanfovx / W) * z[:, None]).astype(np.float32)
    q = rng.normal(size=(P, 4))
    rots = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opac = rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)
    shs = rng.normal(size=(P, 16, 3))
    shs[:, 0, :] *= 0.5
    shs[:, 1:, :] *= 0.1
    shs = shs.astype(np.float32)
    gt = rng.uniform(1.0, 6.0, (H, W)).astype(np.float32)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    gC = (rng.normal(size=(3, H, W)) / (H * W)).astype(np.float32)
    gD = (rng.normal(size=(H, W)) / (H * W)).astype(np.float32)
    gM = (rng.normal(size=(H, W)) / (H * W)).astype(np.float32)
    gV = (rng.normal(size=(H, W)) / (H * W)).astype(np.float32)
    if view_index:
        tanfovx, tanfovy, Rm, t, view, proj, persp, campos = camera(W, H, 0.05 * (view_index + 1))
    return Scene(P, W, H, tanfovx, tanfovy, view, proj, persp, campos, means, scales, rots,
                 opac, shs, gt, bg, gC, gD, gM, gV)


def cluster_scene(s: Scene, frac=0.6, shrink=0.35, shift=(0.5, 0.3), seed=1) -> Scene:
    """A non-uniform variant of a synth-v1 scene (not a BASELINE configuration): `frac` of the Gaussians are pulled towards
    one region of the frame (their x, y scaled by `shrink` about the group's centroid and moved by `shift`), so that tile
    lists range from a few dozen to over a thousand entries at config 3's size (mean 222, max 1135) instead of 203 +- 20 %.
    Used by the schedule tests and by `bench.py --scene clustered`."""
    rng = np.random.default_rng(seed)
    pick = rng.random(s.P) < frac
    m = s.means.copy()
    c = m[pick].mean(axis=0)
    m[pick, :2] = c[:2] + shrink * (m[pick, :2] - c[:2]) + np.asarray(shift, np.float32)
    return s._replace(means=m.astype(np.float32))


def heavy_tail_scene(s: Scene, frac=0.01, sigma_px=(20.0, 150.0), seed=2) -> Scene:
    """A heavy-tailed variant of a synth-v1 scene (not a BASELINE configuration): `frac` of the Gaussians get an isotropic
    extent whose standard deviation ON SCREEN is log-uniform in `sigma_px` pixels (3-sigma rectangles of 8 .. 56 tiles a side,
    up to the whole frame at small sizes); everything else is synth-v1 (sigma 0.7 .. 4 px).  What a SLAM map looks like to the
    front end: a few splats that touch hundreds or thousands of tiles each among many that touch three -- the case the
    reference's duplicateWithKeys walks with ONE thread per Gaussian (L/cuda_rasterizer/rasterizer_impl.cu:70-111).  At config
    3's size 1 % of the Gaussians then own ~70 % of the tile instances.  Used by tests/test_hip_heavy_tail.py and
    `bench.py --scene heavy_tail`."""
    rng = np.random.default_rng(seed)
    pick = rng.random(s.P) < frac
    n = int(pick.sum())
    z = (np.concatenate([s.means, np.ones((s.P, 1), np.float32)], 1).astype(np.float64) @ s.view.astype(np.float64))[:, 2]
    sig = np.exp(rng.uniform(np.log(sigma_px[0]), np.log(sigma_px[1]), n))
    sc = s.scales.copy()
    # (isotropic up to +-10 %: the extent on screen must not depend on the rotation drawn for the Gaussian)
    sc[pick] = (sig[:, None] * rng.uniform(0.9, 1.1, (n, 3)) * (2 * s.tanfovx / s.W) * np.maximum(z[pick], 0.2)[:, None]).astype(np.float32)
    return s._replace(scales=sc)


def sha16(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
