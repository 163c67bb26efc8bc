"""Shared helpers for the parity tests."""
import numpy as np

from dgr_amd.synth import make_scene  # noqa: F401


def frac_outside(a, ref, atol, rtol):
    """Fraction of elements with |a-ref| > atol*max(1,|ref|)... generalised: atol + rtol*|ref|."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.mean(np.abs(a - ref) > (atol + rtol * np.abs(ref))))


def assert_image_close(a, ref, name, tol=1e-5, max_outliers=1e-4):
    """north_star tolerance: |d| <= 1e-5 * max(1, |ref|) per value, for all but a bounded fraction of
    values.  The outlier budget exists because one-ulp differences in exp() flip the reference's hard
    thresholds (alpha < 15/255, T < 1e-4) for a handful of pixel/Gaussian pairs per frame; the reference
    itself moves by this much between FMA-on and FMA-off builds (SURVEY.md s7 "Hard parts")."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    bad = np.abs(a - ref) > tol * np.maximum(1.0, np.abs(ref))
    frac = float(bad.mean())
    assert frac <= max_outliers, f"{name}: {frac:.2e} of values off by > {tol} (max |d| = {np.abs(a - ref).max():.3e})"


def assert_grad_close(a, ref, name, rel_to_max=2e-4, elem_rtol=2e-3, elem_frac=2e-3, outlier_rows=0):
    """Gradients are sums over many pixels of float atomics (order-nondeterministic in the reference too)
    and inherit the forward's threshold flips: bound the worst element relative to the tensor's scale,
    and the fraction of elements that miss a per-element relative tolerance.
    `outlier_rows`: rows (Gaussians) allowed to miss the worst-element bound.  One (pixel, Gaussian) pair whose alpha sits
    within rounding of 15/255 is blended by one implementation and skipped by the other -- a whole term of that
    Gaussian's sums; among the 4e8 pair evaluations of a 1080p frame that happens to a handful of Gaussians."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max()
    if scale == 0:
        assert np.abs(a).max() == 0, f"{name}: reference is all zero, got max {np.abs(a).max():.3e}"
        return
    row_err = np.abs(a - ref).reshape(a.shape[0], -1).max(1) / scale if a.ndim > 1 else np.abs(a - ref) / scale
    n_out = int((row_err > rel_to_max).sum())
    assert n_out <= outlier_rows, (f"{name}: {n_out} rows with max |d| / max |ref| > {rel_to_max} "
                                   f"(worst {row_err.max():.3e}, allowed rows {outlier_rows})")
    frac = float(np.mean(np.abs(a - ref) > (1e-7 * scale + elem_rtol * np.abs(ref))))
    assert frac <= elem_frac, f"{name}: {frac:.2e} of elements off by > {elem_rtol} relative"


FLIP_LOG = []  # (test id, pixels masked, pixels) of every gradient comparison that went through mask_flipped_pixels


def mask_flipped_pixels(grads, n_contrib_a, n_contrib_b, W, H, what="", images=(), median_margin=None):
    """Pixels on which the two implementations decided a hard threshold differently walk different lists in the two
    backward passes: a flipped termination (`T < 1e-4`, seen as a different n_contrib) or a flipped alpha >= 15/255 test
    in the middle of the list (seen as an image value off by far more than the 1e-5 bar; `images` = [(a, ref), ...] of
    shape [C, H, W]).  Instead of skipping the gradient comparison or granting outlier rows, zero the pixel-gradient
    images at exactly those pixels for BOTH implementations: a pixel whose incoming gradients are all zero contributes
    exactly 0 to every output gradient, so everything else is still compared, with no allowance.  The number of masked
    pixels is bounded (the forward tests bound the same fraction) and logged.
    `median_margin` ([H, W], oracle.light_median_margin): the backward re-finds the median-depth Gaussian from a
    transmittance it reconstructs by up to ~150 divisions (every blended alpha is >= 15/255 and T stays >= 1e-4), so two
    correct implementations may pick different Gaussians when some T_k is within that reconstruction error of 0.5 -- and
    no image shows it when the two have the same depth.  Such pixels (margin < 1e-5: 150 steps x 2 ulp x 0.5) are masked
    too, under their own bound (T steps are >= 0.03 wide near 0.5, so at most ~1e-3 of the pixels qualify)."""
    a = np.asarray(n_contrib_a).reshape(H, W)
    b = np.asarray(n_contrib_b).reshape(H, W)
    bad = a != b
    for x, ref in images:
        x = np.asarray(x, np.float64).reshape(-1, H, W)
        ref = np.asarray(ref, np.float64).reshape(-1, H, W)
        bad |= (np.abs(x - ref) > 1e-5 * np.maximum(1.0, np.abs(ref))).any(0)
    n = int(bad.sum())
    FLIP_LOG.append((what, n, W * H))
    assert n <= max(2, int(3e-4 * W * H)), f"{what}: the forward passes disagree on {n} of {W * H} pixels"
    if median_margin is not None:
        close = np.asarray(median_margin).reshape(H, W) < 1e-5
        assert int(close.sum()) <= max(4, int(2e-3 * W * H)), f"{what}: {int(close.sum())} pixels with T within 1e-5 of 0.5"
        bad = bad | close
        n = int(bad.sum())
    if n == 0:
        return tuple(grads), 0
    out = []
    for g in grads:
        g = np.array(g, dtype=np.float32, copy=True)
        g[..., bad] = 0.0
        out.append(g)
    return tuple(out), n
