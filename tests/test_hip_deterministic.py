"""dgr_set_option("deterministic_grads", 1): the light backward without order-dependent float atomics (csrc/render_light.hip: DET).

The default backward -- like the reference's (L/cuda_rasterizer/backward.cu:593-596, 666-680) -- sums a Gaussian's per-tile totals
with float atomics in arrival order: two runs differ in the last bits (~1e-7 of a row, which computeCov2DCUDA's backward amplifies
to 4e-3 on single ill-conditioned rows), and a parity bar near that noise is set by luck.  With the option on
  * two runs of the same backward give the same BITS in every gradient tensor, including the pose gradient;
  * the sums are taken in a fixed order (waves of a tile, tiles of a Gaussian ascending, blocks ascending), so the distance to the
    oracle -- which sums in double and rounds once -- is arithmetic, not arrival order: the 1e-5-of-scale bars of the default path
    tighten to 2e-6 at configs 1-3;
  * the full variant and the batched entry points refuse the option (light variant, one-view backward only)."""
import numpy as np
import pytest
import torch

from dgr_amd import _capi
from util import make_scene
import hip_helpers as hh
from test_hip_light_parity import GRAD_NAMES, check_backward

pytestmark = pytest.mark.gpu

CONFIGS = [(10000, 256, 256, 0, 0), (100000, 640, 480, 3, 0), (500000, 1920, 1080, 3, 0)]  # BASELINE configs 1-3 (sizes)


@pytest.fixture
def deterministic():
    _capi.load()
    _capi.set_option("deterministic_grads", 1)
    yield
    _capi.set_option("deterministic_grads", 0)


def backward_twice(s, deg, **kw):
    grads = tuple(g * (s.W * s.H) ** 0.5 for g in (s.gC, s.gD, s.gM, s.gV))
    out, d = hh.hip_forward(s, deg)
    a = hh.hip_backward(s, deg, out, grads=grads, **kw)
    torch.cuda.synchronize()
    b = hh.hip_backward(s, deg, out, grads=grads, **kw)
    return a, b


@pytest.mark.parametrize("case", CONFIGS + [(3835, 16, 5, 3, 5)])
@pytest.mark.parametrize("mode", [dict(), dict(map_off=True), dict(track_off=True)])
def test_two_runs_give_the_same_bits(deterministic, case, mode):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    a, b = backward_twice(s, deg, **mode)
    for k in list(GRAD_NAMES) + ["dL_dview"]:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (k, int((a[k] != b[k]).sum()))
    if not mode.get("map_off"):
        assert np.abs(a["dL_dmeans3D"]).max() > 0
    if not mode.get("track_off"):
        assert np.abs(a["dL_dview"]).max() > 0


def test_the_default_backward_is_not_bit_reproducible_which_is_what_the_option_is_for():
    """Informational (never fails on equality): how far two default runs are apart at config 2's size."""
    s = make_scene(100000, 640, 480, 0)
    a, b = backward_twice(s, 3)
    worst = {k: float(np.abs(a[k].astype(np.float64) - b[k]).max() / max(np.abs(a[k]).max(), 1e-30)) for k in list(GRAD_NAMES) + ["dL_dview"]}
    print("\n[default backward, two runs, max |difference| / max |value|]", {k: "%.1e" % v for k, v in worst.items()})
    assert max(worst.values()) < 1e-4


@pytest.mark.parametrize("case", CONFIGS)
def test_the_oracle_bar_tightens(deterministic, oracle, case):
    P, W, H, deg, seed = case
    s = make_scene(P, W, H, seed)
    check_backward(oracle, s, deg, rel_to_max=2e-6, view_rel_to_max=2e-6, what=f"deterministic P={P}")


def test_one_tile_frame_at_the_tight_bar(deterministic, oracle):
    """3 835 Gaussians on 16 x 5 pixels -- one tile, every Gaussian's total a single row: the frame whose default-path error
    (1.02e-5, arrival order of one row's float atomics) moved tests/test_hip_random_sweep.py's bar to 2e-5.  Deterministic: 2.1e-6
    (float sums of up to 80 pixels in the butterfly's order against the oracle's double sums), the same on every run."""
    s = make_scene(3835, 16, 5, 5)
    check_backward(oracle, s, 3, rel_to_max=4e-6, view_rel_to_max=4e-6, what="deterministic one-tile frame")


def test_the_other_entry_points_refuse_the_option(deterministic):
    from dgr_amd import full as F  # noqa: F401
    s = make_scene(2000, 64, 48, 1)
    out, d = hh.hip_full_forward(s, 0)
    with pytest.raises(RuntimeError, match="deterministic_grads"):
        hh.hip_full_backward(s, 0, out)
    import test_hip_batch as tb
    ss = tb.scenes(2000, 64, 48, 2, 0)
    outb, cams = tb.batch_forward(ss, 0)
    grads = [(x.gC, x.gD, x.gM, x.gV) for x in ss]
    with pytest.raises(RuntimeError, match="deterministic_grads"):
        tb.batch_backward(ss, 0, outb, cams, grads)
