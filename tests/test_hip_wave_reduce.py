"""The wave64 multi-value butterfly reductions (csrc/wave_reduce.h) against plain sums.

They rest on v_permlane32_swap / v_permlane16_swap emitted as inline asm and on DPP adds written through bank masks, so they
get their own check: random data, plus one-hot inputs that would expose any lane / row permutation error."""
import numpy as np
import pytest
import torch

from dgr_amd import _capi

pytestmark = pytest.mark.gpu


def run(x):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    outs = [torch.zeros(64, device=dev) for _ in range(3)]
    comps = [torch.zeros(64, dtype=torch.int32, device=dev) for _ in range(3)]
    rc = lib.dgr_debug_wave_reduce(_capi.stream_handle(), t.data_ptr(), *[o.data_ptr() for o in outs],
                                   *[c.data_ptr() for c in comps])
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs], [c.cpu().numpy() for c in comps]


def test_component_maps_cover_every_value():
    _, (c16, c12, c4) = run(np.zeros((16, 64)))
    assert sorted(set(c16.tolist())) == list(range(16))
    assert sorted(set(c12.tolist())) == list(range(12))
    for c in (c16, c12):
        assert all(len(set(c[q * 4:(q + 1) * 4])) == 1 for q in range(16))  # one value per lane quad
    assert np.array_equal(c12[32:48], c12[48:64])  # rows 2 and 3 hold the same four totals (values 8..11)
    assert sorted(set(c4.tolist())) == list(range(4))
    assert all(len(set(c4[r * 16:(r + 1) * 16])) == 1 for r in range(4))  # one value per 16-lane row


def test_random_integers_are_summed_exactly():
    rng = np.random.default_rng(0)
    x = rng.integers(-64, 64, size=(16, 64)).astype(np.float32)  # integer-valued: every order sums exactly
    (o16, o12, o4), (c16, c12, c4) = run(x)
    assert np.array_equal(o16, x.sum(1)[c16])
    assert np.array_equal(o12, x[:12].sum(1)[c12])
    assert np.array_equal(o4, x[:4].sum(1)[c4])


@pytest.mark.parametrize("comp", range(16))
def test_one_hot_lane_and_component(comp):
    for lane in (0, 3, 5, 12, 17, 31, 32, 46, 50, 63):
        x = np.zeros((16, 64), np.float32)
        x[comp, lane] = 3.0
        (o16, o12, o4), (c16, c12, c4) = run(x)
        assert np.array_equal(o16, np.where(c16 == comp, 3.0, 0.0))
        assert np.array_equal(o12, np.where(c12 == comp, 3.0, 0.0) if comp < 12 else np.zeros(64))
        assert np.array_equal(o4, np.where(c4 == comp, 3.0, 0.0) if comp < 4 else np.zeros(64))
