"""The wave64 multi-value butterfly reductions (csrc/wave_reduce.h) against plain sums.

They rest on v_permlane32_swap / v_permlane16_swap emitted as inline asm, so they get their own check:
random data, plus one-hot inputs that would expose any lane/row permutation error."""
import numpy as np
import pytest
import torch

from dgr_amd import _capi

pytestmark = pytest.mark.gpu


def run(x):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    o16 = torch.zeros(64, device=dev)
    o4 = torch.zeros(64, device=dev)
    c16 = torch.zeros(64, dtype=torch.int32, device=dev)
    c4 = torch.zeros(64, dtype=torch.int32, device=dev)
    rc = lib.dgr_debug_wave_reduce(_capi.stream_handle(), t.data_ptr(), o16.data_ptr(), o4.data_ptr(), c16.data_ptr(),
                                   c4.data_ptr())
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return o16.cpu().numpy(), o4.cpu().numpy(), c16.cpu().numpy(), c4.cpu().numpy()


def test_component_maps_cover_every_value():
    _, _, c16, c4 = run(np.zeros((16, 64)))
    assert sorted(set(c16.tolist())) == list(range(16))
    assert all(len(set(c16[q * 4:(q + 1) * 4])) == 1 for q in range(16))  # one value per lane quad
    assert sorted(set(c4.tolist())) == list(range(4))
    assert all(len(set(c4[r * 16:(r + 1) * 16])) == 1 for r in range(4))  # one value per 16-lane row


def test_random_integers_are_summed_exactly():
    rng = np.random.default_rng(0)
    x = rng.integers(-64, 64, size=(16, 64)).astype(np.float32)  # integer-valued: every order sums exactly
    o16, o4, c16, c4 = run(x)
    assert np.array_equal(o16, x.sum(1)[c16])
    assert np.array_equal(o4, x[:4].sum(1)[c4])


@pytest.mark.parametrize("comp", range(16))
def test_one_hot_lane_and_component(comp):
    for lane in (0, 5, 17, 31, 32, 46, 63):
        x = np.zeros((16, 64), np.float32)
        x[comp, lane] = 3.0
        o16, o4, c16, c4 = run(x)
        assert np.array_equal(o16, np.where(c16 == comp, 3.0, 0.0))
        if comp < 4:
            assert np.array_equal(o4, np.where(c4 == comp, 3.0, 0.0))


# ---- round 3: the networks with the within-row DPP stages first (wave_reduce16d / wave_reduce12d), the ones the blend
# kernels use
def run_d(x):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    o16 = torch.zeros(128, device=dev)
    o12 = torch.zeros(128, device=dev)
    c16 = torch.zeros(128, dtype=torch.int32, device=dev)
    c12 = torch.zeros(128, dtype=torch.int32, device=dev)
    rc = lib.dgr_debug_wave_reduce_d(_capi.stream_handle(), t.data_ptr(), o16.data_ptr(), o12.data_ptr(), c16.data_ptr(),
                                     c12.data_ptr())
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return o16.cpu().numpy()[64:], o12.cpu().numpy()[64:], c16.cpu().numpy()[64:], c12.cpu().numpy()[64:]


def test_dpp_first_component_maps():
    _, _, c16, c12 = run_d(np.zeros((16, 64)))
    assert sorted(set(c16.tolist())) == list(range(16))
    assert sorted(set(c12.tolist())) == list(range(12))
    for c in (c16, c12):
        assert all(len(set(c[q * 4:(q + 1) * 4])) == 1 for q in range(16))  # one value per lane quad
    assert np.array_equal(c12[32:48], c12[48:64])  # rows 2 and 3 hold the same four totals (values 8..11)


def test_dpp_first_random_integers_are_summed_exactly():
    rng = np.random.default_rng(1)
    x = rng.integers(-64, 64, size=(16, 64)).astype(np.float32)
    o16, o12, c16, c12 = run_d(x)
    assert np.array_equal(o16, x.sum(1)[c16])
    assert np.array_equal(o12, x[:12].sum(1)[c12])


@pytest.mark.parametrize("comp", range(16))
def test_dpp_first_one_hot_lane_and_component(comp):
    for lane in (0, 3, 5, 12, 17, 31, 32, 46, 50, 63):
        x = np.zeros((16, 64), np.float32)
        x[comp, lane] = 3.0
        o16, o12, c16, c12 = run_d(x)
        assert np.array_equal(o16, np.where(c16 == comp, 3.0, 0.0))
        if comp < 12:
            assert np.array_equal(o12, np.where(c12 == comp, 3.0, 0.0))
        else:
            assert not o12.any()


# ---- the 16-lane row reduction of the rows backward (csrc/render_light_rows.hip): twelve values per row of 16 lanes
def run_rows(x):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    out = torch.zeros(64, device=dev)
    comp = torch.zeros(64, dtype=torch.int32, device=dev)
    rc = lib.dgr_debug_row_reduce(_capi.stream_handle(), t.data_ptr(), out.data_ptr(), comp.data_ptr())
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return out.cpu().numpy(), comp.cpu().numpy()


def test_row_reduce_sums_every_row_separately():
    rng = np.random.default_rng(1)
    x = rng.integers(-64, 64, size=(12, 64)).astype(np.float32)
    out, comp = run_rows(x)
    for r in range(4):
        c = comp[16 * r:16 * r + 16]
        assert sorted(v for v in c.tolist() if v >= 0) == list(range(12))  # every value has exactly one delivering lane
        want = x[:, 16 * r:16 * r + 16].sum(1)
        for lane in range(16):
            if c[lane] >= 0:
                assert out[16 * r + lane] == want[c[lane]], (r, lane)


@pytest.mark.parametrize("comp_idx", range(12))
def test_row_reduce_one_hot(comp_idx):
    for lane in (0, 3, 7, 8, 13, 15, 16, 37, 63):
        x = np.zeros((12, 64), np.float32)
        x[comp_idx, lane] = 5.0
        out, comp = run_rows(x)
        row = lane >> 4
        for q in range(64):
            if comp[q] < 0:
                continue
            assert out[q] == (5.0 if (q >> 4) == row and comp[q] == comp_idx else 0.0), (lane, q)
