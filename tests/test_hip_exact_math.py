"""csrc/exact_math.h against the CPU restatement: exp_ref must return the bits of oracle/dgr_oracle.cpp's expf_restated -- the
same IEEE double operation sequence, so the agreement does not depend on the C library of the box (tests/test_oracle_expf.py
pins that function itself) -- and div_ref the correctly rounded quotient, over the whole range the blend kernels use them
on.  The default alpha path rests on these two: the light backward amplifies a last-bit difference of one
alpha by 1 / T_final and by alpha / (1 - alpha) per division."""
import numpy as np
import pytest
import torch

from dgr_amd import _capi

pytestmark = pytest.mark.gpu


def oracle_expf(x):
    from oracle import oracle as O
    O.use_cmath(False)
    return O.exp_as_the_oracle_calls_it(x)


def device(x, a, b):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    tx, ta, tb = (torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev) for v in (x, a, b))
    oe, od = torch.empty_like(tx), torch.empty_like(tx)
    rc = lib.dgr_debug_exact_math(_capi.stream_handle(), tx.numel(), tx.data_ptr(), ta.data_ptr(), tb.data_ptr(),
                                  oe.data_ptr(), od.data_ptr())
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return oe.cpu().numpy(), od.cpu().numpy()


def test_exp_ref_returns_the_restatements_bits():
    rng = np.random.default_rng(0)
    n = 1 << 20
    # the blend loops evaluate exp on [ln(15/255), 0]; the function supports (-87, 0]
    # the blend loops evaluate exp on [ln(15/255), 0] in the forward and on anything <= 0 in the backward (a needle-shaped
    # Gaussian seen from far off its axis: -1e4 and below; the kernel clamps the argument at -104, where the result is 0)
    x = np.concatenate([rng.uniform(-2.9, 0.0, n), rng.uniform(-87.0, 0.0, n // 4), -np.exp(rng.uniform(-30, 1, n // 4)),
                        -np.exp(rng.uniform(np.log(104.5), np.log(3e38), n // 8)),
                        [0.0, -0.0, -1e-30, -2.8332133, -86.9, -104.5, -708.0, -709.0, -1e4, -3e38]]).astype(np.float32)
    got, _ = device(x, np.ones_like(x), np.ones_like(x))
    want = oracle_expf(x)
    bad = np.nonzero(got.view(np.int32) != want.view(np.int32))[0]
    assert bad.size == 0, (bad.size, x[bad][:5], got[bad][:5], want[bad][:5])  # the same operations: no allowance
    # results between the smallest normal float and 0 (arguments in [-104, -87.3]) are subnormal in the restatement; the
    # kernels may flush them -- either way alpha = o * that is far below 15/255
    xs = rng.uniform(-104.0, -87.4, 4096).astype(np.float32)
    gs, _ = device(xs, np.ones_like(xs), np.ones_like(xs))
    ws = oracle_expf(xs)
    assert np.all((gs == ws) | (gs == 0.0)) and np.all(ws < 1.2e-38)


def test_div_ref_is_the_correctly_rounded_quotient():
    rng = np.random.default_rng(1)
    n = 1 << 22
    T = np.exp(rng.uniform(np.log(1e-5), 0.0, n)).astype(np.float32)         # transmittances
    om = (1.0 - rng.uniform(15.0 / 255.0, 0.99, n)).astype(np.float32)      # 1 - alpha
    om[: n // 8] = np.float32(1.0) - np.float32(0.99)                       # the clamp value, a frequent divisor
    _, got = device(np.zeros_like(T), T, om)
    want = (T.astype(np.float64) / om.astype(np.float64)).astype(np.float32)  # correctly rounded: the double quotient of two
    #   floats rounds to float without a double-rounding error (2 p + 2 <= 53)
    assert np.array_equal(want, T / om)
    bad = np.nonzero(got.view(np.int32) != want.view(np.int32))[0]
    assert bad.size <= 2, (bad.size, T[bad][:5], om[bad][:5])
