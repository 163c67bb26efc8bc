"""csrc/exact_math.h against the HOST: exp_ref must return the bits of the C library's expf -- the function the CPU
restatement (oracle/dgr_oracle.cpp) calls -- and div_ref the correctly rounded quotient, over the whole range the blend
kernels use them on.  The default alpha path rests on these two: the light backward amplifies a last-bit
difference of one alpha by 1 / T_final and by alpha / (1 - alpha) per division."""
import numpy as np
import pytest
import torch

from dgr_amd import _capi

pytestmark = pytest.mark.gpu


def host_expf(x):
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


def test_exp_ref_returns_the_hosts_bits():
    rng = np.random.default_rng(0)
    n = 1 << 20
    # the blend loops evaluate exp on [ln(15/255), 0]; the function supports (-87, 0]
    x = np.concatenate([rng.uniform(-2.9, 0.0, n), rng.uniform(-87.0, 0.0, n // 4), -np.exp(rng.uniform(-30, 1, n // 4)),
                        [0.0, -0.0, -1e-30, -2.8332133, -86.9]]).astype(np.float32)
    got, _ = device(x, np.ones_like(x), np.ones_like(x))
    want = host_expf(x)
    bad = np.nonzero(got.view(np.int32) != want.view(np.int32))[0]
    # (a float result can differ only where the double result lies within ~1e-16 of a rounding boundary: ~1 in 2^28)
    assert bad.size <= 1, (bad.size, x[bad][:5], got[bad][:5], want[bad][:5])


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
