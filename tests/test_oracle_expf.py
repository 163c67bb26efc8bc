"""The oracle's exponential (oracle/dgr_oracle.cpp: expf_restated): glibc's expf algorithm written out in IEEE double
operations, so that "the oracle's bits" -- which the default alpha path of the HIP kernels reproduces, csrc/exact_math.h --
do not depend on the C library of the machine the tests run on.

  * known-answer vectors (committed here: bits of the result for fixed arguments, edge cases included) -- the same on every host;
  * against the host's expf where that is glibc >= 2.27: same table and cubic, summed in a different order (glibc:
    (C0 r + C1) r^2 + (C2 r + 1), fused or not as its build decided; here Horner with fused multiply-adds), so the two agree
    except where the double result lies within ~1e-16 of a rounding boundary of the float grid: about one argument in 2^28.
The GPU side of the same pin is tests/test_hip_exact_math.py (exp_ref against THIS function, bit for bit)."""
import ctypes as C

import numpy as np
import pytest

KNOWN = [  # (x, bits of expf_restated(x))
    (0.0, 0x3f800000), (-0.0, 0x3f800000), (-1.0000000031710769e-30, 0x3f800000), (-9.999999974752427e-07, 0x3f7fffef),
    (-0.5, 0x3f1b4598), (-1.0, 0x3ebc5ab2), (-2.8332133293151855, 0x3d70f0f1), (-2.079441547393799, 0x3e000000),
    (-10.0, 0x383e6bce), (-50.0, 0x1b692beb), (-86.9000015258789, 0x00c60f89), (-87.5, 0x006cb2bc), (-100.0, 0x0000001b),
    (-103.9000015258789, 0x00000001), (-103.9800033569336, 0x00000000), (-10000.0, 0x00000000),
    (-0.6931471824645996, 0x3f000000), (-3.1415927410125732, 0x3d310113), (-17.25, 0x330a7a4f), (-0.0009765625, 0x3f7fc008),
    (-7.0, 0x3a6f0b5d), (-23.5, 0x2e88ded2),
]


def test_known_answers(oracle):
    x = np.array([k[0] for k in KNOWN], np.float32)
    want = np.array([k[1] for k in KNOWN], np.uint32)
    got = oracle.expf_restated(x).view(np.uint32)
    assert np.array_equal(got, want), [(float(a), hex(int(b)), hex(int(c))) for a, b, c in zip(x, got, want) if b != c]


def test_special_values(oracle):
    y = oracle.expf_restated(np.array([np.nan, 89.0, 88.0, -np.inf, -200.0], np.float32))
    assert np.isnan(y[0]) and np.isinf(y[1]) and y[1] > 0 and np.isfinite(y[2]) and y[3] == 0.0 and y[4] == 0.0


def test_it_is_what_the_blend_loops_call(oracle):
    """The float build (the checker) routes the reference's unqualified exp() to the restated function."""
    oracle.use_cmath(False)
    x = np.random.default_rng(3).uniform(-20.0, 0.0, 4096).astype(np.float32)
    assert np.array_equal(oracle.exp_as_the_oracle_calls_it(x).view(np.uint32), oracle.expf_restated(x).view(np.uint32))


def test_against_the_hosts_expf(oracle):
    try:
        libc = C.CDLL("libc.so.6")
        libc.gnu_get_libc_version.restype = C.c_char_p
        ver = tuple(int(v) for v in libc.gnu_get_libc_version().decode().split(".")[:2])
        libm = C.CDLL("libm.so.6")
    except (OSError, AttributeError, ValueError):
        pytest.skip("not a glibc host")
    if ver < (2, 27):
        pytest.skip(f"glibc {ver}: expf predates the table algorithm")
    # a vectorised call into libm's expf: numpy's own float32 exp is a different (SIMD) implementation
    libm.expf.restype, libm.expf.argtypes = C.c_float, [C.c_float]
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-2.9, 0.0, 150000), rng.uniform(-103.0, 0.0, 50000), -np.exp(rng.uniform(-30, 1, 50000))]).astype(np.float32)
    host = np.fromiter((libm.expf(float(v)) for v in x), np.float32, x.size)
    got = oracle.expf_restated(x)
    bad = np.nonzero(host.view(np.uint32) != got.view(np.uint32))[0]
    assert bad.size <= 1, (bad.size, x[bad][:4], host[bad][:4], got[bad][:4])
