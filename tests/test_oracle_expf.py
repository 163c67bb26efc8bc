"""The oracle's two exponentials (oracle/dgr_oracle.cpp), both written out in IEEE operations so that "the oracle's bits" -- which
the HIP kernels' exact alpha paths reproduce, csrc/exact_math.h -- do not depend on the C library of the machine the tests run on:

expf_p32 (the blend loops' default since round 8; the kernels' alpha_mode 0): an fp32-only expf -- magic-shift rounding, two fused
Cody-Waite steps, a degree-6 polynomial, an exponent-field add:
  * known-answer vectors, special values;
  * an EXHAUSTIVE scan of all 1 120 927 745 floats of [-104, -0] against exp() in double: within 1 ulp everywhere (measured 0.892,
    0.858 where the result is denormal), more than 99.5 % correctly rounded;
  * it is what the blend loops call, and set_exp_mode(1) switches them to the other one.

expf_restated (set_exp_mode(1); the kernels' alpha_mode 2; rounds 5-7's default): glibc's expf algorithm in IEEE double operations:

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
    """The float build (the checker) routes the reference's unqualified exp() to expf_p32, or to expf_restated on request."""
    oracle.use_cmath(False)
    x = np.random.default_rng(3).uniform(-20.0, 0.0, 4096).astype(np.float32)
    assert oracle.set_exp_mode(0) == 0  # the default
    assert np.array_equal(oracle.exp_as_the_oracle_calls_it(x).view(np.uint32), oracle.expf_p32(x).view(np.uint32))
    try:
        oracle.set_exp_mode(1)
        assert np.array_equal(oracle.exp_as_the_oracle_calls_it(x).view(np.uint32), oracle.expf_restated(x).view(np.uint32))
    finally:
        assert oracle.set_exp_mode(0) == 1
    # the two differ (in the last bit, on a fraction of a percent of the arguments): the mode is not a no-op
    assert np.count_nonzero(oracle.expf_p32(x).view(np.uint32) != oracle.expf_restated(x).view(np.uint32)) > 0


KNOWN_P32 = [  # (x, bits of expf_p32(x)); the last rows are arguments where it is NOT the correctly rounded result
    (0.0, 0x3f800000), (-0.0, 0x3f800000), (-1.0000000031710769e-30, 0x3f800000), (-9.999999974752427e-07, 0x3f7fffef),
    (-0.5, 0x3f1b4598), (-1.0, 0x3ebc5ab2), (-2.8332133293151855, 0x3d70f0f1), (-2.079441547393799, 0x3e000000),
    (-10.0, 0x383e6bce), (-50.0, 0x1b692beb), (-86.9000015258789, 0x00c60f89), (-87.5, 0x006cb2bc), (-100.0, 0x0000001b),
    (-103.9000015258789, 0x00000001), (-103.9800033569336, 0x00000000), (-104.0, 0x00000000), (-10000.0, 0x00000000),
    (-0.6931471824645996, 0x3f000000), (-3.1415927410125732, 0x3d310113), (-17.25, 0x330a7a4f), (-0.0009765625, 0x3f7fc008),
    (-7.0, 0x3a6f0b5d), (-23.5, 0x2e88ded2),
]


def test_p32_known_answers(oracle):
    x = np.array([k[0] for k in KNOWN_P32], np.float32)
    want = np.array([k[1] for k in KNOWN_P32], np.uint32)
    got = oracle.expf_p32(x).view(np.uint32)
    assert np.array_equal(got, want), [(float(a), hex(int(b)), hex(int(c))) for a, b, c in zip(x, got, want) if b != c]
    # where it differs from the correctly rounded result it is the neighbouring float (the worst arguments of the exhaustive scan)
    worst = np.array([0xC0BB2813, 0xC2AF647B], np.uint32).view(np.float32)  # -5.848642 (0.891 ulp), -87.696251 (0.858 ulp, denormal)
    y = oracle.expf_p32(worst)
    t = np.exp(worst.astype(np.float64))
    assert np.all(np.abs(y.astype(np.float64) - t) <= np.maximum(np.spacing(t.astype(np.float32)).astype(np.float64), 2.0 ** -149))


def test_p32_special_values(oracle):
    y = oracle.expf_p32(np.array([np.nan, -np.inf, -200.0, -3e38, -104.0, -103.98], np.float32))
    assert np.all(y == 0.0)  # arguments below -104 and NaN are evaluated at -104 (v_max_f32's NaN rule): exactly 0


def test_p32_is_within_one_ulp_of_exp_on_every_float_of_its_range(oracle):
    """All floats from -0.0 down to -104.0 (1.12e9 arguments, a few seconds with OpenMP) against exp() in double."""
    r = oracle.expf_p32_scan(0x80000000, 0xC2D00000)
    print("\n[expf_p32, exhaustive]", r)
    assert r["scanned"] == 0xC2D00000 - 0x80000000 + 1
    assert r["max_ulp_normal"] < 0.9 and r["max_ulp_denormal"] < 0.9
    assert r["not_correctly_rounded"] < 0.005 * r["scanned"]
    # and the positive zero / tiny positive arguments a blend loop can still pass (power == +0)
    assert oracle.expf_p32(np.array([0.0], np.float32))[0] == 1.0


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
