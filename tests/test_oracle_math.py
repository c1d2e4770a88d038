"""es_math.h (shared by oracle and kernels) must equal the host C library bit for bit, and
NumPy's logaddexp loop (what rtwm/fastpolar.py:18-23 evaluates) must equal our formula."""
import ctypes

import numpy as np


def _libm():
    m = ctypes.CDLL("libm.so.6")
    for f in ("exp", "log1p"):
        getattr(m, f).restype = ctypes.c_double
        getattr(m, f).argtypes = [ctypes.c_double]
    return m


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def test_exp_matches_libm(oracle):
    m = _libm()
    rng = np.random.default_rng(11)
    x = np.concatenate([
        -np.abs(rng.normal(0, 8, 150_000)), -rng.uniform(0, 760, 60_000), -rng.uniform(500, 1100, 10_000),
        -np.ldexp(rng.uniform(0.5, 1, 20_000), -rng.integers(0, 70, 20_000)),
        np.array([0.0, -0.0, -1e-300, -745.2, -745.14, -708.4, -709.8, -1023.99, -1024.0, -5000.0, -np.inf, -512.0])])
    ref = np.array([m.exp(float(v)) for v in x])
    assert np.array_equal(_bits(oracle.exp_vec(x)), _bits(ref))


def test_log1p_matches_libm(oracle):
    m = _libm()
    rng = np.random.default_rng(12)
    x = np.concatenate([
        rng.uniform(0, 1, 150_000), np.exp(-np.abs(rng.normal(0, 8, 60_000))),
        np.ldexp(rng.uniform(0.5, 1, 20_000), -rng.integers(0, 80, 20_000)), rng.uniform(-0.9, 4.0, 20_000),
        np.array([0.0, 1.0, 2.0 ** -54, 2.0 ** -29, 0.41421356237309503, 0.4142135623730951, 0.5, 1e-320])])
    ref = np.array([m.log1p(float(v)) for v in x])
    assert np.array_equal(_bits(oracle.log1p_vec(x)), _bits(ref))


def test_logaddexp_and_f_match_numpy(oracle):
    rng = np.random.default_rng(13)
    a = np.clip(rng.normal(0, 6, 200_000), -12, 12)
    b = np.clip(rng.normal(0, 6, 200_000), -12, 12)
    a[:1000] = b[:1000]                         # x == y branch
    a[1000:2000] = np.round(a[1000:2000])       # many exact ties / integers
    b[1000:2000] = np.round(b[1000:2000])
    big_a = rng.normal(0, 3000, 50_000); big_b = rng.normal(0, 3000, 50_000)    # deep-tree magnitudes
    a = np.concatenate([a, big_a]); b = np.concatenate([b, big_b])
    assert np.array_equal(_bits(oracle.logaddexp_vec(a, b)), _bits(np.logaddexp(a, b)))
    f_ref = np.logaddexp(a, b) - np.logaddexp(0.0, a + b)          # rtwm/fastpolar.py:23
    assert np.array_equal(_bits(oracle.polar_f_vec(a, b)), _bits(f_ref))


def test_penalty_close_to_numpy(oracle):
    """np.exp / np.log1p on scalars may use AVX-512 / SVML kernels (<= 1 ulp from libm); the
    oracle uses libm-exact arithmetic, so equality is to 2 ulp here and exact in glibc mode
    (covered by the golden metrics of tests/test_oracle_polar.py)."""
    rng = np.random.default_rng(14)
    l = np.clip(rng.normal(0, 5, 20_000), -12, 12)
    for bit in (0, 1):
        pen = np.log1p(np.exp(-np.abs(l)))
        pref = (l >= 0).astype(int)
        ref = np.where(pref != bit, pen + np.abs(l), pen)
        got = oracle.penalty_vec(l, bit)
        assert np.allclose(got, ref, rtol=5e-16, atol=0)


def test_softplus_of_exact_zero(oracle):
    """t == 0 (two LLRs clipped to the same value meet in f, or a leaf LLR is exactly 0) is answered by a constant
    in the branch-free softplus: it must be libm's log1p(exp(0)) = log1p(1), for +0 and -0 and for |t| < 2^-54."""
    import math
    want = np.float64(math.log1p(math.exp(-0.0)))
    for l in (0.0, -0.0, 2.0 ** -60, -(2.0 ** -60), 5e-324):
        for bit in (0, 1):
            got = oracle.penalty_vec(np.array([l]), bit)[0]
            ref = want + abs(l) if (1 if l >= 0 else 0) != bit else want
            assert _bits(np.array([got]))[0] == _bits(np.array([np.float64(ref)]))[0]
    a = np.array([12.0, -12.0, 3.5, 0.0]); b = np.array([12.0, 12.0, -3.5, 0.0])
    assert np.array_equal(_bits(oracle.polar_f_vec(a, b)), _bits(np.logaddexp(a, b) - np.logaddexp(0.0, a + b)))


def test_softplus_in_the_log1p_corner(oracle):
    """|t| between 1.1e-16 and 2.86e-6 puts 1 + exp(t) within 3 * 2^-20 below 2, where fdlibm's log1p takes its |f| < 2^-20 form.
    Operands a few f levels down the tree differ (and sum) by that little all the time (f(x, y) ~ x*y/2 for small x, y), so the straight-line
    softplus of es_math.h carries that form as a select (round 3; before, it was left to the generic fall-back): both edges of the corner,
    its inside on a log scale, and what lies below it, against libm and against NumPy's logaddexp formula, bit for bit."""
    m = _libm()
    rng = np.random.default_rng(15)
    mags = np.concatenate([10.0 ** rng.uniform(-18, -4, 200_000),
                           2.86e-6 + rng.uniform(-1e-7, 1e-7, 50_000),         # the upper edge (u's high word 0x3ffffffc | 0x3ffffffd)
                           1.1e-16 + rng.uniform(-1e-16, 2e-16, 50_000)])      # the lower edge (exp(t) rounds to 1.0 below it)
    for bit in (0, 1):
        got = oracle.penalty_vec(mags, bit)                                     # l >= 0: log1p(exp(-l)) (+ l when the bit disagrees)
        pen = np.array([m.log1p(m.exp(-x)) for x in mags[:60_000]])
        ref = pen + mags[:60_000] if bit == 0 else pen
        assert np.array_equal(_bits(got[:60_000]), _bits(ref))
    base = np.clip(rng.normal(0, 2, mags.size), -12, 12)
    for a, b in ((base, base + mags), (base, -base + mags), (mags, mags * rng.uniform(0, 2, mags.size)), (mags, np.zeros_like(mags))):
        assert np.array_equal(_bits(oracle.polar_f_vec(a, b)), _bits(np.logaddexp(a, b) - np.logaddexp(0.0, a + b)))
