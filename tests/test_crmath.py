"""crmath.h (correctly rounded sin/cos/atan/atan2 in double-double): the host build against mpmath
(must be the correctly rounded result every time) and against glibc (may differ only where glibc
itself is not correctly rounded); the device build (-m gpu) must be bit-identical to the host build."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def crm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("crm") / "libcrm_host.so")
    flags = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + flags +
                   ["-o", so, os.path.join(ROOT, "tests", "crmath_host.cpp"), "-lm"], check=True)
    return C.CDLL(so)


def samples():
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-np.pi, np.pi, 6000), rng.uniform(-7, 7, 2000),
                         np.array([np.pi / 2, np.pi, -np.pi, -np.pi / 2, 1.5 * np.pi, 2 * np.pi, np.pi / 4, 1e-300, 1e-10,
                                   -1e-5, 0.0, -0.0, np.pi / 32, 5 * np.pi / 32, 1.0, 0.5, 6.0, 63.9]),
                         np.arange(-64, 65) * (np.pi / 32)])
    th = rng.uniform(-np.pi, np.pi, 3000)
    n = rng.integers(1, 2000, 3000)
    ys = np.concatenate([rng.normal(0, 10, 4000), np.sin(th) * n, rng.uniform(-1, 1, 2000) * 10.0 ** rng.uniform(-12, 3, 2000),
                         np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1e-20, 1.0, 3.0, 1.0])])
    xx = np.concatenate([rng.normal(0, 10, 4000), np.cos(th) * n, rng.uniform(-1, 1, 2000) * 10.0 ** rng.uniform(-12, 3, 2000),
                         np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, -0.0, 1.0, -1.0, 1.0, 1e-20, -4.0, 64.0])])
    return xs, ys, xx


def test_host_build_is_correctly_rounded(crm):
    import mpmath as mp
    mp.mp.prec = 300
    xs, ys, xx = samples()
    n = len(xs)
    s, c = np.zeros(n), np.zeros(n)
    assert crm.crm_sincos_n(P(xs), P(s), P(c), n) == 0
    es = np.array([float(mp.sin(mp.mpf(float(v)))) for v in xs])
    ec = np.array([float(mp.cos(mp.mpf(float(v)))) for v in xs])
    assert np.array_equal(s, es) and np.array_equal(c, ec)
    n = len(ys)
    o = np.zeros(n)
    assert crm.crm_atan2_n(P(ys), P(xx), P(o), n) == 0
    e = np.array([float(mp.atan2(mp.mpf(float(a)), mp.mpf(float(b)))) for a, b in zip(ys, xx)])
    nz = ys != 0                                                  # mpmath has no signed zero
    assert np.array_equal(o[nz], e[nz]) and np.array_equal(np.signbit(o[nz]), np.signbit(e[nz]))
    o2 = np.zeros(n)
    crm.libm_atan2_n(P(ys), P(xx), P(o2), n)                      # IEEE special cases: like glibc
    assert np.array_equal(o[~nz], o2[~nz]) and np.array_equal(np.signbit(o[~nz]), np.signbit(o2[~nz]))
    v = np.concatenate([ys[:3000] / np.where(xx[:3000] == 0, 1, xx[:3000]), [np.inf, -np.inf, 0.0, -0.0, 1e308, 1e-200]])
    o = np.zeros(len(v))
    assert crm.crm_atan_n(P(v), P(o), len(v)) == 0
    e = np.array([float(mp.atan(mp.mpf(float(a)))) if np.isfinite(a) else np.sign(a) * float(mp.pi / 2) for a in v])
    assert np.array_equal(o, e)


def test_glibc_differs_only_by_its_own_misroundings(crm):
    """glibc's sin/cos/atan2 are within ~0.52 ulp: they disagree with the correctly rounded value in a
    fraction of a percent of the calls, always by exactly one ulp."""
    xs, ys, xx = samples()
    n = len(xs)
    s, c, s2, c2 = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    crm.crm_sincos_n(P(xs), P(s), P(c), n)
    crm.libm_sincos_n(P(xs), P(s2), P(c2), n)
    for a, b in ((s, s2), (c, c2)):
        d = np.abs(a.view(np.int64) - b.view(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 0.01
    n = len(ys)
    o, o2 = np.zeros(n), np.zeros(n)
    crm.crm_atan2_n(P(ys), P(xx), P(o), n)
    crm.libm_atan2_n(P(ys), P(xx), P(o2), n)
    d = np.abs(o.view(np.int64) - o2.view(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def _rn(mp, v):
    """an mpf rounded to the nearest double, ties to even, subnormal results and overflow included (float() of an mpf does not
    promise the single rounding below 2^-1022)"""
    import math
    if v == 0:
        return 0.0
    sgn = -1 if v < 0 else 1
    v = abs(v)
    e = int(mp.floor(mp.log(v, 2)))
    while mp.ldexp(1, e) > v:
        e -= 1
    while mp.ldexp(1, e + 1) <= v:
        e += 1
    q = max(e, -1022) - 52                                       # exponent of the unit in the last place
    t = mp.ldexp(v, -q)
    n = int(mp.floor(t))
    f = t - n
    if f > mp.mpf(1) / 2 or (f == mp.mpf(1) / 2 and n % 2 == 1):
        n += 1
    try:
        return sgn * math.ldexp(n, q)
    except OverflowError:
        return sgn * math.inf


def explog_samples(n):
    import math
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-746, 710, n), rng.uniform(-745.2, -707, n // 4), rng.uniform(-1, 1, n // 4) * 10.0 ** rng.uniform(-20, 0, n // 4),
                         np.array([0.0, -0.0, 709.782712893384, 709.7827128933841, -745.1332191019411, -745.1332191019412, -744.44,
                                   -708.3964185322641, 1e-320, -1e-320, 1.0, -1.0]),
                         np.arange(-1075, 1025) * math.log(2), rng.integers(-68000, 65000, n // 4) * (math.log(2) / 64)])
    xl = np.concatenate([10.0 ** rng.uniform(-323, 308, n), rng.uniform(0.5, 2, n), 1 + rng.normal(0, 1, n // 4) * 10.0 ** rng.uniform(-16, -1, n // 4),
                         10.0 ** rng.integers(-300, 300, 600).astype(float),
                         np.array([1.0, 2.0, 0.5, 10.0, 1e-310, 5e-324, 1.7976931348623157e308, math.sqrt(2), 1.4140625, 1.4139, 1.9921875, 1.99218749]),
                         1 + np.arange(0, 129) / 128.0, (1 + np.arange(0, 129) / 128.0) * (1 - 2.0 ** -52)])
    xl = np.abs(xl)
    xl = xl[xl > 0]
    # pow as the NFA calls it: a ratio in (0, 1) to a pixel count (myLSD.cpp:1052); and ordinary arguments
    px = np.concatenate([rng.uniform(0, 1, n), 1 - 10.0 ** rng.uniform(-12, 0, n // 2), rng.uniform(0.01, 30, n // 2), np.array([0.5, 0.25, 2.0, 1.0, 0.999999, 1e-300])])
    py = np.concatenate([rng.integers(1, 6000, n).astype(float), rng.integers(1, 100000, n // 2).astype(float), rng.uniform(-50, 50, n // 2),
                         np.array([3.0, 1074.0, 1023.0, 55.0, 1e7, 3.0])])
    return xs, xl, px, py


def test_exp_log_pow_are_correctly_rounded(crm):
    """exp / log / log10 / pow of crmath.h (RectangleNFACalculator's libm calls on the device) against mpmath: the correctly rounded
    double every time, subnormal results included; and how often glibc itself is not (its log10 in one call out of seven)."""
    import mpmath as mp
    mp.mp.prec = 500
    xs, xl, px, py = explog_samples(3000)
    o = np.zeros(len(xs))
    crm.crm_exp_n(P(xs), P(o), C.c_long(len(xs)))
    e = np.array([_rn(mp, mp.exp(mp.mpf(float(x)))) for x in xs])
    assert np.array_equal(o, e)
    g = np.zeros(len(xs))
    crm.libm_exp_n(P(xs), P(g), C.c_long(len(xs)))
    assert np.abs(g.view(np.int64) - e.view(np.int64)).max() <= 1 and (g != e).mean() < 0.01
    for fn, mf in ((crm.crm_log_n, mp.log), (crm.crm_log10_n, mp.log10)):
        o = np.zeros(len(xl))
        fn(P(xl), P(o), C.c_long(len(xl)))
        e = np.array([_rn(mp, mf(mp.mpf(float(x)))) for x in xl])
        assert np.array_equal(o, e)
    g = np.zeros(len(xl))
    crm.libm_log10_n(P(xl), P(g), C.c_long(len(xl)))
    assert np.abs(g.view(np.int64) - e.view(np.int64)).max() <= 2 and (g != e).mean() < 0.3
    o = np.zeros(len(px))
    crm.crm_pow_n(P(px), P(py), P(o), C.c_long(len(px)))
    e = np.array([_rn(mp, mp.power(mp.mpf(float(a)), mp.mpf(float(b)))) for a, b in zip(px, py)])
    assert np.array_equal(o, e)
    g = np.zeros(len(px))
    crm.libm_pow_n(P(px), P(py), P(g), C.c_long(len(px)))
    fin = np.isfinite(e) & (e > 0)
    assert np.abs(g[fin].view(np.int64) - e[fin].view(np.int64)).max() <= 1 and (g != e).mean() < 0.01


def test_exp_log10_first_stages(crm):
    """exp_fast / log10_fast (what RectangleNFACalculator's two value-carrying calls run first): every answer equals the double-double
    evaluation bit for bit, nearly every call is answered, and the unrounded values stay within a quarter (exp) / half (log10: 0.2
    measured) of the error bound the rounding test assumes."""
    import math
    import mpmath as mp
    mp.mp.prec = 300
    crm.crm_exp_fast_n.restype = C.c_long
    crm.crm_log10_fast_n.restype = C.c_long
    rng = np.random.default_rng(5)
    n = 1_000_000
    xs = np.concatenate([rng.uniform(-700, 700, n), rng.uniform(-60, 0, n),
                         rng.integers(-64000, 64000, n // 4) * (math.log(2) / 64) * (1 + rng.normal(0, 1e-9, n // 4)),
                         (rng.integers(-64000, 64000, n // 4) + 0.5) * (math.log(2) / 64)])
    bad = C.c_long()
    acc = crm.crm_exp_fast_n(P(xs), C.c_long(len(xs)), C.byref(bad))
    assert bad.value == 0 and acc > 0.9999 * len(xs)
    xl = np.concatenate([10.0 ** rng.uniform(-300, 300, n), rng.uniform(0.5, 2, n), 10.0 ** rng.uniform(-40, 0, n),
                         np.repeat(1 + np.arange(0, 129) / 128.0, 50) * (1 + rng.normal(0, 1e-8, 129 * 50))])
    acc = crm.crm_log10_fast_n(P(xl), C.c_long(len(xl)), C.byref(bad))
    assert bad.value == 0 and acc > 0.999 * len(xl)
    m = 1500
    xe = np.concatenate([rng.uniform(-700, 700, m), rng.uniform(-1, 1, m)])
    o = np.zeros(4 * len(xe))
    crm.crm_exp_fast_raw_n(P(xe), P(o), C.c_long(len(xe)))
    worst = 0.0
    for x, row in zip(xe, o.reshape(-1, 4)):
        if row[1] == 0:
            continue
        kd = round(x * 64 / math.log(2))
        e = (kd - kd % 64) // 64
        t = mp.exp(mp.mpf(float(x))) / mp.mpf(2) ** e
        worst = max(worst, float(abs(mp.mpf(float(row[1])) + mp.mpf(float(row[2])) - t) / mp.mpf(float(row[3]))))
    assert worst < 0.25, worst
    xq = np.abs(np.concatenate([10.0 ** rng.uniform(-300, 300, m), rng.uniform(0.5, 2, m), 1 + rng.uniform(-1 / 256, 1 / 128, m), 10.0 ** rng.uniform(-40, 0, m)]))
    o = np.zeros(4 * len(xq))
    crm.crm_log10_fast_raw_n(P(xq), P(o), C.c_long(len(xq)))
    worst = 0.0
    for x, row in zip(xq, o.reshape(-1, 4)):
        if row[3] == 0:
            continue
        worst = max(worst, float(abs(mp.mpf(float(row[1])) + mp.mpf(float(row[2])) - mp.log10(mp.mpf(float(x)))) / mp.mpf(float(row[3]))))
    assert worst < 0.5, worst


def _adversarial(rng, n):
    """Inputs around the table nodes and interval ends of both first stages, next to ordinary ones."""
    xs = np.concatenate([rng.uniform(-np.pi, np.pi, n), rng.uniform(-7, 7, n // 4),
                         rng.integers(-64, 65, n // 4) * (np.pi / 32) + rng.normal(0, 1, n // 4) * 10.0 ** rng.uniform(-17, -2, n // 4),
                         (rng.integers(-64, 64, n // 4) + 0.5) * (np.pi / 32) * (1 + rng.normal(0, 1e-6, n // 4))])
    th = rng.uniform(-np.pi, np.pi, n)
    r = 10.0 ** rng.uniform(-3, 3, n)
    i = rng.integers(0, 65, n)
    d = rng.uniform(0.1, 100, n)
    g1 = rng.integers(-510, 511, n) / 2.0 * rng.uniform(0, 1, n)          # gradient-like: halves of small integers
    g2 = rng.integers(-510, 511, n) / 2.0
    ys = np.concatenate([rng.normal(0, 10, n), np.sin(th) * r, rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-12, 3, n),
                         d * (i / 64.0) * (1 + rng.normal(0, 1, n) * 10.0 ** rng.uniform(-17, -3, n)) * rng.choice([-1, 1], n),
                         d * ((i + 0.5) / 64.0) * (1 + rng.normal(0, 1e-6, n)), g1, g2])
    xx = np.concatenate([rng.normal(0, 10, n), np.cos(th) * r, rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-12, 3, n),
                         d * rng.choice([-1, 1], n), d * rng.choice([-1, 1], n), g2, g1])
    return xs, ys, xx


def test_first_stage_equals_the_full_evaluation(crm):
    """sincos_fast / atan2_fast (Ziv's first stage, what the gradient pass runs for nearly every pixel) answer only when the
    rounding is certain: every answer must equal the double-double evaluation bit for bit, and nearly all calls are answered."""
    crm.crm_sincos_fast_n.restype = C.c_long
    crm.crm_atan2_fast_n.restype = C.c_long
    xs, ys, xx = _adversarial(np.random.default_rng(21), 1_500_000)
    bad = C.c_long()
    acc = crm.crm_sincos_fast_n(P(xs), C.c_long(len(xs)), C.byref(bad))
    assert bad.value == 0 and acc > 0.99 * len(xs)
    acc = crm.crm_atan2_fast_n(P(ys), P(xx), C.c_long(len(ys)), C.byref(bad))
    assert bad.value == 0 and acc > 0.99 * len(ys)
    # ordinary inputs: the second stage is needed about once in 10^4 calls
    rng = np.random.default_rng(22)
    u = rng.uniform(-np.pi, np.pi, 2_000_000)
    assert crm.crm_sincos_fast_n(P(u), C.c_long(len(u)), C.byref(bad)) > (1 - 5e-4) * len(u) and bad.value == 0
    a, b = rng.normal(0, 10, 2_000_000), rng.normal(0, 10, 2_000_000)
    assert crm.crm_atan2_fast_n(P(a), P(b), C.c_long(len(a)), C.byref(bad)) > (1 - 5e-4) * len(a) and bad.value == 0


def test_first_stage_error_is_well_inside_its_bound(crm):
    """The unrounded first-stage values against mpmath: the error must stay below a quarter of the bound the rounding test
    assumes (2^-68 relative, + 2^-99 absolute for sin/cos)."""
    import mpmath as mp
    mp.mp.prec = 200
    xs, ys, xx = _adversarial(np.random.default_rng(23), 4000)
    o = np.zeros(6 * len(xs))
    crm.crm_sincos_fast_raw_n(P(xs), P(o), C.c_long(len(xs)))
    worst = 0.0
    for x, row in zip(xs, o.reshape(-1, 6)):
        if x == 0 or abs(x) > 64:
            continue
        X = mp.mpf(float(x))
        for h, l, f in ((row[1], row[2], mp.sin), (row[3], row[4], mp.cos)):
            e = abs(mp.mpf(float(h)) + mp.mpf(float(l)) - f(X))
            worst = max(worst, float(e / (abs(mp.mpf(float(h))) * mp.mpf(2) ** -68 + mp.mpf(2) ** -99)))
    assert worst < 0.25, worst
    o = np.zeros(3 * len(ys))
    crm.crm_atan2_fast_raw_n(P(ys), P(xx), P(o), C.c_long(len(ys)))
    worst = 0.0
    for y, x, row in zip(ys, xx, o.reshape(-1, 3)):
        if row[1] == 0 and row[2] == 0:                             # declined before evaluating (zeros, extremes)
            continue
        t = abs(mp.atan2(mp.mpf(float(y)), mp.mpf(float(x))))
        e = abs(mp.mpf(float(row[1])) + mp.mpf(float(row[2])) - t)
        worst = max(worst, float(e / (abs(mp.mpf(float(row[1]))) * mp.mpf(2) ** -68)))
    assert worst < 0.25, worst


@pytest.mark.gpu
def test_device_build_matches_host_build(crm, lsdmod):
    ctx = lsdmod.Context(0)
    xs, ys, xx = samples()
    rng = np.random.default_rng(5)
    xs = np.concatenate([xs, rng.uniform(-np.pi, np.pi, 200000)])
    ys = np.concatenate([ys, rng.normal(0, 30, 200000)])
    xx = np.concatenate([xx, rng.normal(0, 30, 200000)])
    n = len(xs)
    s, c = np.zeros(n), np.zeros(n)
    crm.crm_sincos_n(P(xs), P(s), P(c), n)
    ds, dc = ctx.eval_math(0, xs)
    assert np.array_equal(ds, s) and np.array_equal(dc, c)
    n = len(ys)
    o = np.zeros(n)
    crm.crm_atan2_n(P(ys), P(xx), P(o), n)
    do, _ = ctx.eval_math(1, ys, xx)
    assert np.array_equal(do, o) and np.array_equal(np.signbit(do), np.signbit(o))
    v = ys / np.where(xx == 0, 1, xx)
    o = np.zeros(len(v))
    crm.crm_atan_n(P(v), P(o), len(v))
    do, _ = ctx.eval_math(2, v)
    assert np.array_equal(do, o)
    # exp / log10 / pow (RectangleNFACalculator): subnormal results and arguments included
    ex, xl, px, py = explog_samples(100000)
    for fn, hostf, args in ((4, crm.crm_exp_n, (ex,)), (5, crm.crm_log10_n, (xl,)), (6, crm.crm_pow_n, (px, py))):
        o = np.zeros(len(args[0]))
        hostf(*[P(a) for a in args], P(o), C.c_long(len(o)))
        do, _ = ctx.eval_math(fn, *args)
        assert np.array_equal(do, o), fn
    ctx.close()
