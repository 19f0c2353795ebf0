// crmath.h -- correctly-rounded sin / cos / atan / atan2 / exp / log / log10 / pow for fp64 on gfx950 (and, for testing, on the host).
//
// Why: the reference is built against glibc's libm, whose sin/cos/atan2 return the correctly rounded
// result in all but very rare cases, while the device math library (OCML) differs from it by 1 ulp in
// ~30 % of the calls.  The LSD rectangle test has STRUCTURAL ties (axis-aligned walls put rectangle
// edges exactly on pixel rows, so ceil()/floor() of an edge coordinate flips on a 1-ulp difference in
// cos/sin of the rectangle angle) and a 1-ulp drift there changes accept/reject decisions.  These
// routines evaluate in double-double arithmetic (~2^-100 relative error) and round once, which gives
// the correctly rounded result except when the exact value lies within ~2^-100 of a rounding boundary.
//
// Built two ways: by hipcc as __device__ functions (v_fma_f64), and by g++ for tests/test_crmath.py
// (libm fma()), where every function is compared with mpmath and with glibc.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CRM_FN __device__ inline
#define CRM_CONST static __device__ const
#else
#define CRM_FN static inline
#define CRM_CONST static const
#endif

#include "crmath_tables.h"

namespace crm {

struct dd { double hi, lo; };

CRM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

CRM_FN dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
CRM_FN dd fast_two_sum(double a, double b) {   // |a| >= |b| (or a == 0)
    const double s = a + b;
    return {s, b - (s - a)};
}
CRM_FN dd two_prod(double a, double b) {
    const double p = a * b;
    return {p, fma_(a, b, -p)};
}
CRM_FN dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return fast_two_sum(s.hi, s.lo);
}
CRM_FN dd dd_add_d(dd a, double b) {
    dd s = two_sum(a.hi, b);
    s.lo += a.lo;
    return fast_two_sum(s.hi, s.lo);
}
CRM_FN dd dd_neg(dd a) { return {-a.hi, -a.lo}; }
CRM_FN dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return fast_two_sum(p.hi, p.lo);
}
CRM_FN dd dd_mul_d(dd a, double b) {
    dd p = two_prod(a.hi, b);
    p.lo = fma_(a.lo, b, p.lo);
    return fast_two_sum(p.hi, p.lo);
}
CRM_FN dd dd_div(dd a, dd b) {                 // ~2^-104 relative
    const double q1 = a.hi / b.hi;
    dd r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    dd q = fast_two_sum(q1, q2);
    return dd_add_d(q, q3);
}

// sin(r), cos(r) for |r| <= pi/64 (+ a little), r given as double-double
CRM_FN void sincos_small(dd r, dd& s, dd& c) {
    const dd r2 = dd_mul(r, r);
    const double z = r2.hi;
    // sin: r + r^3 * (S0 + z*(S1 + z*(S2 + ...)))   (S_j = (-1)^(j+1)/(2j+3)!)
    double ps = SIN_C[7][0];
    ps = fma_(ps, z, SIN_C[6][0]);
    ps = fma_(ps, z, SIN_C[5][0]);
    ps = fma_(ps, z, SIN_C[4][0]);
    dd qs = dd_add_d(dd{SIN_C[3][0], SIN_C[3][1]}, ps * z);
    qs = dd_add(dd{SIN_C[2][0], SIN_C[2][1]}, dd_mul(qs, r2));
    qs = dd_add(dd{SIN_C[1][0], SIN_C[1][1]}, dd_mul(qs, r2));
    qs = dd_add(dd{SIN_C[0][0], SIN_C[0][1]}, dd_mul(qs, r2));
    s = dd_add(r, dd_mul(dd_mul(qs, r2), r));
    // cos: 1 + z*(C0 + z*(C1 + ...))                 (C_j = (-1)^(j+1)/(2j+2)!)
    double pc = COS_C[7][0];
    pc = fma_(pc, z, COS_C[6][0]);
    pc = fma_(pc, z, COS_C[5][0]);
    pc = fma_(pc, z, COS_C[4][0]);
    dd qc = dd_add_d(dd{COS_C[3][0], COS_C[3][1]}, pc * z);
    qc = dd_add(dd{COS_C[2][0], COS_C[2][1]}, dd_mul(qc, r2));
    qc = dd_add(dd{COS_C[1][0], COS_C[1][1]}, dd_mul(qc, r2));
    qc = dd_add(dd{COS_C[0][0], COS_C[0][1]}, dd_mul(qc, r2));
    c = dd_add_d(dd_mul(qc, r2), 1.0);
}

// Correctly rounded sin(x) and cos(x), |x| <= 64 (the LSD path only produces |x| < 7); returns false
// when x is outside that range or not finite (caller falls back to the platform libm).
CRM_FN bool sincos_cr(double x, double& s_out, double& c_out) {
    if (!(fabs(x) <= 64.0)) return false;
    if (x == 0.0) { s_out = x; c_out = 1.0; return true; }
    const double kd = rint(x * INV_PIO32);
    const int k = (int)kd;
    // r = x - k*pi/32 with pi/32 = PIO32[0..3] (212 bits): exact product + exact leading cancellation
    const dd p0 = two_prod(kd, PIO32[0]);
    dd r = two_sum(x, -p0.hi);
    r = dd_add_d(r, -p0.lo);
    r = dd_add(r, dd_neg(two_prod(kd, PIO32[1])));
    r = dd_add(r, dd_neg(two_prod(kd, PIO32[2])));
    r = dd_add_d(r, -(kd * PIO32[3]));
    dd sr, cr;
    sincos_small(r, sr, cr);
    const int idx = k & 63;
    const dd S = {SINCOS_TAB[idx][0], SINCOS_TAB[idx][1]}, C = {SINCOS_TAB[idx][2], SINCOS_TAB[idx][3]};
    // sin(a+r) = S cos r + C sin r ; cos(a+r) = C cos r - S sin r
    const dd sv = dd_add(dd_mul(S, cr), dd_mul(C, sr));
    const dd cv = dd_add(dd_mul(C, cr), dd_neg(dd_mul(S, sr)));
    s_out = sv.hi + sv.lo;
    c_out = cv.hi + cv.lo;
    return true;
}

// atan(t) for 0 <= t <= 1 (t double-double) as double-double
CRM_FN dd atan_01(dd t) {
    const int i = (int)rint(t.hi * 64.0);
    dd u;
    if (i == 0) {
        u = t;
    } else {
        const double ci = i * (1.0 / 64.0);                   // exact
        const dd num = dd_add_d(t, -ci);
        const dd den = dd_add_d(dd_mul_d(t, ci), 1.0);
        u = dd_div(num, den);
    }
    const dd u2 = dd_mul(u, u);
    const double z = u2.hi;
    double p = ATAN_C[7][0];
    p = fma_(p, z, ATAN_C[6][0]);
    p = fma_(p, z, ATAN_C[5][0]);
    p = fma_(p, z, ATAN_C[4][0]);
    p = fma_(p, z, ATAN_C[3][0]);
    dd q = dd_add_d(dd{ATAN_C[2][0], ATAN_C[2][1]}, p * z);
    q = dd_add(dd{ATAN_C[1][0], ATAN_C[1][1]}, dd_mul(q, u2));
    q = dd_add(dd{ATAN_C[0][0], ATAN_C[0][1]}, dd_mul(q, u2));
    const dd a = dd_add(u, dd_mul(dd_mul(q, u2), u));
    return dd_add(dd{ATAN_TAB[i][0], ATAN_TAB[i][1]}, a);
}

// Correctly rounded atan2(y, x) for finite, not-both-tiny arguments; IEEE special cases like glibc.
// Returns false for NaN/inf/subnormal-range inputs (caller falls back to the platform libm).
CRM_FN bool atan2_cr(double y, double x, double& out) {
    const double ax = fabs(x), ay = fabs(y);
    if (!(ax <= 1e300) || !(ay <= 1e300)) return false;       // NaN / inf / huge
    if (ay == 0.0) {                                          // atan2(+-0, x)
        if (ax == 0.0) { out = signbit(x) ? copysign(PI3[0], y) : y; return true; }
        out = signbit(x) ? copysign(PI3[0], y) : y;
        return true;
    }
    if (ax == 0.0) { out = copysign(PIO2_3[0], y); return true; }
    if (ax < 1e-290 || ay < 1e-290) return false;             // keep away from subnormal quotients
    dd a;
    if (ay <= ax) {
        const dd t = dd_div(dd{ay, 0.0}, dd{ax, 0.0});
        if (t.hi < 1e-290) return false;
        a = atan_01(t);
    } else {
        const dd t = dd_div(dd{ax, 0.0}, dd{ay, 0.0});
        if (t.hi < 1e-290) return false;
        // pi/2 - atan(t)
        a = dd_add(dd{PIO2_3[0], PIO2_3[1]}, dd_neg(atan_01(t)));
        a = dd_add_d(a, PIO2_3[2]);
    }
    if (signbit(x)) {                                         // pi - a
        a = dd_add(dd{PI3[0], PI3[1]}, dd_neg(a));
        a = dd_add_d(a, PI3[2]);
    }
    const double r = a.hi + a.lo;
    out = signbit(y) ? -r : r;
    return true;
}

// Correctly rounded atan(v) (any finite v); +-inf -> +-pi/2
CRM_FN bool atan_cr(double v, double& out) {
    if (v != v) return false;
    const double av = fabs(v);
    if (av > 1e300) { out = copysign(PIO2_3[0], v); return true; }
    if (av == 0.0) { out = v; return true; }
    return atan2_cr(v, 1.0, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// exp / log / log10 / pow, correctly rounded (RectangleNFACalculator, myLSD.cpp:1028-1058: the binomial tail's first term
// exp(log1Term), the returned -log10(binTail) - logNT, and -- behind a bracket of the fast device functions -- the pow / log10
// of the tail's stopping test).  Same scheme as above: double-double evaluation (~2^-100), one rounding at the end.
// ---------------------------------------------------------------------------------------------------------------------

// e^(xh + xl) = 2^e2 * (m.hi + m.lo) with m in [1, 2) (+-), |xh| <= ~800
CRM_FN dd exp_dd(double xh, double xl, int& e2) {
    const double kd = rint(xh * INV_LN2O64);
    const int k = (int)kd;
    // r = x - k ln2/64 (ln2/64 as three doubles, 159 bits): exact product + exact leading cancellation
    const dd p0 = two_prod(kd, LN2O64_3[0]);
    dd r = two_sum(xh, -p0.hi);
    r = dd_add_d(r, -p0.lo);
    r = dd_add(r, dd_neg(two_prod(kd, LN2O64_3[1])));
    r = dd_add_d(r, -(kd * LN2O64_3[2]));
    r = dd_add_d(r, xl);
    // e^r, |r| <= ln2/128 (+): r^n/n! for n >= 7 is below 2^-65 (double is enough), the r^14 term below 2^-141
    const double z = r.hi;
    double q = EXP_C[13][0];
    q = fma_(q, z, EXP_C[12][0]);
    q = fma_(q, z, EXP_C[11][0]);
    q = fma_(q, z, EXP_C[10][0]);
    q = fma_(q, z, EXP_C[9][0]);
    q = fma_(q, z, EXP_C[8][0]);
    q = fma_(q, z, EXP_C[7][0]);
    dd S = dd_add_d(dd{EXP_C[6][0], EXP_C[6][1]}, q * z);
    S = dd_add(dd{EXP_C[5][0], EXP_C[5][1]}, dd_mul(S, r));
    S = dd_add(dd{EXP_C[4][0], EXP_C[4][1]}, dd_mul(S, r));
    S = dd_add(dd{EXP_C[3][0], EXP_C[3][1]}, dd_mul(S, r));
    S = dd_add(dd{EXP_C[2][0], EXP_C[2][1]}, dd_mul(S, r));
    S = dd_add_d(dd_mul(S, r), 1.0);
    S = dd_add_d(dd_mul(S, r), 1.0);
    const int j = k & 63;
    e2 = (k - j) / 64;
    return dd_mul(dd{EXP2_TAB[j][0], EXP2_TAB[j][1]}, S);
}

// RN((m.hi + m.lo) * 2^e) for m.hi in [0.5, 4): one rounding, also where the result is subnormal
CRM_FN double round_scale(dd m, int e) {
    const dd s = fast_two_sum(m.hi, m.lo);                      // s.hi = RN(m.hi + m.lo)
    if (e > 1100) return __builtin_huge_val();
    if (e < -1200) return 0.0;
    if (e >= -1021 || s.hi >= ldexp(1.0, -1022 - e)) return ldexp(s.hi, e);   // normal (or overflow): the scaling is exact
    // below 2^-1022: round to a multiple of 2^-1074.  a + b = the value in those units (both scalings are exact).
    const int sh = e + 1074;
    if (sh < -3) return 0.0;
    const double a = ldexp(s.hi, sh), b = ldexp(s.lo, sh);
    const double big = 0x1p52;
    const double n = (a + big) - big;                           // RN(a) to an integer, ties to even
    const dd t = two_sum(a - n, b);                             // what is left, exactly
    double nn = n;
    if (t.hi > 0.5 || (t.hi == 0.5 && t.lo > 0)) nn = n + 1.0;
    else if (t.hi < -0.5 || (t.hi == -0.5 && t.lo < 0)) nn = n - 1.0;
    else if (b != 0.0 && t.lo == 0.0 && fabs(t.hi) == 0.5 && fmod(n, 2.0) != 0.0) nn = n + (t.hi > 0 ? 1.0 : -1.0);   // an exact tie made by b
    return ldexp(nn, -1074);
}

CRM_FN double exp_cr(double x) {
    if (x != x) return x;
    if (x > 710.0) return __builtin_huge_val();
    if (x < -746.0) return 0.0;
    int e2;
    const dd m = exp_dd(x, 0.0, e2);
    return round_scale(m, e2);
}

// log(x) as double-double, x > 0 finite (subnormals included)
CRM_FN dd log_dd(double x) {
    int E = 0;
    if (x < 0x1p-1022) { x *= 0x1p54; E = -54; }
    int ex;
    const double M = frexp(x, &ex) * 2.0;                       // [1, 2)
    E += ex - 1;
    const int i = (int)((M - 1.0) * 128.0);                     // the 7 leading mantissa bits
    if (i >= CRM_LOG_ISPLIT) E += 1;                            // M >= ~sqrt 2 counts as M / 2: LOG_T carries the - ln2
    // u = M rc_i - 1 exactly (rc_0 = 1 and rc_127 = 1/2 exactly: no cancellation against T_i next to x = 1)
    const dd p = two_prod(M, LOG_RC[i]);
    const dd u = two_sum(p.hi - 1.0, p.lo);
    // log1p(u) = u (1 + u (-1/2 + u (1/3 - ...))), |u| < 2^-7: u^(k-1)/k for k >= 9 is below 2^-59 (double), the k = 18 term below 2^-123
    const double z = u.hi;
    double q = LOG_C[17][0];
    q = fma_(q, z, LOG_C[16][0]);
    q = fma_(q, z, LOG_C[15][0]);
    q = fma_(q, z, LOG_C[14][0]);
    q = fma_(q, z, LOG_C[13][0]);
    q = fma_(q, z, LOG_C[12][0]);
    q = fma_(q, z, LOG_C[11][0]);
    q = fma_(q, z, LOG_C[10][0]);
    q = fma_(q, z, LOG_C[9][0]);
    dd S = dd_add_d(dd{LOG_C[8][0], LOG_C[8][1]}, q * z);
    S = dd_add(dd{LOG_C[7][0], LOG_C[7][1]}, dd_mul(S, u));
    S = dd_add(dd{LOG_C[6][0], LOG_C[6][1]}, dd_mul(S, u));
    S = dd_add(dd{LOG_C[5][0], LOG_C[5][1]}, dd_mul(S, u));
    S = dd_add(dd{LOG_C[4][0], LOG_C[4][1]}, dd_mul(S, u));
    S = dd_add(dd{LOG_C[3][0], LOG_C[3][1]}, dd_mul(S, u));
    S = dd_add(dd{LOG_C[2][0], LOG_C[2][1]}, dd_mul(S, u));
    S = dd_add_d(dd_mul(S, u), 1.0);
    dd r = dd_mul(S, u);
    r = dd_add(dd{LOG_T[i][0], LOG_T[i][1]}, r);
    if (E != 0) {
        const double Ed = (double)E;
        dd a = two_prod(Ed, LN2_H32[1]);
        a = dd_add_d(a, Ed * LN2_H32[2]);
        a = dd_add_d(a, Ed * LN2_H32[0]);                       // (|E| < 2^11) x (32 bits): exact
        r = dd_add(a, r);
    }
    return r;
}

// What "_cr" claims for log / log10 / pow (and exp): the value is the rounding of a double-double evaluation whose error is ~2^-100
// relative, WITHOUT a final rounding test (Ziv's last stage): it is the correctly rounded double unless the exact value lies within
// ~2^-100 of a rounding boundary, i.e. except with probability ~2^-47 per call (no such argument is known; the functions are
// transcendental at the arguments in question, so none is an exact tie).  tests/test_crmath.py holds every routine against mpmath --
// an independent arbitrary-precision evaluation, not this code -- on 10^5-10^6 arguments each, subnormal results included.
CRM_FN double log_cr(double x) {
    if (x != x || x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_huge_val();
    if (x > 1.7976931348623157e308) return x;
    const dd r = log_dd(x);
    return r.hi + r.lo;
}
CRM_FN double log10_cr(double x) {
    if (x != x || x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_huge_val();
    if (x > 1.7976931348623157e308) return x;
    const dd r = dd_mul(log_dd(x), dd{INV_LN10[0], INV_LN10[1]});
    return r.hi + r.lo;
}
// x^y for finite x > 0 and finite y (the NFA raises a ratio in (0, 1) to a pixel count)
CRM_FN double pow_cr(double x, double y) {
    if (y == 0.0 || x == 1.0) return 1.0;
    const dd t = dd_mul_d(log_dd(x), y);
    if (t.hi > 710.0) return __builtin_huge_val();
    if (t.hi < -746.0) return 0.0;
    int e2;
    const dd m = exp_dd(t.hi, t.lo, e2);
    return round_scale(m, e2);
}

// ---------------------------------------------------------------------------------------------------------------------
// First-stage evaluations (Ziv's strategy).  The double-double routines above cost ~900 fp64 instructions per call; the
// gradient pass calls them for every non-zero-gradient pixel.  The routines below evaluate the same functions to ~2^-66
// with a handful of double-double operations (~150 instructions), and return the rounded value only when every number
// within the error bound rounds to the same double -- which then is the value the full evaluation returns, bit for bit.
// Otherwise (about 1 call in 2000 - 16000) they return false and the caller runs the full evaluation.
// tests/test_crmath.py compares both stages on 10^7 inputs, the neighbourhoods of the table nodes included.
// ---------------------------------------------------------------------------------------------------------------------

// true (and out = the rounded value) if every real within err of h + l rounds to the same double; |h| >= |l|
CRM_FN bool round_certain(double h, double l, double err, double& out) {
    const dd s = fast_two_sum(h, l);
    out = s.hi;
    return (s.hi + (s.lo + err)) == s.hi && (s.hi + (s.lo - err)) == s.hi;
}

CRM_FN dd dd_mul_lite(dd a, dd b) {              // a*b, relative error ~2^-104 (a.lo*b.lo dropped), not renormalised
    dd p = two_prod(a.hi, b.hi);
    p.lo = fma_(a.hi, b.lo, fma_(a.lo, b.hi, p.lo));
    return p;
}

// (raw: the unrounded first-stage values, for the error-bound test only)
CRM_FN bool sincos_fast(double x, double& s_out, double& c_out, dd* raw = nullptr) {
    if (!(fabs(x) <= 64.0)) return false;
    if (x == 0.0) { s_out = x; c_out = 1.0; return true; }
    const double kd = rint(x * INV_PIO32);
    const int k = (int)kd;
    // r = x - k*pi/32 from the first 106 bits of pi/32: absolute error <= 2^-101 (|k| <= 652 -> 2^-104 from the dropped
    // words, the rest from the roundings of the low part)
    const dd p0 = two_prod(kd, PIO32[0]);
    const dd r0 = two_sum(x, -p0.hi);
    const double rl0 = fma_(-kd, PIO32[1], r0.lo - p0.lo);
    const dd r = fast_two_sum(r0.hi, rl0);
    // z = r^2
    dd z = two_prod(r.hi, r.hi);
    z.lo = fma_(2.0 * r.hi, r.lo, z.lo);
    // sin r = r + r^3*S0 + r^5*(S1 + z*(S2 + z*(S3 + z*S4)))      |r| <= pi/64: the r^13 term is 2^-84 |r|
    dd t = two_prod(z.hi, r.hi);                                    // r^3
    t.lo = fma_(z.hi, r.lo, fma_(z.lo, r.hi, t.lo));
    const dd u = dd_mul_lite(t, dd{SIN_C[0][0], SIN_C[0][1]});      // -r^3/6, |u| <= 4.1e-4 |r|
    const double ws = (z.hi * z.hi) * r.hi * fma_(z.hi, fma_(z.hi, fma_(z.hi, SIN_C[4][0], SIN_C[3][0]), SIN_C[2][0]), SIN_C[1][0]);
    const dd s1 = fast_two_sum(r.hi, u.hi);
    const dd sr = fast_two_sum(s1.hi, s1.lo + (u.lo + (r.lo + ws)));
    // cos r = 1 - z/2 + z^2*(C1 + z*(C2 + z*(C3 + z*C4)))          the z^6 term is 2^-81
    const double wc = (z.hi * z.hi) * fma_(z.hi, fma_(z.hi, fma_(z.hi, COS_C[4][0], COS_C[3][0]), COS_C[2][0]), COS_C[1][0]);
    const dd c1 = fast_two_sum(1.0, -0.5 * z.hi);
    const dd cr = fast_two_sum(c1.hi, c1.lo + (wc - 0.5 * z.lo));
    const int idx = k & 63;
    const dd S = {SINCOS_TAB[idx][0], SINCOS_TAB[idx][1]}, C = {SINCOS_TAB[idx][2], SINCOS_TAB[idx][3]};
    // sin(a+r) = S cos r + C sin r ; cos(a+r) = C cos r - S sin r.  Where a table value is not 0 it is >= sin(pi/32) = 0.098,
    // twice the largest |sin r|: the sums cancel at most one bit.
    const dd a1 = dd_mul_lite(S, cr), b1 = dd_mul_lite(C, sr);
    const dd a2 = dd_mul_lite(C, cr), b2 = dd_mul_lite(S, sr);
    const dd sv = two_sum(a1.hi, b1.hi), cv = two_sum(a2.hi, -b2.hi);
    const double svl = sv.lo + (a1.lo + b1.lo), cvl = cv.lo + (a2.lo - b2.lo);
    if (raw) { raw[0] = dd{sv.hi, svl}; raw[1] = dd{cv.hi, cvl}; }
    // relative 2^-68 covers the polynomial and product roundings (~2^-72), absolute 2^-99 the reduction
    const bool oks = round_certain(sv.hi, svl, fabs(sv.hi) * 0x1p-68 + 0x1p-99, s_out);
    const bool okc = round_certain(cv.hi, cvl, fabs(cv.hi) * 0x1p-68 + 0x1p-99, c_out);
    return oks && okc;
}

CRM_FN bool atan2_fast(double y, double x, double& out, dd* raw = nullptr) {
    const double ax = fabs(x), ay = fabs(y);
    if (!(ax >= 1e-140 && ax <= 1e140 && ay >= 1e-140 && ay <= 1e140)) return false;   // zeros, NaN, inf, extremes: full path
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    // t = num/den = q1 + q2 (q1 within 2 ulp of the quotient: the remainder is exact in one fma)
    const double rc = 1.0 / den;
    const double q1 = num * rc;
    const double q2 = fma_(-q1, den, num) * rc;
    const int i = (int)rint(q1 * 64.0);
    double u1 = q1, u2 = q2;
    if (i != 0) {                                              // u = (t - c) / (1 + t*c), c = i/64
        const double c = i * (1.0 / 64.0);
        const double nh = q1 - c;                              // exact (Sterbenz)
        const dd p = two_prod(q1, c);
        const dd d1 = fast_two_sum(1.0, p.hi);
        const double dl = d1.lo + fma_(q2, c, p.lo);
        const double rc2 = 1.0 / d1.hi;
        u1 = nh * rc2;
        u2 = (fma_(-u1, d1.hi, nh) + (q2 - u1 * dl)) * rc2;    // absolute error ~2^-104
    }
    // atan u = u + u^3*A0 + u^5*(A1 + z*(A2 + z*A3)), |u| <= 1/128 (+): the u^11 term is 2^-73 |u|; u^3*A0 (up to 2^-15.6 |u|)
    // in double-double, the rest (<= 2^-37 |u|) in double
    dd z = two_prod(u1, u1);
    z.lo = fma_(2.0 * u1, u2, z.lo);
    dd t3 = two_prod(z.hi, u1);
    t3.lo = fma_(z.hi, u2, fma_(z.lo, u1, t3.lo));
    const dd w0 = dd_mul_lite(t3, dd{ATAN_C[0][0], ATAN_C[0][1]});
    const double w1 = (z.hi * z.hi) * u1 * fma_(z.hi, fma_(z.hi, ATAN_C[3][0], ATAN_C[2][0]), ATAN_C[1][0]);
    const double tail = u2 + (w0.lo + w1);
    dd R;
    if (i == 0) {
        const dd s = fast_two_sum(u1, w0.hi);
        R = fast_two_sum(s.hi, s.lo + tail);
    } else {
        const dd s = fast_two_sum(ATAN_TAB[i][0], u1);
        const dd s2 = fast_two_sum(s.hi, w0.hi);
        R = fast_two_sum(s2.hi, s2.lo + (s.lo + (ATAN_TAB[i][1] + tail)));
    }
    if (swap) {                                                // pi/2 - R
        const dd s = fast_two_sum(PIO2_3[0], -R.hi);
        R = fast_two_sum(s.hi, s.lo + (PIO2_3[1] - R.lo));
    }
    if (signbit(x)) {                                          // pi - R
        const dd s = fast_two_sum(PI3[0], -R.hi);
        R = fast_two_sum(s.hi, s.lo + (PI3[1] - R.lo));
    }
    if (raw) *raw = R;
    double r;
    const bool ok = round_certain(R.hi, R.lo, fabs(R.hi) * 0x1p-68, r);
    out = signbit(y) ? -r : r;
    return ok;
}

// exp(x), first stage: the same reduction and table as exp_dd with the series in double beyond r^2 / 2 (~2^-74 relative).  Answers only
// where the result is a normal number and its rounding is certain; then it is the value of exp_cr, bit for bit.
CRM_FN bool exp_fast(double x, double& out, dd* raw = nullptr) {
    if (!(x > -700.0 && x < 700.0)) return false;                // (NaN, overflow, subnormal results: full evaluation)
    const double kd = rint(x * INV_LN2O64);
    const int k = (int)kd;
    // r = x - k ln2/64 from the first 106 bits of ln2/64: absolute error <= 2^-96 (|k| < 2^16: 2^-100 from the dropped word, the rest
    // from the rounding of the low part)
    const dd p0 = two_prod(kd, LN2O64_3[0]);
    const dd r0 = two_sum(x, -p0.hi);
    const double rl0 = fma_(-kd, LN2O64_3[1], r0.lo - p0.lo);
    const dd r = fast_two_sum(r0.hi, rl0);
    const double z = r.hi;
    // e^r = 1 + r + r^2/2 + z^3 (1/6 + z (1/24 + ... + z^4/5040)): |r| <= ln2/128 (+), the z^8 term is 2^-75; r^2/2 in double-double
    const double P = (z * z) * z * fma_(z, fma_(z, fma_(z, fma_(z, EXP_C[7][0], EXP_C[6][0]), EXP_C[5][0]), EXP_C[4][0]), EXP_C[3][0]);
    const dd q = two_prod(z, z);
    const double sq_hi = 0.5 * q.hi, sq_lo = 0.5 * fma_(2.0 * z, r.lo, q.lo);
    const dd s1 = fast_two_sum(1.0, z);
    const dd s2 = two_sum(s1.hi, sq_hi);
    const double lo = (s1.lo + s2.lo) + ((r.lo + sq_lo) + P);
    const int j = k & 63;
    const dd m = dd_mul_lite(dd{EXP2_TAB[j][0], EXP2_TAB[j][1]}, dd{s2.hi, lo});
    if (raw) *raw = m;
    double v;
    if (!round_certain(m.hi, m.lo, fabs(m.hi) * 0x1p-70, v)) return false;
    out = ldexp(v, (k - j) / 64);                                // |x| < 700: a normal number, the scaling is exact
    return true;
}

// log10(x), first stage, for normal x > 0 (2^-65 |u| + 2^-84 (|E| + 1) absolute before the conversion; see the error bound below)
CRM_FN bool log10_fast(double x, double& out, dd* raw = nullptr, double* errb = nullptr) {
    if (!(x >= 0x1p-1022 && x <= 1.7976931348623157e308)) return false;
    int ex;
    const double M = frexp(x, &ex) * 2.0;                       // [1, 2)
    int E = ex - 1;
    const int i = (int)((M - 1.0) * 128.0);
    if (i >= CRM_LOG_ISPLIT) E += 1;
    const dd p = two_prod(M, LOG_RC[i]);
    const dd u = two_sum(p.hi - 1.0, p.lo);                      // exact, |u| < 2^-7
    const double z = u.hi;
    // log1p(u) = u - u^2/2 + z^3 (1/3 - z/4 + ... - z^7/10): the u^11 term is 2^-70 |u| / 11; u^2/2 in double-double
    double P = LOG_C[10][0];
    P = fma_(P, z, LOG_C[9][0]); P = fma_(P, z, LOG_C[8][0]); P = fma_(P, z, LOG_C[7][0]); P = fma_(P, z, LOG_C[6][0]);
    P = fma_(P, z, LOG_C[5][0]); P = fma_(P, z, LOG_C[4][0]); P = fma_(P, z, LOG_C[3][0]);
    P = (z * z) * z * P;
    const dd q = two_prod(z, z);
    const double sq_hi = 0.5 * q.hi, sq_lo = 0.5 * fma_(2.0 * z, u.lo, q.lo);
    const dd b = two_sum(z, -sq_hi);
    const double blo = b.lo + ((u.lo - sq_lo) + P);
    // + T_i + E ln2 (E x the 32-bit head of ln2 is exact)
    const double Ed = (double)E;
    const dd a = two_sum(Ed * LN2_H32[0], LOG_T[i][0]);
    const double alo = a.lo + fma_(Ed, LN2_H32[1], LOG_T[i][1]);
    const dd s = two_sum(a.hi, b.hi);
    const dd R = fast_two_sum(s.hi, s.lo + (alo + blo));
    const dd R10 = dd_mul_lite(R, dd{INV_LN10[0], INV_LN10[1]});
    // error: series + roundings of the log1p part < 2^-66 |u| (the cubic term's rounding, 2^-52 |u|^3 / 3, leads; measured: 0.2 of the
    // bound below), the table and E ln2 part <= 2^-84 (|E| + 1), the conversion 2^-100 relative
    const double err = (fabs(z) * 0x1p-65 + (fabs(Ed) + 1.0) * 0x1p-84) * 0.5 + fabs(R10.hi) * 0x1p-98;
    if (raw) *raw = R10;
    if (errb) *errb = err;
    return round_certain(R10.hi, R10.lo, err, out);
}

}  // namespace crm
