// crmath.h -- correctly-rounded sin / cos / atan / atan2 for fp64 on gfx950 (and, for testing, on the host).
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

}  // namespace crm
