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

}  // namespace crm
