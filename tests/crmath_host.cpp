// Host build of crmath.h for tests/test_crmath.py (g++, libm fma): C entry points over arrays.
#include "../linesegmentdetector-slam_amd/csrc/crmath.h"
extern "C" {
int crm_sincos_n(const double* x, double* s, double* c, long n) {
    int bad = 0;
    for (long i = 0; i < n; i++) if (!crm::sincos_cr(x[i], s[i], c[i])) { s[i] = sin(x[i]); c[i] = cos(x[i]); bad++; }
    return bad;
}
int crm_atan2_n(const double* y, const double* x, double* out, long n) {
    int bad = 0;
    for (long i = 0; i < n; i++) if (!crm::atan2_cr(y[i], x[i], out[i])) { out[i] = atan2(y[i], x[i]); bad++; }
    return bad;
}
int crm_atan_n(const double* v, double* out, long n) {
    int bad = 0;
    for (long i = 0; i < n; i++) if (!crm::atan_cr(v[i], out[i])) { out[i] = atan(v[i]); bad++; }
    return bad;
}
// first stage against the full evaluation: returns the number of inputs the first stage answered; *bad counts answers that
// differ from the full evaluation in any bit
long crm_sincos_fast_n(const double* x, long n, long* bad) {
    long acc = 0; *bad = 0;
    for (long i = 0; i < n; i++) {
        double s, c, s2, c2;
        if (!crm::sincos_fast(x[i], s, c)) continue;
        acc++;
        if (!crm::sincos_cr(x[i], s2, c2)) { s2 = sin(x[i]); c2 = cos(x[i]); }
        if (__builtin_memcmp(&s, &s2, 8) || __builtin_memcmp(&c, &c2, 8)) ++*bad;
    }
    return acc;
}
long crm_atan2_fast_n(const double* y, const double* x, long n, long* bad) {
    long acc = 0; *bad = 0;
    for (long i = 0; i < n; i++) {
        double r, r2;
        if (!crm::atan2_fast(y[i], x[i], r)) continue;
        acc++;
        if (!crm::atan2_cr(y[i], x[i], r2)) r2 = atan2(y[i], x[i]);
        if (__builtin_memcmp(&r, &r2, 8)) ++*bad;
    }
    return acc;
}
// the unrounded first-stage values (hi, lo), for the error-bound test
void crm_sincos_fast_raw_n(const double* x, double* out, long n) {      // out[6*i..]: answered, s.hi, s.lo, c.hi, c.lo, -
    for (long i = 0; i < n; i++) {
        double s, c; crm::dd raw[2] = {{0, 0}, {0, 0}};
        out[6 * i] = crm::sincos_fast(x[i], s, c, raw) ? 1.0 : 0.0;
        out[6 * i + 1] = raw[0].hi; out[6 * i + 2] = raw[0].lo; out[6 * i + 3] = raw[1].hi; out[6 * i + 4] = raw[1].lo; out[6 * i + 5] = 0;
    }
}
void crm_atan2_fast_raw_n(const double* y, const double* x, double* out, long n) {   // out[3*i..]: answered, |r|.hi, |r|.lo
    for (long i = 0; i < n; i++) {
        double r; crm::dd raw = {0, 0};
        out[3 * i] = crm::atan2_fast(y[i], x[i], r, &raw) ? 1.0 : 0.0;
        out[3 * i + 1] = raw.hi; out[3 * i + 2] = raw.lo;
    }
}
void crm_exp_n(const double* x, double* o, long n) { for (long i = 0; i < n; i++) o[i] = crm::exp_cr(x[i]); }
void crm_log_n(const double* x, double* o, long n) { for (long i = 0; i < n; i++) o[i] = crm::log_cr(x[i]); }
void crm_log10_n(const double* x, double* o, long n) { for (long i = 0; i < n; i++) o[i] = crm::log10_cr(x[i]); }
void crm_pow_n(const double* x, const double* y, double* o, long n) { for (long i = 0; i < n; i++) o[i] = crm::pow_cr(x[i], y[i]); }
// first stages against the full evaluations (as for sincos / atan2 above)
long crm_exp_fast_n(const double* x, long n, long* bad) {
    long acc = 0; *bad = 0;
    for (long i = 0; i < n; i++) {
        double r;
        if (!crm::exp_fast(x[i], r)) continue;
        acc++;
        const double r2 = crm::exp_cr(x[i]);
        if (__builtin_memcmp(&r, &r2, 8)) ++*bad;
    }
    return acc;
}
long crm_log10_fast_n(const double* x, long n, long* bad) {
    long acc = 0; *bad = 0;
    for (long i = 0; i < n; i++) {
        double r;
        if (!crm::log10_fast(x[i], r)) continue;
        acc++;
        const double r2 = crm::log10_cr(x[i]);
        if (__builtin_memcmp(&r, &r2, 8)) ++*bad;
    }
    return acc;
}
// the unrounded first-stage values: out[4*i..] = answered, hi, lo, error bound (exp: the bound is |hi| 2^-70, the value is the mantissa part)
void crm_exp_fast_raw_n(const double* x, double* out, long n) {
    for (long i = 0; i < n; i++) {
        double r; crm::dd raw = {0, 0};
        out[4 * i] = crm::exp_fast(x[i], r, &raw) ? 1.0 : 0.0;
        out[4 * i + 1] = raw.hi; out[4 * i + 2] = raw.lo; out[4 * i + 3] = fabs(raw.hi) * 0x1p-70;
    }
}
void crm_log10_fast_raw_n(const double* x, double* out, long n) {
    for (long i = 0; i < n; i++) {
        double r, e = 0; crm::dd raw = {0, 0};
        out[4 * i] = crm::log10_fast(x[i], r, &raw, &e) ? 1.0 : 0.0;
        out[4 * i + 1] = raw.hi; out[4 * i + 2] = raw.lo; out[4 * i + 3] = e;
    }
}
void libm_exp_n(const double* x, double* o, long n) { for (long i = 0; i < n; i++) o[i] = exp(x[i]); }
void libm_log10_n(const double* x, double* o, long n) { for (long i = 0; i < n; i++) o[i] = log10(x[i]); }
void libm_pow_n(const double* x, const double* y, double* o, long n) { for (long i = 0; i < n; i++) o[i] = pow(x[i], y[i]); }
void libm_sincos_n(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) { s[i] = sin(x[i]); c[i] = cos(x[i]); } }
void libm_atan2_n(const double* y, const double* x, double* out, long n) { for (long i = 0; i < n; i++) out[i] = atan2(y[i], x[i]); }
void libm_atan_n(const double* v, double* out, long n) { for (long i = 0; i < n; i++) out[i] = atan(v[i]); }
}
