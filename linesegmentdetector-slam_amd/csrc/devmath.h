// devmath.h -- the transcendental functions used on the LSD path, glibc-compatible rounding.
// sin/cos/atan/atan2 go through crmath.h (correctly rounded, see there for why); the rare inputs it
// declines (non-finite, |x| > 64, subnormal range) fall back to the device math library.
// exp / log10 / pow (RectangleNFACalculator): correctly rounded likewise (exp_g, log10_g, pow_g).
#pragma once
#include "crmath.h"

namespace lsdhip {

// Out of line on purpose: the double-double bodies are ~2-3 KB of code each and the region kernel has to fit the
// instruction cache; they are called a few times per region, never per pixel of the serial chain.
// (the pair comes back in registers: results handed back through references travel through scratch memory)
__device__ __noinline__ double2 sincos_g2(double x) {
    double s, c;
    if (!crm::sincos_fast(x, s, c))                            // first stage (Ziv): certain roundings only
        if (!crm::sincos_cr(x, s, c)) { s = sin(x); c = cos(x); }
    return make_double2(s, c);
}
__device__ __forceinline__ void sincos_g(double x, double& s, double& c) {
    const double2 v = sincos_g2(x);
    s = v.x; c = v.y;
}
__device__ inline double sin_g(double x) { double s, c; sincos_g(x, s, c); return s; }
__device__ inline double cos_g(double x) { double s, c; sincos_g(x, s, c); return c; }
__device__ __noinline__ double atan2_g(double y, double x) {
    double r;
    if (crm::atan2_fast(y, x, r)) return r;                    // first stage (Ziv): certain roundings only
    if (!crm::atan2_cr(y, x, r)) r = atan2(y, x);
    return r;
}
__device__ inline double atan_g(double v) {
    double r;
    if (!crm::atan_cr(v, r)) r = atan(v);
    return r;
}

// RectangleNFACalculator's libm calls (myLSD.cpp:1034, :1040, :1052-1053, :1057), correctly rounded.  Out of line: ~1.5 KB each, called
// twice per rectangle rating (the pow / log10 of the tail's stopping test only where the fast device functions cannot decide it).
__device__ __noinline__ double exp_g(double x) {
    double r;
    if (crm::exp_fast(x, r)) return r;                         // first stage (Ziv): certain roundings only
    return crm::exp_cr(x);
}
__device__ __noinline__ double log10_g(double x) {
    double r;
    if (crm::log10_fast(x, r)) return r;
    return crm::log10_cr(x);
}
__device__ __noinline__ double pow_g(double x, double y) { return crm::pow_cr(x, y); }

}  // namespace lsdhip
