// k_grad.hip -- K2: gradient magnitude, level-line angle, threshold map and per-image max (gfx950).
//
// Replaces the gradient loop of myLineSegmentDetector (LSD/myLSD.cpp:152-174).
// Algorithmic traffic: 8 B read + 8+8+1 B written per scaled pixel; this kernel
// writes the usedMap value as the low bits of the packed pixel word pw (lsd_internal.h: fp32 angle +
// usedMap code, the only thing RegionGrower reads per neighbour), i.e. 8 + 20 B per pixel actually
// move, plus one 16-byte (sin, cos) pair for the ~7 % of pixels that stay growable.
#include "lsd_internal.h"
#include "devmath.h"

namespace lsdhip {

constexpr int GX = 64;     // columns per wavefront (one lane per column)
constexpr int GR = 8;      // rows walked by a wavefront

// fp32 angle with the two lowest mantissa bits replaced by the usedMap code (lsd_internal.h)
__device__ __forceinline__ uint32_t pack_pw(double d, uint32_t code) {
    return (__float_as_uint(__double2float_rn(d)) & ~3u) | code;
}

// One 64-lane workgroup owns a 64-column x 8-row strip.  Each lane walks its column downwards keeping the row
// above in registers; the left neighbour comes from the adjacent lane (lane 0 loads it), so every gauss value is
// fetched from HBM once per strip (+1/8 for the row above the strip, +1/64 for the column left of it) with fully
// coalesced 512-byte rows and no LDS staging.
// Two phases so that the expensive, correctly rounded atan2 / sincos (devmath.h) never run in a half-empty
// wavefront: phase 1 does the dense part (gradient, magnitude, threshold, max, and the angle of the ~93 %
// zero-gradient pixels) and pushes the indices of the non-zero-gradient pixels of the strip into an LDS list;
// phase 2 walks that list with all lanes busy.
__global__ __launch_bounds__(GX) void k_gradient(const double* __restrict__ gauss, double* __restrict__ mag,
                                                 double* __restrict__ deg, double2* __restrict__ sc,
                                                 uint32_t* __restrict__ pw,
                                                 unsigned long long* __restrict__ maxbits, int w, int h,
                                                 double gradThre) {
    __shared__ uint32_t l_px[GX * GR];       // (strip-local pixel index << 1) | growable
    const size_t img = blockIdx.z;
    const size_t base = img * (size_t)w * h;
    const int lane = threadIdx.x;
    const int x = blockIdx.x * GX + lane, y0 = blockIdx.y * GR;
    const bool colok = x < w;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    int cnt = 0;                                               // entries in the LDS list (wave-uniform)
    double mx = 0;

    // row above the strip: C = G[y-1][x], D = G[y-1][x-1]
    double up = 0, upl = 0;
    if (y0 >= 1 && colok) {
        up = gauss[base + (size_t)(y0 - 1) * w + x];
        if (lane == 0 && x >= 1) upl = gauss[base + (size_t)(y0 - 1) * w + x - 1];
    }
    {
        const double t = __shfl_up(up, 1);
        if (lane != 0) upl = t;
    }
    // all rows of the strip are requested before the first one is used
    double rowA[GR], rowB0[GR];
    #pragma unroll
    for (int r = 0; r < GR; r++) {
        const int y = y0 + r;
        rowA[r] = 0; rowB0[r] = 0;
        if (colok && y < h) {
            rowA[r] = gauss[base + (size_t)y * w + x];
            if (lane == 0 && x >= 1) rowB0[r] = gauss[base + (size_t)y * w + x - 1];
        }
    }
    #pragma unroll
    for (int r = 0; r < GR; r++) {
        const int y = y0 + r;
        if (y >= h) break;
        const double A = rowA[r];
        double B = rowB0[r];
        {
            const double t = __shfl_up(A, 1);
            if (lane != 0) B = t;
        }
        const double C = up, D = upl;
        double m = 0, d = 0, gradX = 0, gradY = 0;
        uint32_t u = 0;
        bool heavy = false;
        if (colok && x >= 1 && y >= 1) {                       // Q3: row 0 / col 0 stay mag=0, deg=0, used=0
            gradX = (B + D - A - C) / 2.0;                     // myLSD.cpp:161
            gradY = (C + D - A - B) / 2.0;                     // :162
            m = sqrt(gradX * gradX + gradY * gradY);           // :163 (pow(.,2) == x*x, Q12)
            if (m < gradThre) u = 1;                           // :165-166
            if (gradX == 0.0 && gradY == 0.0) {
                // atan2(+-0, -(+-0)) (:169) followed by the "pi -> 0" rule (:170-171): IEEE special cases
                if (signbit(-gradY)) d = signbit(gradX) ? -kPi : 0.0;
                else d = gradX;
            } else heavy = true;
        }
        if (colok) {
            const size_t p = base + (size_t)y * w + x;
            mag[p] = m;
            if (!heavy) {
                deg[p] = d;
                pw[p] = pack_pw(d, u);
                if (u == 0) sc[p] = make_double2(0.0, 1.0);    // row 0 / col 0: angle 0 exactly, growable (Q3)
            }
        }
        const unsigned long long hm = __ballot(heavy);
        if (heavy) {
            const int slot = cnt + __builtin_popcountll(hm & ltmask);
            l_px[slot] = ((uint32_t)(r * GX + lane) << 1) | (u == 0 ? 1u : 0u);
        }
        cnt += __builtin_popcountll(hm);
        mx = fmax(mx, m);
        up = A; upl = B;
    }
    // per-image max (myLSD.cpp:167-168): non-negative doubles order like their bit patterns
    unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(bits, off);
        bits = o > bits ? o : bits;
    }
    if (lane == 0 && bits != 0ull) atomicMax(&maxbits[img], bits);

    // phase 2: level-line angle (+ sin/cos where the pixel stays growable) of the non-zero-gradient pixels
    for (int i = lane; i < cnt; i += GX) {
        const uint32_t e = l_px[i];
        const int lt = (int)(e >> 1);
        const size_t p = base + (size_t)(y0 + lt / GX) * w + (blockIdx.x * GX + lt % GX);
        // the 2x2 stencil again (L2-resident: this strip has just been read); same expressions, same bits
        const double A = gauss[p], B = gauss[p - 1], C = gauss[p - w], D = gauss[p - w - 1];
        const double gradX = (B + D - A - C) / 2.0, gradY = (C + D - A - B) / 2.0;
        double d = atan2_g(gradX, -gradY);                     // :169
        if (fabs(d - kPi) < 0.000001) d = 0;                   // :170-171
        deg[p] = d;
        pw[p] = pack_pw(d, (e & 1u) ? kPwFree : kPwStatic);
        if (e & 1u) {                                          // sin/cos(deg) for RegionGrower (:545-546)
            double sv, cv;
            sincos_g(d, sv, cv);
            sc[p] = make_double2(sv, cv);
        }
    }
}

void launch_gradient(const Geom& g, const Buffers& b, int n, hipStream_t s) {
    dim3 grid((g.w + GX - 1) / GX, (g.h + GR - 1) / GR, n);
    hipLaunchKernelGGL(k_gradient, grid, dim3(GX), 0, s, b.gauss, b.mag, b.deg, b.sc, b.pw, b.maxbits, g.w, g.h,
                       g.gradThre);
}

}  // namespace lsdhip
