// k_grad.hip -- K2: gradient magnitude, level-line angle, threshold map and per-image max (gfx950).
//
// Replaces the gradient loop of myLineSegmentDetector (LSD/myLSD.cpp:152-174).  One thread per
// scaled pixel, 64x4 tiles; the 2x2 stencil is read through LDS (65x5 window, coalesced rows).
// Algorithmic traffic: 8 B read + 8+8+1 B written per scaled pixel; this kernel
// writes the usedMap value into the low bits of a 32-bit state word (the upper bits later hold
// the curMap stamp of the region stage), i.e. 8 + 20 B per pixel actually move, plus 16 B for the
// ~7 % of pixels that stay growable (sin/cos of their angle, consumed by the region stage).
#include "lsd_internal.h"
#include "devmath.h"

namespace lsdhip {

constexpr int GX = 64, GY = 4;

__global__ __launch_bounds__(GX* GY) void k_gradient(const double* __restrict__ gauss, double* __restrict__ mag,
                                                     double* __restrict__ deg, double* __restrict__ sn,
                                                     double* __restrict__ cs, uint32_t* __restrict__ state,
                                                     unsigned long long* __restrict__ maxbits, int w, int h,
                                                     double gradThre) {
    __shared__ double t[GY + 1][GX + 1];
    const size_t img = blockIdx.z;
    const size_t base = img * (size_t)w * h;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x0 = blockIdx.x * GX, y0 = blockIdx.y * GY;
    const int tid = ty * GX + tx;

    // window rows y0-1 .. y0+GY-1, cols x0-1 .. x0+GX-1
    for (int i = tid; i < (GY + 1) * (GX + 1); i += GX * GY) {
        const int r = i / (GX + 1), c = i % (GX + 1);
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        double v = 0;
        if (gy >= 0 && gx >= 0 && gy < h && gx < w) v = gauss[base + (size_t)gy * w + gx];
        t[r][c] = v;
    }
    __syncthreads();

    const int x = x0 + tx, y = y0 + ty;
    double m = 0;
    if (x < w && y < h) {
        double d = 0;
        uint32_t u = 0;
        if (x >= 1 && y >= 1) {                                // Q3: row 0 / col 0 stay mag=0, deg=0, used=0
            const double A = t[ty + 1][tx + 1], B = t[ty + 1][tx], C = t[ty][tx + 1], D = t[ty][tx];
            const double gradX = (B + D - A - C) / 2.0;        // myLSD.cpp:161
            const double gradY = (C + D - A - B) / 2.0;        // :162
            m = sqrt(gradX * gradX + gradY * gradY);           // :163 (pow(.,2) == x*x, Q12)
            if (m < gradThre) u = 1;                           // :165-166
            d = atan2_g(gradX, -gradY);                        // :169
            if (fabs(d - kPi) < 0.000001) d = 0;               // :170-171
        }
        const size_t p = base + (size_t)y * w + x;
        mag[p] = m;
        deg[p] = d;
        state[p] = u;
        if (u == 0) {                      // growable pixels: sin/cos of the level-line angle for RegionGrower (:545-546)
            double sv, cv;
            sincos_g(d, sv, cv);
            sn[p] = sv;
            cs[p] = cv;
        }
    }
    // per-image max (myLSD.cpp:167-168): non-negative doubles order like their bit patterns
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(bits, off);
        bits = o > bits ? o : bits;
    }
    if (((tid & 63) == 0) && bits != 0ull) atomicMax(&maxbits[img], bits);
}

void launch_gradient(const Geom& g, const Buffers& b, int n, hipStream_t s) {
    dim3 grid((g.w + GX - 1) / GX, (g.h + GY - 1) / GY, n);
    hipLaunchKernelGGL(k_gradient, grid, dim3(GX, GY), 0, s, b.gauss, b.mag, b.deg, b.sn, b.cs, b.state, b.maxbits, g.w, g.h,
                       g.gradThre);
}

}  // namespace lsdhip
