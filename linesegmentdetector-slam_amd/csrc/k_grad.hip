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
// coalesced 512-byte rows and no LDS staging.  The Gaussian image comes with a row pitch of gp doubles (a multiple of
// 16): every 512-byte row segment a wavefront loads covers exactly four 128-byte lines.
// Three phases so that the expensive, correctly rounded atan2 / sincos (devmath.h) never run in a half-empty
// wavefront and every map is stored in whole coalesced rows:
//   1  the dense part: gradient, magnitude (stored), threshold, max, and the angle of the ~93 % zero-gradient pixels,
//      kept in registers; the non-zero-gradient pixels of the strip go into an LDS list (index + gradient);
//   2  the list, all lanes busy: angle (left in the list slot) and, where the pixel stays growable, (sin, cos) -- the only
//      scattered store; every lane then picks the angles of its own pixels out of the list;
//   3  angle map and packed pixel words, whole rows from registers.
// (The list holds CAP entries; a strip with more non-zero gradients runs phase 2 in between.)
// Measured alternatives, all slower: a workgroup walking the whole width or a whole column band (fewer
// partly written lines / better filled phase 2, but too few independent loads in flight), with and without loads one step ahead.
constexpr int CAP = 256;

__global__ __launch_bounds__(GX) void k_gradient(const double* __restrict__ gauss, double* __restrict__ mag,
                                                 double* __restrict__ deg, double2* __restrict__ sc,
                                                 uint32_t* __restrict__ pw,
                                                 unsigned long long* __restrict__ maxbits, int32_t* __restrict__ ties, int w, int h, int gp,
                                                 double gradThre, unsigned gx, unsigned gy, unsigned strips) {
    __shared__ uint32_t l_px[CAP];           // (strip-local pixel index << 1) | growable
    __shared__ double2 l_g[CAP];             // in: (gradX, gradY) of the listed pixel; out: (angle, -)
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own L2), so the strips are
    // numbered such that one XCD walks a contiguous eighth of the batch: the row above a strip and the column left of it
    // then come out of the L2 that the neighbouring strip has just filled instead of from HBM again.
    const unsigned per = (strips + 7u) >> 3;
    const unsigned t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (t >= strips) return;
    const unsigned bx = t % gx, tq = t / gx, by = tq % gy;
    const size_t img = tq / gy;
    const size_t base = img * (size_t)w * h;
    const double* __restrict__ gim = gauss + img * (size_t)gp * h;
    const int lane = threadIdx.x;
    const int x = (int)bx * GX + lane, y0 = (int)by * GR;
    const bool colok = x < w;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    int cnt = 0;                                               // entries in the LDS list (wave-uniform)
    double mx = 0;
    double rowD[GR];                                           // angle of this lane's pixel in row r ...
    uint32_t rowU[GR];                                         // ... its usedMap code ...
    int rowSlot[GR];                                           // ... and, while the angle is still to come, its list slot (else -1)

    auto flush = [&]() {
        // phase 2: level-line angle (+ sin/cos where the pixel stays growable) of the listed pixels
        for (int i = lane; i < cnt; i += GX) {
            const uint32_t e = l_px[i];
            const double2 gr = l_g[i];
            double d;                                          // :169 (first stage inline, second stage out of line: devmath.h)
            if (!crm::atan2_fast(gr.x, -gr.y, d)) d = atan2_g(gr.x, -gr.y);
            if (fabs(fabs(d - kPi) - 0.000001) <= 1e-14) atomicAdd(&ties[img], 1);   // within an ulp of atan2 of the rule's threshold: a decision another libm could take differently (lsd_last_sensitivity)
            if (fabs(d - kPi) < 0.000001) d = 0;               // :170-171
            l_g[i] = make_double2(d, 0.0);
            if (e & 1u) {                                      // sin/cos(deg) for RegionGrower (:545-546)
                const int lt = (int)(e >> 1);
                const size_t p = base + (size_t)(y0 + lt / GX) * w + ((int)bx * GX + lt % GX);
                double sv, cv;
                if (!crm::sincos_fast(d, sv, cv)) sincos_g(d, sv, cv);
                sc[p] = make_double2(sv, cv);
            }
        }
        #pragma unroll
        for (int r = 0; r < GR; r++)
            if (rowSlot[r] >= 0) { rowD[r] = l_g[rowSlot[r]].x; rowSlot[r] = -1; }
        cnt = 0;
    };

    // row above the strip: C = G[y-1][x], D = G[y-1][x-1]
    double up = 0, upl = 0;
    if (y0 >= 1 && colok) {
        up = gim[(size_t)(y0 - 1) * gp + x];
        if (lane == 0 && x >= 1) upl = gim[(size_t)(y0 - 1) * gp + x - 1];
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
            rowA[r] = gim[(size_t)y * gp + x];
            if (lane == 0 && x >= 1) rowB0[r] = gim[(size_t)y * gp + x - 1];
        }
    }
    #pragma unroll
    for (int r = 0; r < GR; r++) {
        const int y = y0 + r;
        rowD[r] = 0; rowU[r] = 0; rowSlot[r] = -1;
        if (y >= h) continue;
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
            if (!heavy && u == 0) sc[p] = make_double2(0.0, 1.0);   // row 0 / col 0: angle 0 exactly, growable (Q3)
        }
        rowD[r] = d; rowU[r] = u;
        const unsigned long long hm = __ballot(heavy);
        if (heavy) {
            const int slot = cnt + __builtin_popcountll(hm & ltmask);
            l_px[slot] = ((uint32_t)(r * GX + lane) << 1) | (u == 0 ? 1u : 0u);
            l_g[slot] = make_double2(gradX, gradY);
            rowSlot[r] = slot;
        }
        cnt += __builtin_popcountll(hm);
        mx = fmax(mx, m);
        up = A; upl = B;
        if (cnt > CAP - GX) flush();                           // (the next row may add up to GX entries)
    }
    // per-image max (myLSD.cpp:167-168): non-negative doubles order like their bit patterns
    unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(bits, off);
        bits = o > bits ? o : bits;
    }
    if (lane == 0 && bits != 0ull) atomicMax(&maxbits[img], bits);

    flush();
    // phase 3: degMap and the packed pixel words, whole rows
    if (colok) {
        #pragma unroll
        for (int r = 0; r < GR; r++) {
            const int y = y0 + r;
            if (y < h) {
                const size_t p = base + (size_t)y * w + x;
                deg[p] = rowD[r];
                pw[p] = pack_pw(rowD[r], rowU[r]);
            }
        }
    }
}

void launch_gradient(const Geom& g, const Buffers& b, int n, hipStream_t s) {
    const unsigned gx = (g.w + GX - 1) / GX, gy = (g.h + GR - 1) / GR, strips = gx * gy * (unsigned)n;
    hipLaunchKernelGGL(k_gradient, dim3(((strips + 7u) >> 3) * 8u), dim3(GX), 0, s, b.gauss, b.mag, b.deg, b.sc, b.pw, b.maxbits, b.ties,
                       g.w, g.h, g.gp, g.gradThre, gx, gy, strips);
}

}  // namespace lsdhip
