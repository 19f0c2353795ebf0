// k_gauss.hip -- K1: fused value remap + separable Gaussian x0.3 downsample (gfx950).
//
// Replaces the prologue remap (LSD/myLSD.cpp:135-142) and GaussianSampler (LSD/myLSD.cpp:378-484).
// One workgroup produces a 32x24 tile (TW x TH) of the scaled image: the (reflected, remapped) u8 source
// window is staged in LDS with coalesced row-major loads, the x-pass writes an fp64 LDS strip,
// the y-pass reads it back column-wise.  The reference's aux[H][w] image never exists in HBM.
//
// Bit-exactness: same taps (computed on the host with the host libm), same accumulation order
// (newVal += pix * ker[i], i ascending), fp64, no FMA contraction (-ffp-contract=off).
#include "lsd_internal.h"
#include <algorithm>

namespace lsdhip {

// Output tile: 32 wide, 24 high.  The height sets the LDS a workgroup needs (the x-pass sums of the window's rows: 38 KB at 24, 49 KB
// at 32, 27 KB at 16) against the rows of the window that neighbouring tiles compute twice; the kernel is bound by the latency of its
// staging, so workgroups per CU count: 32 -> 24 rows (four workgroups per CU instead of three) 2.58 -> 2.35 ms on the bench batch, 16
// rows (five) 2.83 (profiles/r06g_k1_tile_heights.log; same bits).
#ifndef LSD_K1_TH
#define LSD_K1_TH 24
#endif
constexpr int TW = 32, TH = LSD_K1_TH, NT = 256;

__device__ __forceinline__ int reflect_idx(int j, int lim) {  // myLSD.cpp:436-443
    const int dou = 2 * lim;
    while (j < 0) j += dou;
    while (j >= dou) j -= dou;
    if (j >= lim) j = dou - j - 1;
    return j;
}

// centre_of(x) = (int)floor(x / sca + 0.5) (myLSD.cpp:428 / :460) comes from a table the host fills with exactly that expression
// (lsd_ctx.hip: ensure_tables): an fp64 division per use was a fifth of this kernel's vector instructions.

// HS > 0: the tap count is a compile-time constant (17 for the reference's sca = 0.3, sig = 0.6): the x-pass keeps its
// column's taps in registers and both passes are fully unrolled.  HS == 0: any tap count.
template <int HS>
__global__ __launch_bounds__(NT) void k_gauss(const uint8_t* __restrict__ in, double* __restrict__ out,
                                              const double* __restrict__ taps_g, const int* __restrict__ centre_of, int W, int H, int w, int h, int gp,
                                              int tapR, int IWp, int IHmax, unsigned gx, unsigned gy, unsigned tiles,
                                              uint8_t* __restrict__ clr) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int hSize = HS > 0 ? HS : 2 * tapR + 1;
    double* aux = reinterpret_cast<double*>(smem);                // [IHmax][TW]
    double* taps = aux + (size_t)IHmax * TW;                      // [3][hSize]
    uint8_t* tile = reinterpret_cast<uint8_t*>(taps + 3 * hSize); // [IHmax][IWp]

    const int tid = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own L2): the tiles are numbered
    // such that one XCD walks a contiguous eighth of the batch, so the 15-pixel halo a tile shares with its neighbours comes out
    // of the L2 the neighbouring tile has just filled.
    const unsigned per = (tiles + 7u) >> 3;
    const unsigned t = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (t >= tiles) return;
    const unsigned bx = t % gx, tq = t / gx, by = tq % gy;
    const int X0 = (int)bx * TW, Y0 = (int)by * TH;
    const size_t img = tq / gy;
    const uint8_t* src = in + img * (size_t)W * H;
    double* dst = out + img * (size_t)gp * h;                     // rows padded to gp doubles (128-byte aligned rows for K2)

    if (clr) {
        // lineIm = Mat::zeros (myLSD.cpp:215), fused: this kernel is bound by its fp64 filter and LDS staging, its store path idles, so every
        // tile clears the t-th of its image's `tiles per image` contiguous shares of the raster on the way (16 bytes per lane and store,
        // non-temporal; launch_gauss only passes the raster when its size and address are multiples of 16).  As a kernel of its own the
        // clear took 0.44 ms of the bench step.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const unsigned tpi = gx * gy, ti = t % tpi;
        const size_t units = ((size_t)W * H) >> 4;
        const size_t a = units * ti / tpi, e = units * (ti + 1) / tpi;
        u32x4* const v = reinterpret_cast<u32x4*>(clr + img * (size_t)W * H);
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (size_t u = a + tid; u < e; u += NT) __builtin_nontemporal_store(z, &v[u]);
    }
    const int Xl = min(X0 + TW - 1, w - 1), Yl = min(Y0 + TH - 1, h - 1);
    const int c0 = centre_of[X0] - tapR, c1 = centre_of[Xl] + tapR;
    const int r0 = centre_of[Y0] - tapR, r1 = centre_of[Yl] + tapR;
    const int IH = r1 - r0 + 1;

    for (int i = tid; i < 3 * hSize; i += NT) taps[i] = taps_g[i];

    // stage the source window as 32-bit words (window columns start at a0 = c0 rounded down to a multiple of 4): every
    // thread first issues all its loads (up to 16 words in flight), then remaps (myLSD.cpp:135-142) and stores to LDS
    const int a0 = c0 - (((c0 % 4) + 4) % 4);
    // bytes == 1 -> 255, bytes == 255 -> 0 (myLSD.cpp:135-142) on the four bytes of a word, except where `keep` has a byte of ones
    auto remap4 = [](uint32_t x, uint32_t keep) -> uint32_t {
        uint32_t t1 = x ^ 0x01010101u, t2 = ~x;                   // zero bytes mark the two cases
        t1 = ~(((t1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t1 | 0x7f7f7f7fu);   // 0x80 in every byte that was zero
        t2 = ~(((t2 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t2 | 0x7f7f7f7fu);
        const uint32_t m1 = (t1 | (t1 - (t1 >> 7))) & ~keep, m255 = (t2 | (t2 - (t2 >> 7))) & ~keep;   // 0x80 -> 0xff (no multiply)
        return (x | m1) & ~m255;
    };
    uint32_t wany = 0u;                                           // has this thread staged anything but zeros?
    {
        // 32 word columns x 8 rows of threads: a thread keeps its word column and walks down the window 8 rows at a time, so the
        // column work (bounds, reflection, "column 0 keeps raw values") is done once and nothing is divided
        const int DW = (c1 - a0 + 4) >> 2;                        // words per window row
        uint32_t* tile32 = reinterpret_cast<uint32_t*>(tile);
        const int DWp = IWp >> 2;
        const int tx = tid & 31, ty = tid >> 5;
        // A window that lies inside the image, off its row 0 and column 0 (which keep their raw values, Q2), with aligned words: nothing
        // is reflected and nothing exempt -- all but the tiles on the border (wave-uniform)
        const bool plain = a0 >= 4 && a0 + 4 * DW <= W && r0 >= 1 && r1 < H && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0;
        if (plain) {
            // whole groups of 8 window rows (one per row of threads) under wave-uniform conditions: a scalar row base plus one 32-bit
            // lane offset; the window's last IH % 8 rows under a lane mask, once
            const int nfull = IH >> 3, tail = IH & 7;
            const uint8_t* const sb = src + (size_t)r0 * W + a0;
            for (int cw = tx; cw < DW; cw += 32) {
                const uint32_t voff = (uint32_t)ty * (uint32_t)W + 4u * (uint32_t)cw;
                uint32_t* const t0 = tile32 + ty * DWp + cw;
                for (int g0 = 0; g0 < nfull; g0 += 16) {
                    uint32_t v[16];
                    #pragma unroll
                    for (int j = 0; j < 16; j++) {
                        v[j] = 0u;
                        if (g0 + j < nfull) v[j] = *reinterpret_cast<const uint32_t*>(sb + (size_t)(g0 + j) * 8u * (size_t)W + voff);
                    }
                    #pragma unroll
                    for (int j = 0; j < 16; j++) {
                        // words of 0 (free) and of 255s (unknown: 0 after the remap) are most of an occupancy map: where a whole wavefront
                        // sees nothing else there is nothing to compute
                        if (g0 + j < nfull) { const uint32_t rm = __ballot(v[j] + 1u > 1u) != 0ull ? remap4(v[j], 0u) : 0u; t0[(g0 + j) * 8 * DWp] = rm; wany |= rm; }
                    }
                }
                if (ty < tail) { const uint32_t rm = remap4(*reinterpret_cast<const uint32_t*>(sb + (size_t)nfull * 8u * (size_t)W + voff), 0u); t0[nfull * 8 * DWp] = rm; wany |= rm; }
            }
        } else
        for (int cw = tx; cw < DW; cw += 32) {
            const int gx0 = a0 + 4 * cw;
            const bool fast_col = gx0 >= 0 && gx0 + 3 < W;        // the whole word lies inside the image
            int gxr[4];
            uint32_t colkeep = 0u;                                // bytes of source column 0: exempt from the remap (Q2)
            #pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                gxr[k2] = reflect_idx(gx0 + k2, W);
                if (gxr[k2] == 0) colkeep |= 0xffu << (8 * k2);
            }
            for (int rb = 0; rb < IH; rb += 8 * 16) {
                uint32_t v[16];
                #pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int r = rb + ty + 8 * j;
                    v[j] = 0u;
                    if (r < IH) {
                        const int gy = reflect_idx(r0 + r, H);
                        const size_t off = (size_t)gy * W + gx0;
                        if (fast_col && (off & 3) == 0) v[j] = *reinterpret_cast<const uint32_t*>(src + off);
                        else {
                            const uint8_t* row = src + (size_t)gy * W;
                            v[j] = (uint32_t)row[gxr[0]] | ((uint32_t)row[gxr[1]] << 8) | ((uint32_t)row[gxr[2]] << 16) | ((uint32_t)row[gxr[3]] << 24);
                        }
                    }
                }
                #pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int r = rb + ty + 8 * j;
                    if (r < IH) {
                        // row 0 and column 0 keep their raw values (Q2)
                        const uint32_t x = v[j];
                        const uint32_t rm = x != 0u ? remap4(x, reflect_idx(r0 + r, H) == 0 ? 0xffffffffu : colkeep) : 0u;
                        tile32[r * DWp + cw] = rm; wany |= rm;
                    }
                }
            }
        }
    }
    // A window of zeros -- free and unknown cells only: 47 % of the bench maps' windows -- gives a tile of +0.0 (every product is +0.0 * tap,
    // every sum +0.0 + +0.0): both passes and their barrier are skipped (K1 2.92 -> 2.59 ms on the bench batch, same bits).  Measured with
    // it and dropped: one word per window row saying which of its words / x-pass sums are non-zero, so that the passes' zero tests
    // read one word instead of 5 words / 17 doubles -- 0.19 ms SLOWER: the kernel is bound by the latency of its staging, not by those tests.
    const int X = tid & (TW - 1);
    const int gX = X0 + X;
    if (!__syncthreads_or((int)(wany != 0u))) {
        if (gX < w)
            for (int Y = tid / TW; Y < TH && Y0 + Y < h; Y += NT / TW) dst[(size_t)(Y0 + Y) * gp + gX] = 0.0;
        return;
    }
    {
        const int cb = (gX < w ? centre_of[gX] : 0) - tapR - a0;            // first tap's column inside the window
        const double* ker = taps + (gX % 3) * hSize;
        if (gX < w) {
            if (HS > 0) {
                static_assert(HS <= 17, "the x-pass reads 5 words = 17 bytes at any byte offset");
                double kr[HS > 0 ? HS : 1];
                #pragma unroll
                for (int i = 0; i < HS; i++) kr[i] = ker[i];
                // the 17 window bytes start at any byte offset: read the 5 aligned words that hold them (conflict-free: the
                // lanes of a row spread over ~27 banks and the odd row pitch separates the wave's two rows) and shift them into place
                const uint32_t* trow = reinterpret_cast<const uint32_t*>(tile) + (cb >> 2);
                const uint32_t sh = (uint32_t)(cb & 3);
                const int DWp = IWp >> 2;
                for (int r = tid / TW; r < IH; r += NT / TW) {
                    const uint32_t* t4 = trow + r * DWp;
                    const uint32_t d0 = t4[0], d1 = t4[1], d2 = t4[2], d3 = t4[3], d4 = t4[4];
                    // occupancy maps are mostly zeros after the remap (free and unknown cells): where the whole wavefront sees
                    // zeros the sum is +0.0 exactly (every term is +0.0 * tap = +0.0, and +0.0 + +0.0 = +0.0)
                    if (__ballot((d0 | d1 | d2 | d3 | d4) != 0u) == 0ull) { aux[r * TW + X] = 0.0; continue; }
                    uint32_t wv[5];
                    wv[0] = __builtin_amdgcn_alignbyte(d1, d0, sh); wv[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    wv[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); wv[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
                    wv[4] = d4 >> (8u * sh);
                    double v = 0;
                    #pragma unroll
                    for (int i = 0; i < HS; i++) v += (double)(int)((wv[i >> 2] >> (8 * (i & 3))) & 0xffu) * kr[i];

                    aux[r * TW + X] = v;
                }
            } else {
                for (int r = tid / TW; r < IH; r += NT / TW) {
                    const uint8_t* t = tile + r * IWp + cb;
                    double v = 0;
                    for (int i = 0; i < hSize; i++) v += (double)(int)t[i] * ker[i];
                    aux[r * TW + X] = v;
                }
            }
        }
    }
    __syncthreads();

    // y-pass (myLSD.cpp:452-482): out[Y][X] = sum_i aux[yc-h+i][X] * ker[Y%3][i]
    if (gX < w) {
        for (int Y = tid / TW; Y < TH; Y += NT / TW) {
            const int gY = Y0 + Y;
            if (gY >= h) break;
            const int rb = centre_of[gY] - tapR - r0;
            const double* ker = taps + (gY % 3) * hSize;
            double v = 0;
            if (HS > 0) {
                double a[HS > 0 ? HS : 1];
                uint32_t bits = 0u;                            // (the x-pass writes sums of non-negative terms: +0.0 is the only zero)
                #pragma unroll
                for (int i = 0; i < HS; i++) { a[i] = aux[(rb + i) * TW + X]; bits |= (uint32_t)__double2hiint(a[i]) | (uint32_t)__double2loint(a[i]); }
                if (__ballot(bits != 0u) != 0ull) {                    // (all zeros: the sum is +0.0, as in the x-pass)
                    #pragma unroll
                    for (int i = 0; i < HS; i++) v += a[i] * ker[i];
                }
            } else {
                for (int i = 0; i < hSize; i++) v += aux[(rb + i) * TW + X] * ker[i];
            }
            dst[(size_t)gY * gp + gX] = v;
        }
    }
}

// Observable side effect of the reference: the caller's image is rewritten in place (myLSD.cpp:135-142): 1 -> 255, 255 -> 0 for y >= 1,
// x >= 1.  16 bytes per lane where the rows are whole 16-byte units (the byte masks of the staging above, four words at a time), and a
// unit is written back only if it changes -- free space is most of an occupancy map, so most units are only read: 512 maps of 2048^2
// 3.2 -> see DESIGN.md section 5 (one byte per thread with a 64-bit division each before).
__device__ __forceinline__ uint32_t remap_word(uint32_t x, uint32_t keep) {
    uint32_t t1 = x ^ 0x01010101u, t2 = ~x;                   // zero bytes mark the two cases
    t1 = ~(((t1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t1 | 0x7f7f7f7fu);   // 0x80 in every byte that was zero
    t2 = ~(((t2 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t2 | 0x7f7f7f7fu);
    const uint32_t m1 = (t1 | (t1 - (t1 >> 7))) & ~keep, m255 = (t2 | (t2 - (t2 >> 7))) & ~keep;
    return (x | m1) & ~m255;
}
__global__ __launch_bounds__(256) void k_remap_inplace16(uint8_t* __restrict__ img, uint32_t W, uint32_t H, size_t units) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* const v = reinterpret_cast<u32x4*>(img);
    const uint32_t upr = W >> 4;                                  // units per row
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += stride) {
        const u32x4 x = v[u];
        if ((x.x | x.y | x.z | x.w) == 0u) continue;              // free space: nothing to rewrite
        const uint32_t row = (uint32_t)(u / upr), col = (uint32_t)(u - (size_t)row * upr);
        if (row % H == 0u) continue;                              // row 0 of an image keeps its raw values (Q2)
        u32x4 r;
        r.x = remap_word(x.x, col == 0u ? 0xffu : 0u);            // ... and so does column 0
        r.y = remap_word(x.y, 0u); r.z = remap_word(x.z, 0u); r.w = remap_word(x.w, 0u);
        if ((r.x ^ x.x) | (r.y ^ x.y) | (r.z ^ x.z) | (r.w ^ x.w)) v[u] = r;
    }
}
__global__ __launch_bounds__(256) void k_remap_inplace(uint8_t* __restrict__ img, int W, int H, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t per = (size_t)W * H;
    for (; i < total; i += stride) {
        const size_t p = i % per;
        const int y = (int)(p / W), x = (int)(p % W);
        if (y >= 1 && x >= 1) {
            uint8_t v = img[i];
            if (v == 1) img[i] = 255; else if (v == 255) img[i] = 0;
        }
    }
}

// clr: lineIm to be cleared on the way (null: none; the caller has checked that every image's raster is a whole number of 16-byte words)
void launch_gauss(const Geom& g, const Buffers& b, int n, uint8_t* clr, hipStream_t s) {
    const int span = (int)floor((TW - 1) / g.sca) + 2;            // bound on centre(X0+31) - centre(X0) + 1
    const int IWmax = span + 2 * g.tapR + 1;
    int IWp = ((IWmax + 3) & ~3) + 8;                             // + the alignment slack of the word-wise staging and of the x-pass's 5-word reads
    if (((IWp >> 2) & 1) == 0) IWp += 4;                          // odd pitch in 32-bit words: consecutive rows start in different LDS banks
    const int IHmax = (int)floor((TH - 1) / g.sca) + 2 + 2 * g.tapR + 1;   // (as IWmax, for the tile's height)
    const int hSize = 2 * g.tapR + 1;
    const size_t lds = (size_t)IHmax * TW * sizeof(double) + 3 * hSize * sizeof(double) + (size_t)IHmax * IWp;
    auto kern = hSize == 17 ? k_gauss<17> : k_gauss<0>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned gx = (g.w + TW - 1) / TW, gy = (g.h + TH - 1) / TH, tiles = gx * gy * (unsigned)n;
    hipLaunchKernelGGL(kern, dim3(((tiles + 7u) >> 3) * 8u), dim3(NT), lds, s, b.in, b.gauss, b.taps, b.centres, g.W, g.H, g.w, g.h, g.gp,
                       g.tapR, IWp, IHmax, gx, gy, tiles, clr);
}

void launch_remap_writeback(const Geom& g, const Buffers& b, int n, hipStream_t s) {
    if (!b.in_rw) return;
    const size_t total = (size_t)n * g.W * g.H;
    if ((g.W & 15) == 0 && (reinterpret_cast<uintptr_t>(b.in_rw) & 15) == 0) {      // rows of whole 16-byte units
        const size_t units = total >> 4;
        const int blocks = (int)std::min<size_t>((units + 255) / 256, 16384);
        hipLaunchKernelGGL(k_remap_inplace16, dim3(blocks), dim3(256), 0, s, b.in_rw, (uint32_t)g.W, (uint32_t)g.H, units);
        return;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_remap_inplace, dim3(blocks), dim3(256), 0, s, b.in_rw, g.W, g.H, total);
}

}  // namespace lsdhip
