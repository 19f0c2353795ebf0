// k_lines.hip -- K5: accepted rectangles -> structLinesInfo records + lineIm raster (gfx950).
//
// Replaces myLSD.cpp:280-368 (and sind/cosd/atand, LSD/baseFunc.cpp:6-16).  One workgroup per image,
// one wavefront per line (strided); lanes walk the samples of the longer axis.  lineIm has been
// cleared by a memset node on the same stream; marking 255 is idempotent so store order is free.
#include "lsd_internal.h"
#include "devmath.h"

namespace lsdhip {

__global__ __launch_bounds__(256) void k_lines(const double* __restrict__ recs_scaled, const int32_t* __restrict__ counts,
                                               lsd_line* __restrict__ lines, uint8_t* __restrict__ line_im,
                                               int max_lines, int W, int H) {
    const size_t img = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int n = counts[img];
    if (n > max_lines) n = max_lines;
    const double* rs = recs_scaled + img * (size_t)max_lines * 4;
    lsd_line* out = lines + img * (size_t)max_lines;
    uint8_t* im = line_im ? line_im + img * (size_t)W * H : nullptr;

    for (int i = wave; i < n; i += 4) {
        const double x1 = rs[i * 4 + 0], y1 = rs[i * 4 + 1], x2 = rs[i * 4 + 2], y2 = rs[i * 4 + 3];
        const double k = (y2 - y1) / (x2 - x1);                                    // :289
        double ang = atan_g(k) * 180.0 / kPi;                                        // atand, baseFunc.cpp:14-16
        int orient = 1;
        if (ang < 0) { ang += 180; orient = -1; }                                  // :292-295
        int xLow, xHigh, yLow, yHigh;
        if (x1 > x2) { xLow = cvt_x86(floor(x2)); xHigh = cvt_x86(ceil(x1)); }     // :298-305
        else         { xLow = cvt_x86(floor(x1)); xHigh = cvt_x86(ceil(x2)); }
        if (y1 > y2) { yLow = cvt_x86(floor(y2)); yHigh = cvt_x86(ceil(y1)); }     // :306-313
        else         { yLow = cvt_x86(floor(y1)); yHigh = cvt_x86(ceil(y2)); }
        const double xRang = fabs(x2 - x1), yRang = fabs(y2 - y1);
        const int xx_len = xHigh - xLow + 1, yy_len = yHigh - yLow + 1;
        if (im) {
            // Q10: only the sampled part of the reference's marking loop is defined behaviour
            if (xRang > yRang) {                                                   // :319-330
                for (int j = lane; j < xx_len; j += 64) {
                    const int xx = j + xLow;
                    const int yy = cvt_x86(round((xx - x1) * k + y1));
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    if (xx != 0 && yy != 0) im[(size_t)yy * W + xx] = 255;        // :346
                }
            } else {                                                               // :331-342
                for (int j = lane; j < yy_len; j += 64) {
                    const int yy = j + yLow;
                    const int xx = cvt_x86(round((yy - y1) / k + x1));
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    if (xx != 0 && yy != 0) im[(size_t)yy * W + xx] = 255;        // :352
                }
            }
        }
        if (lane == 0) {
            lsd_line L;
            L.k = k;
            L.b = (y1 + y2) / 2.0 - k * (x1 + x2) / 2.0;                           // :359
            sincos_g(ang / 180.0 * kPi, L.dy, L.dx);                               // sind / cosd
            L.x1 = x1; L.y1 = y1; L.x2 = x2; L.y2 = y2;
            const double ey = y2 - y1, ex = x2 - x1;
            L.len = sqrt(ey * ey + ex * ex);                                       // :366
            L.orient = orient;
            out[i] = L;
            reinterpret_cast<uint32_t*>(&out[i])[19] = 0u;      // the tail padding of structLinesInfo: defined bytes (records are compared and moved as words)
        }
    }
}

// lineIm = Mat::zeros (myLSD.cpp:215): the raster is cleared by the library itself, 16 bytes per lane and store, each wavefront a
// run of consecutive 1 KB pieces (whole 128-byte lines, no read), ~2 workgroups per CU; head and tail bytes that do not fill a
// 16-byte word are written singly.  (hipMemsetAsync's fill kernel reached 2.3 TB/s on the 2.1 GB of the bench batch.)
__global__ __launch_bounds__(256) void k_clear16(uint8_t* __restrict__ p, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const size_t head = min((size_t)((16 - (a & 15)) & 15), bytes);
    const size_t nvec = (bytes - head) >> 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* v = reinterpret_cast<u32x4*>(p + head);
    const u32x4 z = {0u, 0u, 0u, 0u};
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    // a wavefront writes 4 consecutive 1 KB pieces per step (4 stores in flight per lane)
    const size_t wv = tid >> 6, lane = tid & 63, nwv = nth >> 6;
    for (size_t base = wv * 256; base < nvec; base += nwv * 256) {
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t i = base + 64 * j + lane;
            if (i < nvec) __builtin_nontemporal_store(z, &v[i]);
        }
    }
    if (tid < head) p[tid] = 0;
    const size_t tail0 = head + (nvec << 4);
    if (tail0 + tid < bytes) p[tail0 + tid] = 0;
}

void launch_clear(uint8_t* p, size_t bytes, int num_cus, hipStream_t s) {
    if (!p || !bytes) return;
    size_t blocks = (bytes + 16 * 256 * 4 - 1) / (16 * 256 * 4);
    const size_t cap = (size_t)num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_clear16, dim3((unsigned)blocks), dim3(256), 0, s, p, bytes);
}

// Host entry points: the per-image line arrays (max_lines apart) compacted into one flat array + prefix offsets, so that
// the host fetches exactly the lines there are in one copy.  offsets[n + 3]: offsets[n] = total, offsets[n + 1] = number of
// images with more than max_lines lines (they are clamped; the host reports LSD_ERR_CAPACITY), offsets[n + 2] = 1 + index of
// the first image the region stage gave up (counts == -1, its watchdog; 0: none -- such an image contributes no lines and the
// host reports LSD_ERR_INTERNAL).  Two launches: a one-workgroup scan of the counts, then one workgroup per image moves its
// records (size_t indices: n * max_lines * 10 words may pass 2^31).
__device__ __forceinline__ int kept_lines(int count, int max_lines) { return max(0, min(count, max_lines)); }
__global__ __launch_bounds__(256) void k_scan_counts(const int32_t* __restrict__ counts, int max_lines, int n, int32_t* __restrict__ offsets) {
    __shared__ int s_part[256];
    __shared__ int s_over[256];
    __shared__ int s_bad[256];
    const int t = threadIdx.x;
    const int per = (n + 255) / 256, lo = min(t * per, n), hi = min(lo + per, n);
    int sum = 0, over = 0, bad = 0;
    for (int i = lo; i < hi; i++) {
        sum += kept_lines(counts[i], max_lines); over += counts[i] > max_lines ? 1 : 0;
        if (counts[i] < 0 && !bad) bad = i + 1;
    }
    s_part[t] = sum; s_over[t] = over; s_bad[t] = bad;
    __syncthreads();
    if (t == 0) {
        int run = 0, ov = 0, bad = 0;
        for (int j = 0; j < 256; j++) { const int v = s_part[j]; s_part[j] = run; run += v; ov += s_over[j]; if (!bad) bad = s_bad[j]; }
        offsets[n] = run; offsets[n + 1] = ov; offsets[n + 2] = bad;
    }
    __syncthreads();
    int run = s_part[t];
    for (int i = lo; i < hi; i++) { offsets[i] = run; run += kept_lines(counts[i], max_lines); }
}
__global__ __launch_bounds__(256) void k_compact_lines(const lsd_line* __restrict__ lines, const int32_t* __restrict__ counts,
                                                       int max_lines, const int32_t* __restrict__ offsets, lsd_line* __restrict__ flat) {
    const size_t i = blockIdx.x;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(lines) + i * (size_t)max_lines * 10;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(flat) + (size_t)offsets[i] * 10;
    const size_t k = (size_t)kept_lines(counts[i], max_lines) * 10;
    for (size_t j = threadIdx.x; j < k; j += 256) dst[j] = src[j];
}

void launch_compact_lines(const lsd_line* lines, const int32_t* counts, int max_lines, int n, lsd_line* flat, int32_t* offsets, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(256), 0, s, counts, max_lines, n, offsets);
    hipLaunchKernelGGL(k_compact_lines, dim3(n), dim3(256), 0, s, lines, counts, max_lines, offsets, flat);
}

void launch_lines(const Geom& g, const Buffers& b, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_lines, dim3(n), dim3(256), 0, s, b.recs_scaled, b.counts, b.lines, b.line_im, b.max_lines,
                       g.W, g.H);
}

}  // namespace lsdhip
