// k_sort.hip -- K3: pseudo-ordering of the gradient (bin + compact + stable descending sort), gfx950.
//
// Replaces myLSD.cpp:177-204: zoom = pseBin/maxGrad, v = floor(mag*zoom) clipped to pseBin, keep
// v != 0 in raster order, then qsort(binCell, Comp).  glibc's qsort is a stable merge sort and the
// comparator never returns 0, so the reference order is "descending v, raster order among ties"
// (SURVEY 8a-Q4).  Here that order is produced directly by a stable counting sort:
//
//   one 1024-thread workgroup per image; wave s owns the s-th contiguous raster segment;
//   pass 1  streams magMap ONCE: bins, per-wave LDS histograms hist[s][pseBin - v] (LDS atomics), and the kept pixels
//           (v != 0: one in ten on an occupancy map) compacted in raster order into the wave's own stretch of a scratch
//           list (pixel, v) -- ballot + popcount, no global atomics;
//   scan    start[s][b] = sum_{b'<b} tot[b'] + sum_{s'<s} hist[s'][b]        (block prefix sum)
//   pass 2  each wave walks ITS compacted list 64 kept pixels at a time (every lane busy).  The lanes of a chunk that share a
//           bin find each other with one ballot per bit of the bin number (peers = AND over the bits of "ballot of the bit" or
//           its complement) and take consecutive ranks in lane (= raster) order: one LDS read and one LDS write per chunk, no
//           loop over the distinct bins.
//   (Until round 5 pass 2 streamed magMap a second time and ranked the ~4.5 kept lanes of a raw 64-pixel chunk with a loop
//    over their distinct bins: 1.5 ms per 512 maps and 2.4 x the algorithmic traffic.)
//
// The result is deterministic and independent of wave scheduling.
#include "lsd_internal.h"

namespace lsdhip {

constexpr int SNT = 1024, SWAVES = SNT / 64;

__device__ __forceinline__ int bin_of(double m, double zoom, int pseBin) {
    int v = cvt_x86(floor(m * zoom));                              // myLSD.cpp:182
    if (v > pseBin) v = pseBin;                                    // :183-184
    return (int)(uint16_t)v;                                       // pseIdx is CV_16UC1 (:178,:187)
}

__global__ __launch_bounds__(SNT) void k_sort(const double* __restrict__ mag,
                                              const unsigned long long* __restrict__ maxbits,
                                              uint32_t* __restrict__ ord,
                                              int32_t* __restrict__ nb, uint32_t* __restrict__ kept_px, uint32_t* __restrict__ kept_v32,
                                              int npx, int pseBin) {
    extern __shared__ uint32_t hist[];                             // [SWAVES][pseBin] then [SWAVES] scratch
    uint32_t* wsum = hist + SWAVES * pseBin;
    const size_t img = blockIdx.x;
    const double* m = mag + img * (size_t)npx;
    uint32_t* o = ord + img * (size_t)npx;
    uint32_t* kp = kept_px + img * (size_t)npx;                    // the kept pixels in raster order, wave s's from kp[beg_s] on
    uint16_t* kv = reinterpret_cast<uint16_t*>(kept_v32 + img * (size_t)npx);   // ... and their bin values
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    const double maxGrad = __longlong_as_double((long long)maxbits[img]);
    if (!(maxGrad > 0)) {                                          // blank image: reference reads garbage; define as empty
        if (tid == 0) nb[img] = 0;
        return;
    }
    const double zoom = 1.0 * pseBin / maxGrad;                    // :179
    const int segLen = ((npx + SNT - 1) / SNT) * 64;
    const int beg = min(wave * segLen, npx), end = min(beg + segLen, npx);

    for (int i = tid; i < SWAVES * pseBin; i += SNT) hist[i] = 0;
    __syncthreads();

    uint32_t* myh = hist + wave * pseBin;
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int UN = 4;                                          // 64-pixel chunks per step: independent loads in flight per wave
    int cnt = 0;                                                   // kept pixels of this wave so far (wave-uniform)
    for (int base = beg; base < end; base += 64 * UN) {
        double mv[UN];
        #pragma unroll
        for (int j = 0; j < UN; j++) {
            const int p = base + 64 * j + lane;
            mv[j] = p < end ? m[p] : 0.0;
        }
        #pragma unroll
        for (int j = 0; j < UN; j++) {                             // chunks in raster order
            const int v = bin_of(mv[j], zoom, pseBin);             // (0.0 gives bin 0)
            const unsigned long long km = __ballot(v != 0);
            if (v != 0) {
                atomicAdd(&myh[pseBin - v], 1u);
                const int k2 = beg + cnt + __builtin_popcountll(km & lt);
                kp[k2] = (uint32_t)(base + 64 * j + lane);
                kv[k2] = (uint16_t)v;
            }
            cnt += __builtin_popcountll(km);
        }
    }
    __syncthreads();                                               // (also: this wave's own kept list is visible to it below)

    // exclusive scan over bins (descending value == ascending b), then over waves inside a bin
    {
        uint32_t tot = 0;
        if (tid < pseBin)
            for (int s = 0; s < SWAVES; s++) tot += hist[s * pseBin + tid];
        uint32_t inc = tot;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (int s = 0; s < wave; s++) wbase += wsum[s];
        uint32_t run = wbase + inc - tot;
        if (tid < pseBin)
            for (int s = 0; s < SWAVES; s++) {
                const uint32_t c = hist[s * pseBin + tid];
                hist[s * pseBin + tid] = run;
                run += c;
            }
        if (tid == SNT - 1) nb[img] = (int32_t)(wbase + inc);     // total (bins >= pseBin contribute 0)
    }
    __syncthreads();

    const int nbits = 32 - __builtin_clz((unsigned)max(pseBin - 1, 1));   // bits of a bin number b = pseBin - v in [0, pseBin)
    for (int base = 0; base < cnt; base += 64 * UN) {
        uint32_t pv[UN];
        int vv[UN];
        #pragma unroll
        for (int j = 0; j < UN; j++) {
            const int k2 = base + 64 * j + lane;
            pv[j] = 0u; vv[j] = pseBin;                            // (idle lanes: bin 0, kept apart from the others by the act mask)
            if (k2 < cnt) { pv[j] = kp[beg + k2]; vv[j] = (int)kv[beg + k2]; }
        }
        #pragma unroll
        for (int j = 0; j < UN; j++) {                             // chunks in raster order
            if (base + 64 * j >= cnt) break;                       // (wave-uniform)
            const bool act = base + 64 * j + lane < cnt;
            const int b = pseBin - vv[j];
            unsigned long long peers = __ballot(act);              // lanes of this chunk with the same bin as this lane
            for (int bit = 0; bit < nbits; bit++) {
                const bool one = ((b >> bit) & 1) != 0;
                const unsigned long long mk = __ballot(one);
                peers &= one ? mk : ~mk;
            }
            const uint32_t start = myh[b];
            if (act) {
                const uint32_t r = start + (uint32_t)__builtin_popcountll(peers & lt);
                o[r] = pv[j];                                  // (the bin values themselves are not kept: k_ordv recomputes them for the debug fetch)
                if (((peers >> lane) >> 1) == 0ull) myh[b] = start + (uint32_t)__builtin_popcountll(peers);   // the group's last lane
            }
        }
    }
}

// Heaviest images first: the region stage's workgroups are dispatched in index order, so the batch is handed to it in descending
// order of expected cost (longest-processing-time-first keeps the tail of the launch short).  The expectation: the number of sortable
// pixels (correlation with the measured cost ~0.4) -- or, when the caller says the batch holds the same maps as the context's last one
// (lsd_set_cost_history: a site's maps change little from step to step), the shader clocks the stage spent on each image last time
// (`hist`: the previous launch's counter records, still in place when this kernel runs).  One workgroup; counting sort over 256 keys.
__global__ __launch_bounds__(256) void k_order(const int32_t* __restrict__ nb, const long long* __restrict__ hist, uint32_t* __restrict__ order, int n, int npx) {
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t start[256];
    __shared__ unsigned long long hmax;
    const int tid = threadIdx.x;
    cnt[tid] = 0;
    if (tid == 0) hmax = 1ull;
    __syncthreads();
    if (hist) {
        unsigned long long m = 0;
        for (int i = tid; i < n; i += 256) m = max(m, (unsigned long long)max(hist[(size_t)i * kStatWords + kStatTotalWord], 0ll));
        atomicMax(&hmax, m);
        __syncthreads();
    }
    const unsigned long long hm = hmax;
    auto key = [&](int i) {
        if (hist) return 255 - (int)(((unsigned long long)max(hist[(size_t)i * kStatWords + kStatTotalWord], 0ll) * 255ull) / hm);
        return 255 - (int)min(255ll, (long long)nb[i] * 2048 / (npx + 1));   // (nb is ~7 % of npx on occupancy maps)
    };
    for (int i = tid; i < n; i += 256) atomicAdd(&cnt[key(i)], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k2 = 0; k2 < 256; k2++) { start[k2] = run; run += cnt[k2]; }
    }
    __syncthreads();
    // stable within a key (ascending image index), so the order is deterministic: one thread per key walks the images
    uint32_t pos = start[tid];
    for (int i = 0; i < n; i++)
        if (key(i) == tid) order[pos++] = (uint32_t)i;
}

void launch_order(const Buffers& b, int n, int npx, const long long* hist, hipStream_t s) {
    hipLaunchKernelGGL(k_order, dim3(1), dim3(256), 0, s, b.nb, hist, b.order, n, npx);
}

// Debug fetch (LSD_DBG_ORDER_VAL): the bin value of every entry of one image's sorted list, recomputed with the sort's own bin_of()
// (the sort does not store them: the region stage never reads them, and the scattered 2-byte stores cost a 32-byte sector each).
__global__ __launch_bounds__(256) void k_ordv(const double* __restrict__ mag, const unsigned long long* __restrict__ maxbits, const uint32_t* __restrict__ ord,
                                              uint16_t* __restrict__ out, int count, int pseBin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double maxGrad = __longlong_as_double((long long)maxbits[0]);
    if (i < count && maxGrad > 0) out[i] = (uint16_t)bin_of(mag[ord[i]], 1.0 * pseBin / maxGrad, pseBin);
}
void launch_ordv(const double* mag, const unsigned long long* maxbits, const uint32_t* ord, uint16_t* out, int count, int pseBin, hipStream_t s) {
    if (count > 0) hipLaunchKernelGGL(k_ordv, dim3((count + 255) / 256), dim3(256), 0, s, mag, maxbits, ord, out, count, pseBin);
}

void launch_sort(const Geom& g, const Buffers& b, int n, hipStream_t s) {
    const size_t lds = ((size_t)SWAVES * g.pseBin + SWAVES) * sizeof(uint32_t);
    // 16 x 1024 bins x 4 B is just over the 64 KiB default; gfx950 has 160 KiB of LDS per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    // (the kept lists live in the region stage's seed arrays, which that stage fills at its start: they are free until then)
    hipLaunchKernelGGL(k_sort, dim3(n), dim3(SNT), lds, s, b.mag, b.maxbits, b.ord, b.nb, b.seedidx, b.seedpos, g.npx, g.pseBin);
}

}  // namespace lsdhip
