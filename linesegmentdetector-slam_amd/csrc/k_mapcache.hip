// k_mapcache.hip -- createMapCache on the device (SURVEY 8f "next" #1), gfx950.
//
// Replaces mylsd::createMapCache (LSD/myLSD.cpp:11-127): a FIFO breadth-first flood from every occupied
// cell (value 1) in raster order; a popped node (src, cur) offers each of its 4 neighbours (up, left, down,
// right -- :48, :67, :86, :105) that nobody has claimed yet the value dist(cur, src) * res -- the PARENT's
// distance (:49-54) -- provided that distance is <= cell_radius = floor(z_occ_max_dis / res); the neighbour
// joins the queue with the same src.  Cells never reached keep z_occ_max_dis, occupied cells 0.
//
// FIFO order == level order, and inside a level "queue order" == (rank of the parent in its level, direction).
// So the flood is run level by level by ONE 1024-thread workgroup per image:
//   claim  every expanding node i of the level does atomicMin(claim[nb], level<<40 | i*4+d) on its unclaimed
//          neighbours: the FIFO winner is the smallest key (earlier levels have smaller keys and stay);
//   emit   the winners are compacted IN KEY ORDER (block prefix sum over "won" flags, 1024 nodes per round)
//          into the next level's frontier -- that position is exactly the FIFO position -- and write the value.
// Everything is integer/ordering work plus one exactly rounded sqrt and multiply per cell: results are
// bit-identical to the reference.
#include "lsd_internal.h"

namespace lsdhip {

constexpr int MNT = 1024;
constexpr unsigned long long kUnclaimed = ~0ull;

__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();                       // wsum may still be read by the previous round
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int s = 0; s < MNT / 64; s++) { const int x = wsum[s]; if (s < wave) base += x; tot += x; }
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(MNT) void k_mapcache(const uint8_t* __restrict__ maps, double* __restrict__ out,
                                                  unsigned long long* __restrict__ claim, uint32_t* __restrict__ fr_a,
                                                  uint32_t* __restrict__ fr_b, int W, int H, double res, double zmax,
                                                  int cell_radius) {
    __shared__ int wsum[MNT / 64];
    const size_t img = blockIdx.x;
    const size_t npx = (size_t)W * H;
    const uint8_t* map = maps + img * npx;
    double* mc = out + img * npx;
    unsigned long long* cl = claim + img * npx;
    uint32_t* cur = fr_a + img * npx * 2;          // frontier entries: (cell, src) pairs
    uint32_t* nxt = fr_b + img * npx * 2;
    const int tid = threadIdx.x;

    // level 0: the occupied cells in raster order (:22-40)
    int n0 = 0;
    for (size_t base = 0; base < npx; base += MNT) {
        const size_t p = base + tid;
        const bool occ = p < npx && map[p] == 1;
        int tot;
        const int pos = block_excl_scan(occ ? 1 : 0, wsum, tot);
        if (p < npx) {
            mc[p] = occ ? 0.0 : zmax;
            cl[p] = occ ? 0ull : kUnclaimed;
            if (occ) { cur[2 * (size_t)(n0 + pos)] = (uint32_t)p; cur[2 * (size_t)(n0 + pos) + 1] = (uint32_t)p; }
        }
        n0 += tot;
    }
    __threadfence_block();
    __syncthreads();

    int n = n0;
    for (unsigned long long level = 1; n > 0; level++) {
        // ---- claim ----
        for (int i = tid; i < n; i += MNT) {
            const uint32_t c = cur[2 * (size_t)i], s = cur[2 * (size_t)i + 1];
            const int ci = (int)(c / (uint32_t)W), cj = (int)(c % (uint32_t)W);
            const int si = (int)(s / (uint32_t)W), sj = (int)(s % (uint32_t)W);
            const double di = abs(ci - si), dj = abs(cj - sj);                   // :49-50
            const double distance = sqrt(di * di + dj * dj);                    // :51
            if (distance <= cell_radius) {                                      // :53
                const unsigned long long key = (level << 40) | ((unsigned long long)i << 2);
                // a neighbour is open if nobody claimed it in an EARLIER level (claims of this level compete by key)
                const unsigned long long open = level << 40;
                if (ci >= 1 && cl[c - W] >= open) atomicMin(&cl[c - W], key | 0ull);                    // up    :48
                if (cj >= 1 && cl[c - 1] >= open) atomicMin(&cl[c - 1], key | 1ull);                    // left  :67
                if (ci < H - 1 && cl[c + W] >= open) atomicMin(&cl[c + W], key | 2ull);                 // down  :86
                if (cj < W - 1 && cl[c + 1] >= open) atomicMin(&cl[c + 1], key | 3ull);                 // right :105
            }
        }
        __threadfence_block();
        __syncthreads();
        // ---- emit the winners in key order ----
        int nn = 0;
        for (int base = 0; base < n; base += MNT) {
            const int i = base + tid;
            uint32_t c = 0, s = 0;
            int won = 0;                                   // bit d set: direction d won its neighbour
            double val = 0;
            if (i < n) {
                c = cur[2 * (size_t)i]; s = cur[2 * (size_t)i + 1];
                const int ci = (int)(c / (uint32_t)W), cj = (int)(c % (uint32_t)W);
                const int si = (int)(s / (uint32_t)W), sj = (int)(s % (uint32_t)W);
                const double di = abs(ci - si), dj = abs(cj - sj);
                const double distance = sqrt(di * di + dj * dj);
                if (distance <= cell_radius) {
                    val = distance * res;                                        // :54
                    const unsigned long long key = (level << 40) | ((unsigned long long)i << 2);
                    if (ci >= 1 && cl[c - W] == (key | 0ull)) won |= 1;
                    if (cj >= 1 && cl[c - 1] == (key | 1ull)) won |= 2;
                    if (ci < H - 1 && cl[c + W] == (key | 2ull)) won |= 4;
                    if (cj < W - 1 && cl[c + 1] == (key | 3ull)) won |= 8;
                }
            }
            int tot;
            int pos = nn + block_excl_scan(__builtin_popcount(won), wsum, tot);
            if (won & 1) { const uint32_t nb = c - W; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
            if (won & 2) { const uint32_t nb = c - 1; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
            if (won & 4) { const uint32_t nb = c + W; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
            if (won & 8) { const uint32_t nb = c + 1; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
            nn += tot;
        }
        __threadfence_block();
        __syncthreads();
        uint32_t* t = cur; cur = nxt; nxt = t;
        n = nn;
    }
}

void launch_mapcache(const uint8_t* maps, double* out, unsigned long long* claim, uint32_t* fr_a, uint32_t* fr_b, int n,
                     int W, int H, double res, double zmax, int cell_radius, hipStream_t s) {
    hipLaunchKernelGGL(k_mapcache, dim3(n), dim3(MNT), 0, s, maps, out, claim, fr_a, fr_b, W, H, res, zmax, cell_radius);
}

}  // namespace lsdhip
