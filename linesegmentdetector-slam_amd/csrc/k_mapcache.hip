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

// The level loop of one workgroup on one map, from frontier `cur` (n nodes) at `level` until the flood dies out.
// claim words are decided by atomics performed in L2: read them back from there, not from a line the CU's L1 may still hold
__device__ __forceinline__ unsigned long long ld_claim(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void mc_run_levels(unsigned long long* cl, double* mc, uint32_t* cur, uint32_t* nxt, int n,
                                              unsigned long long level0, int W, int H, double res, int cell_radius, int* wsum) {
    const int tid = threadIdx.x;
    for (unsigned long long level = level0; n > 0; level++) {
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
                    if (ci >= 1 && ld_claim(&cl[c - W]) == (key | 0ull)) won |= 1;
                    if (cj >= 1 && ld_claim(&cl[c - 1]) == (key | 1ull)) won |= 2;
                    if (ci < H - 1 && ld_claim(&cl[c + W]) == (key | 2ull)) won |= 4;
                    if (cj < W - 1 && ld_claim(&cl[c + 1]) == (key | 3ull)) won |= 8;
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

    mc_run_levels(cl, mc, cur, nxt, n0, 1ull, W, H, res, cell_radius, wsum);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same flood spread over G workgroups per map (small batches: one workgroup per map is a chain of ~8 k dependent global
// round trips, 20 ms at 2048^2).  The level barrier becomes a kernel boundary, so nothing spins: per level
//   claim  (frontier chunk g of G)   atomicMin of the FIFO key, global parent ranks
//   count  winners per chunk
//   emit   winners in key order at (sum of the counts of the chunks before) + block prefix, values written
// and the host enqueues the three for every level the flood can have (a node expands only within cell_radius of its source,
// so the depth is bounded by sqrt(2) * cell_radius + 2); launches on an empty frontier return at once.
// ctl[img] = { n of frontier A, n of frontier B }, cnt[img][G] = per-chunk counts.
// ---------------------------------------------------------------------------------------------------------------------
struct McArgs {
    const uint8_t* maps; double* out; unsigned long long* claim; uint32_t* fr_a; uint32_t* fr_b; int* ctl; int* cnt;
    int W, H, G; double res, zmax; int cell_radius;
};

__global__ __launch_bounds__(MNT) void k_mc_init_count(McArgs a) {
    __shared__ int wsum[MNT / 64];
    const size_t img = blockIdx.y, npx = (size_t)a.W * a.H;
    const int g = blockIdx.x;
    const size_t chunk = (npx + a.G - 1) / a.G, p0 = g * chunk, p1 = p0 + chunk < npx ? p0 + chunk : npx;
    const uint8_t* map = a.maps + img * npx;
    int total = 0;
    for (size_t base = p0; base < p1; base += MNT) {
        const size_t p = base + threadIdx.x;
        const bool occ = p < p1 && map[p] == 1;
        if (p < p1) {
            a.out[img * npx + p] = occ ? 0.0 : a.zmax;
            a.claim[img * npx + p] = occ ? 0ull : kUnclaimed;
        }
        int tot;
        (void)block_excl_scan(occ ? 1 : 0, wsum, tot);
        total += tot;
    }
    if (threadIdx.x == 0) a.cnt[img * a.G + g] = total;
}

__global__ __launch_bounds__(MNT) void k_mc_init_emit(McArgs a) {
    __shared__ int wsum[MNT / 64];
    const size_t img = blockIdx.y, npx = (size_t)a.W * a.H;
    const int g = blockIdx.x;
    const size_t chunk = (npx + a.G - 1) / a.G, p0 = g * chunk, p1 = p0 + chunk < npx ? p0 + chunk : npx;
    const uint8_t* map = a.maps + img * npx;
    uint32_t* cur = a.fr_a + img * npx * 2;
    int pos0 = 0, all = 0;
    for (int j = 0; j < a.G; j++) { const int v = a.cnt[img * a.G + j]; if (j < g) pos0 += v; all += v; }
    for (size_t base = p0; base < p1; base += MNT) {
        const size_t p = base + threadIdx.x;
        const bool occ = p < p1 && map[p] == 1;
        int tot;
        const int pos = pos0 + block_excl_scan(occ ? 1 : 0, wsum, tot);
        if (occ) { cur[2 * (size_t)pos] = (uint32_t)p; cur[2 * (size_t)pos + 1] = (uint32_t)p; }
        pos0 += tot;
    }
    if (g == 0 && threadIdx.x == 0) { a.ctl[img * 2] = all; a.ctl[img * 2 + 1] = 0; }
}

// phase 0: claim, 1: count winners, 2: emit.  parity = level & 1 selects which frontier is current (level 1 reads A).
template <int PHASE>
__global__ __launch_bounds__(MNT) void k_mc_level(McArgs a, unsigned long long level) {
    __shared__ int wsum[MNT / 64];
    const size_t img = blockIdx.y, npx = (size_t)a.W * a.H;
    const int g = blockIdx.x, W = a.W, H = a.H;
    const bool odd = (level & 1ull) != 0ull;
    const int n = a.ctl[img * 2 + (odd ? 0 : 1)];
    if (n == 0) {
        if (PHASE == 2 && g == 0 && threadIdx.x == 0) a.ctl[img * 2 + (odd ? 1 : 0)] = 0;
        return;
    }
    const uint32_t* cur = (odd ? a.fr_a : a.fr_b) + img * npx * 2;
    uint32_t* nxt = (odd ? a.fr_b : a.fr_a) + img * npx * 2;
    unsigned long long* cl = a.claim + img * npx;
    double* mc = a.out + img * npx;
    const int chunk = (n + a.G - 1) / a.G, i0 = g * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    int pos0 = 0, all = 0;
    if (PHASE == 2) for (int j = 0; j < a.G; j++) { const int v = a.cnt[img * a.G + j]; if (j < g) pos0 += v; all += v; }
    int total = 0;
    for (int base = i0; base < i1 || (PHASE != 0 && base == i0); base += MNT) {
        const int i = base + threadIdx.x;
        uint32_t c = 0, s = 0;
        int won = 0;
        double val = 0;
        if (i < i1) {
            c = cur[2 * (size_t)i]; s = cur[2 * (size_t)i + 1];
            const int ci = (int)(c / (uint32_t)W), cj = (int)(c % (uint32_t)W);
            const int si = (int)(s / (uint32_t)W), sj = (int)(s % (uint32_t)W);
            const double di = abs(ci - si), dj = abs(cj - sj);                   // :49-50
            const double distance = sqrt(di * di + dj * dj);                    // :51
            if (distance <= a.cell_radius) {                                    // :53
                const unsigned long long key = (level << 40) | ((unsigned long long)i << 2);
                if (PHASE == 0) {
                    const unsigned long long open = level << 40;                // claims of earlier levels stand
                    if (ci >= 1 && cl[c - W] >= open) atomicMin(&cl[c - W], key | 0ull);        // up    :48
                    if (cj >= 1 && cl[c - 1] >= open) atomicMin(&cl[c - 1], key | 1ull);        // left  :67
                    if (ci < H - 1 && cl[c + W] >= open) atomicMin(&cl[c + W], key | 2ull);     // down  :86
                    if (cj < W - 1 && cl[c + 1] >= open) atomicMin(&cl[c + 1], key | 3ull);     // right :105
                } else {
                    val = distance * a.res;                                     // :54
                    if (ci >= 1 && cl[c - W] == (key | 0ull)) won |= 1;
                    if (cj >= 1 && cl[c - 1] == (key | 1ull)) won |= 2;
                    if (ci < H - 1 && cl[c + W] == (key | 2ull)) won |= 4;
                    if (cj < W - 1 && cl[c + 1] == (key | 3ull)) won |= 8;
                }
            }
        }
        if (PHASE != 0) {
            int tot;
            int pos = pos0 + block_excl_scan(__builtin_popcount(won), wsum, tot);
            if (PHASE == 2) {
                if (won & 1) { const uint32_t nb = c - W; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
                if (won & 2) { const uint32_t nb = c - 1; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
                if (won & 4) { const uint32_t nb = c + W; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
                if (won & 8) { const uint32_t nb = c + 1; mc[nb] = val; nxt[2 * (size_t)pos] = nb; nxt[2 * (size_t)pos + 1] = s; pos++; }
            }
            pos0 += tot; total += tot;
        }
    }
    if (PHASE == 1 && threadIdx.x == 0) a.cnt[img * a.G + g] = total;
    if (PHASE == 2 && g == 0 && threadIdx.x == 0) a.ctl[img * 2 + (odd ? 1 : 0)] = all;      // size of the next frontier
}

// Whatever the planned number of levels left over (the depth of the flood has no useful static bound: trees of different
// sources block each other into detours): one workgroup per map finishes it.  Normally the frontier is already empty.
__global__ __launch_bounds__(MNT) void k_mc_finish(McArgs a, unsigned long long level) {
    __shared__ int wsum[MNT / 64];
    const size_t img = blockIdx.x, npx = (size_t)a.W * a.H;
    const bool odd = (level & 1ull) != 0ull;
    const int n = a.ctl[img * 2 + (odd ? 0 : 1)];
    if (n == 0) return;
    mc_run_levels(a.claim + img * npx, a.out + img * npx, (odd ? a.fr_a : a.fr_b) + img * npx * 2, (odd ? a.fr_b : a.fr_a) + img * npx * 2,
                  n, level, a.W, a.H, a.res, a.cell_radius, wsum);
}

void launch_mapcache_spread(const uint8_t* maps, double* out, unsigned long long* claim, uint32_t* fr_a, uint32_t* fr_b, int* ctl,
                            int* cnt, int n, int G, int W, int H, double res, double zmax, int cell_radius, hipStream_t s) {
    McArgs a{maps, out, claim, fr_a, fr_b, ctl, cnt, W, H, G, res, zmax, cell_radius};
    const dim3 grid(G, n), blk(MNT);
    hipLaunchKernelGGL(k_mc_init_count, grid, blk, 0, s, a);
    hipLaunchKernelGGL(k_mc_init_emit, grid, blk, 0, s, a);
    const int levels = (int)(1.5 * (cell_radius > 0 ? cell_radius : 0)) + 3;          // covers the bulk; the rest below
    for (int level = 1; level <= levels; level++) {
        hipLaunchKernelGGL(k_mc_level<0>, grid, blk, 0, s, a, (unsigned long long)level);
        hipLaunchKernelGGL(k_mc_level<1>, grid, blk, 0, s, a, (unsigned long long)level);
        hipLaunchKernelGGL(k_mc_level<2>, grid, blk, 0, s, a, (unsigned long long)level);
    }
    hipLaunchKernelGGL(k_mc_finish, dim3(n), blk, 0, s, a, (unsigned long long)(levels + 1));
}

void launch_mapcache(const uint8_t* maps, double* out, unsigned long long* claim, uint32_t* fr_a, uint32_t* fr_b, int n,
                     int W, int H, double res, double zmax, int cell_radius, hipStream_t s) {
    hipLaunchKernelGGL(k_mapcache, dim3(n), dim3(MNT), 0, s, maps, out, claim, fr_a, fr_b, W, H, res, zmax, cell_radius);
}

}  // namespace lsdhip
