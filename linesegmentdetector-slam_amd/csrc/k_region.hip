// k_region.hip -- K4: the region stage (seed loop) of LSD, one workgroup of NW wavefronts per image (gfx950).
//
// Replaces the seed loop of myLineSegmentDetector (LSD/myLSD.cpp:219-272) and its callees
// RegionGrower (:491-590), CenterGetter/OrientationGetter/RectangleConverter (:592-734),
// RegionRadiusReducer (:736-802), Refiner (:804-880), LogGammaCalculator (:882-924),
// RectangleNFACalculator (:926-1059) and RectangleImprover (:1061-1158).
//
// The reference is strictly sequential: seeds are visited in sorted order and each one sees the
// usedMap left by all earlier ones.  What actually couples two seeds is small, though: a seed's whole
// evaluation (grow, rectangle, refine, NFA) reads usedMap only through "is this pixel banned
// (== 1)?", and only ACCEPTED lines ever set a pixel to 1 (rejected regions set 2, which stays
// growable; small/failed regions set nothing).  So the wavefronts of a workgroup evaluate seeds
// SPECULATIVELY ahead of a commit cursor and commit in seed order:
//   * a wave takes a block of 8 consecutive seeds, notes the accept epoch, and grows the 8 regions together
//     in group mode (grow8(): 8 lanes per region, one frontier entry per step, private lists / stamps in HBM);
//   * it evaluates the grown regions one by one with all 64 lanes (rectangle, Refiner incl. its regrow with
//     the wave-wide batched grow(), NFA); results that mark nothing are published in an LDS ring, results
//     that mark usedMap are stashed (record + pixel list) and committed when the cursor reaches them;
//   * at its turn a result is valid if its seed is still unused and no pixel it EXAMINED (a member of one of
//     its grown lists or one of their 8 neighbours) was banned since its snapshot -- accepted pixels carry
//     their line's epoch above the usedMap bits, the boxes of recently accepted lines are a first filter;
//     an invalid result is evaluated again at the cursor, where everything earlier is committed;
//   * commit = mark usedMap (1 + epoch, or 2), append the rectangle, advance the cursor.
// The committed sequence of decisions is therefore exactly the sequential one (DESIGN.md section 4 has the
// measurements behind every choice, and the variants that were tried and dropped).
// Inside a wavefront the lanes cooperate where the order of evaluation can be kept:
//   * region growing: 8 regions x 8 neighbours per step (grow8) or 8 frontier pixels x 8 neighbours of ONE
//     region per batch (grow); angle sums in reference order, candidates classified against an estimate of
//     the region angle with a rigorous margin, exact angle only when too close to call;
//   * rectangle moments: products per lane, SERIAL accumulation in list order (bit-exact sums);
//   * NFA pixel count: the rectangle's columns are flattened with a wave prefix sum and counted
//     with ballot/popcount;
//   * usedMap marking: only the region's pixels are visited (the reference scans the whole image).
#include "lsd_internal.h"
#include "devmath.h"

// The file is compiled twice (Makefile): LSD_REGION_NW = 4 (two images per CU: the batch path, all 512 workgroups of the
// bench batch resident at once) and LSD_REGION_NW = 8 (one image per CU, 8 speculative wavefronts per image: ~1.5x lower
// latency per image, chosen when the batch leaves CUs idle anyway).  Everything lives in a per-variant namespace.
#ifndef LSD_REGION_NW
#define LSD_REGION_NW 4
#endif
#if LSD_REGION_NW == 8
#define RVAR w8
#else
#define RVAR w4
#endif

namespace lsdhip {
namespace RVAR {

#ifndef LSD_REGION_NW
#define LSD_REGION_NW 4
#endif
#ifndef LSD_REGION_WAVES_PER_SIMD
#define LSD_REGION_WAVES_PER_SIMD 2
#endif
constexpr int NW = LSD_REGION_NW;        // wavefronts (concurrent speculative seeds) per image
constexpr int LCAP = 1024;   // region-list entries kept in LDS per wave; the rest spills to HBM
constexpr int NSLOT = 4;     // tile-cache slots per wave (2 x 2 tiles)
constexpr int RING = 128;    // remembered bounding boxes of recently accepted lines
#ifndef LSD_REGION_NB
#define LSD_REGION_NB 1
#endif
#if LSD_REGION_NB != 1
#error "more than one block buffer per wave is not supported (the list-withdrawal at refetch assumes one; 2-4 buffers were measured slower and are untested since)"
#endif
constexpr int NB = LSD_REGION_NB;        // blocks of 8 seeds a wave may have in flight (evaluated, waiting for their turn to commit)

struct Rec {  // structRec, myLSD.h:80-93 (+ pk = number of halvings of p, indexes the host log tables)
    double x1, y1, x2, y2, wid, cX, cY, deg, dx, dy, p, prec;
    int pk;
};

struct RCtx {
    int w, h, lane;
    const double* mag;
    const double* deg;
    uint32_t* state;     // usedMap values (shared by the workgroup)
    uint32_t* stamp;     // this wave's curMap stamps
    uint32_t* spill;
    uint32_t* gcopy;
    float2* meta;        // HBM [mcap]: (angle estimate, slack) of the last full test of a list entry, see grow()
    int mcap;
    const double* sn;
    const double* cs;
    uint32_t* lst;       // LDS [LCAP]
    uint16_t* wl0;       // LDS [LCAP] sweep worklists
    uint16_t* wl1;
    uint32_t* t_st;      // LDS [NSLOT][64] tile cache: state words
    double* t_deg;       // LDS [NSLOT][64]
    double* t_sn;        // LDS [NSLOT][64]
    double* t_cs;        // LDS [NSLOT][64]
    int* t_tag;          // LDS [NSLOT]
    int tilesX;
    bool dirty;          // stamps stored to HBM since the last fence
    int* s_incl;         // LDS [64]
    int* s_lo;           // LDS [64]
    int* s_x;            // LDS [64]
    uint32_t cur_id;     // stamp of the current grow (id_base + running number)
    int gnum;            // size of the last grow (grow order)
    bool has_copy;       // gcopy holds the grow-order list (RegionRadiusReducer reordered lst)
    double logNT;
    const double* lgamma;
    const double* ptab;
    unsigned long long* stat;   // LDS [32] per-wave counters (see ST_* below); kept out of registers
};

enum { ST_GROW = 0, ST_GROWN, ST_NFA, ST_RRR, ST_RRRPASS, ST_SENT, ST_OOB, ST_SPILL, ST_TOTAL, ST_TGROW, ST_TRECT, ST_TNFA,
       ST_TMARK, ST_MAXREG, ST_NFAPX, ST_SEEDS, ST_EXACT, ST_TILEFETCH, ST_BATCHES, ST_TTILES, ST_REDO, ST_DISCARD,
       ST_WAIT, ST_RESWEEP, ST_TGROUP, ST_TEVAL, ST_NHANDED, ST_PXHANDED, ST_TIDLE, ST_TSELECT, ST_THANDED, ST_SKIPPED, ST_COUNT };
#define STAT(i, v) do { if (c.lane == 0) c.stat[i] += (unsigned long long)(v); } while (0)

__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

__device__ __forceinline__ double rl(double v, int l) {  // broadcast lane l (l wave-uniform)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t pack_xy(int x, int y) { return ((uint32_t)y << 16) | (uint32_t)x; }
__device__ __forceinline__ uint32_t lget(const RCtx& c, int i) { return i < LCAP ? c.lst[i] : c.spill[i - LCAP]; }
__device__ __forceinline__ void lset(const RCtx& c, int i, uint32_t v) {
    if (i < LCAP) c.lst[i] = v; else c.spill[i - LCAP] = v;
}
__device__ __forceinline__ double angle_diff(double a, double b) {  // myLSD.cpp:540-542 / :1009-1011
    double d = fabs(a - b);
    if (d > kPi * 3 / 2.0) d = fabs(d - 2.0 * kPi);
    return d;
}

// ---------------------------------------------------------------------------------------------
// LDS tile cache: 8x8-pixel tiles of (state, deg, sin deg, cos deg), direct-mapped over a
// 16x16-pixel window (slot = tx&1 | (ty&1)<<1).  RegionGrower reads its 3x3 neighbourhoods from
// here, so a batch costs LDS latency instead of two dependent HBM round trips plus a store fence.
// Accepted pixels are stamped in the LDS copy AND in HBM (write-through, never waited for); the
// cache is dropped whenever usedMap marks change (mark_region).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_slot(int tx, int ty) { return (tx & 1) | ((ty & 1) << 1); }

// Makes the tiles of every lane with need==true resident.  Returns false when two needed tiles map
// to the same slot (the caller retries with a smaller batch; a single 3x3 neighbourhood never conflicts).
__device__ __forceinline__ bool ensure_tiles(RCtx& c, bool need, int px, int py) {
    const int lane = c.lane, w = c.w, h = c.h;
    const int tx = px >> 3, ty = py >> 3;
    const int tile = need ? ty * c.tilesX + tx : -1;
    const int slot = tile_slot(tx, ty);
    unsigned long long todo = __ballot(need && c.t_tag[slot] != tile);
    if (!todo) return true;
    // conflict check over all needed tiles (resident ones included)
    {
        unsigned long long chk = __ballot(need);
        while (chk) {
            const int l = __builtin_ctzll(chk);
            const int T = __builtin_amdgcn_readlane(tile, l), S = __builtin_amdgcn_readlane(slot, l);
            if (__ballot(need && slot == S && tile != T)) return false;
            chk &= ~__ballot(tile == T);
        }
    }
    const long long tt0 = (long long)__builtin_amdgcn_s_memtime();
    if (c.dirty) { wg_fence(); c.dirty = false; }     // earlier stamps must have landed before a tile is (re)read
    STAT(ST_TILEFETCH, 1);
    while (todo) {
        // up to 4 missing tiles per round, all loads in flight together
        int T[4], S[4];
        int nt = 0;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            T[j] = -1; S[j] = 0;
            if (todo) {
                const int l = __builtin_ctzll(todo);
                T[j] = __builtin_amdgcn_readlane(tile, l);
                S[j] = __builtin_amdgcn_readlane(slot, l);
                todo &= ~__ballot(tile == T[j]);
                nt++;
            }
        }
        double vd[4], vs[4], vc[4];
        uint32_t vw[4];
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            vd[j] = 0; vs[j] = 0; vc[j] = 1; vw[j] = 1u;          // outside the image: banned
            if (j < nt) {
                const int ttx = T[j] % c.tilesX, tty = T[j] / c.tilesX;
                const int x = ttx * 8 + (lane & 7), y = tty * 8 + (lane >> 3);
                if (x < w && y < h) {
                    const size_t q = (size_t)y * w + x;
                    vw[j] = (c.stamp[q] << 2) | (c.state[q] & 3u);
                    vd[j] = c.deg[q];
                    vs[j] = c.sn[q];                              // garbage where usedMap == 1: never read
                    vc[j] = c.cs[q];
                }
            }
        }
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < nt) {
                c.t_st[S[j] * 64 + lane] = vw[j];
                c.t_deg[S[j] * 64 + lane] = vd[j];
                c.t_sn[S[j] * 64 + lane] = vs[j];
                c.t_cs[S[j] * 64 + lane] = vc[j];
                if (lane == 0) c.t_tag[S[j]] = T[j];
            }
        }
    }
    wg_fence();
    STAT(ST_TTILES, (long long)__builtin_amdgcn_s_memtime() - tt0);
    return true;
}

__device__ __forceinline__ void invalidate_tiles(RCtx& c) {
    if (c.lane < NSLOT) c.t_tag[c.lane] = -1;
    wg_fence();
}

// ---------------------------------------------------------------------------------------------
// RegionGrower, myLSD.cpp:491-590.  Leaves the region in c.lst (grow order); returns the size and
// the angle sums (the region angle atan2(sinS, cosS) is evaluated by the caller only when needed).
//
// The reference tests every candidate against regDeg = atan2(sinDeg, cosDeg) recomputed after each
// accepted pixel (:545-547), in list order / row-major neighbour order.  Here a batch of 8 frontier
// pixels x 8 neighbours is classified at once against an ESTIMATE of regDeg with a rigorous margin:
//   margin = (error of the fp32 estimate) + (largest drift regDeg can undergo while the up-to-m
//            candidates of this batch are accepted: each accepted unit vector lies within tol of the
//            current sum vector of norm L, so it turns it by at most sin(tol)/L)
//   * |dif| < tol - margin : passes whatever happens earlier in the batch  -> accepted in bulk
//   * |dif| > tol + margin : fails whatever happens                        -> ignored
//   * otherwise            : resolved one by one in reference order against a fresh estimate, and
//                            against the correctly rounded angle when still too close to call.
// The sums are accumulated in the reference order in every case, so they are bit-identical, and every
// accept/reject decision is the one the exact angle would give.
// Sweeps after the first revisit only entries that still had a non-member, non-banned neighbour
// (membership and bans only grow during one call, so the others cannot accept anything).
// ---------------------------------------------------------------------------------------------

// an upper bound of 1 / v for v >= 0.9 (v_rcp_f32 is good to 1 ulp; the margins it feeds are themselves upper bounds)
__device__ __forceinline__ float inv_ub(float v) { return __builtin_amdgcn_rcpf(v) * 1.000001f; }

constexpr double kAngEps = 8e-6;   // >= error of (double)atan2f((float)s,(float)c) incl. input rounding

__device__ __forceinline__ void grow(RCtx& c, int sx, int sy, double regDeg0, double tol, int& out_num,
                                     double& out_sin, double& out_cos) {
    const int lane = c.lane, w = c.w, h = c.h;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const uint32_t id = ++c.cur_id;                          // fresh curMap (:519)
    ensure_tiles(c, lane == 0, sx, sy);
    double sinS, cosS;
    {
        const int slot = tile_slot(sx >> 3, sy >> 3), ti = ((sy & 7) << 3) | (sx & 7);
        sinS = c.t_sn[slot * 64 + ti];                       // sin/cos(regDeg0): regDeg0 is degMap[seed] at both call sites (:225, :857)
        cosS = c.t_cs[slot * 64 + ti];
        if (lane == 0) {
            lset(c, 0, pack_xy(sx, sy));
            c.t_st[slot * 64 + ti] = (id << 2) | (c.t_st[slot * 64 + ti] & 3u);   // :520
            c.stamp[(size_t)sy * w + sx] = id;
        }
        c.dirty = true;
    }
    // R estimates the reference's regDeg; |R - regDeg| <= eps at any time.  eps is 0 while R is the exact
    // value, kAngEps right after an fp32 refresh, and grows by the worst-case turn of every pixel accepted
    // since (so the estimate is refreshed only when some candidate is too close to call).
    double R = regDeg0;
    double eps = 0.0;
    bool fresh = true;                                       // R was computed from the current sums
    const bool tol_small = tol < 1.5;                        // accepted vectors then never shorten the sum
    const double turn = tol < 1.1 ? tol : 1.1;               // >= sin(tol) resp. the asin(1/L)*L bound
    // lower bound of the norm of the angle-sum vector: every accepted unit vector lies within tol (< pi/2) of
    // it, so it lengthens it by at least cos(tol)
    const float cos_lb = tol_small ? cosf((float)tol) * 0.999f - 1e-6f : 0.0f;
    float Llb = 0.999f;
    int n = 1;
    const int e = lane >> 3, k = lane & 7;
    const int kk = k + (k >= 4);                             // 3x3 neighbourhood, row-major, centre skipped (:533-534)
    const int ox = kk % 3 - 1, oy = kk / 3 - 1;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    uint16_t* wl_cur = c.wl0;
    uint16_t* wl_nxt = c.wl1;
    int wl_cnt = 0;                                          // entries of wl_cur (sweep >= 2)
    bool filter = true;                                      // false once the list outgrew the worklists
    // Re-sweeps: an entry whose remaining candidates all failed by more than the region angle has moved since cannot
    // accept anything now either (membership and bans only grow); it is carried over to the next worklist without
    // touching its neighbourhood.  meta[entry] = (angle estimate its candidates were compared with, smallest
    // "distance - tol" among the candidates left), checked 64 entries at a time.
    unsigned long long flt_need = 0;                         // chunk [flt_base, flt_base + 64) of wl_cur: entries to test in full
    int flt_base = 0;
    bool flt_valid = false;
    int sweep = 1, ex;
    do {                                                     // :525 sweeps to fixpoint (Q7)
        ex = n;
        int nxt_cnt = 0;
        const int n_start = (sweep == 1) ? 0 : n;            // entries below n_start come from the worklist
        int wi = 0;                                          // worklist cursor
        int i = (sweep == 1 || !filter) ? 0 : n_start;       // contiguous cursor
        bool in_wl = (sweep > 1) && filter;
        while (true) {
            // ---- pick up to 8 entries ----
            int cnt, eidx;
            if (in_wl) {
                if (wi >= wl_cnt) { in_wl = false; continue; }
                cnt = min(8, wl_cnt - wi);
                if (tol_small && filter) {
                    if (!flt_valid || wi >= flt_base + 64) {
                        if (!fresh) { R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true; }
                        flt_base = wi;
                        bool nd = false;
                        if (wi + lane < wl_cnt) {
                            const int ei = (int)wl_cur[wi + lane];
                            nd = true;
                            if (ei < c.mcap) {
                                const float2 mt = c.meta[ei];
                                float dr = fabsf((float)R - mt.x);
                                if (dr > 3.14159265f) dr = 6.28318531f - dr;
                                nd = !(dr + (float)eps + 1e-5f < mt.y);
                            }
                        }
                        flt_need = __ballot(nd);
                        flt_valid = true;
                    }
                    const int off = wi - flt_base;
                    const int nval = min(64, wl_cnt - flt_base) - off;         // entries of the chunk from wi on
                    const unsigned long long rest = flt_need >> off;          // bit 0 = entry wi
                    const int nskip = rest ? min(__builtin_ctzll(rest), nval) : nval;
                    if (nskip > 0) {                         // a run of entries that cannot accept anything: carry them over
                        if (nxt_cnt + nskip > LCAP) filter = false;
                        else {
                            if (lane < nskip) wl_nxt[nxt_cnt + lane] = wl_cur[wi + lane];
                            nxt_cnt += nskip;
                        }
                        wi += nskip;
                        STAT(ST_SKIPPED, nskip);
                        continue;
                    }
                    // the run of consecutive entries to test (no skipped entry in between: its check would be stale after an accept)
                    const unsigned long long inv = ~rest;
                    cnt = min(cnt, inv ? __builtin_ctzll(inv) : 64);
                }
                eidx = e < cnt ? (int)wl_cur[wi + e] : 0;
            } else {
                if (i >= n) break;                           // n is live (:529)
                cnt = min(8, n - i);
                eidx = i + e;
            }
            bool valid = e < cnt;
            const uint32_t pk = valid ? lget(c, eidx) : 0u;
            const int nx = (int)(pk & 0xffffu) + ox, ny = (int)(pk >> 16) + oy;
            bool inb = valid && nx >= 0 && ny >= 0 && nx < w && ny < h;            // :536
            const int slot = tile_slot(nx >> 3, ny >> 3), ti = ((ny & 7) << 3) | (nx & 7);
            const int cell = slot * 64 + ti;
            if (__ballot(inb && c.t_tag[slot] != (ny >> 3) * c.tilesX + (nx >> 3))) {
                if (!ensure_tiles(c, inb, nx, ny)) {         // slot conflict: one entry at a time
                    cnt = 1;
                    valid = e < cnt;
                    inb = inb && valid;
                    ensure_tiles(c, inb, nx, ny);
                }
            }
            const int q = ny * w + nx;
            // all per-pixel reads of the batch are issued together (cell is in range even for !inb lanes)
            const uint32_t word_r = c.t_st[cell];
            const double d = c.t_deg[cell], sd = c.t_sn[cell], cd = c.t_cs[cell];
            const uint32_t word = inb ? word_r : 1u;
            const bool cand = inb && (word >> 2) != id && (word & 3u) != 1u;       // :537 (2 is growable, Q5)
            const unsigned long long candm = __ballot(cand);
            unsigned long long acc = 0;                      // accepted lanes (one per accepted pixel)
            unsigned long long gone = 0;                     // every lane whose pixel became a member in this batch
            if (candm) {
                // first occurrence of every candidate pixel: a lane is a repeat iff an EARLIER entry of the batch
                // has the pixel in its 3x3 neighbourhood (that entry's lane for it comes first in reference order)
                bool winner = cand;
                {
                    const int ex0 = (int)(pk & 0xffffu), ey0 = (int)(pk >> 16);
                    for (int e2 = 0; e2 + 1 < cnt; e2++) {
                        const int px2 = __builtin_amdgcn_readlane(ex0, e2 * 8), py2 = __builtin_amdgcn_readlane(ey0, e2 * 8);
                        if (e > e2 && abs(nx - px2) <= 1 && abs(ny - py2) <= 1) winner = false;
                    }
                }
                // margin: error bound of R + worst-case drift while this batch's m candidates are accepted
                const int m = __builtin_popcountll(__ballot(winner));
                if (!tol_small) {                            // (rare: Refiner asked for a huge tolerance) no cheap norm bound
                    if (!fresh) { R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true; }
                    const float sf = (float)sinS, cf = (float)cosS;
                    Llb = sqrtf(sf * sf + cf * cf) * 0.999f - (float)m;
                }
                double margin = Llb >= (tol_small ? 0.9f : 3.0f)
                                    ? eps + (double)(1.002f * (float)m * (float)turn * inv_ub(Llb)) + 1e-7
                                    : 1e30;
                double raw = fabs(R - d);
                double dif = raw > kPi * 3 / 2.0 ? fabs(raw - 2.0 * kPi) : raw;          // :540-542
                bool cut = fabs(R) > kPi - margin || fabs(raw - kPi * 3 / 2.0) <= margin;
                bool pc = !cut && dif < tol - margin;
                bool amb = cut || (!pc && !(dif > tol + margin));
                if (!fresh && tol_small && __ballot(cand && amb)) {
                    // some candidate is too close to call with the drifted estimate: refresh it once and reclassify
                    R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true;
                    margin = Llb >= 0.9f ? eps + (double)(1.002f * (float)m * (float)turn * inv_ub(Llb)) + 1e-7 : 1e30;
                    raw = fabs(R - d);
                    dif = raw > kPi * 3 / 2.0 ? fabs(raw - 2.0 * kPi) : raw;
                    cut = fabs(R) > kPi - margin || fabs(raw - kPi * 3 / 2.0) <= margin;
                    pc = !cut && dif < tol - margin;
                    amb = cut || (!pc && !(dif > tol + margin));
                }
                const double Rcls = R;                       // the angle dif was taken against
                const unsigned long long P = __ballot(winner && pc);              // accepted whatever the order
                const unsigned long long A = __ballot(cand && amb);               // every occurrence, resolved in order
                unsigned long long todo = P | A;
                STAT(ST_BATCHES, 1);
                if (sweep > 1) STAT(ST_RESWEEP, 1);
                while (todo) {
                    const int l = __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    bool take = (P >> l) & 1ull;
                    if (!take) {
                        if ((gone >> l) & 1ull) continue;    // the same pixel was accepted a moment ago
                        const double dl = rl(d, l);
                        if (!fresh) { R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true; }
                        double rw = fabs(R - dl);
                        double df = rw > kPi * 3 / 2.0 ? fabs(rw - 2.0 * kPi) : rw;
                        if (eps != 0.0 && (fabs(R) > kPi - eps || fabs(df - tol) <= eps || fabs(rw - kPi * 3 / 2.0) <= eps ||
                                           !(tol == tol))) {
                            R = atan2_g(sinS, cosS);         // :547, too close to call with the estimate
                            eps = 0.0;
                            fresh = true;
                            STAT(ST_EXACT, 1);
                            rw = fabs(R - dl);
                            df = rw > kPi * 3 / 2.0 ? fabs(rw - 2.0 * kPi) : rw;
                        }
                        take = df < tol;                     // :543
                    }
                    if (take) {
                        cosS += rl(cd, l);                   // :545
                        sinS += rl(sd, l);                   // :546
                        // the estimate drifts by at most turn / |sum| per accepted pixel
                        eps = (eps < 1e29 && Llb >= 0.9f) ? eps + (double)(1.002f * (float)turn * inv_ub(Llb)) + 1e-9 : 1e30;
                        Llb += cos_lb;
                        fresh = false;
                        acc |= 1ull << l;
                        if (!((P >> l) & 1ull)) gone |= __ballot(cand && q == __builtin_amdgcn_readlane(q, l));
                    }
                }
                if (acc) {
                    const bool mine = (acc >> lane) & 1ull;
                    if (mine) {
                        c.t_st[cell] = (id << 2) | (word & 3u);   // :549
                        c.stamp[q] = id;
                        lset(c, n + __builtin_popcountll(acc & ltmask), pack_xy(nx, ny));   // :551-556
                    }
                    n += __builtin_popcountll(acc);
                    c.dirty = true;
                    flt_valid = false;                       // the region angle moved
                }
                // entries that still have a growable non-member neighbour go to the next sweep's worklist
                const unsigned long long left = candm & ~gone & ~__ballot(cand && pc);
                if (filter && left) {
                    const bool has = valid && ((left >> (8 * e)) & 0xffull) != 0ull;
                    if (tol_small) {                         // slack of this entry's remaining candidates (circular distance - tol)
                        float mg = 3.0e38f;
                        if ((left >> lane) & 1ull) {
                            const double rw = fabs(Rcls - d);
                            mg = (float)((rw > kPi ? 2.0 * kPi - rw : rw) - tol);
                        }
                        mg = fminf(mg, __shfl_xor(mg, 1)); mg = fminf(mg, __shfl_xor(mg, 2)); mg = fminf(mg, __shfl_xor(mg, 4));
                        if (has && k == 0 && eidx < c.mcap) c.meta[eidx] = make_float2((float)Rcls, mg - 4e-6f);
                    }
                    const unsigned long long hm = __ballot(has && k == 0);
                    const int add = __builtin_popcountll(hm);
                    if (nxt_cnt + add > LCAP || n > 65535) filter = false;
                    else {
                        if (has && k == 0) wl_nxt[nxt_cnt + __builtin_popcountll(hm & ltmask)] = (uint16_t)eidx;
                        nxt_cnt += add;
                    }
                }
            }
            if (in_wl) wi += cnt; else i += cnt;
        }
        uint16_t* t = wl_cur; wl_cur = wl_nxt; wl_nxt = t;
        wl_cnt = nxt_cnt;
        sweep++;
        flt_valid = false;
        wg_fence();                                          // meta[] written in this sweep is read in the next
    } while (n != ex);
    c.gnum = n;
    c.has_copy = false;
    STAT(ST_GROW, 1);
    STAT(ST_GROWN, n);
    if (n > LCAP) STAT(ST_SPILL, 1);
    if (c.lane == 0 && (unsigned long long)n > c.stat[ST_MAXREG]) c.stat[ST_MAXREG] = (unsigned long long)n;
    STAT(ST_TGROW, (long long)__builtin_amdgcn_s_memtime() - t0);
    out_num = n;
    out_sin = sinS;
    out_cos = cosS;
}

// ---------------------------------------------------------------------------------------------
// grow8(): RegionGrower for EIGHT seeds at once, one seed per group of 8 lanes (lane = group*8 + neighbour).
//
// The serial chain of RegionGrower keeps one wavefront busy with a single frontier pixel per step in the
// common case (thin structures grow one pixel at a time), which leaves 56 of 64 lanes idle.  Here every group
// runs the same state machine on its own region -- one frontier entry per step, its 8 neighbours on the
// group's 8 lanes -- so one instruction stream advances eight regions.  Per-group state lives in registers
// (identical in the 8 lanes of a group); the region list, the sweep worklists and the curMap stamps of a group
// are private arrays in HBM, the last 64 list entries are mirrored in an LDS ring (the frontier is at the tail).
// Semantics per group are exactly those of grow(): reference order of the tests (entry by entry, neighbours
// row-major), sums accumulated in that order, estimate + rigorous margin with exact fallback, worklist sweeps.
// ---------------------------------------------------------------------------------------------
constexpr int NG = 8;
#ifndef LSD_G8_MIN_STEPS
#define LSD_G8_MIN_STEPS 48
#endif
#ifndef LSD_G8_MIN_ACTIVE
#define LSD_G8_MIN_ACTIVE 1
#endif
constexpr int G8_MIN_STEPS = LSD_G8_MIN_STEPS;    // group mode always runs this many steps ...
constexpr int G8_MIN_ACTIVE = LSD_G8_MIN_ACTIVE;  // ... and goes on while at least this many of the 8 regions are still growing
                                                  // (1 = to the end: since the group step got cheaper than a wave-wide batch per pixel,
                                                  //  only regions that outgrow their list slot are handed over)

struct G8 {                      // per-lane results (identical within a group)
    int n;                       // region size; -1: list capacity exceeded (caller falls back to grow())
    double sinS, cosS;
    int bx0, by0, bx1, by1;      // bounding box of the region
};

__device__ __forceinline__ unsigned grp_bits(unsigned long long m, int g) { return (unsigned)((m >> (8 * g)) & 0xffull); }
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }

// (out of line on purpose: inlined into the seed loop its step loop spills to scratch, and a scratch reload per step is a memory
//  round trip on the critical path; as a function it gets the register file to itself and saves the caller's registers once)
struct G8Ctx {                   // what grow8 needs of RCtx, passed by value
    int w, h, lane;
    const double* deg;
    const double* sn;
    const double* cs;
    const uint32_t* state;
    unsigned long long* stat;
};

__device__ __noinline__ G8 grow8(G8Ctx c, bool act, int sx, int sy, uint32_t* glist, uint32_t* gwl, uint16_t* gstamp,
                                 uint16_t id, uint32_t* ring, int gcap, double tol) {
    G8 out;
    const int lane = c.lane, w = c.w, h = c.h;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const int g = lane >> 3, j = lane & 7, gbase = g * 8;
    const int kk = j + (j >= 4);                             // 3x3 neighbourhood, row-major, centre skipped (:533-534)
    const int ox = kk % 3 - 1, oy = kk / 3 - 1;
    uint32_t* rg8 = ring + g * 64;
    uint32_t* wl_cur = gwl;
    uint32_t* wl_nxt = gwl + gcap;

    // seed (:512-520)
    double sinS = 0, cosS = 1, R = 0;
    if (act) {
        const size_t q0 = (size_t)sy * w + sx;
        R = c.deg[q0]; sinS = c.sn[q0]; cosS = c.cs[q0];     // regDeg, sin/cos(regDeg) (:515-516)
        if (j == 0) {
            __hip_atomic_store(&gstamp[q0], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            glist[0] = pack_xy(sx, sy); rg8[0] = pack_xy(sx, sy);
        }
    }
    double eps = 0.0;
    bool fresh = true;
    const bool tol_small = tol < 1.5;
    const double turn = tol < 1.1 ? tol : 1.1;
    const float cos_lb = tol_small ? cosf((float)tol) * 0.999f - 1e-6f : 0.0f;
    float Llb = 0.999f;
    int n = 1, i = 0, ex = 1, wi = 0, wl_cnt = 0, nxt_cnt = 0;
    bool in_wl = false, filter = true, done = !act, overflow = false;
    wg_fence();

    int steps = 0;
    while (true) {
        const unsigned long long live = __ballot(!done);
        if (!live) break;
        // Eight regions share every step; once most of them are finished the few long ones are cheaper in the
        // batched wave-wide grow() (8 frontier pixels per step for ONE region): hand them over (n = -1).
        if (++steps > G8_MIN_STEPS && __builtin_popcountll(live) < 8 * G8_MIN_ACTIVE) {
            if (!done) { overflow = true; done = true; }
            break;
        }
        // ---- pick this group's next entry (one per step) ----
        bool have = false;
        int eidx = 0;
        if (!done) {
            if (in_wl) {
                if (wi < wl_cnt) { eidx = (int)wl_cur[wi]; have = true; }
                else in_wl = false;
            }
            if (!have && !in_wl) {
                if (i < n) { eidx = i; have = true; }        // n is live (:529)
                else {                                       // sweep finished (:525)
                    uint32_t* t = wl_cur; wl_cur = wl_nxt; wl_nxt = t;
                    wl_cnt = nxt_cnt; nxt_cnt = 0; wi = 0;
                    if (n == ex) done = true;
                    else { ex = n; in_wl = filter; i = filter ? n : 0; }
                }
            }
        }
        uint32_t pk = 0;
        if (have) pk = (n - eidx <= 64) ? rg8[eidx & 63] : glist[eidx];
        const int nx = (int)(pk & 0xffffu) + ox, ny = (int)(pk >> 16) + oy;
        const bool inb = have && nx >= 0 && ny >= 0 && nx < w && ny < h;                  // :536
        const size_t q = (size_t)(inb ? ny : 0) * w + (inb ? nx : 0);
        uint32_t uw = 1u; uint16_t st = 0;
        double d = 0, sd = 0, cd = 0;
        if (inb) {
            uw = c.state[q];
            // the group's own 16-bit stamps are re-read right after being stored: go to L2 (sc1), not through the CU's L1
            st = __hip_atomic_load(&gstamp[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d = c.deg[q]; sd = c.sn[q]; cd = c.cs[q];
        }
        const bool cand = inb && st != id && (uw & 3u) != 1u;                             // :537 (2 is growable, Q5)
        const unsigned cb = grp_bits(__ballot(cand), g);
        const int m = __builtin_popcount(cb);
        if (!tol_small && cb) {                              // (rare) no cheap norm bound: refresh every step
            if (!fresh) { R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true; }
            const float sf = (float)sinS, cf = (float)cosS;
            Llb = sqrtf(sf * sf + cf * cf) * 0.999f - (float)m;
        }
        double margin = Llb >= (tol_small ? 0.9f : 3.0f) ? eps + (double)(1.002f * (float)m * (float)turn * inv_ub(Llb)) + 1e-7 : 1e30;
        double raw = fabs(R - d);
        double dif = raw > kPi * 3 / 2.0 ? fabs(raw - 2.0 * kPi) : raw;                   // :540-542
        bool cut = fabs(R) > kPi - margin || fabs(raw - kPi * 3 / 2.0) <= margin;
        bool pc = !cut && dif < tol - margin;
        bool amb = cut || (!pc && !(dif > tol + margin));
        unsigned ab = grp_bits(__ballot(cand && amb), g);
        if (!fresh && tol_small && ab) {                     // too close to call with the drifted estimate: refresh once
            R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true;
            margin = Llb >= 0.9f ? eps + (double)(1.002f * (float)m * (float)turn * inv_ub(Llb)) + 1e-7 : 1e30;
            raw = fabs(R - d);
            dif = raw > kPi * 3 / 2.0 ? fabs(raw - 2.0 * kPi) : raw;
            cut = fabs(R) > kPi - margin || fabs(raw - kPi * 3 / 2.0) <= margin;
            pc = !cut && dif < tol - margin;
            amb = cut || (!pc && !(dif > tol + margin));
        }
        const unsigned long long pm = __ballot(cand && pc), am = __ballot(cand && amb);
        const unsigned pb = grp_bits(pm, g);
        ab = grp_bits(am, g);
        unsigned accb = 0;
        // ---- the chain: neighbour jj of every group takes its turn (reference order inside a group) ----
        // (only the neighbour positions some group has work at: the union over the groups, folded on the scalar unit)
        unsigned long long fold = pm | am;
        fold |= fold >> 32; fold |= fold >> 16; fold |= fold >> 8;
        unsigned any8 = (unsigned)fold & 0xffu;
        #pragma unroll 1
        while (any8) {
            const int jj = __builtin_ctz(any8);
            any8 &= any8 - 1u;
            const unsigned bit = 1u << jj;
            const bool isp = (pb & bit) != 0u, isa = (ab & bit) != 0u;
            bool take = isp;
            if (isa) {
                const double dl = shfl_d(d, gbase + jj);
                if (!fresh) { R = (double)atan2f((float)sinS, (float)cosS); eps = kAngEps; fresh = true; }
                double rw = fabs(R - dl);
                double df = rw > kPi * 3 / 2.0 ? fabs(rw - 2.0 * kPi) : rw;
                if (eps != 0.0 && (fabs(R) > kPi - eps || fabs(df - tol) <= eps || fabs(rw - kPi * 3 / 2.0) <= eps || !(tol == tol))) {
                    R = atan2_g(sinS, cosS);                 // :547, too close to call with the estimate
                    eps = 0.0; fresh = true;
                    STAT(ST_EXACT, 1);
                    rw = fabs(R - dl);
                    df = rw > kPi * 3 / 2.0 ? fabs(rw - 2.0 * kPi) : rw;
                }
                take = df < tol;                             // :543
            }
            const double cdl = shfl_d(cd, gbase + jj), sdl = shfl_d(sd, gbase + jj);
            if (take) {
                cosS += cdl;                                 // :545
                sinS += sdl;                                 // :546
                eps = (eps < 1e29 && Llb >= 0.9f) ? eps + (double)(1.002f * (float)turn * inv_ub(Llb)) + 1e-9 : 1e30;
                Llb += cos_lb;
                fresh = false;
                accb |= bit;
            }
        }
        // ---- commit the accepted neighbours (:549-556) ----
        if (accb) {
            const int na = __builtin_popcount(accb);
            if (n + na > gcap) { overflow = true; done = true; }
            else {
                if (accb & (1u << j)) {
                    const int idx = n + __builtin_popcount(accb & ((1u << j) - 1u));
                    const uint32_t pv = pack_xy(nx, ny);
                    __hip_atomic_store(&gstamp[q], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    glist[idx] = pv;
                    rg8[idx & 63] = pv;
                }
                n += na;
            }
        }
        // entries that still have a growable non-member neighbour go to the next sweep's worklist
        if (have && filter && (cb & ~(pb | accb))) {
            if (nxt_cnt >= gcap) filter = false;
            else { if (j == 0) wl_nxt[nxt_cnt] = (uint32_t)eidx; nxt_cnt++; }
        }
        if (have) { if (in_wl) wi++; else i++; }
        wg_fence();                                          // this step's stamps / list / worklist stores precede the next step's loads
    }
    // bounding boxes (for the speculation check), 8 lanes per region
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    if (act && !overflow) {
        for (int e2 = j; e2 < n; e2 += 8) {
            const uint32_t pv = (n - e2 <= 64) ? rg8[e2 & 63] : glist[e2];
            const int x = (int)(pv & 0xffffu), y = (int)(pv >> 16);
            x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
        }
    }
    for (int off = 4; off >= 1; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
        x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
    }
    if (act && j == 0) {
        atomicAdd(&c.stat[ST_GROW], 1ull);
        atomicAdd(&c.stat[ST_GROWN], (unsigned long long)n);
        atomicMax(&c.stat[ST_MAXREG], (unsigned long long)n);
    }
    out.n = overflow ? -1 : n;
    out.sinS = sinS; out.cosS = cosS;
    out.bx0 = x0; out.by0 = y0; out.bx1 = x1; out.by1 = y1;
    STAT(ST_TGROW, (long long)__builtin_amdgcn_s_memtime() - t0);
    return out;
}

// ---------------------------------------------------------------------------------------------
// CenterGetter (:592-619) + OrientationGetter (:621-667) + RectangleConverter (:669-734)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rect_convert(RCtx& c, int num, double regdeg, double aliPro, int pk, double tol,
                                          Rec& r) {
    const int lane = c.lane, w = c.w;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    double cenX = 0, cenY = 0, ws = 0;
    for (int base = 0; base < num; base += 64) {                                   // :608-613
        const int kx = base + lane;
        const bool valid = kx < num;
        const uint32_t pkx = valid ? lget(c, kx) : 0u;
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        const double wgt = valid ? c.mag[(size_t)y * w + x] : 0.0;
        const double ax = wgt * x, ay = wgt * y;
        const int cnt = min(64, num - base);
        if (cnt == 64) {                         // full chunk: constant lane indices (no scalar loop overhead)
            #pragma unroll
            for (int j = 0; j < 64; j++) { cenX += rl(ax, j); cenY += rl(ay, j); ws += rl(wgt, j); }
        } else
        for (int j = 0; j < cnt; j++) {          // serial accumulation in list order (bit-exact)
            cenX += rl(ax, j);
            cenY += rl(ay, j);
            ws += rl(wgt, j);
        }
    }
    cenX = cenX / ws;
    cenY = cenY / ws;

    double Ixx = 0, Iyy = 0, Ixy = 0;
    ws = 0;
    for (int base = 0; base < num; base += 64) {                                   // :637-643
        const int kx = base + lane;
        const bool valid = kx < num;
        const uint32_t pkx = valid ? lget(c, kx) : 0u;
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        const double wgt = valid ? c.mag[(size_t)y * w + x] : 0.0;
        const double ddy = y - cenY, ddx = x - cenX;
        const double a = wgt * (ddy * ddy), b = wgt * (ddx * ddx), cc = wgt * ddx * ddy;
        const int cnt = min(64, num - base);
        if (cnt == 64) {
            #pragma unroll
            for (int j = 0; j < 64; j++) { Ixx += rl(a, j); Iyy += rl(b, j); Ixy -= rl(cc, j); ws += rl(wgt, j); }
        } else
        for (int j = 0; j < cnt; j++) {
            Ixx += rl(a, j);
            Iyy += rl(b, j);
            Ixy -= rl(cc, j);
            ws += rl(wgt, j);
        }
    }
    Ixx /= ws; Iyy /= ws; Ixy /= ws;
    const double dI = Ixx - Iyy;
    const double lamb = (Ixx + Iyy - sqrt(dI * dI + 4 * Ixy * Ixy)) / 2.0;          // :647
    double inertiaDeg;
    {
        const bool xx = fabs(Ixx) > fabs(Iyy);                                    // :649-652
        inertiaDeg = atan2_g(xx ? lamb - Ixx : Ixy, xx ? Ixy : lamb - Iyy);
    }
    double regDif = inertiaDeg - regdeg;                                          // :655-665
    while (regDif <= -kPi) regDif += 2 * kPi;
    while (regDif > kPi) regDif -= 2 * kPi;
    if (regDif < 0) regDif = -regDif;
    if (regDif > tol) inertiaDeg += kPi;

    double dx, dy;
    sincos_g(inertiaDeg, dy, dx);                                                  // :699-700
    double lenMin = 0, lenMax = 0, widMin = 0, widMax = 0;                         // Q9: start at 0 (:701)
    for (int base = 0; base < num; base += 64) {
        const int kx = base + lane;
        if (kx < num) {
            const uint32_t pkx = lget(c, kx);
            const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
            const double len = (x - cenX) * dx + (y - cenY) * dy;                  // :704
            const double wid = -(x - cenX) * dy + (y - cenY) * dx;                 // :705
            lenMin = fmin(lenMin, len); lenMax = fmax(lenMax, len);
            widMin = fmin(widMin, wid); widMax = fmax(widMax, wid);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {   // min/max are order-independent: plain wave reduction
        lenMin = fmin(lenMin, __shfl_xor(lenMin, off));
        lenMax = fmax(lenMax, __shfl_xor(lenMax, off));
        widMin = fmin(widMin, __shfl_xor(widMin, off));
        widMax = fmax(widMax, __shfl_xor(widMax, off));
    }
    r.x1 = cenX + lenMin * dx; r.y1 = cenY + lenMin * dy;                          // :717-720
    r.x2 = cenX + lenMax * dx; r.y2 = cenY + lenMax * dy;
    r.wid = widMax - widMin;
    r.cX = cenX; r.cY = cenY; r.deg = inertiaDeg; r.dx = dx; r.dy = dy;
    r.p = aliPro; r.prec = tol; r.pk = pk;
    if (r.wid < 1) r.wid = 1;                                                      // :730
    STAT(ST_TRECT, (long long)__builtin_amdgcn_s_memtime() - t0);
}

__device__ __forceinline__ double rec_density(int num, const Rec& r) {             // :757,:798,:827,:867
    const double ex = r.x1 - r.x2, ey = r.y1 - r.y2;
    return num / (sqrt(ex * ex + ey * ey) * r.wid);
}

// ---------------------------------------------------------------------------------------------
// RegionRadiusReducer, myLSD.cpp:736-802 (incl. the `i <= num` sentinel behaviour)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool radius_reduce_body(RCtx& c, int sx, int sy, int& num, double regdeg, Rec& rec,
                                                double denThre) {
    const int lane = c.lane, w = c.w;
    STAT(ST_RRR, 1);
    double den = rec_density(num, rec);
    if (den > denThre) return true;                                                // :760
    // keep the grow-order list for the marking loops before it gets reordered
    for (int k2 = lane; k2 < num; k2 += 64) c.gcopy[k2] = lget(c, k2);
    c.has_copy = true;
    wg_fence();
    c.dirty = false;
    const double ax = sx - rec.x1, ay = sy - rec.y1, bx = sx - rec.x2, by = sy - rec.y2;
    const double rad1 = sqrt(ax * ax + ay * ay), rad2 = sqrt(bx * bx + by * by);    // :768-769
    double rad = rad1 > rad2 ? rad1 : rad2;
    bool removed_any = false;
    while (den < denThre) {                                                        // :775
        rad *= 0.75;
        STAT(ST_RRRPASS, 1);
        int i = 0;
        while (i <= num) {                                                         // :779 (`<=`)
            int px, py;
            if (i == num) {
                if (!removed_any) { STAT(ST_OOB, 1); break; }   // the reference reads out of bounds here (UB): no removal
                px = 0; py = 0;                            // slot holds the NULL written at :784-785
            } else {
                const uint32_t pkx = lget(c, i);
                px = (int)(pkx & 0xffffu); py = (int)(pkx >> 16);
            }
            const double ddx = sx - px, ddy = sy - py;
            if (sqrt(ddx * ddx + ddy * ddy) > rad) {                               // :780
                if (lane == 0) {
                    const size_t q = (size_t)py * w + px;
                    c.stamp[q] = 0u;                                               // curMap = 0 (:781)
                    if (i == num) { lset(c, num - 1, 0u); }
                    else { lset(c, i, lget(c, num - 1)); lset(c, num - 1, 0u); }   // :782-785
                }
                if (i == num) STAT(ST_SENT, 1);
                wg_fence();
                removed_any = true;
                i--;
                num--;
            }
            i++;
        }
        if (num < 2) return false;                                                 // :792
        rect_convert(c, num, regdeg, rec.p, rec.pk, rec.prec, rec);                // :797
        den = rec_density(num, rec);
    }
    return true;
}

// Cold path (a handful of calls per image): kept out of line, with the context passed BY VALUE so that the
// caller's context stays in registers.  *copied_out tells the caller that lst was reordered (gcopy holds the
// grow-order list).
__device__ __noinline__ bool radius_reduce(RCtx c, int sx, int sy, int* num_io, double regdeg, Rec* rec_io,
                                           double denThre, int* copied_out) {
    int num = *num_io;
    Rec rec = *rec_io;
    c.has_copy = false;
    const bool ok = radius_reduce_body(c, sx, sy, num, regdeg, rec, denThre);
    *num_io = num;
    *rec_io = rec;
    *copied_out = c.has_copy ? 1 : 0;
    return ok;
}

// ---------------------------------------------------------------------------------------------
// LogGammaCalculator (:882-924): integers below kLgTable come from the host-computed table
// ---------------------------------------------------------------------------------------------
__device__ double log_gamma_dev(const RCtx& c, int x) {
    if (x >= 0 && x < kLgTable) return c.lgamma[x];
    const double xd = x;
    return 0.918938533204673 + (xd - 0.5) * log(xd) - xd +
           0.5 * xd * log(xd * sinh(1.0 / xd) + 1.0 / (810 * pow(xd, 6.0)));
}

// ---------------------------------------------------------------------------------------------
// RectangleNFACalculator, myLSD.cpp:926-1059 (the full-image pass :940-945 is a no-op, not restated)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rect_nfa_impl(RCtx& c, const Rec& rec) {
    const int lane = c.lane, xLim = c.w, yLim = c.h;
    const double logNT = c.logNT;
    STAT(ST_NFA, 1);
    double verX[4], verY[4];
    verX[0] = rec.x1 - rec.dy * rec.wid / 2.0;                                     // :949-956
    verX[1] = rec.x2 - rec.dy * rec.wid / 2.0;
    verX[2] = rec.x2 + rec.dy * rec.wid / 2.0;
    verX[3] = rec.x1 + rec.dy * rec.wid / 2.0;
    verY[0] = rec.y1 + rec.dx * rec.wid / 2.0;
    verY[1] = rec.y2 + rec.dx * rec.wid / 2.0;
    verY[2] = rec.y2 - rec.dx * rec.wid / 2.0;
    verY[3] = rec.y1 - rec.dx * rec.wid / 2.0;
    int offset;
    if ((rec.x1 < rec.x2) && (rec.y1 <= rec.y2)) offset = 0;                       // :959-966
    else if ((rec.x1 >= rec.x2) && (rec.y1 < rec.y2)) offset = 1;
    else if ((rec.x1 > rec.x2) && (rec.y1 >= rec.y2)) offset = 2;
    else offset = 3;
    const double vx0 = verX[offset & 3], vx1 = verX[(offset + 1) & 3], vx2 = verX[(offset + 2) & 3],
                 vx3 = verX[(offset + 3) & 3];
    const double vy0 = verY[offset & 3], vy1 = verY[(offset + 1) & 3], vy2 = verY[(offset + 2) & 3],
                 vy3 = verY[(offset + 3) & 3];
    const double cx0 = ceil(vx0);
    int xlen = cvt_x86(cx0 - floor(vx2));                                          // :973
    if (xlen < 0 && xlen != (int)0x80000000) xlen = -xlen;
    xlen = (int)((unsigned)xlen + 1u);
    const double k0 = (vy1 - vy0) / (vx1 - vx0);                                   // :979-982
    const double k1 = (vy2 - vy1) / (vx2 - vx1);
    const double k2 = (vy2 - vy3) / (vx2 - vx3);
    const double k3 = (vy3 - vy0) / (vx3 - vx0);
    int all = 0, ali = 0;
    for (int cb = 0; cb < xlen; cb += 64) {
        const int i = cb + lane;
        int cntc = 0, lo = 0, xr = 0;
        if (i < xlen) {
            xr = cvt_x86(i + cx0);                                                 // :976
            int yLow, yHigh;
            if (xr < vx3) yLow = cvt_x86(ceil(vy0 + (xr - vx0) * k3));             // :988-989
            else          yLow = cvt_x86(ceil(vy3 + (xr - vx3) * k2));             // :992-993
            if (xr < vx1) yHigh = cvt_x86(floor(vy0 + (xr - vx0) * k0));           // :998-999
            else          yHigh = cvt_x86(floor(vy1 + (xr - vx1) * k1));           // :1002-1003
            if (xr >= 0 && xr < xLim) {                                            // :1007
                lo = yLow < 0 ? 0 : yLow;
                const int hi = yHigh > yLim - 1 ? yLim - 1 : yHigh;
                if (hi >= lo) cntc = hi - lo + 1;
            }
        }
        int inc = cntc;                                       // inclusive wave scan of the column heights
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        const int tot = __builtin_amdgcn_readlane(inc, 63);
        if (tot == 0) continue;
        c.s_incl[lane] = inc; c.s_lo[lane] = lo; c.s_x[lane] = xr;
        wg_fence();
        all += tot;
        STAT(ST_NFAPX, tot);
        for (int t0 = 0; t0 < tot; t0 += 64) {                // flattened (column, row) pairs, 64 per step
            const int t = t0 + lane;
            bool hit = false;
            if (t < tot) {
                int ci = 0;                                    // smallest ci with s_incl[ci] > t
                for (int step = 32; step >= 1; step >>= 1)
                    if (c.s_incl[ci + step - 1] <= t) ci += step;
                const int ex = ci ? c.s_incl[ci - 1] : 0;
                const int j = c.s_lo[ci] + (t - ex);
                const double dv = c.deg[(size_t)j * xLim + c.s_x[ci]];
                hit = angle_diff(rec.deg, dv) < rec.prec;                          // :1009-1013
            }
            ali += __builtin_popcountll(__ballot(hit));
        }
        wg_fence();
    }
    if (all == 0 || ali == 0) return -logNT;                                       // :1019-1022
    const double logp = c.ptab[rec.pk * 3 + 0], log10p = c.ptab[rec.pk * 3 + 1], log1mp = c.ptab[rec.pk * 3 + 2];
    if (all == ali) return -logNT - all * log10p;                                  // :1023-1026
    const double proTerm = rec.p / (1.0 - rec.p);
    const double log1Coef = log_gamma_dev(c, all + 1) - log_gamma_dev(c, ali + 1) - log_gamma_dev(c, all - ali + 1);
    const double log1Term = log1Coef + ali * logp + (all - ali) * log1mp;          // :1033
    double term = exp(log1Term);
    const double eps = 2.2204e-16;
    if (fabs(term) < 100 * eps) {                                                  // :1037-1043
        if (ali > all * rec.p) return -log10(term) - logNT;
        return -logNT;
    }
    double binTail = term;
    const double tole = 0.1;
    for (int i = ali + 1; i <= all; i++) {                                         // :1046-1056
        const double binTerm = (all - i + 1) / (i * 1.0);
        const double multTerm = binTerm * proTerm;
        term *= multTerm;
        binTail += term;
        if (binTerm < 1) {
            const double err = term * ((1 - pow(multTerm, (double)(all - i + 1))) / (1.0 - multTerm) - 1);
            if (err < tole * fabs(-log10(binTail) - logNT) * binTail) break;
        }
    }
    return -log10(binTail) - logNT;
}

__device__ __forceinline__ double rect_nfa(RCtx& c, const Rec& rec) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const double v = rect_nfa_impl(c, rec);
    STAT(ST_TNFA, (long long)__builtin_amdgcn_s_memtime() - t0);
    return v;
}

// RectangleImprover, myLSD.cpp:1061-1158.  The reference's five hand-unrolled phases are walked by one
// loop (step 0 = the initial evaluation, then 5 phases x 5 tries) so that the NFA code is inlined once.
__device__ __forceinline__ double improve(RCtx& c, Rec& rec_io) {
    const double delt = 0.5, delt2 = delt / 2.0;
    Rec best = rec_io, r = rec_io;
    double bestNFA = 0;
    for (int step = 0; step <= 25; step++) {
        const int phase = step == 0 ? -1 : (step - 1) / 5;
        if (step > 0 && (step - 1) % 5 == 0) {              // phase boundary (:1078,:1093,:1108,:1126,:1144)
            if (bestNFA > 0) break;
            r = best;
        }
        bool eval = true;
        if (phase == 0 || phase == 4) {                     // :1084-1092 / :1148-1156  halve p
            r.p /= 2.0; r.prec = r.p * kPi; r.pk++;
        } else if (phase == 1) {                            // :1097-1107  reduce width
            if (r.wid - delt >= 0.5) r.wid -= delt; else eval = false;
        } else if (phase == 2) {                            // :1112-1125  move one side
            if (r.wid - delt >= 0.5) {
                r.x1 -= r.dy * delt2; r.y1 += r.dx * delt2;
                r.x2 -= r.dy * delt2; r.y2 += r.dx * delt2;
                r.wid -= delt;
            } else eval = false;
        } else if (phase == 3) {                            // :1130-1143  move the other side
            if (r.wid - delt >= 0.5) {
                r.x1 += r.dy * delt2; r.y1 -= r.dx * delt2;
                r.x2 += r.dy * delt2; r.y2 -= r.dx * delt2;
                r.wid -= delt;
            } else eval = false;
        }
        if (!eval) continue;
        const double v = rect_nfa(c, r);
        if (step == 0) { bestNFA = v; if (v > 0) break; }   // :1075-1079
        else if (v > bestNFA) { bestNFA = v; best = r; }
    }
    rec_io = best;
    return bestNFA;
}

// Refiner, myLSD.cpp:804-880, first half: the re-estimated angle tolerance (:833-855).  The regrow (:857),
// the refit (:866) and the density checks are in the caller's two-pass loop so that grow() and
// rect_convert() are inlined once.
__device__ __forceinline__ double refine_tol(RCtx& c, int sx, int sy, int num, const Rec& rec, double cenDeg) {
    const int lane = c.lane, w = c.w;
    double difSum = 0, squSum = 0;
    int ptNum = 0;
    for (int base = 0; base < num; base += 64) {                                   // :839-853
        const int kx = base + lane;
        bool flag = false;
        double degDif = 0;
        if (kx < num) {
            const uint32_t pkx = lget(c, kx);
            const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
            const double ddx = sx - x, ddy = sy - y;
            if (sqrt(ddx * ddx + ddy * ddy) < rec.wid) {
                flag = true;
                degDif = c.deg[(size_t)y * w + x] - cenDeg;
                while (degDif <= -kPi) degDif += 2 * kPi;
                while (degDif > kPi) degDif -= 2 * kPi;
            }
        }
        const double sq = degDif * degDif;
        unsigned long long m = __ballot(flag);
        while (m) {                               // serial accumulation in list order
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            difSum += rl(degDif, j);
            squSum += rl(sq, j);
            ptNum++;
        }
    }
    const double meanDif = difSum / (ptNum * 1.0);
    return 2.0 * sqrt((squSum - 2 * meanDif * difSum) / (ptNum * 1.0) + meanDif * meanDif);   // :855
}

// ---------------------------------------------------------------------------------------------
// seed loop, myLSD.cpp:219-272
// ---------------------------------------------------------------------------------------------
// usedMap marking (:243-248 / :259-265) restricted to the grown pixels; returns their bounding box.
__device__ __forceinline__ void mark_region(RCtx& c, uint32_t val, const uint32_t* src, int src_cnt, int& bx0, int& by0, int& bx1,
                                            int& by1) {
    const int w = c.w;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    wg_fence();                                   // the stamps written by grow() must have landed
    c.dirty = false;
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    const int cnt = src ? src_cnt : c.gnum;
    for (int k2 = c.lane; k2 < cnt; k2 += 64) {
        const uint32_t pkx = src ? src[k2] : (c.has_copy ? c.gcopy[k2] : lget(c, k2));
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        const size_t q = (size_t)y * w + x;
        if (src || c.stamp[q] == c.cur_id) {      // curMap == 1 only (src: a stashed list holds exactly those)
            c.state[q] = val;
            x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
        x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
    }
    bx0 = x0; by0 = y0; bx1 = x1; by1 = y1;
    STAT(ST_TMARK, (long long)__builtin_amdgcn_s_memtime() - t0);
}

// bounding box of the pixels currently in the region list
__device__ __forceinline__ void list_bbox(const RCtx& c, int num, int& bx0, int& by0, int& bx1, int& by1) {
    int x0 = bx0, y0 = by0, x1 = bx1, y1 = by1;
    for (int k2 = c.lane; k2 < num; k2 += 64) {
        const uint32_t pkx = lget(c, k2);
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
        x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
    }
    bx0 = x0; by0 = y0; bx1 = x1; by1 = y1;
}

__device__ __forceinline__ int lds_ld(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Commit ring: one record per seed in flight (index = seed number & (RW-1)).
//   R_EMPTY  not evaluated yet, or evaluated with a result that marks usedMap ("heavy": its owner keeps it
//            in registers and commits it itself when the cursor reaches it)
//   R_SKIP   the seed pixel was already used when it was looked at (monotone, so final): nothing to do
//   R_LIGHT  evaluated, no marks to make (small region :228 or refine failed :237); whoever advances the
//            cursor checks that no line accepted since the record's snapshot touches its box
//   R_REDO   a speculative result was invalidated (or abandoned): must be evaluated again at the cursor
//   R_BUSY   being re-evaluated at the cursor
enum { R_EMPTY = 0, R_SKIP = 1, R_LIGHT = 2, R_REDO = 3, R_BUSY = 4 };
constexpr int RW = 256;

struct Ring {
    int state[RW];
    int snap[RW];
    short box[RW][4];
    uint32_t lref[RW];   // list slot of the region ((wave * NB + buf) * NG + slot), ~0u: no lists kept (box check only)
    uint32_t lcnt[RW];   // n1 | n2 << 16: sizes of the two lists in the slot (first grow, Refiner's regrow)
};

__global__ __launch_bounds__(64 * NW, LSD_REGION_WAVES_PER_SIMD) void k_region(Geom g, Buffers b, uint32_t id_base, uint32_t id_base16) {
    __shared__ uint32_t lst[NW][LCAP];
    __shared__ uint16_t wl0[NW][LCAP], wl1[NW][LCAP];
    __shared__ double t_deg[NW][NSLOT * 64], t_sn[NW][NSLOT * 64], t_cs[NW][NSLOT * 64];
    __shared__ uint32_t t_st[NW][NSLOT * 64];
    __shared__ int t_tag[NW][NSLOT];
    __shared__ int s_incl[NW][64], s_lo[NW][64], s_x[NW][64];
    __shared__ unsigned long long s_stat[NW][ST_COUNT];
    __shared__ uint32_t g_ring[NW][NG * 64];             // tails of the 8 group-mode region lists of a wave
    __shared__ int s_next, s_commit, s_epoch, s_lines, s_ntrace, s_nseeds, s_lock;
    __shared__ short s_ring[RING][4];
    __shared__ int s_q[NW][2][NB];                       // in-flight blocks of a wave: first seed, accept epoch at fetch
    __shared__ Ring rg;

    const size_t img = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = g.w, h = g.h;
    const size_t npx = (size_t)g.npx;

    RCtx c;
    c.w = w; c.h = h; c.lane = lane;
    c.mag = b.mag + img * npx; c.deg = b.deg + img * npx; c.state = b.state + img * npx;
    c.sn = b.sn + img * npx; c.cs = b.cs + img * npx;
    c.stamp = b.stamps + (img * NW + wave) * npx;
    c.spill = b.spill + (img * NW + wave) * npx; c.gcopy = b.gcopy + (img * NW + wave) * npx;
    c.meta = reinterpret_cast<float2*>(b.wmeta) + (img * NW + wave) * (size_t)b.mcap; c.mcap = b.mcap;
    c.lst = lst[wave]; c.wl0 = wl0[wave]; c.wl1 = wl1[wave];
    c.t_st = t_st[wave]; c.t_deg = t_deg[wave]; c.t_sn = t_sn[wave]; c.t_cs = t_cs[wave]; c.t_tag = t_tag[wave];
    c.s_incl = s_incl[wave]; c.s_lo = s_lo[wave]; c.s_x = s_x[wave];
    c.tilesX = (w + 7) >> 3; c.dirty = false;
    c.cur_id = id_base; c.gnum = 0; c.has_copy = false;
    c.logNT = g.logNT; c.lgamma = b.lgamma; c.ptab = b.ptab;
    c.stat = s_stat[wave];
    if (lane < ST_COUNT) c.stat[lane] = 0ull;
    const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
    if (lane < NSLOT) c.t_tag[lane] = -1;
    for (int j = threadIdx.x; j < RW; j += 64 * NW) rg.state[j] = R_EMPTY;

    const uint32_t* ord = b.ord + img * npx;
    uint32_t* seedidx = b.seedidx + img * npx;
    const int nb = b.nb[img];
    double* recs = b.recs + img * (size_t)b.max_lines * 12;
    double* recs_scaled = b.recs_scaled + img * (size_t)b.max_lines * 4;
    SeedRec* trace = b.seeds ? reinterpret_cast<SeedRec*>(b.seeds) + img * npx : nullptr;
    int* rnum = b.rnum + img * (size_t)RW * 2;            // (num0, final_num << 2 | outcome) of published records: read by the seed trace only

    // potential seeds: sorted entries whose pixel is not below the gradient threshold (usedMap == 0 after K2)
    if (wave == 0) {
        int cnt = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int base = 0; base < nb; base += 64) {
            const int idx = base + lane;
            const bool ok = idx < nb && (c.state[ord[idx < nb ? idx : 0]] & 3u) == 0u;   // :222
            const unsigned long long m = __ballot(ok);
            if (ok) seedidx[cnt + __builtin_popcountll(m & lt)] = (uint32_t)idx;
            cnt += __builtin_popcountll(m);
        }
        if (lane == 0) { s_next = 0; s_commit = 0; s_epoch = 0; s_lines = 0; s_ntrace = 0; s_nseeds = cnt; s_lock = 0; }
        wg_fence();
    }
    __syncthreads();
    const int nseeds = s_nseeds;

    // box overlap of record-style boxes against the lines accepted in epochs [snap, now)
    auto hit_since = [&](int snap, int now, int x0, int y0, int x1, int y1) -> bool {
        if (now - snap > RING) return true;
        bool hit = false;
        for (int ep = snap; ep < now; ep++) {
            const short* r = s_ring[ep & (RING - 1)];
            if (!(r[2] < x0 || r[0] > x1 || r[3] < y0 || r[1] > y1)) hit = true;
        }
        return hit;
    };
    auto write_trace = [&](int k, int num0, int fnum, int outcome, double logNFA) {
        if (lane == 0) {
            if (trace) {
                const int oidx = (int)seedidx[k];
                const uint32_t pp = ord[oidx];
                SeedRec tr;
                tr.order_idx = oidx; tr.x = (int)(pp % (uint32_t)w); tr.y = (int)(pp / (uint32_t)w);
                tr.num = num0; tr.outcome = outcome; tr.final_num = fnum; tr.logNFA = logNFA;
                trace[s_ntrace] = tr;
            }
            s_ntrace = s_ntrace + 1;
        }
    };
    // True if a pixel examined by a region (a member or one of its 8 neighbours) was banned by a line accepted in epoch
    // >= snap: only then can the region's evaluation differ from what it would be now (it read usedMap only as
    // "banned?", and only accepted lines ban).  Accepted pixels carry their line's epoch + 1 above the usedMap bits.
    auto examined_hit = [&](const uint32_t* lp, int cnt, int snap) -> bool {
        bool hit = false;
        for (int base = 0; base < cnt; base += 64) {
            const int k2 = base + lane;
            if (k2 < cnt) {
                const uint32_t pkx = lp[k2];
                const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
                #pragma unroll
                for (int t9 = 0; t9 < 9; t9++) {
                    const int xx = x + t9 % 3 - 1, yy = y + t9 / 3 - 1;
                    if (xx >= 0 && yy >= 0 && xx < w && yy < h) {
                        const uint32_t sw = c.state[(size_t)yy * w + xx];
                        if ((sw & 3u) == 1u && (int)(sw >> 2) > snap) hit = true;
                    }
                }
            }
        }
        return __ballot(hit) != 0ull;
    };
    // Moves the commit cursor over finished records (one wave at a time).
    auto advance = [&]() {
        int got = 0;
        if (lane == 0) got = atomicCAS(&s_lock, 0, 1) == 0 ? 1 : 0;
        got = __builtin_amdgcn_readfirstlane(got);
        if (!got) return;
        while (true) {
            const int f = lds_ld(&s_commit);
            if (f >= nseeds) break;
            const int st = lds_ld(&rg.state[f & (RW - 1)]);
            if (st == R_SKIP) {
                if (lane == 0) { rg.state[f & (RW - 1)] = R_EMPTY; lds_st(&s_commit, f + 1); }
                continue;
            }
            if (st == R_LIGHT) {
                const int snap = rg.snap[f & (RW - 1)], now = lds_ld(&s_epoch);
                const short* bx = rg.box[f & (RW - 1)];
                if (now != snap && hit_since(snap, now, bx[0], bx[1], bx[2], bx[3])) {
                    const uint32_t lr = rg.lref[f & (RW - 1)], lc = rg.lcnt[f & (RW - 1)];
                    bool conflict = true;
                    if (lr != ~0u) {                       // the lists are still in their slot: look at the pixels themselves
                        wg_fence();
                        conflict = examined_hit(b.glist + (img * (size_t)(NW * NB * NG) + lr) * b.gcap, (int)(lc & 0xffffu) + (int)(lc >> 16), snap);
                    }
                    if (conflict) {
                        if (lane == 0) lds_st(&rg.state[f & (RW - 1)], R_REDO);
                        break;
                    }
                }
                bool used_now = false;
                if (trace) used_now = (c.state[ord[seedidx[f]]] & 3u) != 0u;     // the reference skips it then (:222): no record
                if (trace && !used_now) {
                    const int no = rnum[(f & (RW - 1)) * 2 + 1];
                    write_trace(f, rnum[(f & (RW - 1)) * 2], no >> 2, no & 3, 0.0);
                } else if (!trace) write_trace(f, 0, 0, 0, 0.0);
                if (lane == 0) { rg.state[f & (RW - 1)] = R_EMPTY; lds_st(&s_commit, f + 1); }
                continue;
            }
            break;
        }
        if (lane == 0) lds_st(&s_lock, 0);
    };

    // group-mode storage of this wave's 8 groups (lane -> group lane>>3)
    const int grp = lane >> 3;
    uint32_t* const wave_glist = b.glist + (img * NW + wave) * (size_t)NB * NG * b.gcap;   // [NB][NG][gcap]
    uint32_t* my_gwl = b.gwl + ((img * NW + wave) * NG + grp) * 2 * (size_t)b.gcap;
    uint16_t* my_gstamp = b.gstamp + ((img * NW + wave) * NG + grp) * npx;
    uint32_t gid_local = 0;                                // grows of this group in this run (16-bit stamp = id_base16 + it)

    int forced_k = -1;                                     // seed to (re)evaluate non-speculatively at the cursor
    int blk_k0 = -NG, blk_g = NG, blk_snap = 0;            // current block of 8 consecutive seeds, next group to hand over
    G8 blk;                                                // per-lane results of the block (group = lane>>3)
    blk.n = 0; blk.sinS = 0; blk.cosS = 0; blk.bx0 = blk.by0 = 0; blk.bx1 = blk.by1 = -1;
    bool blk_skip = true;
    // Results that mark usedMap ("heavy") are stashed -- record in pend[], pixel list in the seed's glist slot of the
    // block's buffer -- and the wave carries on: with the rest of its block, then with up to NB - 1 further blocks.
    // Stashed results are committed in seed order as the cursor reaches them (the oldest block of the wave first).
    int q_head = 0, q_cnt = 0, cur_buf = 0;                // ring of NB block buffers: oldest, blocks in flight, newest
    unsigned pend32 = 0;                                   // byte i = slots of buffer i holding a stashed result
    int* const qk = s_q[wave][0];
    int* const qs = s_q[wave][1];
    double* const wave_pend = b.pend + (img * NW + wave) * (size_t)NB * NG * 24;
    const unsigned long long ltm = (1ull << lane) - 1ull;
    long long tl = (long long)__builtin_amdgcn_s_memtime();
    // coarse accounting of this wave's time (s_memtime ticks since the last stamp go to slot i)
#define LT(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); STAT((i), t_ - tl); tl = t_; } while (0)
    while (true) {
        // ---- choose the next job ----
        int k;
        bool spec, from_group = false, from_stash = false;
        int epoch_snap;
        int st_buf = 0, st_k0 = 0;                         // buffer / first seed of the block a stashed result belongs to
        LT(ST_TSELECT);
        // retire blocks whose results are all handed over and committed
        while (q_cnt > 0 && ((pend32 >> (8 * q_head)) & 0xffu) == 0u && (q_cnt > 1 || blk_g >= NG)) {
            q_head = (q_head + 1) % NB; q_cnt--;
        }
        int kp = -1;                                       // earliest stashed result of this wave
        if (q_cnt > 0) {
            const unsigned m = (pend32 >> (8 * q_head)) & 0xffu;
            if (m) { st_buf = q_head; st_k0 = qk[q_head]; kp = st_k0 + __builtin_ctz(m); }
        }
        if (forced_k >= 0) { k = forced_k; spec = false; forced_k = -1; epoch_snap = lds_ld(&s_epoch); }
        else {
            // a record waiting to be redone at the cursor has priority
            const int f = lds_ld(&s_commit);
            int won = 0;
            if (f < nseeds && lds_ld(&rg.state[f & (RW - 1)]) == R_REDO) {
                if (lane == 0) won = atomicCAS(&rg.state[f & (RW - 1)], R_REDO, R_BUSY) == R_REDO ? 1 : 0;
                won = __builtin_amdgcn_readfirstlane(won);
            }
            if (won) { k = f; spec = false; epoch_snap = lds_ld(&s_epoch); }
            else if (kp == f) {
                // ---- the cursor is at a stashed result of this wave: commit it ----
                k = kp;
                from_stash = true; spec = true; epoch_snap = qs[st_buf];
            }
            else if (q_cnt > 0 && blk_g < NG) {
                // ---- hand over the next result of the current block ----
                const int gq = blk_g++;
                k = blk_k0 + gq;
                if (k >= nseeds) { blk_g = NG; continue; }
                const int src = gq * 8;
                const bool gskip = __builtin_amdgcn_readlane((int)blk_skip, src) != 0;
                const int gn = __builtin_amdgcn_readlane(blk.n, src);
                if (gskip) {
                    if (lane == 0) lds_st(&rg.state[k & (RW - 1)], R_SKIP);
                    if (gq == NG - 1) advance();
                    continue;
                }
                if (gn >= 0 && gn < g.regThre) {           // small region (:228): nothing to evaluate, nothing to mark
                    // cross-lane reads stay in uniform control flow (values of inactive lanes are not defined inside a branch)
                    const int qx0 = __builtin_amdgcn_readlane(blk.bx0, src), qy0 = __builtin_amdgcn_readlane(blk.by0, src);
                    const int qx1 = __builtin_amdgcn_readlane(blk.bx1, src), qy1 = __builtin_amdgcn_readlane(blk.by1, src);
                    if (lane == 0) {
                        const int r = k & (RW - 1);
                        rg.snap[r] = blk_snap;
                        rg.box[r][0] = (short)(qx0 - 1);
                        rg.box[r][1] = (short)(qy0 - 1);
                        rg.box[r][2] = (short)(qx1 + 1);
                        rg.box[r][3] = (short)(qy1 + 1);
                        rg.lref[r] = ~0u; rg.lcnt[r] = 0u;
                        if (trace) { rnum[r * 2] = gn; rnum[r * 2 + 1] = (gn << 2) | 0; }
                        lds_st(&rg.state[r], R_LIGHT);
                    }
                    if (gq == NG - 1) advance();
                    continue;
                }
                spec = true;
                epoch_snap = blk_snap;
                from_group = gn >= 0;                      // gn < 0: list overflow in group mode -> plain grow() below
            } else if (q_cnt < NB && lds_ld(&s_next) < nseeds && lds_ld(&s_next) - f < RW - NW * NG) {
                // ---- fetch a block of 8 consecutive seeds and grow them together ----
                if (NB == 1 && blk_k0 >= 0 && f < blk_k0 + NG) {
                    // records of the previous block are still ahead of the cursor and its list slots are about to be reused:
                    // withdraw their lists (under the cursor lock), their validation falls back to the bounding box
                    while (true) {
                        int got = 0;
                        if (lane == 0) got = atomicCAS(&s_lock, 0, 1) == 0 ? 1 : 0;
                        if (__builtin_amdgcn_readfirstlane(got)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (lane < NG && blk_k0 + lane < nseeds) rg.lref[(blk_k0 + lane) & (RW - 1)] = ~0u;
                    wg_fence();
                    if (lane == 0) lds_st(&s_lock, 0);
                }
                int k0 = 0;
                if (lane == 0) k0 = atomicAdd(&s_next, NG);
                k0 = __builtin_amdgcn_readfirstlane(k0);
                if (k0 >= nseeds) continue;
                cur_buf = (q_head + q_cnt) % NB; q_cnt++;
                blk_k0 = k0; blk_g = 0;
                blk_snap = lds_ld(&s_epoch);               // before anything of usedMap is read for these seeds
                if (lane == 0) { qk[cur_buf] = k0; qs[cur_buf] = blk_snap; }
                wg_fence();
                const int kg = k0 + grp;
                bool gact = kg < nseeds;
                int gsx = 0, gsy = 0;
                if (gact) {
                    const uint32_t ppg = ord[seedidx[kg]];
                    gsx = (int)(ppg % (uint32_t)w); gsy = (int)(ppg / (uint32_t)w);
                    if ((c.state[ppg] & 3u) != 0u) gact = false;          // monotone: once used, always used (:222)
                }
                blk_skip = !gact;
                if (gid_local >= 2047u) {                  // 16-bit stamp range of this run exhausted: plain grow() for the rest
                    blk.n = -1;
                } else {
                    gid_local++;
                    G8Ctx gc;
                    gc.w = w; gc.h = h; gc.lane = lane; gc.deg = c.deg; gc.sn = c.sn; gc.cs = c.cs; gc.state = c.state; gc.stat = c.stat;
                    blk = grow8(gc, gact, gsx, gsy, wave_glist + ((size_t)cur_buf * NG + grp) * b.gcap, my_gwl, my_gstamp,
                                (uint16_t)(id_base16 + gid_local), g_ring[wave], b.gcap, g.degThre);
                }
                LT(ST_TGROUP);
                continue;
            } else if (kp >= 0) {
                // ---- nothing else to do: wait for the turn of the earliest stashed result ----
                const long long tw0 = (long long)__builtin_amdgcn_s_memtime();
                while (true) {
                    advance();
                    const int f2 = lds_ld(&s_commit);
                    if (f2 == kp) break;
                    if (lds_ld(&rg.state[f2 & (RW - 1)]) == R_REDO) break;   // somebody has to redo f2 (top of the loop)
                    __builtin_amdgcn_s_sleep(4);
                }
                tl = (long long)__builtin_amdgcn_s_memtime();
                STAT(ST_WAIT, tl - tw0);
                continue;
            } else {
                advance();
                if (lds_ld(&s_next) >= nseeds && lds_ld(&s_commit) >= nseeds) break;   // everything is committed
                __builtin_amdgcn_s_sleep(8);               // nothing left to hand out, or the run-ahead window is full
                LT(ST_TIDLE);
                continue;
            }
        }
        const int oidx = (int)seedidx[k];
        const uint32_t pp = ord[oidx];
        const int sx = (int)(pp % (uint32_t)w), sy = (int)(pp / (uint32_t)w);

        int outcome = 0, num = 0, num0 = 0;
        double logNFA = 0;
        Rec rec;
        bool skip = false;
        const uint32_t* m_src = nullptr;                   // stashed pixel list to mark at commit (else the wave's own list)
        int m_cnt = 0;
        int x0 = 0, y0 = 0, x1 = -1, y1 = -1;              // box of everything the evaluation examined
        int st_n1 = -1, st_n2 = 0;                         // stashed result: sizes of the lists kept in its slot (-1: none)
        const uint32_t* st_list = nullptr;
        if (from_stash) {
            const int slot = k - st_k0;
            pend32 &= ~(1u << (8 * st_buf + slot));
            wg_fence();
            const double pv = wave_pend[(st_buf * NG + slot) * 24 + (lane < 24 ? lane : 0)];
            rec.x1 = rl(pv, 0); rec.y1 = rl(pv, 1); rec.x2 = rl(pv, 2); rec.y2 = rl(pv, 3); rec.wid = rl(pv, 4); rec.cX = rl(pv, 5);
            rec.cY = rl(pv, 6); rec.deg = rl(pv, 7); rec.dx = rl(pv, 8); rec.dy = rl(pv, 9); rec.p = rl(pv, 10); rec.prec = rl(pv, 11);
            logNFA = rl(pv, 12);
            rec.pk = (int)rl(pv, 13); outcome = (int)rl(pv, 14); num0 = (int)rl(pv, 15); num = (int)rl(pv, 16); m_cnt = (int)rl(pv, 17);
            x0 = (int)rl(pv, 18); y0 = (int)rl(pv, 19); x1 = (int)rl(pv, 20); y1 = (int)rl(pv, 21);
            {
                const long long pk3 = (long long)rl(pv, 22);
                st_n1 = (int)(pk3 % 32768ll) - 1; st_n2 = (int)((pk3 / 32768ll) % 32768ll);
                st_list = wave_glist + ((size_t)st_buf * NG + slot) * b.gcap;
                m_src = st_list + (int)(pk3 / (32768ll * 32768ll));
            }
        } else {
        // ---- evaluate ----
        if (!from_group) { wg_fence(); }
        invalidate_tiles(c);
        skip = from_group ? false : (c.state[pp] & 3u) != 0u;   // monotone: once used, always used (:222)
        int fx0 = 0x7fffffff, fy0 = 0x7fffffff, fx1 = -1, fy1 = -1;   // box of a first grow that refine() replaced
        // list slot of a speculative evaluation: [first grow (n1)][Refiner's regrow (n2)][pixels to mark, if not one of those]
        uint32_t* const gl0 = wave_glist + ((size_t)cur_buf * NG + (spec ? k - blk_k0 : 0)) * b.gcap;
        int n1 = -1;                                       // -1: the lists are not kept (validation by bounding box only)
        bool regrown = false;
        if (!skip) {
            // RegionGrower -> RectangleConverter -> Refiner (:225-238) as a two-pass loop: pass 0 grows with the
            // global tolerance (or takes the region grown in group mode), pass 1 (only when the rectangle is too
            // sparse, :829) regrows with the tolerance re-estimated by Refiner (:833-857).
            const double seedDeg = c.deg[pp];
            double tol = g.degThre, regdeg = seedDeg, gs, gc;
            bool done = false;
            for (int pass = 0; pass < 2 && !done; pass++) {
                if (pass == 0 && from_group) {
                    // the region of seed k was grown by group (k - blk_k0): bring its list into this wave's list storage
                    const int src = (k - blk_k0) * 8;
                    num = __builtin_amdgcn_readlane(blk.n, src);
                    gs = rl(blk.sinS, src); gc = rl(blk.cosS, src);
                    const uint32_t* gl = wave_glist + ((size_t)cur_buf * NG + (k - blk_k0)) * b.gcap;
                    for (int k2 = lane; k2 < num && k2 < LCAP; k2 += 64) c.lst[k2] = gl[k2];
                    for (int k2 = LCAP + lane; k2 < num; k2 += 64) c.spill[k2 - LCAP] = gl[k2];
                    c.gnum = num; c.has_copy = false;
                    n1 = num;
                    wg_fence();
                } else {
                    const long long tg0 = (long long)__builtin_amdgcn_s_memtime();
                    grow(c, sx, sy, seedDeg, tol, num, gs, gc);                       // :225 / :857
                    if (pass == 0) { STAT(ST_THANDED, (long long)__builtin_amdgcn_s_memtime() - tg0); STAT(ST_NHANDED, 1); STAT(ST_PXHANDED, num); }
                    if (pass == 0 && spec && num <= b.gcap) {          // keep the first list for the validation at the cursor
                        for (int k2 = lane; k2 < num; k2 += 64) gl0[k2] = lget(c, k2);
                        n1 = num;
                    }
                    if (pass == 1) regrown = true;

                }
                if (pass == 0) {
                    num0 = num;
                    if (num < g.regThre) { done = true; break; }                      // :228 (not marked, Q5)
                } else if (num < 2) { outcome = 1; done = true; break; }              // :861
                regdeg = num > 1 ? atan2_g(gs, gc) : seedDeg;                         // reg.deg (:547, :581)
                rect_convert(c, num, regdeg, g.aliPro, 0, g.degThre, rec);            // :232 / :866 (p, prec still the defaults)
                const double den = rec_density(num, rec);
                if (pass == 0) {
                    if (den >= g.denThre) break;                                      // :829 dense enough
                    if (spec) list_bbox(c, num, fx0, fy0, fx1, fy1);                  // the regrow replaces this list
                    tol = refine_tol(c, sx, sy, num, rec, seedDeg);                   // :833-855
                } else if (den < g.denThre) {                                         // :869-877
                    int copied = 0;
                    const bool ok = radius_reduce(c, sx, sy, &num, regdeg, &rec, g.denThre, &copied);
                    c.has_copy = copied != 0;                // lst was reordered: gcopy holds the grow-order list
                    if (!ok) { outcome = 1; done = true; }
                }
            }
            if (!done) {
                logNFA = improve(c, rec);                                             // :240
                outcome = logNFA <= 0 ? 2 : 3;                                        // :242
            }
        }

        // ---- hand the result over ----
        LT(ST_TEVAL);
        if (spec) {
            if (skip) {
                if (lane == 0) lds_st(&rg.state[k & (RW - 1)], R_SKIP);
                advance();
                continue;
            }
            // box of everything this evaluation examined (region pixels and their 8-neighbourhoods)
            x0 = fx0; y0 = fy0; x1 = fx1; y1 = fy1;
            if (c.has_copy) {                              // RegionRadiusReducer reordered/shrunk lst: use the grow-order copy
                for (int k2 = lane; k2 < c.gnum; k2 += 64) {
                    const uint32_t pkx = c.gcopy[k2];
                    const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
                    x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
                }
                for (int off = 32; off >= 1; off >>= 1) {
                    x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
                    x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
                }
            } else list_bbox(c, c.gnum, x0, y0, x1, y1);
            x0 -= 1; y0 -= 1; x1 += 1; y1 += 1;
            bool precise = n1 >= 0;
            int n2 = 0;
            if (regrown) {                                 // keep Refiner's regrow (in grow order, before any reduction) behind the first list
                if (precise && n1 + c.gnum <= b.gcap) {
                    for (int k2 = lane; k2 < c.gnum; k2 += 64) gl0[n1 + k2] = c.has_copy ? c.gcopy[k2] : lget(c, k2);
                    n2 = c.gnum;
                } else precise = false;
            }
            const int slot = k - blk_k0;
            if (outcome <= 1) {                            // nothing to mark: publish and move on
                if (lane == 0) {
                    const int r = k & (RW - 1);
                    rg.snap[r] = epoch_snap;
                    rg.box[r][0] = (short)x0; rg.box[r][1] = (short)y0; rg.box[r][2] = (short)x1; rg.box[r][3] = (short)y1;
                    rg.lref[r] = precise ? (uint32_t)((wave * NB + cur_buf) * NG + slot) : ~0u;
                    rg.lcnt[r] = precise ? ((uint32_t)n1 | ((uint32_t)n2 << 16)) : 0u;
                    if (trace) { rnum[r * 2] = num0; rnum[r * 2 + 1] = (num << 2) | outcome; }
                    lds_st(&rg.state[r], R_LIGHT);
                }
                advance();
                continue;
            }
            // marks to make: stash the result (record in pend[], the pixels to mark in the list slot) and carry on with the block
            int m_off = 0, mcnt = num;                     // not regrown: the first list is exactly the region
            if (!regrown) {
                if (!precise) {                            // (larger than a list slot) evaluate again at the cursor
                    if (lane == 0) lds_st(&rg.state[k & (RW - 1)], R_REDO);
                    STAT(ST_REDO, 1);
                    advance();
                    continue;
                }
            } else if (precise && !c.has_copy) { m_off = n1; mcnt = n2; }      // the regrow as it is
            else {
                m_off = precise ? n1 + n2 : 0;
                if (m_off + c.gnum > b.gcap) { precise = false; m_off = 0; }
                if (c.gnum > b.gcap) {
                    if (lane == 0) lds_st(&rg.state[k & (RW - 1)], R_REDO);
                    STAT(ST_REDO, 1);
                    advance();
                    continue;
                }
                wg_fence();                                // the stamps written by grow() must have landed
                c.dirty = false;
                mcnt = 0;
                for (int base = 0; base < c.gnum; base += 64) {
                    const int k2 = base + lane;
                    uint32_t pkx = 0;
                    bool keep = false;
                    if (k2 < c.gnum) {
                        pkx = c.has_copy ? c.gcopy[k2] : lget(c, k2);
                        keep = c.stamp[(size_t)(pkx >> 16) * w + (pkx & 0xffffu)] == c.cur_id;   // curMap == 1 only
                    }
                    const unsigned long long km = __ballot(keep);
                    if (keep) gl0[m_off + mcnt + __builtin_popcountll(km & ltm)] = pkx;
                    mcnt += __builtin_popcountll(km);
                }
            }
            if (lane == 0) {
                double* P = wave_pend + (cur_buf * NG + slot) * 24;
                P[0] = rec.x1; P[1] = rec.y1; P[2] = rec.x2; P[3] = rec.y2; P[4] = rec.wid; P[5] = rec.cX; P[6] = rec.cY;
                P[7] = rec.deg; P[8] = rec.dx; P[9] = rec.dy; P[10] = rec.p; P[11] = rec.prec; P[12] = logNFA;
                P[13] = (double)rec.pk; P[14] = (double)outcome; P[15] = (double)num0; P[16] = (double)num; P[17] = (double)mcnt;
                P[18] = (double)x0; P[19] = (double)y0; P[20] = (double)x1; P[21] = (double)y1;
                P[22] = (double)((long long)(precise ? n1 + 1 : 0) + 32768ll * n2 + 32768ll * 32768ll * m_off);
            }
            wg_fence();
            pend32 |= 1u << (8 * cur_buf + slot);
            continue;
        }
        }   // !from_stash

        if (spec) {
            // ---- a stashed result at the cursor (k == s_commit): is it still what the sequential run would get? ----
            wg_fence();
            bool bad = (c.state[pp] & 3u) != 0u;           // an earlier seed marked the pixel meanwhile
            if (bad) {
                STAT(ST_DISCARD, 1);
                if (lane == 0) { lds_st(&s_commit, k + 1); }
                continue;
            }
            const int now = lds_ld(&s_epoch);
            if (now != epoch_snap && hit_since(epoch_snap, now, x0, y0, x1, y1)) {
                bool conflict = true;
                if (st_n1 >= 0) conflict = examined_hit(st_list, st_n1 + st_n2, epoch_snap);   // the pixels themselves
                if (conflict) {
                    STAT(ST_REDO, 1);
                    forced_k = k;                          // evaluate again; everything earlier is committed now
                    continue;
                }
            }
        }

        // ---- commit at the cursor (k == s_commit, nobody else can commit) ----
        if (!skip) {
            write_trace(k, num0, outcome == 0 ? num0 : num, outcome, logNFA);
            int bx0, by0, bx1, by1;
            if (outcome == 2) {                                                      // :242-250
                mark_region(c, 2u, m_src, m_cnt, bx0, by0, bx1, by1);
            } else if (outcome == 3) {
                const int li = s_lines;
                if (li < b.max_lines && lane == 0) {
                    double* rr = recs + (size_t)li * 12;
                    rr[0] = rec.x1; rr[1] = rec.y1; rr[2] = rec.x2; rr[3] = rec.y2; rr[4] = rec.wid; rr[5] = rec.cX;
                    rr[6] = rec.cY; rr[7] = rec.deg; rr[8] = rec.dx; rr[9] = rec.dy; rr[10] = rec.p; rr[11] = rec.prec;
                    double x1 = rec.x1, y1 = rec.y1, x2 = rec.x2, y2 = rec.y2;
                    if (g.sca != 1) {                                                // :252-258
                        x1 = (x1 - 1.0) / g.sca + 1; y1 = (y1 - 1.0) / g.sca + 1;
                        x2 = (x2 - 1.0) / g.sca + 1; y2 = (y2 - 1.0) / g.sca + 1;
                    }
                    double* rs = recs_scaled + (size_t)li * 4;
                    rs[0] = x1; rs[1] = y1; rs[2] = x2; rs[3] = y2;
                }
                mark_region(c, 1u | ((uint32_t)(lds_ld(&s_epoch) + 1) << 2), m_src, m_cnt, bx0, by0, bx1, by1);   // :259-265 (+ the line's epoch)
                wg_fence();                               // the marks must be visible before the epoch moves
                if (lane == 0) {
                    const int ep = s_epoch;
                    short* r = s_ring[ep & (RING - 1)];
                    r[0] = (short)bx0; r[1] = (short)by0; r[2] = (short)bx1; r[3] = (short)by1;
                    s_lines = li + 1;
                    lds_st(&s_epoch, ep + 1);
                }
            }
        }
        wg_fence();                                       // marks + ring visible before the cursor moves
        if (lane == 0) { rg.state[k & (RW - 1)] = R_EMPTY; lds_st(&s_commit, k + 1); }
    }

    __syncthreads();
    if (threadIdx.x == 0) {
        b.counts[img] = s_lines;
        if (b.nseed) b.nseed[img] = s_ntrace;
    }
    if (b.stats) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(b.stats + img * 32);
        if (lane == 0 && wave == 0) { c.stat[ST_TOTAL] = (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - t_begin); c.stat[ST_SEEDS] = (unsigned long long)nseeds; }
        if (lane < ST_COUNT) {
            if (lane == ST_MAXREG) atomicMax(&st[lane], c.stat[lane]);
            else atomicAdd(&st[lane], c.stat[lane]);
        }
    }
}

}  // namespace RVAR

#if LSD_REGION_NW == 8
void launch_region_w8(const Geom& g, const Buffers& b, int n, uint32_t id_base, uint32_t id_base16, hipStream_t s) {
    hipLaunchKernelGGL(w8::k_region, dim3(n), dim3(64 * w8::NW), 0, s, g, b, id_base, id_base16);
}
// workspace is sized for the wider variant
int region_groups() { return w8::NW * w8::NG; }
int region_waves() { return w8::NW; }
int region_blocks() { return w8::NB; }
int region_ring() { return w8::RW; }
#else
void launch_region_w4(const Geom& g, const Buffers& b, int n, uint32_t id_base, uint32_t id_base16, hipStream_t s) {
    hipLaunchKernelGGL(w4::k_region, dim3(n), dim3(64 * w4::NW), 0, s, g, b, id_base, id_base16);
}
#endif

}  // namespace lsdhip
