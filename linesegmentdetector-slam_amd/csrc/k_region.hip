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
//   * seeds are handed out in chunks of 32; every seed has a record in an LDS ring (state byte, snapshot epoch, a result word);
//   * a wavefront first CLASSIFIES the seeds of its chunk, eight at a time: its eight 8-lane groups each grow one seed's region
//     out of a 16x16-pixel window in LDS.  Nine regions in ten stay below regThre pixels and end there (nothing to mark, :228);
//   * the others wait for a FULL evaluation by whichever wave is free: grow() with all 64 lanes (8 frontier pixels x 8
//     neighbours per batch out of an LDS tile cache of packed pixel words), rectangle, Refiner incl. its regrow, NFA
//     (eval_seed()); results that mark nothing are published in the ring, results that mark usedMap are stashed (record +
//     pixel list in one of the wave's result slots in HBM) and committed when the cursor reaches them;
//   * at its turn a result is valid if its seed is still unused and no MEMBER of its grown lists was banned since its
//     snapshot -- accepted pixels carry their line's epoch (epochmap), per-tile accept epochs and the boxes of recently
//     accepted lines are the first filters; an invalid result is evaluated again at the cursor, where everything earlier is
//     committed;
//   * commit = mark usedMap (code 3 + epoch, or 2), append the rectangle, advance the cursor (up to 64 records per step);
//   * how far ahead the waves work adapts (deeper while waves idle, shallower after a redo);
//   * a workgroup whose image is done lends its wavefronts to images that are still busy evaluating ("Help from other
//     workgroups" in the kernel): requests and answers through HBM, the owner's cursor commits.
// The committed sequence of decisions is therefore exactly the sequential one (DESIGN.md section 4 has the
// measurements behind every choice, DESIGN_NOTES.md the variants that were tried and dropped).
// Inside a wavefront the lanes cooperate where the order of evaluation can be kept:
//   * region growing: candidates are classified against the ESTIMATED sum vector of the region with a rigorous
//     margin, the exact fp64 angle sums (reference order = list order) are caught up lazily from the list;
//   * rectangle moments: products per lane, SERIAL accumulation in list order (bit-exact sums);
//   * NFA pixel count: the rectangle's columns are flattened with a wave prefix sum and counted
//     with ballot/popcount;
//   * usedMap marking: only the region's pixels are visited (the reference scans the whole image).
#include "lsd_internal.h"
#include "devmath.h"

// The file is compiled twice (Makefile): LSD_REGION_NW = 4 (two images per CU: the batch path, all 512 workgroups of the
// bench batch resident at once) and LSD_REGION_NW = 8 (one image per CU, 8 speculative wavefronts per image: lower
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

#ifndef LSD_REGION_WAVES_PER_SIMD
#define LSD_REGION_WAVES_PER_SIMD 2
#endif
#ifndef LSD_REGION_NS
#define LSD_REGION_NS 16
#endif
#ifndef LSD_REGION_KATTR
#define LSD_REGION_KATTR                 // experiments: extra attributes of the kernel (e.g. __attribute__((amdgpu_num_vgpr(168))))
#endif
#ifndef LSD_REGION_WAIT_SLEEP
#define LSD_REGION_WAIT_SLEEP 127      // x 64 clocks
#endif
#ifndef LSD_REGION_WATCHDOG
#define LSD_REGION_WATCHDOG 600000     // looks of one wave (a sleep of LSD_REGION_WAIT_SLEEP x 64 clocks, ~3.4 us, each) that found nothing to do
                                       // while the cursor, as that wave saw it, never moved
#endif
constexpr int NW = LSD_REGION_NW;        // wavefronts (concurrent speculative seeds) per image
constexpr int NS = LSD_REGION_NS;        // result slots per wave: seeds a wave may have evaluated ahead of the cursor
#ifndef LSD_REGION_LCAP
#define LSD_REGION_LCAP 512
#endif
#ifndef LSD_REGION_NT
#define LSD_REGION_NT 16
#endif
// The region list of a grow lives in LDS as a RING of the LCAP entries appended last (slot = index mod LCAP): the sweep that appends
// entries reads them again a frontier's width later, which for the thin structures of an occupancy map is a handful of entries,
// whatever the length of the region.  A list that outgrows the ring is also written through to HBM (`spill`, all entries, from the
// moment the ring would wrap), where the few readers of older entries find them (re-sweeps, the sums over the whole list).
constexpr int LCAP = LSD_REGION_LCAP;   // entries of the list ring (a power of two)
constexpr int LMASK = LCAP - 1;
constexpr int NT = LSD_REGION_NT;       // tile-cache slots per wave (8x8-pixel tiles of packed pixel words; a power of two; 16 measured as good as 32)
static_assert((NT & (NT - 1)) == 0 && NT >= 8 && LCAP >= 256 && (LCAP & (LCAP - 1)) == 0, "tile slots and list ring: powers of two");
constexpr int RING = 128;    // remembered bounding boxes of recently accepted lines
constexpr int kSetMax = 255;           // certified sets per image and launch (labels 1 .. kSetMax)
constexpr int kSetMinPixels = 64;      // ... of at least this many pixels
constexpr uint32_t kSetPending = 0x80000000u;
// The label of a growable pixel (in its epochmap word): bit 31 | the launch's tag << 8 | set number.  Accept epochs (small integers) and
// the labels of earlier launches in the same buffers never look like one of THIS launch (the tag is the run number, as for the stamps).
__device__ __forceinline__ uint32_t label_make(uint32_t tag, uint32_t id) { return 0x80000000u | (tag << 8) | id; }
__device__ __forceinline__ uint32_t label_set(uint32_t tag, uint32_t word) { return (word >> 8) == (0x800000u | tag) ? (word & 0xffu) : 0u; }

struct Rec {  // structRec, myLSD.h:80-93 (+ pk = number of halvings of p, indexes the host log tables)
    double x1, y1, x2, y2, wid, cX, cY, deg, dx, dy, p, prec;
    int pk;
};

// Mutable per-wave state.  It lives in LDS (not in registers) so that the out-of-line stages below can take the
// context by value and still share it; none of it is touched inside the inner loops.
struct WState {
    uint32_t cur_id;     // stamp of the current grow (id_base + running number)
    int gnum;            // size of the last grow (grow order)
    int has_copy;        // gcopy holds the grow-order list (RegionRadiusReducer reordered lst)
    int tm_pending;      // member masks of evicted tiles stored to HBM since the last fence
    int cache_epoch;     // accept epoch the tile cache was (re)started at; -1: empty
    int members_cached;  // the cache may hold member bits of the last grow
    int ex_upto;         // exact angle sums of the last grow, caught up lazily in list order (myLSD.cpp:545-546)
    double ex_sin, ex_cos;
    Rec rec;             // the rectangle of the region being evaluated
};

struct RCtx {
    int w, h, lane, wave;
    const double* mag;
    const double* deg;
    uint32_t* pw;        // packed pixel words: fp32 angle | usedMap code (shared by the workgroup)
    uint32_t* epochmap;  // accept epoch of code-3 pixels; for growable pixels (code 0 / 2) the LABEL of the certified set they belong to (0: none)
    uint32_t ltag;       // the launch's label tag (see label_make)
    uint32_t* sets;      // this image's certified sets (see "Certified uniform sets" below): [kSetMax + 1] sizes, 0 = dead / unused; null for a helper
    uint32_t* tep;       // per 8x8-pixel tile: epoch + 1 of the latest accepted line with a pixel in it (0: none)
    uint32_t* tmask;     // this wave's member masks of evicted tiles: 4 words per 8x8 tile (grow id, -, 64 member bits)
    uint32_t* spill;
    uint32_t* gcopy;
    float4* meta;        // HBM [mcap]: (unit sum vector, sin of the smallest slack) of the last full test of a list entry, see grow()
    int mcap;
    const double2* sc;   // (sin, cos)(deg)
    int tilesX;
    uint32_t id_base;
    uint32_t id_budget;  // grows a wave may number before it has to clear its member masks (< 2^20: the next run's ids start there)
    double logNT;
    const double* lgamma;
    int lg_count;
    const double* ptab;
    uint32_t* wslist;    // this wave's result slots: [NS][gcap] list entries
    int gcap;
    int llo;             // entries [llo, n) of the current region list are in the LDS ring, entries below in `spill` (grow() keeps g_ctx[wave].llo current)
};

enum { ST_GROW = 0, ST_GROWN, ST_NFA, ST_RRR, ST_RRRPASS, ST_SENT, ST_OOB, ST_TREFILL, ST_TOTAL, ST_TGROW, ST_TRECT, ST_TNFA,
       ST_TMARK, ST_SMALLBAIL, ST_WNOSLOT, ST_SEEDS, ST_EXACT, ST_WRING, ST_BATCHES, ST_TTILES, ST_REDO, ST_DISCARD,
       ST_WAIT, ST_SMALLSTEPS, ST_SLOW, ST_TEVAL, ST_TSUMS, ST_TREFINE, ST_TSMALL, ST_TSELECT, ST_TCOMMIT, ST_WNOSEED,
       ST_DEPTHUP, ST_DEPTHDN, ST_DEPTHEND, ST_MINNFA, ST_MINGAP, ST_XEXP, ST_XHELP, ST_TIES, ST_NFASLOW, ST_SETHIT, ST_SETNEW, ST_NFACNT, ST_NFAITER, ST_COUNT };
static_assert(ST_TOTAL == kStatTotalWord, "lsd_last_region_cycles reads this word");
static_assert(ST_TIES == kStatTiesWord, "lsd_last_sensitivity reads this word");
// STAT: the few per-region counters the parity tests and the bench read (always on).  DSTAT / NOW(): per-batch counters and
// s_memtime stopwatches of the developer build (make STATS=1): they cost ~10 % of the kernel, so the product build has none.
// (every active lane adds the same value to the same word -- no lane-0 branch: a lane-dependent branch whose join block
//  coincides with a join of wave-uniform control flow makes the compiler treat the uniform loop state as divergent)
#ifdef LSD_REGION_STATS
constexpr int kStatSlots = ST_COUNT;
__device__ constexpr int sslot(int i) { return i; }
#else
// the product build keeps the always-on counters only (LDS is the scarce resource of this kernel)
constexpr int kStatSlots = 19;
__device__ constexpr int sslot(int i) {
    return i == ST_GROW ? 0 : i == ST_GROWN ? 1 : i == ST_NFA ? 2 : i == ST_RRR ? 3 : i == ST_RRRPASS ? 4 : i == ST_SENT ? 5 : i == ST_OOB ? 6 :
           i == ST_TOTAL ? 7 : i == ST_SEEDS ? 8 : i == ST_REDO ? 9 : i == ST_DISCARD ? 10 : i == ST_MINNFA ? 12 : i == ST_MINGAP ? 13 : i == ST_XEXP ? 14 : i == ST_XHELP ? 15 : i == ST_SETHIT ? 16 : i == ST_SETNEW ? 17 : i == ST_TIES ? 18 : 11;
}
#endif
#define STAT(i, v) do { g_stat[c.wave][sslot(i)] += (unsigned long long)(v); } while (0)
// ... and two running maxima (every lane the same value): the smallest |logNFA| RectangleImprover has compared with 0, and the
// smallest non-zero difference between two NFA values it has compared with each other, both kept as kInfBits - bit pattern so
// that the zero-initialised counters work with max (tests/test_parity_gpu.py::test_nfa_decisions_are_far_from_ties)
#define STATMAX(i, v) do { const unsigned long long n_ = (v); if (n_ > g_stat[c.wave][sslot(i)]) g_stat[c.wave][sslot(i)] = n_; } while (0)
constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;
// ... and ST_TIES: the number of DECISIONS this image's evaluations took within the noise of the reference's libm (lsd_last_sensitivity,
// include/lsd_hip.h).  The reference's accept / reject decisions hang on glibc's sin / cos / atan2 / exp / log10 / pow, which differ from the
// correctly rounded values computed here by at most one ulp.  A decision "a < b" whose operands are closer than what those ulps can
// move them could come out differently under another libm; each such decision adds one.  0 for an image: every libm within one ulp
// yields the same decisions, hence the same usedMap and lines.  The bounds (upper bounds of the operands' noise, generous: a false
// count costs nothing but information):
//   kTieAng    angles: regDeg = atan2(sum sin, sum cos) of n libm terms -> (n / |V|) 6e-16 + 1e-15 (grow()'s exact test adds n / |V|)
//   kTieFlip   OrientationGetter's comparison of the inertia angle with regDeg (:655-665) and Refiner's wraps
//   kTieRel    the density of a rectangle against denThre, distances against the rectangle's width / Reducer's radius (relative)
//   kTieCoord  a rectangle edge against a pixel row / column (:973-1004), relative: a corner is c + t (dx, dy) with t up to the rectangle's
//              length and (dx, dy) a few ulps of sin / cos off, so it moves by kTieCoord (|c| + length); an edge's height in a column by
//              that times (1 + |slope|) -- the END edges of a rectangle that is almost axis-parallel are steep
// A rectangle whose direction is EXACTLY axis-parallel (min(|dx|, |dy|) < 1e-15: inertiaDeg is 0, pi or +-pi/2 to the last bit, which
// every libm returns alike, and a cosine of 6e-17 moves nothing) has libm-independent coordinates: its exact ties -- edges on pixel
// rows are the rule there -- are not counted.
constexpr double kTieAng = 1e-15, kTieFlip = 1e-13, kTieRel = 1e-12, kTieCoord = 4e-15;
// (developer: -DLSD_TIE_SITES makes the counter a decimal record of WHERE the ties are: three digits per site, see the call sites)
#ifdef LSD_TIE_SITES
__device__ constexpr unsigned long long tie_weight(int site) { unsigned long long w = 1; for (int i = 0; i < site; i++) w *= 1000ull; return w; }
#define TIE_UNIT(site) tie_weight(site)
#else
#define TIE_UNIT(site) 1ull
#endif
#define TIES_AT(site, v) STAT(ST_TIES, (unsigned long long)(v) * TIE_UNIT(site))
enum { TS_GROW = 0, TS_FLIP, TS_DENS, TS_DIST, TS_EDGE, TS_ALIGN, TS_NFA };
__device__ __forceinline__ bool axis_exact(double dx, double dy) { return fmin(fabs(dx), fabs(dy)) < 1e-15; }
#ifdef LSD_REGION_STATS
#define DSTAT(i, v) STAT(i, v)
#define NOW() ((long long)__builtin_amdgcn_s_memtime())
#else
#define DSTAT(i, v) do { } while (0)
#define NOW() 0ll
#endif
// developer experiment (with LSD_REGION_STATS): the time of one grow() batch by segment, in the counters of the per-stage stopwatches
// (rect: entry -> neighbour words read; nfa: -> classified; mark: -> accepted; refine: -> worklist done; sums: between batches)
#ifdef LSD_REGION_BATCHPROF
#define BSTAT(i, v) DSTAT(i, v)
#define PSTAT(i, v) do { } while (0)
#else
#define BSTAT(i, v) do { } while (0)
#define PSTAT(i, v) DSTAT(i, v)
#endif

// Per-wave LDS storage.  Declared at namespace scope (not inside the kernel) so that the out-of-line stages address it
// as LDS (ds_ instructions) instead of through generic pointers carried in the context (flat_ instructions).
// One arena of 32-bit words per wave, used in two ways.  A full evaluation: [list ring, LCAP words (packed y<<16 | x)][the sweep
// worklist, WCAP 16-bit entries + a dummy slot for predicated stores: list indices of the entries that still have a growable
// neighbour; the next sweep's worklist is written IN PLACE behind the read cursor][tile cache, NT x 64 words: (fp32 angle & ~3) |
// member << 1 | banned].  The small-region grower (seed loop): [eight 16x16-pixel windows][eight lists of SCAP entries] from the
// start of the arena -- nothing of a full evaluation survives it (tw_small in the seed loop).
constexpr int SCAP = 16;                  // list entries of a small-region group
constexpr int kSmallWords = 8 * 256 + 8 * SCAP;
#ifndef LSD_REGION_WLW
#define LSD_REGION_WLW (kSmallWords - LCAP - NT * 64 >= 256 ? kSmallWords - LCAP - NT * 64 : 256)
#endif
constexpr int WLW = (LSD_REGION_WLW + 3) & ~3;               // words of the worklist (16-byte multiple: windows and tiles are written as uint4)
constexpr int WCAP = 2 * WLW - 2;                            // its entries; [WCAP]: the dummy slot
constexpr int kTwOff = LCAP + WLW;
constexpr int kArenaWords = kTwOff + NT * 64 > kSmallWords ? kTwOff + NT * 64 : kSmallWords;
#ifdef LSD_REGION_DYN_ARENA
constexpr unsigned kDynLds = NW * kArenaWords * 4;
#else
constexpr unsigned kDynLds = 0;
#endif
constexpr int kMvCap = kArenaWords - LCAP - 1;               // RegionRadiusReducer's scratch: worklist + tile cache (+ a dummy slot)
static_assert(WLW >= 192, "the NFA's column scan keeps 3 x 64 ints in the worklist's place");
#ifdef LSD_REGION_DYN_ARENA
// The arenas as dynamic LDS (the launch passes NW * kArenaWords * 4 bytes): the compiler then sees ~16 KB of static LDS and accepts a
// register budget for four wavefronts per SIMD although the 50 KB a workgroup really uses admit three workgroups per CU -- the fourth
// wavefront slot of every SIMD (128 registers) stays free for the kernels of other steps (K1, K2, K3, K5) that run beside this one.
extern __shared__ __attribute__((aligned(16))) uint32_t g_arena_dyn[];
#define G_ARENA(w) (g_arena_dyn + (w) * kArenaWords)
#else
__shared__ __attribute__((aligned(16))) uint32_t g_arena[NW][kArenaWords];
#define G_ARENA(w) (&g_arena[w][0])
#endif
#define G_LST(w) (G_ARENA(w))
#define G_WL(w) (reinterpret_cast<uint16_t*>(G_ARENA(w) + LCAP))
#define G_TW(w) (G_ARENA(w) + kTwOff)
__shared__ int g_ttag[NW][NT];
__shared__ unsigned long long g_stat[NW][kStatSlots];      // per-wave counters (see ST_* above); kept out of registers
__shared__ WState g_ws[NW];
__shared__ RCtx g_ctx[NW];                                // the wave's context: the out-of-line stages get the wave number and read it here
                                                          // (a struct passed by value travels through scratch memory at every call)
__shared__ double g_tol0[3];                              // the global tolerance (degThre) with its sine and cosine: every first grow uses it
__shared__ double g_acc[NW][32 * 4];                      // staging of the serial (bit-exact) sums: 32 list elements x up to 4 terms
// what eval_seed() leaves for its caller (the rectangle itself stays in g_ws[wave].rec)
struct EvalOut {
    int skip, outcome, num, num0, rec_pk;
    int x0, y0, x1, y1;      // box of the pixels of the grown lists (speculative evaluations only)
    int n1, n2, precise;     // sizes of the first grow and of Refiner's regrow kept in the slot (precise == 0: not kept)
    int m_off, mcnt, redo;   // where the pixels to mark sit in the slot; redo: the result does not fit a slot
    int setid;               // != 0: the result was taken from certified set `setid` without growing anything (outcome 1)
    int cert;                // 1: this evaluation went the way every seed of a uniform set goes (see certify_set): its first list may found a set
    double logNFA;
};
__shared__ EvalOut g_eo[NW];
__shared__ double g_par[4];                               // degThre, regThre, aliPro, denThre of the launch (Geom)

// The reference's sums over a region (moments, angle sums, Refiner's statistics) are plain left-to-right fp64 additions, and
// their rounding decides accept/reject ties, so they are added in exactly that order: the lanes compute the terms of 32 list
// elements at a time and stage them in LDS, then lane j (j < 4) adds term j of the elements one after the other.  (One
// ds_read + one v_add per element and sum, all sums at once, instead of broadcasting every term to every lane.)
__device__ __forceinline__ void stage4(int wave, int lane, int half, double t0, double t1, double t2, double t3) {
    if ((lane >> 5) == half) {
        double* q = &g_acc[wave][(lane & 31) * 4];
        q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3;
    }
}
__device__ __forceinline__ double acc32(int wave, int lane, int cnt, double S) {   // cnt (wave-uniform) <= 32 staged elements
    const double* q = &g_acc[wave][lane & 3];
    int e = 0;
    for (; e + 8 <= cnt; e += 8) {
        S += q[(e + 0) * 4]; S += q[(e + 1) * 4]; S += q[(e + 2) * 4]; S += q[(e + 3) * 4];
        S += q[(e + 4) * 4]; S += q[(e + 5) * 4]; S += q[(e + 6) * 4]; S += q[(e + 7) * 4];
    }
    for (; e < cnt; e++) S += q[e * 4];
    return S;
}

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// a load that does not stop at the CU's vector cache: for words other wavefronts change with ATOMICS (performed in L2, they leave a
// stale line in the L1 behind; plain stores of the same CU do not)
__device__ __forceinline__ uint32_t ld_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
// words shared with OTHER workgroups (the help protocol of the seed loop): written and read in L2, ordered by agent-scope fences
__device__ __forceinline__ void st_l2(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void agent_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// Function arguments arrive in vector registers even when they are the same in every lane; the inner loop wants them on
// the scalar unit (scalar compares and branches, SGPR-base addressing of global memory with 32-bit lane offsets).
#define AS1 __attribute__((address_space(1)))
typedef float nf4 __attribute__((ext_vector_type(4)));       // (HIP's float4 class cannot be reached through an address-space pointer)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <class T>
__device__ __forceinline__ AS1 T* uglobal(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (AS1 T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int mbcnt(unsigned long long m) {   // number of set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// minimum over the 8 lanes of a group (lane >> 3), every lane gets it: three DPP steps, no LDS traffic
__device__ __forceinline__ float min8(float v) {
    int t = __float_as_int(v);
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
    t = __float_as_int(v);
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
    t = __float_as_int(v);
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0x141, 0xf, 0xf, false)));  // row_half_mirror
    return v;
}

// sum over the 8 lanes of a group, every lane gets it (used where at most one lane holds a non-zero value: the sum is that value)
__device__ __forceinline__ float sum8(float v) {
    int t = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    t = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    t = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0x141, 0xf, 0xf, false));  // row_half_mirror
    return v;
}

__device__ __forceinline__ double rl(double v, int l) {  // broadcast lane l (l wave-uniform)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rlf(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ uint32_t pack_xy(int x, int y) { return ((uint32_t)y << 16) | (uint32_t)x; }
__device__ __forceinline__ uint32_t lget(const RCtx& c, int i) { return i >= c.llo ? G_LST(c.wave)[i & LMASK] : c.spill[i]; }
__device__ __forceinline__ void lset(const RCtx& c, int i, uint32_t v) {           // (after the grow: every entry has one home)
    if (i >= c.llo) G_LST(c.wave)[i & LMASK] = v; else c.spill[i] = v;
}
__device__ __forceinline__ double angle_diff(double a, double b) {  // myLSD.cpp:540-542 / :1009-1011
    double d = fabs(a - b);
    if (d > kPi * 3 / 2.0) d = fabs(d - 2.0 * kPi);
    return d;
}

// ---------------------------------------------------------------------------------------------
// LDS tile cache: 8x8-pixel tiles of packed pixel words, NT slots, slot = (tx + 5 ty) mod NT (rows, columns
// and diagonals of tiles spread over all slots).  RegionGrower reads its 3x3 neighbourhoods from here, so a
// batch costs LDS latency instead of dependent HBM round trips.  A cached word is the pixel's pw with the code
// replaced by two flags: bit 0 = banned (code 1 or 3), bit 1 = member of the current grow (curMap).  The member
// flags live in the cache; a tile that is evicted with members leaves them in HBM as a 64-bit mask tagged with the
// grow's id (`tmask`, 16 bytes per tile and wave), and takes them back when it returns.  (Until round 4 every accepted
// pixel was stamped in a 4-byte-per-pixel map instead: a scattered store per pixel and a fence in front of most tile
// fetches.)  The cache survives from seed to seed while no line is accepted in the image (a tile fetched before an
// accept could miss a ban that the snapshot of a later seed no longer flags).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_slot(int tx, int ty) { return (tx + 5 * ty) & (NT - 1); }

// Makes the tiles of every lane with need==true resident.  Returns false when two needed tiles map
// to the same slot (the caller retries with a smaller batch; a single 3x3 neighbourhood never conflicts).
// A tile's tag is (tile row << 16 | tile column).
__device__ __forceinline__ int tile_key(int tx, int ty) { return (ty << 16) | tx; }
__device__ __forceinline__ uint32_t tm_index(const RCtx& c, int key) { return 4u * (uint32_t)((key >> 16) * c.tilesX + (key & 0xffff)); }
// curMap of a pixel whose tile is NOT in the cache (the stages after RegionRadiusReducer, which empties the cache into tmask first):
// read past the L1, the reducer clears bits with atomics
__device__ __forceinline__ bool tm_member(const RCtx& c, int x, int y, uint32_t id) {
    const uint32_t* t = c.tmask + tm_index(c, tile_key(x >> 3, y >> 3));
    const int b = ((y & 7) << 3) | (x & 7);
    return __hip_atomic_load(&t[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == id &&
           ((__hip_atomic_load(&t[2 + (b >> 5)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (b & 31)) & 1u) != 0u;
}
__device__ __forceinline__ void tm_clear(const RCtx& c, int x, int y) {           // curMap(x, y) = 0
    uint32_t* t = c.tmask + tm_index(c, tile_key(x >> 3, y >> 3));
    const int b = ((y & 7) << 3) | (x & 7);
    atomicAnd(&t[2 + (b >> 5)], ~(1u << (b & 31)));
}
__device__ __forceinline__ bool ensure_tiles(const RCtx& c, bool need, int px, int py) {
    const int lane = c.lane, w = c.w, h = c.h, wave = c.wave;
    const int tx = px >> 3, ty = py >> 3;
    const int tile = need ? tile_key(tx, ty) : -1;
    const int slot = tile_slot(tx, ty);
    unsigned long long todo = ballot64(need & (g_ttag[wave][slot] != tile));
    if (!todo) return true;
    // conflict check over all needed tiles (resident ones included)
    {
        unsigned long long chk = ballot64(need);
        while (chk) {
            const int l = __builtin_ctzll(chk);
            const int T = __builtin_amdgcn_readlane(tile, l), S = __builtin_amdgcn_readlane(slot, l);
            if (ballot64(need & (slot == S) & (tile != T))) return false;
            chk &= ~ballot64(tile == T);
        }
    }
    [[maybe_unused]] const long long tt0 = NOW();
    if (__builtin_amdgcn_readfirstlane(g_ws[wave].tm_pending)) { wg_fence(); g_ws[wave].tm_pending = 0; }   // masks of tiles evicted earlier must have landed before one of them is read back
    const uint32_t id = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_ws[wave].cur_id);
    AS1 const uint32_t* const pw = uglobal(c.pw);
    AS1 uint32_t* const tm = uglobal(c.tmask);
    const int lx = lane & 7, ly = lane >> 3;
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
                todo &= ~ballot64(tile == T[j]);
                nt++;
            }
        }
        DSTAT(ST_WRING, nt);                               // (developer build: tiles fetched)
        // the tiles that make room leave their member flags in HBM (most have none: nothing is stored for them)
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < nt) {
                const int old = __builtin_amdgcn_readfirstlane(g_ttag[wave][S[j]]);
                const unsigned long long om = old != -1 ? ballot64((G_TW(wave)[S[j] * 64 + lane] & 2u) != 0u) : 0ull;
                if (om) {
                    if (lane == 0) {
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 rec = {id, 0u, (uint32_t)om, (uint32_t)(om >> 32)};
                        *reinterpret_cast<AS1 u32x4*>(tm + tm_index(c, old)) = rec;
                    }
                    g_ws[wave].tm_pending = 1;                 // (all lanes, same value)
                }
            }
        }
        uint32_t vw[4], vi[4], vm[4];
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            vw[j] = kPwStatic; vi[j] = 0u; vm[j] = 0u;            // outside the image: banned
            if (j < nt) {
                const int x = (T[j] & 0xffff) * 8 + lx, y = (T[j] >> 16) * 8 + ly;
                if ((x < w) & (y < h)) vw[j] = pw[(uint32_t)(y * w + x)];
                AS1 const uint32_t* t = tm + tm_index(c, T[j]);
                vi[j] = t[0]; vm[j] = t[2 + (lane >> 5)];          // (two addresses per tile for the whole wave)
            }
        }
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < nt) {
                const uint32_t mem = (vi[j] == id) ? ((vm[j] >> (lane & 31)) & 1u) : 0u;
                G_TW(wave)[S[j] * 64 + lane] = (vw[j] & ~3u) | (vw[j] & 1u) | (mem << 1);
                g_ttag[wave][S[j]] = T[j];                 // (all lanes, same value)
            }
        }
    }
    DSTAT(ST_TTILES, NOW() - tt0);
    return true;
}

__device__ __forceinline__ void invalidate_tiles(const RCtx& c) {
    if (c.lane < NT) g_ttag[c.wave][c.lane] = -1;
    g_ws[c.wave].members_cached = 0;
}
// Empties the cache into tmask: afterwards curMap of the current grow is in HBM in full (RegionRadiusReducer clears bits there, the
// marking stages read them there).
__device__ __forceinline__ void flush_tiles(const RCtx& c) {
    const uint32_t id = g_ws[c.wave].cur_id;
    for (int sl = 0; sl < NT; sl++) {
        const int old = __builtin_amdgcn_readfirstlane(g_ttag[c.wave][sl]);
        if (old == -1) continue;
        const unsigned long long om = ballot64((G_TW(c.wave)[sl * 64 + c.lane] & 2u) != 0u);
        if (om && c.lane == 0) {
            uint32_t* t = c.tmask + tm_index(c, old);
            t[0] = id; t[1] = 0u; t[2] = (uint32_t)om; t[3] = (uint32_t)(om >> 32);
        }
    }
    invalidate_tiles(c);
    wg_fence();
    g_ws[c.wave].tm_pending = 0;
}

// ---------------------------------------------------------------------------------------------
// The exact angle sums of the current region (sinDeg, cosDeg of RegionGrower, :545-546) over the list prefix
// [0, n): the reference adds cos/sin(deg) of every accepted pixel in the order of acceptance, which is the list
// order, so the sums can be caught up at any time from the list and the (sin, cos) map K2 wrote.
// ---------------------------------------------------------------------------------------------
// (out of line, like every per-region stage below: each gets the register file to itself, and the seed loop keeps only
//  what it needs across the calls; the context travels by value, the mutable state sits in LDS)
__device__ __noinline__ void exact_sums(int cw_, int n_) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane(c.wave), n = __builtin_amdgcn_readfirstlane(n_);
    const int from = __builtin_amdgcn_readfirstlane(g_ws[wave].ex_upto);
    if (from >= n) return;
    [[maybe_unused]] const long long t0 = NOW();
    const int lane = c.lane, w = c.w;
    double S = lane == 0 ? g_ws[wave].ex_cos : g_ws[wave].ex_sin;      // lane 0: cosDeg, lane 1: sinDeg (:545-546)
    for (int base = from; base < n; base += 64) {
        const int kx = base + lane;
        double vs = 0, vc = 0;
        if (kx < n) {
            const uint32_t pk = lget(c, kx);
            const double2 v = c.sc[(size_t)(pk >> 16) * w + (pk & 0xffffu)];
            vs = v.x; vc = v.y;
        }
        for (int half = 0; half < 2; half++) {
            const int cnt = min(32, n - base - 32 * half);
            if (cnt <= 0) break;
            stage4(wave, lane, half, vc, vs, 0.0, 0.0);
            S = acc32(wave, lane, cnt, S);
        }
    }
    if (lane == 0) { g_ws[wave].ex_cos = S; g_ws[wave].ex_upto = n; }
    if (lane == 1) g_ws[wave].ex_sin = S;
    PSTAT(ST_TSUMS, NOW() - t0);
}

// ---------------------------------------------------------------------------------------------
// RegionGrower, myLSD.cpp:491-590.  Leaves the region in c.lst (grow order) and returns its size; the angle
// sums are available through exact_sums() (the caller needs them only for regions that go on to the rectangle).
//
// The reference tests every candidate against regDeg = atan2(sinDeg, cosDeg) recomputed after each accepted
// pixel (:545-547), in list order / row-major neighbour order.  For tol < pi/2 "|regDeg - deg| (wrapped) < tol"
// is the circular distance between the candidate's direction u and the direction of the sum vector V, i.e.
// u.V > cos(tol) |V|.  A batch of 8 frontier pixels x 8 neighbours is classified at once in that form with fp32
// ESTIMATES of u (hardware sin/cos of the packed fp32 angle) and V (their running sum), and a rigorous margin:
//   eps_c  error of the estimated cosine: |u - u_true| <= kEpsU per vector, so V is off by <= n kEpsU
//   delta  largest turn of V while the up-to-m winners of this batch are accepted: each accepted unit vector lies
//          within tol of the current sum of norm L, so it turns it by at most sin(tol)/L
//   * cos > cos(tol) + delta sin(tol) + eps_c      : passes whatever happens earlier in the batch -> accepted in bulk
//   * cos < cos(tol) - delta (sin(tol)+delta) - eps_c : fails whatever happens                    -> ignored
//   * otherwise the batch is resolved pixel by pixel in reference order against the then-current estimate, and
//     against the correctly rounded angle of the exact sums when still too close to call.
// Every accept/reject decision is therefore the one the exact angle would give; the exact sums are accumulated in
// reference order (exact_sums()).  Larger tolerances (Refiner may ask for any) take the pixel-by-pixel path with the
// reference's own wrapped-difference test.
// Sweeps after the first revisit only entries that still had a non-member, non-banned neighbour
// (membership and bans only grow during one call, so the others cannot accept anything).
// ---------------------------------------------------------------------------------------------

// an upper bound of 1 / v for v >= 0.9 (v_rcp_f32 is good to 1 ulp; the margins it feeds are themselves upper bounds)
__device__ __forceinline__ float inv_ub(float v) { return __builtin_amdgcn_rcpf(v) * 1.000001f; }

constexpr float kEpsU = 6e-6f;     // >= |(cos, sin) estimate - exact| per accepted pixel: 2-bit truncation of the fp32 angle (1e-6) + v_sin/v_cos_f32
                                   //    (together <= 3e-6: tests/test_parity_gpu.py::test_fast_sincos_error_bound) + the fp32 partial sums of
                                   //    a batch (<= 64 terms: <= 2^-24 * 32 = 1.9e-6 per term); the running sums themselves are fp64
constexpr float kInv2Pi = 0.15915494309189535f;

__device__ __forceinline__ void fast_sincos(float a, float& s, float& co) {   // hardware sin/cos take revolutions
    const float r = a * kInv2Pi;
    s = __builtin_amdgcn_sinf(r);
    co = __builtin_amdgcn_cosf(r);
}

__device__ __noinline__ int grow(int cw_, int sx_, int sy_, double regDeg0_, double tol_) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int lane = c.lane;
    const double regDeg0 = uni(regDeg0_), tol = uni(tol_);
    const int w = uni(c.w), h = uni(c.h), wave = uni(c.wave), mcap = uni(c.mcap);
    const int sx = uni(sx_), sy = uni(sy_);
    AS1 nf4* const meta = (AS1 nf4*)uglobal(c.meta);
    c.w = w; c.h = h; c.wave = wave;                         // (what the helpers below read)
    [[maybe_unused]] const long long t0 = NOW();
    // curMap of the previous grow: drop its member flags from the cache (:519 starts from zeros)
    if (uni(g_ws[wave].members_cached)) {
        const int gprev = uni(g_ws[wave].gnum);
        if (uni(g_ws[wave].has_copy) || gprev > 4 * LCAP) invalidate_tiles(c);
        else {
            for (int k2 = lane; k2 < gprev; k2 += 64) {
                const uint32_t pk = lget(c, k2);
                const int x = (int)(pk & 0xffffu), y = (int)(pk >> 16);
                const int slot = tile_slot(x >> 3, y >> 3);
                if (g_ttag[wave][slot] == tile_key(x >> 3, y >> 3)) G_TW(wave)[slot * 64 + ((y & 7) << 3) + (x & 7)] &= ~2u;
            }
        }
    }
    uint32_t id = (uint32_t)uni((int)g_ws[wave].cur_id);
    if ((id - (uint32_t)uni((int)c.id_base)) >= (uint32_t)uni((int)c.id_budget)) {                      // the run's 2^20 stamp ids are used up: start over on clean stamps
        const uint32_t tmw = 4u * (uint32_t)(c.tilesX * ((h + 7) >> 3));
        for (uint32_t q = lane; q < tmw; q += 64) c.tmask[q] = 0u;
        wg_fence();
        id = c.id_base;
    }
    id = (uint32_t)uni((int)id + 1);                         // fresh curMap (:519)
    if (lane == 0) {
        WState& ws = g_ws[wave];
        ws.cur_id = id; ws.members_cached = 1; ws.has_copy = 0; ws.ex_upto = 0; ws.ex_sin = 0.0; ws.ex_cos = 0.0;
        g_ctx[wave].llo = 0;                                 // the new list starts inside the ring
    }
    c.llo = 0;
    AS1 uint32_t* const spill = uglobal(c.spill);
    ensure_tiles(c, lane == 0, sx, sy);
    double Ce, Se;                                           // estimated sum vector (fp64 accumulation of the fp32 unit vectors)
    {
        const int slot = tile_slot(sx >> 3, sy >> 3), ti = ((sy & 7) << 3) | (sx & 7);
        const uint32_t sw = G_TW(wave)[slot * 64 + ti];
        float s0, c0;
        fast_sincos(__uint_as_float(sw & ~3u), s0, c0);
        Ce = (double)c0; Se = (double)s0;
        if (lane == 0) {
            G_LST(wave)[0] = pack_xy(sx, sy);
            G_TW(wave)[slot * 64 + ti] = sw | 2u;            // :520
        }
    }
    int n = 1;
    bool wt = false;                                         // the list has outgrown the ring: entries are also written through to `spill`
    if (!(tol == tol)) {                                     // NaN tolerance (Refiner, :855): no test ever passes
        if (lane == 0) g_ws[wave].gnum = 1;
        STAT(ST_GROW, 1); STAT(ST_GROWN, 1);
        return 1;
    }
    const bool tol_small = tol < 1.5;                        // the circular-distance form applies, and accepted vectors never shorten the sum
    const float turn = (float)(tol < 1.1 ? tol : 1.1) * 1.0032f;   // >= sin(tol) resp. the asin(1/L)*L bound, x (|V| estimate / its lower bound)
    const float tolf_lo = (float)tol * 0.9999999f;           // <= tol
    float cos_tol, sin_tol;
    {
        double st, ct;
        if (tol == g_tol0[0]) { st = g_tol0[1]; ct = g_tol0[2]; }           // (wave-uniform)
        else sincos_g(tol_small ? tol : 1.0, st, ct);
        cos_tol = (float)ct; sin_tol = (float)st * 1.0000002f + 1e-7f;      // sin_tol >= sin(tol)
    }
    const int e = lane >> 3, k = lane & 7;
    const int kk = k + (k >= 4);                             // 3x3 neighbourhood, row-major, centre skipped (:533-534)
    const int ox = kk % 3 - 1, oy = kk / 3 - 1;
    int wl_cnt = 0;                                          // entries of this sweep's worklist (sweep >= 2)
    bool filter = true;                                      // false once the list outgrew the worklist
    // Re-sweeps: an entry whose remaining candidates all failed by more than the sum vector has turned since cannot
    // accept anything now either (membership and bans only grow); it is carried over to the next worklist without
    // touching its neighbourhood.  meta[entry] = (unit sum vector its candidates were compared with, sine of the
    // smallest "distance - tol" among the candidates left), checked 64 entries at a time.
    unsigned long long flt_need = 0;                         // chunk [flt_base, flt_base + 64) of the worklist: entries to test in full
    int flt_base = 0;
    bool flt_valid = false;
    int nxt_cnt = 0;                                         // entries of the next sweep's worklist
    [[maybe_unused]] long long bt_last = NOW();
    // One batch: up to 8 list entries (cnt of them, entry e of the batch = list index eidx in its 8 lanes) x 8 neighbours.
    // Returns the number of entries it dealt with (1 instead of cnt when their tiles collide in the cache).
    auto batch = [&](int cnt, const int eidx, const bool direct, const float Cf, const float Sf, const float rV, const float Vn,
                     const float nrat) -> int {
        [[maybe_unused]] const long long bt0 = NOW();
        BSTAT(ST_TSUMS, bt0 - bt_last);
        bool valid = e < cnt;
        // (entries past n: harmless garbage, masked by valid.)  The LDS part of the list is read unconditionally and the HBM part
        // in a block of its own that also waits for it: a load whose register is still pending at the join would make the
        // compiler put an s_waitcnt vmcnt(0) in front of every batch, and that waits for the stamp stores of the batch before.
        uint32_t pk = G_LST(wave)[eidx & LMASK];
        if (!direct) {                                       // entries that have left the ring (wave-uniform: only a list longer than the ring has any)
            const int lo = wt ? n - LCAP : 0;
            if (ballot64(valid & (eidx < lo))) {
                uint32_t t = pk;
                if (valid & (eidx < lo)) t = spill[(uint32_t)eidx];
                asm volatile("; spilled list entry %0" :: "v"(t));
                pk = t;
            }
        }
        const int nx = (int)(pk & 0xffffu) + ox, ny = (int)(pk >> 16) + oy;
        bool inb = valid & ((unsigned)nx < (unsigned)w) & ((unsigned)ny < (unsigned)h);   // :536 (plain &: no short-circuit branches)
        const int tx = nx >> 3, ty = ny >> 3;
        const int slot = tile_slot(tx, ty);
        const int cell = (slot << 6) | ((ny & 7) << 3) | (nx & 7);        // (in range even for !inb lanes)
        uint32_t word_r = G_TW(wave)[cell];
        int tagv = g_ttag[wave][slot];
        // (both reads in flight before the tag is looked at: left alone the compiler moves the word's read behind the check -- it is read
        //  again after a tile fetch anyway -- and a batch pays one more LDS round trip)
        asm volatile("; tile word %0 and tag %1" : "+v"(word_r), "+v"(tagv));
        if (ballot64(inb & (tagv != tile_key(tx, ty)))) {
            if (!ensure_tiles(c, inb, nx, ny)) {         // slot conflict: one entry at a time
                cnt = 1;
                valid = e < cnt;
                inb = inb && valid;
                ensure_tiles(c, inb, nx, ny);
            }
            word_r = G_TW(wave)[cell];
        }
        const bool cand = inb & ((word_r & 3u) == 0u);   // :537: not in curMap, not banned (2 is growable, Q5)
        const unsigned long long candm = ballot64(cand);
        DSTAT(ST_BATCHES, 1);
        [[maybe_unused]] const long long bt1 = NOW();
        BSTAT(ST_TRECT, bt1 - bt0);
        [[maybe_unused]] long long bt2 = bt1, bt3 = bt1;
        if (candm) {
            const int q = ny * w + nx;
            const float af = __uint_as_float(word_r & ~3u);
            float sf, cf;
            fast_sincos(af, sf, cf);
            // first occurrence of every candidate pixel: a lane is a repeat iff an EARLIER entry of the batch
            // has the pixel in its 3x3 neighbourhood (that entry's lane for it comes first in reference order)
            bool winner = cand;
            if (cnt > 1) {
                const int ex0 = (int)(pk & 0xffffu), ey0 = (int)(pk >> 16);
                for (int e2 = 0; e2 + 1 < cnt; e2++) {
                    const int px2 = __builtin_amdgcn_readlane(ex0, e2 * 8), py2 = __builtin_amdgcn_readlane(ey0, e2 * 8);
                    winner = winner & !((e > e2) & ((unsigned)(nx - px2 + 1) <= 2u) & ((unsigned)(ny - py2 + 1) <= 2u));
                }
            }
            unsigned long long gone = 0;                 // every lane whose pixel became a member in this batch
            bool bulk = false;
            float dot = 0.0f;
            bt2 = NOW();
            BSTAT(ST_TNFA, bt2 - bt1);
            if (tol_small) {
                const float m = (float)__builtin_popcountll(ballot64(winner));
                dot = __builtin_fmaf(cf, Cf, sf * Sf);                            // ~ cos(distance) * |V|
                const float eps_c = kEpsU * (1.0f + 2.1f * nrat) + 5e-6f;         // incl. the error of Vn
                // a candidate is compared with the sum after the winners BEFORE it (at most m - 1) have been added, each turning it by
                // at most turn / |V| (|V| >= 1 here: accepted vectors only lengthen the sum); a lone candidate sees no drift at all
                const float delta = (m - 1.0f) * turn * rV + 1e-7f;
                const float t_hi = delta <= tolf_lo ? (cos_tol + delta * sin_tol + eps_c) * Vn : 3e38f;
                const float t_lo = delta <= 1.6f ? (cos_tol - delta * fminf(1.0f, sin_tol + delta) - eps_c) * Vn : -3e38f;
                const unsigned long long pcm = ballot64(cand & (dot > t_hi));     // candidates that clearly pass
                const unsigned long long failm = ballot64(cand & (dot < t_lo));   // ... clearly fail
                bulk = (candm & ~(pcm | failm)) == 0ull;
                if (bulk && pcm) {
                    const unsigned long long P = ballot64(winner) & pcm;
                    const int np = __builtin_popcountll(P);
                    if ((P >> lane) & 1ull) {
                        const int idx = n + mbcnt(P);
                        G_TW(wave)[cell] = word_r | 2u;                           // :549
                        G_LST(wave)[idx & LMASK] = pack_xy(nx, ny);               // :551-556
                        if (wt) spill[(uint32_t)idx] = pack_xy(nx, ny);
                    }
                    float ps = 0.0f, pc2 = 0.0f;
                    unsigned long long todo = P;
                    while (todo) {
                        const int l = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        pc2 += rlf(cf, l); ps += rlf(sf, l);
                    }
                    Ce += (double)pc2; Se += (double)ps;
                    n += np;
                    flt_valid = false;                   // the region angle moved
                    gone = pcm;
                }
            }
            if (!bulk) {
                // ---- pixel by pixel, in reference order (lane order) ----
                unsigned long long todo = candm;
                while (todo) {
                    int l, decided = -1;                 // 1 take, 0 reject, -1 exact test needed
                    if (tol_small) {
                        // All candidates still to come, against the estimate as it stands: the ones that clearly fail BEFORE the
                        // first one that does not are decided for good (nothing is accepted in between, so this is the estimate
                        // they meet at their turn) -- the loop runs once per accepted pixel, not once per candidate.
                        const float Cg = (float)Ce, Sg = (float)Se;
                        const float Vg = __builtin_amdgcn_sqrtf(Cg * Cg + Sg * Sg) * 1.000001f;
                        const float nr = (float)n * inv_ub(fmaxf(Vg, 1e-3f));
                        const float ec = kEpsU * (1.0f + 2.1f * nr) + 5e-6f;
                        const float d1 = cf * Cg + sf * Sg;
                        const unsigned long long failm1 = ballot64(d1 < (cos_tol - ec) * Vg);
                        const unsigned long long passm1 = ballot64(d1 > (cos_tol + ec) * Vg);
                        const unsigned long long nf = todo & ~(failm1 | gone);
                        if (!nf) break;                  // everything left fails
                        l = __builtin_ctzll(nf);
                        todo &= ~((2ull << l) - 1ull);   // (l < 63 or the mask is all ones: 2 << 63 wraps to 0)
                        if ((passm1 >> l) & 1ull) decided = 1;
                    } else {
                        l = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        if ((gone >> l) & 1ull) continue;    // the same pixel was accepted a moment ago
                        const float Cg = (float)Ce, Sg = (float)Se;
                        const float Vg = __builtin_amdgcn_sqrtf(Cg * Cg + Sg * Sg) * 1.000001f;
                        const float nr = (float)n * inv_ub(fmaxf(Vg, 1e-3f));
                        if (Vg > 0.05f) {
                            // any tolerance: the reference's wrapped difference (:540-542) of estimates, exact when near a discontinuity
                            const double R = n == 1 ? regDeg0 : atan2(Se, Ce);
                            const double er = (n == 1 ? 0.0 : (double)(1.05f * kEpsU * nr) + 1e-7) + 1.2e-6;   // + the packed angle's own error
                            const double al = (double)rlf(af, l);
                            const double rw = fabs(R - al);
                            const double df = rw > kPi * 3 / 2.0 ? fabs(rw - 2.0 * kPi) : rw;
                            if (!(fabs(R) > kPi - er || fabs(df - tol) <= er || fabs(rw - kPi * 3 / 2.0) <= er)) decided = df < tol ? 1 : 0;
                        }
                        decided = uni(decided);          // (the same in every lane; computed on the vector unit)
                    }
                    const float cl = rlf(cf, l), sl = rlf(sf, l);
                    const int ql = __builtin_amdgcn_readlane(q, l);
                    if (decided < 0) {
                        g_ctx[wave].llo = wt ? max(n - LCAP, 0) : 0;              // (all lanes, same value: what exact_sums()'s reads go by)
                        exact_sums(c.wave, n);
                        const double R = n == 1 ? regDeg0 : atan2_g(g_ws[wave].ex_sin, g_ws[wave].ex_cos);   // :547 (regDeg is the seed's angle until the first accept)
                        DSTAT(ST_EXACT, 1);
                        const double dq = c.deg[ql], adq = angle_diff(R, dq);
                        decided = uni(adq < tol ? 1 : 0);                                   // :540-543
                        {   // within the libm's noise of the tolerance, or of the wrap at 3 pi / 2?  (tol == 0 and equal angles: an exact 0 < 0 on any libm)
                            const double es = g_ws[wave].ex_sin, ec = g_ws[wave].ex_cos;
                            const double nz = kTieAng * (1.0 + (n == 1 ? 0.0 : (double)n / fmax(sqrt(es * es + ec * ec), 1e-300)));
                            // (the wrap at 3 pi / 2 (:541) maps a difference that fails to one of pi / 2, which fails as well unless tol reaches a quarter turn)
                            // (angles that are 0, +-pi/2 or +-pi to the last bit -- axis-parallel walls -- are the same constants on every libm)
                            const bool quarters = (R == 0.0 || fabs(R) == kPi / 2.0 || fabs(R) == kPi) && (dq == 0.0 || fabs(dq) == kPi / 2.0 || fabs(dq) == kPi);
                            const bool tie = !quarters && ((fabs(adq - tol) <= nz && !(tol == 0.0 && adq == 0.0)) || (tol > 1.5 && fabs(fabs(R - dq) - kPi * 3 / 2.0) <= nz));
                            TIES_AT(TS_GROW, uni(tie ? 1 : 0));
                        }
                    }
                    if (decided == 1) {
                        if (lane == l) {
                            G_TW(wave)[cell] = word_r | 2u;                       // :549
                            G_LST(wave)[n & LMASK] = pack_xy(nx, ny);             // :551-556
                            if (wt) spill[(uint32_t)n] = pack_xy(nx, ny);
                        }
                        Ce += (double)cl; Se += (double)sl;
                        n++;
                        flt_valid = false;
                        gone |= ballot64(cand & (q == ql));
                    }
                }
            }
            bt3 = NOW();
            BSTAT(ST_TMARK, bt3 - bt2);
            // entries that still have a growable non-member neighbour go to the next sweep's worklist
            const unsigned long long left = candm & ~gone;
            if (filter && left) {
                const bool has = valid & (((left >> (8 * e)) & 0xffull) != 0ull);
                if (tol_small) {
                    // slack of this entry's remaining candidates: sin(distance - tol), from the start-of-batch estimate;
                    // after a pixel-by-pixel batch the sum has moved in between, so no slack is claimed (0 = test in full next time)
                    float sg = 2.0f;
                    if ((left >> lane) & 1ull) {
                        if (bulk) {
                            const float ct = fminf(fmaxf(dot * rV, -1.0f), 1.0f);
                            const float st = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - ct * ct));
                            const float cs_ = ct * cos_tol + st * sin_tol;            // cos(distance - tol)
                            sg = cs_ <= 0.0f ? 1.0f : st * cos_tol - ct * sin_tol;    // sin(distance - tol), 1 beyond a quarter turn
                        } else sg = 0.0f;
                    }
                    sg = min8(sg);
                    if (has && k == 0 && eidx < mcap)
                        meta[(uint32_t)eidx] = nf4{Cf * rV, Sf * rV, sg - 1.2e-4f - 8.0f * kEpsU * nrat, 0.0f};
                }
                const unsigned long long hm = ballot64(has & (k == 0));
                const int add = __builtin_popcountll(hm);
                const bool room = nxt_cnt + add <= WCAP && n <= 65535;
                // (in place: the next worklist never passes the read cursor -- every entry written was read before, in this batch or earlier)
                G_WL(wave)[(room & has & (k == 0)) ? nxt_cnt + mbcnt(hm) : WCAP] = (uint16_t)eidx;   // (no branch: dummy slot)
                filter = filter && room;
                nxt_cnt += room ? add : 0;
            }
            BSTAT(ST_TREFINE, NOW() - bt3);
        }
        bt_last = NOW();
        return cnt;
    };
    // The loop state is wave-uniform by construction, but the compiler's divergence analysis gives up on it as soon as the
    // join of some lane-conditional store coincides with a join of the uniform control flow (which its CFG simplifications
    // produce at will) -- and then runs the whole loop as divergent code on the vector unit.  Saying it again at the top of
    // every iteration costs nothing where the analysis already knows, and keeps the control flow scalar where it does not.
#define GROW_ESTIMATE()                                                                                                   \
    n = uni(n); nxt_cnt = uni(nxt_cnt); filter = uni((int)filter) != 0; wt = uni((int)wt) != 0;                          \
    if (!wt && n + 64 > LCAP) {             /* the batch to come may wrap the ring: from here on the list is in HBM as well */ \
        for (int k2 = lane; k2 < n; k2 += 64) spill[(uint32_t)k2] = G_LST(wave)[k2];                                      \
        wg_fence();                                                                                                       \
        wt = true;                                                                                                        \
    }                                                                                                                     \
    const float Cf = (float)Ce, Sf = (float)Se;                      /* the estimate of this batch (same in every lane) */ \
    const float V2 = __builtin_fmaf(Cf, Cf, Sf * Sf);                                                                     \
    const float rV = __builtin_amdgcn_rsqf(fmaxf(V2, 1e-12f)) * 1.000001f;   /* >= 1 / |V| */                             \
    const float Vn = V2 * rV;                                                 /* |V| (to 2e-6) */                          \
    const float nrat = (float)n * rV;                                         /* >= n / |V| */
    int sweep = 1, ex;
    do {                                                     // :525 sweeps to fixpoint (Q7)
        ex = n;
        nxt_cnt = 0;
        int i = n;                                           // contiguous cursor: the entries appended during this sweep ...
        // ... or the whole list: the first sweep; worklists given up; and a region of up to 8 pixels -- one batch sweeps it
        // again, which costs less than fetching the slack records of its worklist (most regions are this small)
        if (sweep == 1 || !filter || n <= 8) i = 0;
        else {
            // ---- entries of earlier sweeps that still had a growable non-member neighbour ----
            int wi = 0;                                      // worklist cursor
            while (true) {
                GROW_ESTIMATE();
                wi = uni(wi); wl_cnt = uni(wl_cnt); flt_base = uni(flt_base); flt_valid = uni((int)flt_valid) != 0;
                flt_need = ((unsigned long long)(uint32_t)uni((int)(uint32_t)(flt_need >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)flt_need);
                if (wi >= wl_cnt) break;
                int cnt = min(8, wl_cnt - wi);
                if (tol_small && filter) {                   // (filter lost in this sweep: the rest of the worklist is tested in full)
                    if (!flt_valid || wi >= flt_base + 64) {
                        flt_base = wi;
                        bool nd = false;
                        if (wi + lane < wl_cnt) {
                            const int ei = (int)G_WL(wave)[wi + lane];
                            nd = true;
                            if (ei < mcap) {
                                const nf4 mt = meta[(uint32_t)ei];
                                const float vx = Cf * rV, vy = Sf * rV;                  // current unit sum vector (norm within 3e-6 of 1)
                                const float dotv = mt.x * vx + mt.y * vy, crs = fabsf(mt.x * vy - mt.y * vx);
                                nd = !(dotv > 0.0f && crs + 1e-5f + 2.0f * kEpsU * nrat < mt.z);
                            }
                        }
                        flt_need = ballot64(nd);
                        flt_valid = true;
                    }
                    const int off = wi - flt_base;
                    const int nval = min(64, wl_cnt - flt_base) - off;         // entries of the chunk from wi on
                    const unsigned long long rest = flt_need >> off;          // bit 0 = entry wi
                    const int nskip = rest ? min(__builtin_ctzll(rest), nval) : nval;
                    if (nskip > 0) {                         // a run of entries that cannot accept anything: carry them over
                        // (lane-dependent branches stay in the MIDDLE of wave-uniform blocks, see STAT)
                        const bool room = nxt_cnt + nskip <= WCAP;
                        // (in place: all 64 lanes read before any of them writes, and nxt_cnt <= wi)
                        G_WL(wave)[room && lane < nskip ? nxt_cnt + lane : WCAP] = G_WL(wave)[min(wi + lane, WCAP - 1)];   // (no branch: dummy slot)
                        filter = filter && room;
                        nxt_cnt += room ? nskip : 0;
                        wi += nskip;
                        continue;
                    }
                    // the run of consecutive entries to test (no skipped entry in between: its check would be stale after an accept)
                    const unsigned long long inv = ~rest;
                    cnt = min(cnt, inv ? __builtin_ctzll(inv) : 64);
                }
                const int eidx = e < cnt ? (int)G_WL(wave)[wi + e] : 0;
                wi += batch(cnt, eidx, false, Cf, Sf, rV, Vn, nrat);
            }
        }
        while (true) {                                       // ---- contiguous entries; n is live (:529) ----
            GROW_ESTIMATE();
            i = uni(i);
            if (i >= n) break;
            const int cnt = min(8, n - i);
            i += batch(cnt, i + e, n - i <= LCAP, Cf, Sf, rV, Vn, nrat);   // (direct: entries i .. n - 1 are all in the ring)
        }
        wl_cnt = nxt_cnt;
        sweep++;
        flt_valid = false;
        if (n != ex && n > 8) wg_fence();                    // meta[] written in this sweep is read in the next
    } while (n != ex);
#undef GROW_ESTIMATE
    if (lane == 0) { g_ws[wave].gnum = n; g_ctx[wave].llo = wt ? max(n - LCAP, 0) : 0; }
    if (wt) wg_fence();                                      // the written-through part is read back by the stages that follow
    STAT(ST_GROW, 1);
    STAT(ST_GROWN, n);
    DSTAT(ST_TGROW, NOW() - t0);
    return n;
}

// ---------------------------------------------------------------------------------------------
// CenterGetter (:592-619) + OrientationGetter (:621-667) + RectangleConverter (:669-734)
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void rect_convert(int cw_, int num, double regdeg, double aliPro, int pk, double tol) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int lane = c.lane, w = c.w;
    [[maybe_unused]] const long long t0 = NOW();
    const int wave = __builtin_amdgcn_readfirstlane(c.wave);
    double S = 0;                                // serial accumulation in list order (bit-exact): lane 0 cenX, 1 cenY, 2 weight sum
    for (int base = 0; base < num; base += 64) {                                   // :608-613
        const int kx = base + lane;
        const bool valid = kx < num;
        const uint32_t pkx = valid ? lget(c, kx) : 0u;
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        const double wgt = valid ? c.mag[(size_t)y * w + x] : 0.0;
        const double ax = wgt * x, ay = wgt * y;
        for (int half = 0; half < 2; half++) {
            const int cnt = min(32, num - base - 32 * half);
            if (cnt <= 0) break;
            stage4(wave, lane, half, ax, ay, wgt, 0.0);
            S = acc32(wave, lane, cnt, S);
        }
    }
    double ws = rl(S, 2);
    const double cenX = rl(S, 0) / ws;
    const double cenY = rl(S, 1) / ws;

    S = 0;                                       // lane 0 Ixx, 1 Iyy, 2 Ixy, 3 weight sum
    for (int base = 0; base < num; base += 64) {                                   // :637-643
        const int kx = base + lane;
        const bool valid = kx < num;
        const uint32_t pkx = valid ? lget(c, kx) : 0u;
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        const double wgt = valid ? c.mag[(size_t)y * w + x] : 0.0;
        const double ddy = y - cenY, ddx = x - cenX;
        const double a = wgt * (ddy * ddy), b = wgt * (ddx * ddx), cc = wgt * ddx * ddy;
        for (int half = 0; half < 2; half++) {
            const int cnt = min(32, num - base - 32 * half);
            if (cnt <= 0) break;
            stage4(wave, lane, half, a, b, -cc, wgt);     // Ixy -= cc (:642): adding the negated term is the same operation
            S = acc32(wave, lane, cnt, S);
        }
    }
    ws = rl(S, 3);
    const double Ixx = rl(S, 0) / ws, Iyy = rl(S, 1) / ws, Ixy = rl(S, 2) / ws;
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
    TIES_AT(TS_FLIP, fabs(regDif - tol) <= kTieFlip ? 1 : 0);     // (the wraps above are continuous in |regDif|: no decision)
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
    if (lane == 0) {
        Rec& r = g_ws[c.wave].rec;
        r.x1 = cenX + lenMin * dx; r.y1 = cenY + lenMin * dy;                      // :717-720
        r.x2 = cenX + lenMax * dx; r.y2 = cenY + lenMax * dy;
        r.wid = widMax - widMin;
        r.cX = cenX; r.cY = cenY; r.deg = inertiaDeg; r.dx = dx; r.dy = dy;
        r.p = aliPro; r.prec = tol; r.pk = pk;
        if (r.wid < 1) r.wid = 1;                                                  // :730
    }
    PSTAT(ST_TRECT, NOW() - t0);
}

__device__ __forceinline__ double rec_density(int num, const Rec& r) {             // :757,:798,:827,:867
    const double ex = r.x1 - r.x2, ey = r.y1 - r.y2;
    return num / (sqrt(ex * ex + ey * ey) * r.wid);
}

// ---------------------------------------------------------------------------------------------
// RegionRadiusReducer, myLSD.cpp:736-802 (incl. the `i <= num` sentinel behaviour)
// ---------------------------------------------------------------------------------------------
// Returns the new region size, or -(size + 1) when the region is given up (:792).  The rectangle is g_ws[c.wave].rec.
__device__ __noinline__ int radius_reduce_impl(int cw_, int sx, int sy, int num, double regdeg, double denThre);
__device__ __noinline__ int radius_reduce(int cw_, int sx, int sy, int num, double regdeg, double denThre) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    [[maybe_unused]] const long long t0 = NOW();
    const int r = radius_reduce_impl(c.wave, sx, sy, num, regdeg, denThre);
    return r;
}
__device__ __noinline__ int radius_reduce_impl(int cw_, int sx, int sy, int num, double regdeg, double denThre) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int lane = c.lane;
    STAT(ST_RRR, 1);
    double den = rec_density(num, g_ws[c.wave].rec);
    const bool axis0 = axis_exact(g_ws[c.wave].rec.dx, g_ws[c.wave].rec.dy);
    // (the comparison with denThre here repeats the caller's, which has counted its tie; the ones after a refit are counted below)
    if (den > denThre) return num;                                                 // :760
    // keep the grow-order list for the marking loops before it gets reordered
    for (int k2 = lane; k2 < num; k2 += 64) c.gcopy[k2] = lget(c, k2);
    // curMap moves to HBM in full: the removals below clear bits there, the marking stages read them there (and the scratch of the
    // parallel passes takes the tile cache's place)
    flush_tiles(c);
    if (lane == 0) g_ws[c.wave].has_copy = 1;
    wg_fence();
    const Rec rec = g_ws[c.wave].rec;
    const double ax = sx - rec.x1, ay = sy - rec.y1, bx = sx - rec.x2, by = sy - rec.y2;
    const double rad1 = sqrt(ax * ax + ay * ay), rad2 = sqrt(bx * bx + by * by);    // :768-769
    double rad = rad1 > rad2 ? rad1 : rad2;
    bool removed_any = false;
    const int wave = __builtin_amdgcn_readfirstlane(c.wave);
    unsigned long long* const msk = reinterpret_cast<unsigned long long*>(g_acc[wave]);   // keep-masks of up to 128 chunks of 64 entries
    uint32_t* const mv = reinterpret_cast<uint32_t*>(G_WL(wave));                          // up to kMvCap moved entries: the worklist and the tile cache behind it
    while (den < denThre) {                                                        // :775
        rad *= 0.75;
        STAT(ST_RRRPASS, 1);
        num = __builtin_amdgcn_readfirstlane(num);
        // The reference walks the list from the front and fills every slot whose point is farther than rad with the LAST
        // point of the list, re-testing it (:779-789).  The outcome is: the K points within rad stay in slots [0, K); the
        // holes among those slots (ascending) receive the kept points of the slots >= K, taken from the back (descending).
        // That is computed 64 entries at a time; lists too long for the scratch arrays take the reference's own loop below.
        const int nchunks = (num + 63) >> 6;
        bool parallel = nchunks <= 128;
        int K = 0;
        if (parallel) {
            for (int ci = 0; ci < nchunks; ci++) {
                const int idx = ci * 64 + lane;
                const bool valid = idx < num;
                const uint32_t pkx = valid ? lget(c, idx) : 0u;
                const int px = (int)(pkx & 0xffffu), py = (int)(pkx >> 16);
                const double ddx = sx - px, ddy = sy - py;
                const double dist = sqrt(ddx * ddx + ddy * ddy);
                const bool far = valid & (dist > rad);                             // :780
                if (!axis0) TIES_AT(TS_DIST, __builtin_popcountll(ballot64(valid & (fabs(dist - rad) <= kTieRel * (1.0 + rad)))));   // (rad derives from the rectangle's corners)
                const unsigned long long nearm = ballot64(valid & !far);
                msk[ci] = nearm;                           // (all lanes, same value)
                K += __builtin_popcountll(nearm);
            }
            if (min(K, num - K) > kMvCap) parallel = false;  // more moves than mv[] holds
        }
        if (parallel) {
            if (K != num) {
                int nm = 0;                                // kept points of the slots >= K, highest slot first
                for (int ci = nchunks - 1; ci >= 0 && ci * 64 + 63 >= K; ci--) {
                    const int idx = ci * 64 + lane;
                    const bool is = (((msk[ci] >> lane) & 1ull) != 0ull) & (idx >= K);
                    const unsigned long long mm = ballot64(is);
                    const int above = __builtin_popcountll((mm >> lane) >> 1);
                    mv[is ? nm + above : kMvCap] = is ? lget(c, idx) : 0u;         // (no branch: dummy slot)
                    nm += __builtin_popcountll(mm);
                }
                int nh = 0;                                // far points of the slots < K, lowest slot first
                for (int ci = 0; ci * 64 < K; ci++) {
                    const int idx = ci * 64 + lane;
                    const bool valid = idx < num;
                    const bool nearb = ((msk[ci] >> lane) & 1ull) != 0ull;
                    const unsigned long long hm = ballot64((idx < K) & !nearb);
                    const uint32_t old = valid ? lget(c, idx) : 0u;
                    if (valid & !nearb) tm_clear(c, (int)(old & 0xffffu), (int)(old >> 16));       // curMap = 0 (:781), every far point of this chunk
                    if ((idx < K) & !nearb) lset(c, idx, mv[nh + mbcnt(hm)]);                      // :782-785
                    nh += __builtin_popcountll(hm);
                }
                for (int ci = (K + 63) >> 6; ci < nchunks; ci++) {                                 // far points of the chunks wholly behind K
                    const int idx = ci * 64 + lane;
                    if ((idx < num) & (((msk[ci] >> lane) & 1ull) == 0ull)) {
                        const uint32_t old = lget(c, idx);
                        tm_clear(c, (int)(old & 0xffffu), (int)(old >> 16));
                    }
                }
                num = K;
                removed_any = true;
            }
            // the extra round at i == num (:779 `<=`): the slot holds the NULL written at :784-785, i.e. the point (0, 0)
            if (!removed_any) STAT(ST_OOB, 1);             // the reference reads out of bounds here (UB): no removal
            else {
                const double ddx = sx, ddy = sy;
                if (!axis0) TIES_AT(TS_DIST, fabs(sqrt(ddx * ddx + ddy * ddy) - rad) <= kTieRel * (1.0 + rad) ? 1 : 0);
                if (sqrt(ddx * ddx + ddy * ddy) > rad) {
                    if (lane == 0) tm_clear(c, 0, 0);      // curMap(0, 0) = 0
                    num--;                                 // the last point is dropped from the list (its curMap bit stays)
                    STAT(ST_SENT, 1);
                }
            }
            wg_fence();
        }
        int i = parallel ? num + 1 : 0;
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
            if (!axis0) TIES_AT(TS_DIST, fabs(sqrt(ddx * ddx + ddy * ddy) - rad) <= kTieRel * (1.0 + rad) ? 1 : 0);
            if (sqrt(ddx * ddx + ddy * ddy) > rad) {                               // :780
                if (lane == 0) {
                    tm_clear(c, px, py);                                           // curMap = 0 (:781)
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
        if (num < 2) return -(num + 1);                                            // :792
        rect_convert(c.wave, num, regdeg, rec.p, rec.pk, rec.prec);                     // :797 (p, prec unchanged)
        den = rec_density(num, g_ws[c.wave].rec);
        TIES_AT(TS_DENS, (fabs(den - denThre) <= kTieRel * denThre && !axis_exact(g_ws[c.wave].rec.dx, g_ws[c.wave].rec.dy)) ? 1 : 0);       // :775
    }
    return num;
}

// ---------------------------------------------------------------------------------------------
// LogGammaCalculator (:882-924): a look-up in the host-computed table (lsd_ctx.hip sizes it for every pixel count a rectangle of
// the image can have; only images of more than kLgTableMax scaled pixels can get past it, and then with the device's own log / sinh / pow)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double log_gamma_dev(const double* lgamma, int lg_count, int x) {
    if (x >= 0 && x < lg_count) return lgamma[x];
    const double xd = x;
    return 0.918938533204673 + (xd - 0.5) * log(xd) - xd +
           0.5 * xd * log(xd * sinh(1.0 / xd) + 1.0 / (810 * pow(xd, 6.0)));
}

// ---------------------------------------------------------------------------------------------
// RectangleNFACalculator, myLSD.cpp:926-1059 (the full-image pass :940-945 is a no-op, not restated), in two parts: the pixel
// count of the rectangle with all 64 lanes (:947-1017), and the value from the two counts (:1019-1058) -- scalar arithmetic that
// RectangleImprover's five tries of a phase run side by side in five lanes (improve(), below).
// ---------------------------------------------------------------------------------------------
// Returns `all`; ali[q] = pixels of the rectangle whose level-line angle is within prec[q] of the rectangle's (NP > 1: the tries of
// a phase that halves p share the rectangle and differ in the precision only).
template <int NP>
__device__ __forceinline__ int nfa_count(const RCtx& c, const Rec& rec, const double (&prec)[NP], int (&ali)[NP]) {
    const int lane = c.lane, xLim = c.w, yLim = c.h;
    STAT(ST_NFA, NP);
    [[maybe_unused]] const long long t00 = NOW();
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
    int all = 0;
    #pragma unroll
    for (int q = 0; q < NP; q++) ali[q] = 0;
    // decisions within the libm's noise (ST_TIES): a corner or an edge within kTieCoord of a pixel column / row (not for rectangles that
    // are axis-parallel to the last bit: their coordinates do not depend on the libm), a pixel's angle within kTieAng-ish of the precision
    const bool axis0 = axis_exact(rec.dx, rec.dy);
    const double ex_ = rec.x2 - rec.x1, ey_ = rec.y2 - rec.y1;
    const double cnz = kTieCoord * (fabs(rec.x1) + fabs(rec.y1) + fabs(ex_) + fabs(ey_) + rec.wid + 1.0);     // what the libm's last place can move a corner
    int ties_a = 0;
    int ties = (lane == 0 && !axis0 && (fabs(vx0 - rint(vx0)) <= cnz || fabs(vx2 - rint(vx2)) <= cnz)) ? 1 : 0;   // (per lane; summed over the wave below)
    // per-column scan results of one 64-column block; the sweep worklists are free while a rectangle is being rated
    int* const s_incl = reinterpret_cast<int*>(G_WL(c.wave));
    int* const s_lo = s_incl + 64;
    int* const s_x = s_incl + 128;
    for (int cb = 0; cb < xlen; cb += 64) {
        const int i = cb + lane;
        int cntc = 0, lo = 0, xr = 0;
        if (i < xlen) {
            xr = cvt_x86(i + cx0);                                                 // :976
            int yLow, yHigh;
            const double eLow = xr < vx3 ? vy0 + (xr - vx0) * k3 : vy3 + (xr - vx3) * k2;     // :988-989 / :992-993
            const double eHigh = xr < vx1 ? vy0 + (xr - vx0) * k0 : vy1 + (xr - vx1) * k1;    // :998-999 / :1002-1003
            yLow = cvt_x86(ceil(eLow));
            yHigh = cvt_x86(floor(eHigh));
            const double kLow = xr < vx3 ? k3 : k2, kHigh = xr < vx1 ? k0 : k1;
            if (!axis0 && (fabs(eLow - rint(eLow)) <= cnz * (1.0 + fabs(kLow)) || fabs(eHigh - rint(eHigh)) <= cnz * (1.0 + fabs(kHigh)) ||
                           fabs(xr - vx3) <= cnz || fabs(xr - vx1) <= cnz)) {
                ties++;
#ifdef LSD_TIE_PRINT
                printf("edge tie: xr %d eLow %.17g eHigh %.17g vx0 %.17g vx1 %.17g vx3 %.17g dx %.17g dy %.17g wid %.17g x1 %.17g y1 %.17g\n", xr, eLow, eHigh, vx0, vx1, vx3, rec.dx, rec.dy, rec.wid, rec.x1, rec.y1);
#endif
            }
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
        s_incl[lane] = inc; s_lo[lane] = lo; s_x[lane] = xr;
        all += tot;
        for (int t0 = 0; t0 < tot; t0 += 64) {                // flattened (column, row) pairs, 64 per step
            const int t = t0 + lane;
            double df = 1e300;                                 // (no pixel: within no precision)
            if (t < tot) {
                int ci = 0;                                    // smallest ci with s_incl[ci] > t
                for (int step = 32; step >= 1; step >>= 1)
                    if (s_incl[ci + step - 1] <= t) ci += step;
                const int ex = ci ? s_incl[ci - 1] : 0;
                const int j = s_lo[ci] + (t - ex);
                df = angle_diff(rec.deg, c.deg[(size_t)j * xLim + s_x[ci]]);       // :1009-1011
            }
            bool near = false;
            #pragma unroll
            for (int q = 0; q < NP; q++) {
                ali[q] += __builtin_popcountll(ballot64(df < prec[q]));            // :1012-1013
                near = near || fabs(df - prec[q]) <= 4.0 * kTieAng;                // (two angles of an ulp each)
            }
            if (near) ties_a++;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { ties += __shfl_xor(ties, off); ties_a += __shfl_xor(ties_a, off); }
    TIES_AT(TS_EDGE, ties);
    TIES_AT(TS_ALIGN, ties_a);
    PSTAT(ST_NFACNT, NOW() - t00);
    return all;
}

// bit 0 of flags: the value is made of host-computed numbers alone (logNT, log10 p: the reference's own libm) -- no device-evaluated
// function; bits 8..: stopping tests of the tail that the bracket (below) could not decide
struct NfaVal { double v; int flags; };
// Scalar code (every active lane for itself: improve() runs five at a time); p = the rectangle's p, pk the number of its halvings.
__device__ __noinline__ NfaVal nfa_tail(int all, int ali, int pk, double p, double logNT, const double* ptab, const double* lgamma, int lg_count) {
    if (all == 0 || ali == 0) return NfaVal{-logNT, 1};                            // :1019-1022
    const double logp = ptab[pk * 3 + 0], log10p = ptab[pk * 3 + 1], log1mp = ptab[pk * 3 + 2];
    if (all == ali) return NfaVal{-logNT - all * log10p, 1};                       // :1023-1026
    const double proTerm = p / (1.0 - p);
    const double log1Coef = log_gamma_dev(lgamma, lg_count, all + 1) - log_gamma_dev(lgamma, lg_count, ali + 1) - log_gamma_dev(lgamma, lg_count, all - ali + 1);
    const double log1Term = log1Coef + ali * logp + (all - ali) * log1mp;          // :1033
    // From here on the reference calls exp, log10 and pow.  What reaches the result -- the first term and the logarithm of the tail --
    // is evaluated correctly rounded (exp_g, log10_g: crmath.h); pow and log10 inside the loop only decide when the sum stops, and that
    // decision is taken from the device math library's values where a bracket around them (kOcmlBracket, many times their error:
    // tests/test_parity_gpu.py::test_device_libm_is_inside_the_nfa_bracket) leaves no doubt, from the correctly rounded values otherwise.
    double term = exp_g(log1Term);
    const double eps = 2.2204e-16;
    if (fabs(term) < 100 * eps) {                                                  // :1037-1043
        if (ali > all * p) return NfaVal{-log10_g(term) - logNT, 0};
        return NfaVal{-logNT, 1};
    }
    int nslow = 0;
    double binTail = term;
    const double tole = 0.1;
    constexpr double kOcmlBracket = 0x1p-44, kTiny = 0x1p-1000;
    for (int i = ali + 1; i <= all; i++) {                                         // :1046-1056
        const double binTerm = (all - i + 1) / (i * 1.0);
        const double multTerm = binTerm * proTerm;
        term *= multTerm;
        binTail += term;
        if (binTerm < 1) {
            // err < tole * |-log10(binTail) - logNT| * binTail ?  (:1052-1053)  Every operation of the two sides is monotone in the value
            // of pow resp. log10, so the sides at the ends of the brackets enclose the sides at the correctly rounded values.
            const double N = (double)(all - i + 1), om = 1.0 - multTerm;
            const double Pd = pow(multTerm, N), Ld = log10(binTail);
            const double Pe = Pd * kOcmlBracket + kTiny, Le = fabs(Ld) * kOcmlBracket + kTiny;
            const double err_hi = term * ((1 - (Pd - Pe)) / om - 1), err_lo = term * ((1 - (Pd + Pe)) / om - 1);
            const double a1 = -(Ld - Le) - logNT, a2 = -(Ld + Le) - logNT;
            const double f1 = fabs(a1), f2 = fabs(a2);
            const double rhs_hi = tole * fmax(f1, f2) * binTail;
            const double rhs_lo = (a1 > 0) == (a2 > 0) && a1 != 0 && a2 != 0 ? tole * fmin(f1, f2) * binTail : 0.0;
            bool stop;
            if (err_hi < rhs_lo) stop = true;
            else if (!(err_lo < rhs_hi)) stop = false;
            else {
                nslow++;                                    // (stopping tests of the tail the bracket could not decide)
                const double err = term * ((1 - pow_g(multTerm, N)) / om - 1);
                stop = err < tole * fabs(-log10_g(binTail) - logNT) * binTail;
            }
            if (stop) break;
        }
    }
    return NfaVal{-log10_g(binTail) - logNT, nslow << 8};
}

// RectangleImprover, myLSD.cpp:1061-1158: the initial evaluation, then five phases of five tries each.  Within a phase the tried
// rectangles do not depend on the values found (only `best` does, and a phase starts from the best so far): the five pixel
// counts are taken one after the other -- one pass for the phases that only halve p -- and the five values computed side by side.
__device__ __noinline__ double improve(int cw_) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int lane = c.lane;
    const double delt = 0.5, delt2 = delt / 2.0;
    Rec& best = g_ws[c.wave].rec;                           // (the best rectangle so far stays in LDS: every lane writes the same values)
    double bestNFA = 0;
    [[maybe_unused]] const long long t0 = NOW();
    // How close the comparisons below come to a tie, as a MARGIN: the distance of the operands over the most the reference's
    // libm (glibc: exp and pow within 1 ulp, log10 within 1 ulp of the correctly rounded values computed here) can move them apart.
    // v = fl(-L - logNT) with L = log10(tail): |dL| <= 2^-51 |L| + 2^-53 (the tail's first term differs by an ulp), and the
    // subtraction rounds to an ulp of max(|v|, logNT): noise(v) = 2^-51 |v + logNT| + 2^-52 (1 + max(|v|, logNT)).  A decision can come
    // out differently on the two libms only where the margin is below 1 (tools/campaign.py enforces a floor of 2).  A value made of
    // the host's numbers alone (-logNT - n log10 p: an exact 0 exists, w h = 6^4, p = 1/6, n = 10) is the reference's own.
    auto margins = [&](double v, bool host_only, bool first) {
        if (fabs(v) <= 1.7976931348623157e308) {
            const double nv = 0x1p-51 * fabs(v + c.logNT) + 0x1p-52 * (1.0 + fmax(fabs(v), c.logNT));
            if (!host_only) STATMAX(ST_MINNFA, kInfBits - (unsigned long long)__double_as_longlong(fabs(v) / nv));       // (v is compared with 0: :1075, :242)
            int tie = (!host_only && fabs(v) / nv < 2.0) ? 1 : 0;              // (the campaigns' floor: a margin below 1 can flip, below 2 is counted)
            if (!first && v != bestNFA) {
                const double nb = 0x1p-51 * fabs(bestNFA + c.logNT) + 0x1p-52 * (1.0 + fmax(fabs(bestNFA), c.logNT));
                STATMAX(ST_MINGAP, kInfBits - (unsigned long long)__double_as_longlong(fabs(v - bestNFA) / (nv + nb)));
                if (fabs(v - bestNFA) / (nv + nb) < 2.0) tie++;
            }
            TIES_AT(TS_NFA, tie);
        }
    };
    {   // :1075-1079
        Rec r = best;
        const double pr[1] = {r.prec};
        int al[1];
        const int all = nfa_count<1>(c, r, pr, al);
        const NfaVal nv = nfa_tail(all, al[0], r.pk, r.p, c.logNT, c.ptab, c.lgamma, c.lg_count);
        STAT(ST_NFASLOW, nv.flags >> 8);
        margins(nv.v, (nv.flags & 1) != 0, true);
        bestNFA = nv.v;
    }
    // one try of phase ph (:1084-1092 / :1148-1156 halve p; :1097-1107 reduce width; :1112-1125 move one side; :1130-1143 the other)
    auto next_try = [&](Rec& r, int ph) -> bool {
        if (ph == 0 || ph == 4) { r.p /= 2.0; r.prec = r.p * kPi; r.pk++; return true; }
        if (!(r.wid - delt >= 0.5)) return false;
        if (ph == 2) { r.x1 -= r.dy * delt2; r.y1 += r.dx * delt2; r.x2 -= r.dy * delt2; r.y2 += r.dx * delt2; }
        else if (ph == 3) { r.x1 += r.dy * delt2; r.y1 -= r.dx * delt2; r.x2 += r.dy * delt2; r.y2 -= r.dx * delt2; }
        r.wid -= delt;
        return true;
    };
    for (int ph = 0; ph < 5 && !(bestNFA > 0); ph++) {      // phase boundary (:1078,:1093,:1108,:1126,:1144)
        // lane t < 5 keeps the counts of try t
        int my_all = 0, my_ali = 0, my_pk = 0;
        double my_p = 0.0;
        bool my_eval = false;
        Rec r = best;
        if (ph == 0 || ph == 4) {
            double pr[5];
            int al[5];
            #pragma unroll
            for (int t = 0; t < 5; t++) pr[t] = r.p / (double)(2 << t) * kPi;      // p halved t + 1 times, exactly as next_try does it
            const int all = nfa_count<5>(c, r, pr, al);
            #pragma unroll
            for (int t = 0; t < 5; t++)
                if (lane == t) { my_all = all; my_ali = al[t]; my_pk = r.pk + t + 1; my_p = r.p / (double)(2 << t); my_eval = true; }
        } else {
            for (int t = 0; t < 5; t++) {
                if (!next_try(r, ph)) continue;
                const double pr[1] = {r.prec};
                int al[1];
                const int all = nfa_count<1>(c, r, pr, al);
                if (lane == t) { my_all = all; my_ali = al[0]; my_pk = r.pk; my_p = r.p; my_eval = true; }
            }
        }
        NfaVal nv{0.0, 0};
        if (my_eval) nv = nfa_tail(my_all, my_ali, my_pk, my_p, c.logNT, c.ptab, c.lgamma, c.lg_count);
        // the five values in the reference's order
        r = best;
        for (int t = 0; t < 5; t++) {
            if (!next_try(r, ph)) continue;
            const double v = rl(nv.v, t);
            const int fl = __builtin_amdgcn_readlane(nv.flags, t);
            STAT(ST_NFASLOW, fl >> 8);
            margins(v, (fl & 1) != 0, false);
            if (v > bestNFA) { bestNFA = v; best = r; }
        }
    }
    PSTAT(ST_TNFA, NOW() - t0);
    return bestNFA;
}

// Refiner, myLSD.cpp:804-880, first half: the re-estimated angle tolerance (:833-855).  The regrow (:857),
// the refit (:866) and the density checks are in the caller's two-pass loop so that grow() and
// rect_convert() are inlined once.
__device__ __noinline__ double refine_tol(int cw_, int sx, int sy, int num, double cenDeg) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int lane = c.lane, w = c.w;
    [[maybe_unused]] const long long t0 = NOW();
    const double rwid = g_ws[c.wave].rec.wid;
    const bool axis0 = axis_exact(g_ws[c.wave].rec.dx, g_ws[c.wave].rec.dy);
    const int wave = __builtin_amdgcn_readfirstlane(c.wave);
    double S = 0;                                 // serial accumulation in list order: lane 0 difSum, 1 squSum
    int ptNum = 0, ties = 0;
    for (int base = 0; base < num; base += 64) {                                   // :839-853
        const int kx = base + lane;
        bool flag = false;
        double degDif = 0;
        if (kx < num) {
            const uint32_t pkx = lget(c, kx);
            const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
            const double ddx = sx - x, ddy = sy - y;
            const double dist = sqrt(ddx * ddx + ddy * ddy);
            if (!axis0 && fabs(dist - rwid) <= kTieRel * (1.0 + rwid)) ties++;    // :845 within the noise of the rectangle's width
            if (dist < rwid) {
                flag = true;
                degDif = c.deg[(size_t)y * w + x] - cenDeg;
                if (fabs(fabs(degDif) - kPi) <= kTieFlip) ties++;                  // :848-851 (two map angles half a turn apart)
                while (degDif <= -kPi) degDif += 2 * kPi;
                while (degDif > kPi) degDif -= 2 * kPi;
            }
        }
        const double sq = degDif * degDif;
        const unsigned long long m = ballot64(flag);
        if (m == 0ull) continue;
        ptNum += __builtin_popcountll(m);
        // (points outside the width contribute +0.0, which leaves a sum that started at +0.0 unchanged to the bit)
        for (int half = 0; half < 2; half++) {
            const int cnt = min(32, num - base - 32 * half);
            if (cnt <= 0) break;
            stage4(wave, lane, half, flag ? degDif : 0.0, flag ? sq : 0.0, 0.0, 0.0);
            S = acc32(wave, lane, cnt, S);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) ties += __shfl_xor(ties, off);
    TIES_AT(TS_DIST, ties);
    const double difSum = rl(S, 0), squSum = rl(S, 1);
    const double meanDif = difSum / (ptNum * 1.0);
    PSTAT(ST_TREFINE, NOW() - t0);
    return 2.0 * sqrt((squSum - 2 * meanDif * difSum) / (ptNum * 1.0) + meanDif * meanDif);   // :855
}

// ---------------------------------------------------------------------------------------------
// seed loop, myLSD.cpp:219-272
// ---------------------------------------------------------------------------------------------
// usedMap marking (:243-248 / :259-265) restricted to the grown pixels; returns their bounding box.
// epoch1 == 0: a rejected region (usedMap = 2); else an accepted line of epoch epoch1 - 1 (usedMap = 1).
struct Box { int x0, y0, x1, y1; };

__device__ __noinline__ Box mark_region(int cw_, uint32_t epoch1, const uint32_t* src, int src_cnt) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    const int w = c.w;
    [[maybe_unused]] const long long t0 = NOW();
    wg_fence();                                   // (after RegionRadiusReducer: its removals from curMap must have landed)
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    const int cnt = src ? src_cnt : g_ws[c.wave].gnum;
    const bool has_copy = g_ws[c.wave].has_copy != 0;
    const uint32_t cur_id = g_ws[c.wave].cur_id;
    for (int k2 = c.lane; k2 < cnt; k2 += 64) {
        const uint32_t pkx = src ? src[k2] : (has_copy ? c.gcopy[k2] : lget(c, k2));
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        const size_t q = (size_t)y * w + x;
        // curMap == 1 only: a stashed list (src) holds exactly those; the last grow's list is curMap unless RegionRadiusReducer has taken
        // pixels out of it -- then curMap is in tmask (flush_tiles) and the grow-order copy is walked
        if (src || !has_copy || tm_member(c, x, y, cur_id)) {
            const uint32_t old = c.pw[q];
            if (epoch1) {
                if (c.sets) {                              // a banned member ends its certified set (the word holds the set's label until now)
                    const uint32_t lb = label_set(c.ltag, c.epochmap[q]);
                    if (lb) st_l2(&c.sets[lb], 0u);
                }
                c.epochmap[q] = epoch1; c.pw[q] = (old & ~3u) | kPwLine; atomicMax(&c.tep[(y >> 3) * c.tilesX + (x >> 3)], epoch1);
            }
            else c.pw[q] = (old & ~3u) | kPwRejected;
            x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
        x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
    }
    PSTAT(ST_TMARK, NOW() - t0);
    Box bx; bx.x0 = x0; bx.y0 = y0; bx.x1 = x1; bx.y1 = y1;
    return bx;
}

// bounding box of `in` and the first num pixels of the region list (of the grow-order copy when from_copy)
__device__ __noinline__ Box list_bbox(int cw_, int num, Box in, bool from_copy) {
    RCtx c = g_ctx[__builtin_amdgcn_readfirstlane(cw_)];
    c.lane = (int)(threadIdx.x & 63u);
    int x0 = in.x0, y0 = in.y0, x1 = in.x1, y1 = in.y1;
    for (int k2 = c.lane; k2 < num; k2 += 64) {
        const uint32_t pkx = from_copy ? c.gcopy[k2] : lget(c, k2);
        const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
        x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
        x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
    }
    Box bx; bx.x0 = x0; bx.y0 = y0; bx.x1 = x1; bx.y1 = y1;
    return bx;
}


// ---------------------------------------------------------------------------------------------
// Certified uniform sets.
//
// A fifth of all pixels the bench batch grows, and nine tenths of its heaviest images', belong to a few sparse structures whose
// pixels all have THE SAME level-line angle, bit for bit (long axis-parallel edges: the seams of the tiled maps, runs of wall),
// grown again from every one of their hundreds of seeds: RegionGrower returns the same ~1 700 pixels, the rectangle around them is
// far too sparse, Refiner re-estimates the tolerance from the pixels near the seed -- all with the seed's own angle, so the new
// tolerance is exactly 0 --, regrows the seed alone and gives up (myLSD.cpp:833-861): nothing is marked, and the next seed of the
// structure starts over (the reference does exactly that).  For such a set S the whole evaluation of ANY seed in S is known without
// growing anything, provided
//   (1) every pixel of S has the angle theta (fp64 equality) and is growable (not banned);
//   (2) every growable pixel next to S that is not in S is further than tol + 1e-5 from theta (circular distance; for tol < pi/2 the
//       reference's test :540-543 IS the circular distance; the 1e-5 covers the packed fp32 angle the check reads, 1.2e-6);
//   (3) S is 8-connected (it is: it was grown as one region);
//   (4) the rectangle of S is sparse by a margin: density < denThre (1 - 1e-3).
// (1)-(3): whatever seed B in S the reference starts from, regDeg is theta at every step (the sum of equal unit vectors has their
// direction; atan2's rounding is many orders below the 1e-5), every member passes its test the first time it comes up and every
// other growable neighbour fails every time: RegionGrower returns S, |S| >= regThre.  (4): the density of the rectangle depends on
// the list order only through the rounding of its sums (relative 1e-10 for 65 535 pixels), so it is below denThre for every seed.
// Refiner then sees angle differences of exactly 0 (:839-853: degDif = theta - theta), tol = 2 sqrt(0) = 0, regrows the seed alone
// (:857, `0 < 0` never holds) and fails at :861.  Outcome: no marks, first region |S| pixels, final region 1 pixel.
//
// Mechanics.  epochmap[] is free for growable pixels (it holds the accept epoch of banned ones): there it carries the LABEL of the
// pixel's set, tagged with the launch's run number (label_make: what earlier launches left in the buffer is no label of this one).  An evaluation that went exactly this way (EvalOut.cert) offers its first list: certify_set() checks
// (1) and labels the members under the cursor lock -- the lock commits hold, so no ban can slip between check and label --, checks
// (2) and publishes the set's size in sets[label].  A line that bans a member clears sets[label] (mark_region: the label is still in
// the word it overwrites with the epoch).  Invariant: every member of a live set carries its label (a new set that takes over a
// labelled pixel ends the older set).  eval_seed() looks at its seed's label first; a live set answers at once (EvalOut.setid).  The
// result waits in the ring as R_SETL and is valid at its turn iff the set is still alive -- alive means no member was ever banned,
// which is exactly "no member banned since the snapshot" for a result that has ALL of S as its list.  Wavefronts that help another
// image neither use nor found sets (labels and table are read through this CU's caches).
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ int certify_set(int cw_, uint32_t pp_, int slot_, int n_, int* lock_, int* nsets_) {
    const int wave = uni(cw_);
    const int lane = (int)(threadIdx.x & 63);
    const RCtx c = g_ctx[wave];
    const int w = uni(c.w), h = uni(c.h), n = uni(n_);
    const uint32_t pp = (uint32_t)uni((int)pp_);
    const uint32_t* const list = c.wslist + (size_t)uni(slot_) * uni(c.gcap);     // the first grow's list, kept for the cursor's validation
    uint32_t* const sets = c.sets;
    {   // the structure has its set already (another wavefront's evaluation of a neighbouring seed got here first): nothing to found
        const uint32_t lb0 = label_set(c.ltag, (uint32_t)uni((int)c.epochmap[pp]));
        if (lb0 && (uint32_t)uni((int)ld_l2(&sets[lb0])) != 0u) return 0;
    }
    int got = 0;
    if (lane == 0) got = atomicCAS(lock_, 0, 1) == 0 ? 1 : 0;                     // (busy: the next seed of the structure will offer again)
    if (!uni(got)) return 0;
    {   // (again under the lock: labels and table only change under it)
        const uint32_t lb0 = label_set(c.ltag, (uint32_t)uni((int)c.epochmap[pp]));
        if (lb0 && (uint32_t)uni((int)ld_l2(&sets[lb0])) != 0u) {
            if (lane == 0) __hip_atomic_store(lock_, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            return 0;
        }
    }
    int id = 0;
    if (lane == 0) { id = *nsets_ + 1; if (id <= kSetMax) *nsets_ = id; }
    id = uni(id);
    bool bad = id > kSetMax;
    if (!bad) {
        // (1) + labels, under the lock
        const double theta = c.deg[pp];
        for (int base = 0; base < n; base += 64) {
            const int k2 = base + lane;
            if (k2 < n) {
                const uint32_t pk = list[k2];
                const size_t q = (size_t)(pk >> 16) * w + (pk & 0xffffu);
                const uint32_t code = c.pw[q] & 3u;
                if (code != kPwFree && code != kPwRejected) bad = true;              // banned meanwhile (its word holds the line's epoch: hands off)
                else {
                    if (c.deg[q] != theta) bad = true;
                    const uint32_t lb = label_set(c.ltag, c.epochmap[q]);
                    if (lb && lb != (uint32_t)id) st_l2(&sets[lb], 0u);   // an older set loses a pixel: it ends
                    c.epochmap[q] = label_make(c.ltag, (uint32_t)id);
                }
            }
        }
        bad = ballot64(bad) != 0ull;
        if (lane == 0) st_l2(&sets[id], bad ? 0u : ((uint32_t)n | kSetPending));
        wg_fence();
    }
    if (lane == 0) __hip_atomic_store(lock_, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (bad) return 0;
    // (2) the growable neighbours outside S, without the lock: a pixel that fails now fails for good (bans only grow, angles never change)
    const float thf = __uint_as_float(c.pw[pp] & ~3u);
    const float lim = (float)g_tol0[0] + 1e-5f;
    const int e = lane >> 3, k8 = lane & 7, kk = k8 + (k8 >= 4);
    const int ox = kk % 3 - 1, oy = kk / 3 - 1;
    for (int base = 0; base < n; base += 8) {
        const int k2 = base + e;
        if (k2 < n) {
            const uint32_t pk = list[k2];
            const int nx = (int)(pk & 0xffffu) + ox, ny = (int)(pk >> 16) + oy;
            if (((unsigned)nx < (unsigned)w) & ((unsigned)ny < (unsigned)h)) {
                const size_t q = (size_t)ny * w + nx;
                const uint32_t wq = c.pw[q], lb = label_set(c.ltag, c.epochmap[q]);
                const uint32_t code = wq & 3u;
                if ((code == kPwFree || code == kPwRejected) && lb != (uint32_t)id) {
                    float d = fabsf(__uint_as_float(wq & ~3u) - thf);
                    if (d > (float)kPi) d = 2.0f * (float)kPi - d;
                    if (d < lim) bad = true;
                }
            }
        }
    }
    bad = ballot64(bad) != 0ull;
    int alive = 0;
    if (lane == 0) {
        if (bad) st_l2(&sets[id], 0u);
        else alive = atomicCAS(&sets[id], (uint32_t)n | kSetPending, (uint32_t)n) == ((uint32_t)n | kSetPending) ? 1 : 0;   // (0 meanwhile: a member was banned)
    }
    return uni(alive) ? id : 0;
}

// One seed's evaluation, RegionGrower ... RectangleImprover (:225-240), with all 64 lanes of the wave; pp = the seed's pixel.
// A speculative evaluation (spec != 0) also leaves in result slot `slot` what the commit at the cursor will need: the first
// grow's list and Refiner's regrow (their pixels decide whether the result is still valid at its turn), and the pixels to mark.
// The outcome goes to g_eo[wave] (all lanes store the same values), the rectangle stays in g_ws[wave].rec.
__device__ __noinline__ void eval_seed(int cw_, uint32_t pp_, int spec_, int slot_) {
    // What lives across this function's calls is wave-uniform and kept in scalar registers (uni()): values in VECTOR registers that
    // survive a call have to sit in callee-saved ones, which this function must then save and restore through scratch memory for ITS
    // caller -- 59 registers per evaluation before this was done, the largest single writer of the stage's HBM traffic.  The wave's
    // context is read from LDS where it is needed instead of being carried along.
    const int wave = uni(cw_);
    const int lane = (int)(threadIdx.x & 63);
    const int w = uni(g_ctx[wave].w), gcap = uni(g_ctx[wave].gcap);
    const uint32_t pp = (uint32_t)uni((int)pp_);
    const bool spec = uni(spec_) != 0;
    const int slot = uni(slot_);
    const double p_degThre = uni(g_par[0]), p_regThre = uni(g_par[1]), p_aliPro = uni(g_par[2]), p_denThre = uni(g_par[3]);
    const int sx = uni((int)(pp % (uint32_t)w)), sy = uni((int)(pp / (uint32_t)w));
    EvalOut& eo = g_eo[wave];
#define EVAL_CTX() RCtx c = g_ctx[wave]; c.lane = lane
#define EVAL_GL0() (g_ctx[wave].wslist + (size_t)slot * gcap)

    int outcome = 0, num = 0, num0 = 0, rec_pk = 0;
    double logNFA = 0;
    const bool skip = uni((int)(g_ctx[wave].pw[pp] & 3u)) != 0;   // monotone: once used, always used (:222)
    eo.setid = 0; eo.cert = 0;
    if (!skip && g_ctx[wave].sets) {
        // a seed of a live certified set: the evaluation is known (see "Certified uniform sets")
        const uint32_t lb = label_set((uint32_t)uni((int)g_ctx[wave].ltag), (uint32_t)uni((int)g_ctx[wave].epochmap[pp]));
        if (lb) {
            const uint32_t ns = (uint32_t)uni((int)ld_l2(&g_ctx[wave].sets[lb]));
            if (ns >= (uint32_t)kSetMinPixels && ns < kSetPending && (double)ns >= p_regThre) {
                RCtx c = g_ctx[wave]; c.lane = lane;
                STAT(ST_GROW, 2); STAT(ST_GROWN, ns + 1u);             // (RegionGrower's two calls of the reference, as the work counters count them)
                STAT(ST_SETHIT, 1);
                eo.skip = 0; eo.outcome = 1; eo.num = 1; eo.num0 = (int)ns; eo.rec_pk = 0; eo.logNFA = 0;
                eo.redo = 0; eo.precise = 0; eo.n1 = 0; eo.n2 = 0; eo.m_off = 0; eo.mcnt = 1;
                eo.x0 = 0; eo.y0 = 0; eo.x1 = -1; eo.y1 = -1;
                eo.setid = (int)lb;
                return;
            }
        }
    }
    int fx0 = 0x7fffffff, fy0 = 0x7fffffff, fx1 = -1, fy1 = -1;   // box of a first grow that refine() replaced
    bool sparse_by_margin = false, tol_zero = false;
    // list slot of a speculative evaluation: [first grow (n1)][Refiner's regrow (n2)][pixels to mark, if not one of those]
    int n1 = -1;                                       // -1: the lists are not kept (validation by bounding box only)
    bool regrown = false;
    if (!skip) {
        // RegionGrower -> RectangleConverter -> Refiner (:225-238) as a two-pass loop: pass 0 grows with the
        // global tolerance, pass 1 (only when the rectangle is too sparse, :829) regrows with the tolerance
        // re-estimated by Refiner (:833-857).
        const double seedDeg = uni(g_ctx[wave].deg[pp]);
        double tol = p_degThre, regdeg = seedDeg;
        bool done = false;
        for (int pass = 0; pass < 2 && !done; pass++) {
            num = uni(grow(wave, sx, sy, seedDeg, tol));                               // :225 / :857
            if (pass == 0 && spec && num <= gcap) {                // keep the first list for the validation at the cursor
                EVAL_CTX();
                uint32_t* const gl0 = EVAL_GL0();
                for (int k2 = lane; k2 < num; k2 += 64) gl0[k2] = lget(c, k2);
                n1 = num;
            }
            if (pass == 1) regrown = true;
            if (pass == 0) {
                num0 = num;
                if (num < p_regThre) { done = true; break; }                      // :228 (not marked, Q5)
            } else if (num < 2) { outcome = 1; done = true; break; }              // :861
            if (num > 1) { exact_sums(wave, num); regdeg = uni(atan2_g(g_ws[wave].ex_sin, g_ws[wave].ex_cos)); }   // reg.deg (:547, :581)
            else regdeg = seedDeg;
            rect_convert(wave, num, regdeg, p_aliPro, 0, p_degThre);                   // :232 / :866 (p, prec still the defaults)
            const double den = uni(rec_density(num, g_ws[wave].rec));
            if (fabs(den - p_denThre) <= kTieRel * p_denThre && !axis_exact(g_ws[wave].rec.dx, g_ws[wave].rec.dy))     // :829 / :869 within the libm's noise
                g_stat[wave][sslot(ST_TIES)] += TIE_UNIT(TS_DENS);
            if (pass == 0) {
                if (den >= p_denThre) break;                                      // :829 dense enough
                sparse_by_margin = den < p_denThre * (1.0 - 1e-3);
                if (spec) {                                                       // the regrow replaces this list
                    Box fb; fb.x0 = fx0; fb.y0 = fy0; fb.x1 = fx1; fb.y1 = fy1;
                    fb = list_bbox(wave, num, fb, false);
                    fx0 = uni(fb.x0); fy0 = uni(fb.y0); fx1 = uni(fb.x1); fy1 = uni(fb.y1);
                }
                tol = uni(refine_tol(wave, sx, sy, num, seedDeg));                     // :833-855
                tol_zero = tol == 0.0;
            } else if (den < p_denThre) {                                         // :869-877
                const int r = uni(radius_reduce(wave, sx, sy, num, regdeg, p_denThre));   // (lst reordered: gcopy holds the grow-order list)
                if (r < 0) { num = -r - 1; outcome = 1; done = true; }
                else num = r;
            }
        }
        if (!done) {
            logNFA = uni(improve(wave));                                               // :240
            outcome = logNFA <= 0 ? 2 : 3;                                        // :242
            rec_pk = uni(g_ws[wave].rec.pk);
        }
    }
    eo.skip = skip ? 1 : 0; eo.outcome = outcome; eo.num = num; eo.num0 = num0; eo.rec_pk = rec_pk; eo.logNFA = logNFA;
    eo.redo = 0; eo.precise = 0; eo.n1 = 0; eo.n2 = 0; eo.m_off = 0; eo.mcnt = num;
    // the way every seed of a uniform set goes: a sparse first region, a re-estimated tolerance of exactly 0, the seed alone, given up
    eo.cert = (spec && !skip && outcome == 1 && regrown && num == 1 && sparse_by_margin && tol_zero && n1 == num0 && num0 >= kSetMinPixels &&
               num0 <= 65535 && p_degThre < 1.5 && g_ctx[wave].sets != nullptr) ? 1 : 0;
    if (!spec || skip) return;

    const int gnum = uni(g_ws[wave].gnum);             // size of the last grow (grow order)
    const bool has_copy = uni(g_ws[wave].has_copy) != 0;
    // box of the pixels of this evaluation's grown lists
    {                                                  // (RegionRadiusReducer reordered/shrunk lst: the grow-order copy then)
        Box fb; fb.x0 = fx0; fb.y0 = fy0; fb.x1 = fx1; fb.y1 = fy1;
        fb = list_bbox(wave, gnum, fb, has_copy);
        eo.x0 = fb.x0; eo.y0 = fb.y0; eo.x1 = fb.x1; eo.y1 = fb.y1;
    }
    EVAL_CTX();
    uint32_t* const gl0 = EVAL_GL0();
    const unsigned long long ltm = (1ull << lane) - 1ull;
    bool precise = n1 >= 0;
    int n2 = 0;
    if (regrown) {                                     // keep Refiner's regrow (in grow order, before any reduction) behind the first list
        if (precise && n1 + gnum <= gcap) {
            for (int k2 = lane; k2 < gnum; k2 += 64) gl0[n1 + k2] = has_copy ? c.gcopy[k2] : lget(c, k2);
            n2 = gnum;
        } else precise = false;
    }
    if (n1 > 32767 || n2 > 32767) precise = false;     // (the sizes travel in 15-bit fields)
    eo.precise = precise ? 1 : 0; eo.n1 = n1; eo.n2 = n2;
    if (outcome <= 1) return;                          // nothing to mark
    // the pixels to mark
    int m_off = 0, mcnt = num;                         // not regrown: the first list is exactly the region
    bool redo = false;
    if (!regrown) {
        if (!precise) redo = true;                     // (larger than a list slot) evaluate again at the cursor
    } else if (precise && !has_copy) { m_off = n1; mcnt = n2; }            // the regrow as it is
    else {
        m_off = precise ? n1 + n2 : 0;
        if (m_off + gnum > gcap) { precise = false; m_off = 0; }
        if (gnum > gcap) redo = true;
        else {
            wg_fence();                                // (RegionRadiusReducer's removals from curMap must have landed)
            const uint32_t cur_id = g_ws[wave].cur_id;
            mcnt = 0;
            for (int base = 0; base < gnum; base += 64) {
                const int k2 = base + lane;
                uint32_t pkx = 0;
                bool keep = false;
                if (k2 < gnum) {
                    pkx = has_copy ? c.gcopy[k2] : lget(c, k2);
                    keep = !has_copy || tm_member(c, (int)(pkx & 0xffffu), (int)(pkx >> 16), cur_id);   // curMap == 1 only
                }
                const unsigned long long km = ballot64(keep);
                if (keep) gl0[m_off + mcnt + __builtin_popcountll(km & ltm)] = pkx;
                mcnt += __builtin_popcountll(km);
            }
        }
    }
    eo.precise = precise ? 1 : 0; eo.m_off = m_off; eo.mcnt = mcnt; eo.redo = redo ? 1 : 0;
#undef EVAL_CTX
#undef EVAL_GL0
}

// (developer build only: the seed loop can be cut short for the cost-probe experiments of DESIGN_NOTES.md; the product runs them all)
#ifdef LSD_REGION_STATS
__device__ __forceinline__ int seed_limit(int cnt, int stop) { return stop > 0 ? min(cnt, stop) : cnt; }
#else
__device__ __forceinline__ int seed_limit(int cnt, int) { return cnt; }
#endif
__device__ __forceinline__ int lds_ld(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Commit ring: one record per seed in flight (index = seed number & (RW-1)).
//   R_EMPTY  reserved by a wave (part of its chunk of seeds), not classified yet
//   R_SKIP   the seed pixel was already used when it was looked at (monotone, so final): nothing to do
//   R_LIGHT  a small region (:228, nothing to mark) grown by the small-region grower: aux = box of what it examined,
//            relative to the seed; whoever advances the cursor checks that no line accepted since the snapshot touches it
//   R_LIGHTL a full evaluation without marks (small region :228 or refine failed :237): aux = result slot (box and list
//            sizes in the slot table, the lists in the slot); checked like R_LIGHT, by the pixels themselves if the box is hit
//   R_BIG    the small-region grower gave the seed up (its region reaches regThre pixels, leaves the seed's window, or a
//            test was too close to call): waits for a full evaluation by any wave
//   R_EVAL   being evaluated in full, ahead of the cursor
//   R_STASH  evaluated with a result that marks usedMap: record and pixel list wait in the owner's result slot (aux);
//            whoever moves the cursor over it validates and commits it
//   R_REDO   a speculative result was invalidated (or abandoned): must be evaluated again at the cursor
//   R_BUSY   being evaluated at the cursor
//   R_REMOTE given to the wavefronts of other workgroups that help with this image (aux = request number); R_XLIGHTL / R_XSTASH:
//            their answers, R_LIGHTL / R_STASH with the lists and the record in the HELPER's result slot (aux = request number,
//            the slot and the box in the request table)
//   R_SETL   the seed belongs to a live certified set (aux = its label): the evaluation is known without growing anything (no marks);
//            valid at the cursor iff the set is still alive
enum { R_EMPTY = 0, R_SKIP = 1, R_LIGHT = 2, R_REDO = 3, R_BUSY = 4, R_STASH = 5, R_BIG = 6, R_EVAL = 7, R_LIGHTL = 8,
       R_REMOTE = 9, R_XLIGHTL = 10, R_XSTASH = 11, R_SETL = 12 };
constexpr int kHelpIdle = 50;             // looks without a request after which a helper wavefront leaves an image
constexpr int RW = 256 * NW;              // records in flight: how far the hand-out may run ahead of the cursor
constexpr int CH = 32;                    // seeds a wave reserves at a time (its chunk)

struct Ring {
    alignas(4) uint8_t state[RW];
    uint16_t snap[RW];   // accept epoch (mod 2^16) the result was computed against
    uint32_t aux[RW];    // see above
};
struct SlotTab {         // per result slot (wave * NS + slot) of a full evaluation published as R_LIGHTL
    short box[NW * NS][4];
    uint32_t lcnt[NW * NS];   // n1 + 1 | n2 << 16 (n1 + 1 == 0: the lists were not kept, box check only)
};
__device__ __forceinline__ int st_ld(uint8_t* p) { return (int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_st(uint8_t* p, int v) { __hip_atomic_store(p, (uint8_t)v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// compare-and-swap of ONE state byte (call it from one lane): the containing word is swapped, and the swap is retried as long
// as only the other three bytes of the word have changed in between
__device__ __forceinline__ bool st_cas(uint8_t* base, int idx, int expect, int desired) {
    uint32_t* const wp = reinterpret_cast<uint32_t*>(base) + (idx >> 2);
    const int sh = (idx & 3) * 8;
    while (true) {
        const uint32_t oldw = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (((oldw >> sh) & 0xffu) != (uint32_t)expect) return false;
        const uint32_t neww = (oldw & ~(0xffu << sh)) | ((uint32_t)desired << sh);
        if (atomicCAS(wp, oldw, neww) == oldw) return true;
    }
}

// minimum / maximum over the 8 lanes of a group for small non-negative integers (exact in fp32)
__device__ __forceinline__ int imin8(int v) { return (int)min8((float)v); }
__device__ __forceinline__ int imax8(int v) { return -(int)min8(-(float)v); }

// One image's region stage by the calling workgroup; `bid` is the workgroup's place in the launch's image order (blockIdx.x of the
// one-workgroup-per-image launch; the next number of the launch's counter for a persistent workgroup, see k_region below).
__device__ __forceinline__ void region_image(const Geom& g, const Buffers& b, uint32_t id_base, const int bid, const int nimg) {
    __shared__ int s_next, s_commit, s_epoch, s_lines, s_ntrace, s_nseeds, s_lock, s_nbig, s_depth, s_abort, s_nsets;
    __shared__ int s_scan[NW];                              // the waves' counts of potential seeds (the seed scan at the start)
    __shared__ short s_ring[RING][4];
    __shared__ Ring rg;
    __shared__ SlotTab stab;
    // seeds given to helpers: request j holds seed s_xk[j] (-1: free, -2: answered -- the answer, in the image's record of the
    // help protocol in HBM, waits for the cursor)
    __shared__ int s_nhelp, s_xout, s_xlock, s_xreg, s_xpub, s_idlecnt, s_xc_last, s_xt_last, s_workbound;
    __shared__ int s_xk[kXReq];

    // The last b.npool workgroups of the launch own no image: they are HELPERS from the start (the host adds them when the images
    // leave workgroup slots of the device free, see "Help from other workgroups" below), with a workspace slot of their own.
    const bool pool = bid >= nimg;
    const size_t img = pool ? (size_t)bid : (size_t)b.order[bid];                         // heaviest images first (k_order); pool: its workspace slot
    uint32_t* const xr = (b.xq && !pool) ? b.xq + img * (size_t)kXStride : nullptr;        // this image's record of the help protocol
    uint32_t* const xhdr = b.xq ? b.xq + (size_t)nimg * kXStride : nullptr;                 // ... and the launch's
    if (threadIdx.x == 0 && xhdr && !pool) atomicAdd(&xhdr[0], 1u);
    // when the launch's first workgroup started (s_memrealtime, 100 MHz, the same on every CU; low 32 bits | 1): what "running long"
    // is measured against (below)
    if (threadIdx.x == 0 && xhdr) atomicCAS(&xhdr[4], 0u, (uint32_t)__builtin_amdgcn_s_memrealtime() | 1u);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = g.w, h = g.h;
    const size_t npx = (size_t)g.npx;

    RCtx c;
    c.w = w; c.h = h; c.lane = lane; c.wave = wave;
    c.mag = b.mag + img * npx; c.deg = b.deg + img * npx; c.pw = b.pw + img * npx; c.epochmap = b.epochmap + img * npx;
    c.sets = (pool || !b.sets) ? nullptr : b.sets + img * (size_t)(kSetMax + 1);
    c.ltag = (id_base >> 20) & 0x3ffu;
    c.tep = b.tepoch + img * (size_t)(((w + 7) >> 3) * ((h + 7) >> 3));
    c.sc = b.sc + img * npx;
    c.tmask = b.stamps + (img * NW + wave) * (size_t)b.tm_stride;
    c.spill = b.spill + (img * NW + wave) * npx; c.gcopy = b.gcopy + (img * NW + wave) * npx;
    c.meta = b.wmeta + (img * NW + wave) * (size_t)b.mcap; c.mcap = b.mcap;
    c.tilesX = (w + 7) >> 3; c.id_base = id_base; c.id_budget = b.id_budget;
    if (lane == 0) {
        WState& ws = g_ws[wave];
        ws.cur_id = id_base; ws.gnum = 0; ws.has_copy = 0; ws.tm_pending = 0; ws.cache_epoch = -1; ws.members_cached = 0;
        ws.ex_upto = 0; ws.ex_sin = 0; ws.ex_cos = 0;
    }
    c.logNT = g.logNT; c.lgamma = b.lgamma; c.lg_count = b.lg_count; c.ptab = b.ptab;
    c.wslist = b.slist + (img * NW + wave) * (size_t)NS * b.gcap; c.gcap = b.gcap;
    if (threadIdx.x == 0) { g_par[0] = g.degThre; g_par[1] = g.regThre; g_par[2] = g.aliPro; g_par[3] = g.denThre; }
    if (lane == 0) g_ctx[wave] = c;                        // (c.lane is set by every reader)
    if (lane < kStatSlots) g_stat[c.wave][lane] = 0ull;
    const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
    [[maybe_unused]] const long long rt_begin = (long long)__builtin_amdgcn_s_memrealtime();   // (100 MHz, the same on every CU)
    if (lane < NT) g_ttag[c.wave][lane] = -1;
    for (int j = threadIdx.x; j < RW / 4; j += 64 * NW) reinterpret_cast<uint32_t*>(rg.state)[j] = 0u;    // R_EMPTY
    if (threadIdx.x < kXReq) s_xk[threadIdx.x] = -1;
    if (c.sets) for (int j = threadIdx.x; j <= kSetMax; j += 64 * NW) st_l2(&c.sets[j], 0u);   // no certified set yet (labels of earlier launches carry another tag)
    if (!pool) {
        // this image's counter record and tile epochs start from zero (cleared here, not by fills in front of the launch: every dispatch
        // of a batch in flight waits for a hardware pipe; lsd_ctx.hip).  Nobody touches them before the barrier behind the seed scan below;
        // helpers of other workgroups only after this image has asked for them.
        if (b.stats) for (int j = threadIdx.x; j < kStatWords; j += 64 * NW) b.stats[img * kStatWords + j] = 0ll;
        const int ntile = c.tilesX * ((h + 7) >> 3);
        for (int j = threadIdx.x; j < ntile; j += 64 * NW) c.tep[j] = 0u;
    }

    const uint32_t* ord = b.ord + img * npx;
    uint32_t* seedidx = b.seedidx + img * npx;
    uint32_t* seedpos = b.seedpos + img * npx;
    const int nb = pool ? 0 : b.nb[img];
    double* recs = b.recs + img * (size_t)b.max_lines * 12;
    double* recs_scaled = b.recs_scaled + img * (size_t)b.max_lines * 4;
    SeedRec* trace = b.seeds ? reinterpret_cast<SeedRec*>(b.seeds) + img * npx : nullptr;
    int* rnum = b.rnum + img * (size_t)RW * 2;            // (num0, final_num << 2 | outcome) of published records: read by the seed trace only

    if (wave == NW - 1) {
        double st, ct;
        sincos_g(g.degThre < 1.5 ? g.degThre : 1.0, st, ct);
        if (lane == 0) { g_tol0[0] = g.degThre; g_tol0[1] = st; g_tol0[2] = ct; }
    }
    // potential seeds: sorted entries whose pixel is not below the gradient threshold (usedMap == 0 after K2), in sorted order.  Every
    // wave takes a contiguous share of the list, four chunks of 64 entries per trip to memory (an entry costs two dependent reads:
    // its position, then that pixel's word): first the counts, then -- the shares' offsets known -- the same walk again, writing
    // (one wave walking the whole list chunk by chunk took ~0.8 ms of a 2048 x 2048 map's 9).
    {
        const int nchunks = (nb + 63) >> 6, per = (nchunks + NW - 1) / NW;
        const int c0 = min(wave * per, nchunks), c1 = min(c0 + per, nchunks);
        const unsigned long long lt = (1ull << lane) - 1ull;
        int cnt = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (int ch = c0; ch < c1; ch += 4) {
                uint32_t pq[4], cw[4];
                bool in[4];
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int idx = (ch + u) * 64 + lane;
                    in[u] = ch + u < c1 && idx < nb;
                    pq[u] = ord[in[u] ? idx : 0];
                }
                #pragma unroll
                for (int u = 0; u < 4; u++) cw[u] = c.pw[pq[u]];
                #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool ok = in[u] && (cw[u] & 3u) == 0u;                   // :222
                    const unsigned long long m = ballot64(ok);
                    if (pass == 1 && ok) { const int o = cnt + __builtin_popcountll(m & lt); seedidx[o] = (uint32_t)((ch + u) * 64 + lane); seedpos[o] = pq[u]; }
                    cnt += __builtin_popcountll(m);
                }
            }
            if (pass == 0) {
                if (lane == 0) s_scan[wave] = cnt;
                __syncthreads();
                int off = 0, tot = 0;
                for (int v = 0; v < NW; v++) { const int t = s_scan[v]; off += v < wave ? t : 0; tot += t; }
                cnt = off;
                if (wave == 0 && lane == 0) { s_next = 0; s_commit = 0; s_epoch = 0; s_lines = 0; s_ntrace = 0; s_nseeds = seed_limit(tot, b.tun_stop); s_lock = 0; s_nbig = 0; s_depth = min(max(b.tun_soft, 2 * CH), RW - 128); s_abort = 0; s_nsets = 0; s_nhelp = 0; s_xout = 0; s_xlock = 0; s_xreg = 0; s_xpub = 0; s_idlecnt = 0; s_xc_last = 0; s_xt_last = (int)__builtin_amdgcn_s_memtime(); s_workbound = 0; }
            }
        }
        wg_fence();
        __syncthreads();
        if (xr) {                                           // helpers on other XCDs read seedpos[] as soon as a request names a seed
            if (wave == 0) agent_release();
            __syncthreads();
        }
    }
    const int nseeds = s_nseeds;

    // How far the hand-out and the full evaluations may run ahead of the cursor (s_depth, in seeds).  The further ahead a region is
    // evaluated, the likelier a line accepted before its turn makes the work void (typical maps: many lines, short evaluations);
    // but where single evaluations take a millisecond (long sparse structures grown again and again without ever marking
    // anything) a short look-ahead leaves the other waves without work.  So the depth adapts: it grows while waves find nothing
    // to do within it, and shrinks when a speculative result is redone or discarded at the cursor.  Ring slots are reused no
    // sooner than 128 commits later.
    const int depth_min = min(max(b.tun_soft, 2 * CH), RW - 128), depth_max = min(max(b.tun_claim, depth_min), RW - 128);
    const int kDepthUp = b.tun_up, kDepthDown = b.tun_down;   // steps of the adaptive look-ahead
    const int kFeed = min(max(b.tun_feed, 1), 8);           // idle groups that make a wave fetch the windows of its next seeds (a memory round trip)
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
                const uint32_t pp = seedpos[k];
                SeedRec tr;
                tr.order_idx = oidx; tr.x = (int)(pp % (uint32_t)w); tr.y = (int)(pp / (uint32_t)w);
                tr.num = num0; tr.outcome = outcome; tr.final_num = fnum; tr.logNFA = logNFA;
                trace[s_ntrace] = tr;
            }
            s_ntrace = s_ntrace + 1;
        }
    };
    // True if a MEMBER of a region's grown lists was banned by a line accepted in epoch >= snap: only then can the region's
    // evaluation differ from what it would be now.  (It read usedMap only as "banned?" of candidate pixels, :537.  A pixel it
    // examined and did NOT take -- the angle test failed every time it came up -- is skipped now instead of failing: the same
    // sequence of accepts, the same sums, the same lists.  A pixel it took would now be skipped: a different region.  So the
    // neighbours of the lists do not matter, the lists do.)  Accepted pixels carry their line's epoch + 1 in epochmap.
    auto examined_hit = [&](const uint32_t* lp, int cnt, int snap) -> bool {
        // first by tiles (tep[] is a few KB and stays in the cache): no line accepted since the snapshot has a pixel in any tile
        // a member lies in
        const int tX = c.tilesX;
        bool thit = false;
        for (int base = 0; base < cnt; base += 64) {
            const int k2 = base + lane;
            if (k2 < cnt) {
                const uint32_t pkx = lp[k2];
                const int x = (int)(pkx & 0xffffu), y = (int)(pkx >> 16);
                if ((int)ld_l2(&c.tep[(y >> 3) * tX + (x >> 3)]) > snap) thit = true;
            }
        }
        if (!ballot64(thit)) return false;
        bool hit = false;
        for (int base = 0; base < cnt; base += 64) {
            const int k2 = base + lane;
            if (k2 < cnt) {
                const uint32_t pkx = lp[k2];
                const size_t q = (size_t)(pkx >> 16) * w + (pkx & 0xffffu);
                if ((c.pw[q] & 3u) == kPwLine && (int)c.epochmap[q] > snap) hit = true;
            }
        }
        return ballot64(hit) != 0ull;
    };
    // the same question for a box: true if a tile overlapping it holds a pixel of a line accepted in epoch >= snap
    auto box_tile_hit = [&](int snap, int x0, int y0, int x1, int y1) -> bool {
        const int tX = c.tilesX;
        const int xa = max(x0, 0) >> 3, xb = min(x1, w - 1) >> 3, ya = max(y0, 0) >> 3, yb = min(y1, h - 1) >> 3;
        bool hit = false;
        for (int ty = ya; ty <= yb; ty++)
            for (int tx = xa; tx <= xb; tx++)
                if ((int)ld_l2(&c.tep[ty * tX + tx]) > snap) hit = true;
        return hit;
    };
    // Commits a result that marks usedMap (accepted line: code 3 + epoch; rejected region: code 2), at the cursor, under
    // the cursor lock: pv = lane j < 12: field j of the rectangle (structRec order); m_src: the pixels to mark (null: this
    // wave's own last grow).
    // A line has just been accepted (seed k, box mb).  A finished result further ahead that it invalidates would be found
    // invalid when the cursor reaches it and evaluated again THERE, with every other wave waiting (images with many lines on
    // the same structures spent 45 % of their time in such evaluations).  Found now, it goes back to the seeds that wait for a
    // full evaluation and is redone by whichever wave is free while the cursor works its way towards it.  (Called by the wave
    // that owns the cursor; nobody else touches a finished record.)
    auto requeue_ahead = [&](int k, const Box& mb) {
        const int lim = min(lds_ld(&s_next), nseeds);
        const int now = lds_ld(&s_epoch);
        for (int base = k + 1; base < lim; base += 64) {
            const int idx = base + lane;
            const int rr = idx & (RW - 1);
            const int stl = idx < lim ? st_ld(&rg.state[rr]) : R_EMPTY;
            int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
            if (stl == R_LIGHT) {
                const uint32_t ax = rg.aux[rr], sp = seedpos[idx];
                const int sxp = (int)(sp % (uint32_t)w), syp = (int)(sp / (uint32_t)w);
                x0 = sxp + (int)(ax & 63u) - 32; y0 = syp + (int)((ax >> 6) & 63u) - 32;
                x1 = sxp + (int)((ax >> 12) & 63u) - 32; y1 = syp + (int)((ax >> 18) & 63u) - 32;
            } else if (stl == R_LIGHTL) {
                const uint32_t ax = rg.aux[rr];
                x0 = stab.box[ax][0]; y0 = stab.box[ax][1]; x1 = stab.box[ax][2]; y1 = stab.box[ax][3];
            } else if (stl == R_STASH) {
                const double* P = b.pend + (img * (size_t)(NW * NS) + rg.aux[rr]) * 24;
                x0 = (int)P[18]; y0 = (int)P[19]; x1 = (int)P[20]; y1 = (int)P[21];
            }
            if (stl == R_SETL && ld_l2(&c.sets[rg.aux[rr]]) == 0u) {           // its set ended (with this line, or earlier): a full evaluation then
                st_st(&rg.state[rr], R_BIG); atomicAdd(&s_nbig, 1);
                DSTAT(ST_DEPTHUP, 1);
            }
            unsigned long long hm = ballot64(!(mb.x1 < x0 || mb.x0 > x1 || mb.y1 < y0 || mb.y0 > y1) && x1 >= x0);
            while (hm) {
                const int l = __builtin_ctzll(hm);
                hm &= hm - 1ull;
                const int st1 = __builtin_amdgcn_readlane(stl, l), r1 = (base + l) & (RW - 1);
                const uint32_t ax1 = rg.aux[r1];
                const int snap = now - ((now - (int)rg.snap[r1]) & 0xffff);
                bool conflict = true;
                if (st1 == R_LIGHT) {
                    conflict = box_tile_hit(snap, __builtin_amdgcn_readlane(x0, l), __builtin_amdgcn_readlane(y0, l), __builtin_amdgcn_readlane(x1, l), __builtin_amdgcn_readlane(y1, l));
                } else {
                    const size_t gs = img * (size_t)(NW * NS) + ax1;
                    int n = -1;
                    if (st1 == R_LIGHTL) { const uint32_t lc = stab.lcnt[ax1]; if (lc & 0xffffu) n = (int)(lc & 0xffffu) - 1 + (int)(lc >> 16); }
                    else { const long long pk3 = (long long)b.pend[gs * 24 + 22]; const int n1 = (int)(pk3 % 32768ll) - 1; if (n1 >= 0) n = n1 + (int)((pk3 / 32768ll) % 32768ll); }
                    if (n >= 0) conflict = examined_hit(b.slist + gs * b.gcap, n, snap);
                }
                if (conflict) {
                    if (lane == 0) { st_st(&rg.state[r1], R_BIG); atomicAdd(&s_nbig, 1); }
                    DSTAT(ST_DEPTHUP, 1);
                }
            }
        }
    };
    auto commit_marks = [&](int k, int num0, int fnum, int outcome, double logNFA, double pv, const uint32_t* m_src, int m_cnt) {
        write_trace(k, num0, fnum, outcome, logNFA);
        if (outcome == 2) {                                                          // :242-250
            (void)mark_region(c.wave, 0u, m_src, m_cnt);
        } else if (outcome == 3) {
            const int li = s_lines;
            if (li < b.max_lines) {
                if (lane < 12) recs[(size_t)li * 12 + lane] = pv;                    // structRec as accepted
                if (lane < 4) recs_scaled[(size_t)li * 4 + lane] = g.sca != 1 ? (pv - 1.0) / g.sca + 1 : pv;   // x1 y1 x2 y2, :252-258
            }
            const Box mb = mark_region(c.wave, (uint32_t)(lds_ld(&s_epoch) + 1), m_src, m_cnt);   // :259-265 (+ the line's epoch)
            wg_fence();                                   // the marks must be visible before the epoch moves
            if (xr) agent_release();                      // ... to the helpers on other CUs as well
            if (lane == 0) {
                const int ep = s_epoch;
                short* r = s_ring[ep & (RING - 1)];
                r[0] = (short)mb.x0; r[1] = (short)mb.y0; r[2] = (short)mb.x1; r[3] = (short)mb.y1;
                s_lines = li + 1;
                lds_st(&s_epoch, ep + 1);
                if (xr) st_l2(&xr[0], (uint32_t)(ep + 1));
            }
            invalidate_tiles(c);                          // this wave's cached ban flags are stale now
            g_ws[wave].cache_epoch = -1;
            wg_fence();
            if (b.tun_requeue) requeue_ahead(k, mb);
        }
        wg_fence();                                       // marks + ring visible before the cursor moves
    };
    // Moves the commit cursor over finished records (one wave at a time, whichever comes by): skipped seeds, results
    // without marks (after checking that they are still valid), and STASHED results of ANY wave -- everything a commit
    // needs sits in the owner's result slot in HBM, so the cursor never waits for an owner that is busy with a long
    // speculative evaluation further ahead.
    auto advance = [&]() {
        int got = 0;
        if (lane == 0) got = atomicCAS(&s_lock, 0, 1) == 0 ? 1 : 0;
        got = __builtin_amdgcn_readfirstlane(got);
        if (!got) return;
        while (true) {
            int f = lds_ld(&s_commit);
            if (f >= nseeds) break;
            if (!trace) {
                // a run of up to 64 records that need nothing but the cursor's nod (skipped seeds, results without marks that no
                // line accepted since their snapshot can have touched), one record per lane
                const int now = lds_ld(&s_epoch);
                const int idx = f + lane;
                const int rr = idx & (RW - 1);
                const int stl = idx < nseeds ? st_ld(&rg.state[rr]) : R_EMPTY;
                const uint32_t sp = idx < nseeds ? seedpos[idx] : 0u;           // (used only when a box has to be placed)
                bool ok = stl == R_SKIP;
                if (stl == R_SETL) ok = ld_l2(&c.sets[rg.aux[rr]]) != 0u;          // (alive: no member of the set was ever banned)
                if (stl == R_LIGHT || stl == R_LIGHTL) {
                    const int d = (now - (int)rg.snap[rr]) & 0xffff;
                    ok = d == 0;
                    if (!ok) {
                        const uint32_t ax = rg.aux[rr];
                        int x0, y0, x1, y1;
                        if (stl == R_LIGHT) {
                            const int sxp = (int)(sp % (uint32_t)w), syp = (int)(sp / (uint32_t)w);
                            x0 = sxp + (int)(ax & 63u) - 32; y0 = syp + (int)((ax >> 6) & 63u) - 32;
                            x1 = sxp + (int)((ax >> 12) & 63u) - 32; y1 = syp + (int)((ax >> 18) & 63u) - 32;
                        } else { x0 = stab.box[ax][0]; y0 = stab.box[ax][1]; x1 = stab.box[ax][2]; y1 = stab.box[ax][3]; }
                        // (the tile test only for the small boxes: a full evaluation's box may span the image, its lists say more)
                        ok = !hit_since(now - d, now, x0, y0, x1, y1) || (stl == R_LIGHT && !box_tile_hit(now - d, x0, y0, x1, y1));
                    }
                }
                const unsigned long long okm = ballot64(ok);
                const int run = okm == ~0ull ? 64 : __builtin_ctzll(~okm);
                if (run > 0) {
                    const unsigned long long runm = run == 64 ? ~0ull : (1ull << run) - 1ull;
                    const int nl = __builtin_popcountll(ballot64(stl != R_SKIP) & runm);
                    if (lane < run) rg.state[rr] = (uint8_t)R_EMPTY;
                    if (lane == 0) { s_ntrace = s_ntrace + nl; lds_st(&s_commit, f + run); }
                    f += run;
                    if (run == 64 || f >= nseeds) continue;
                }
            }
            const int r = f & (RW - 1);
            const int st = st_ld(&rg.state[r]);
            if (st == R_SKIP) {
                if (lane == 0) { rg.state[r] = (uint8_t)R_EMPTY; lds_st(&s_commit, f + 1); }
                continue;
            }
            if (st == R_LIGHT || st == R_LIGHTL || st == R_XLIGHTL) {
                const int now = lds_ld(&s_epoch);
                const int d = (now - (int)rg.snap[r]) & 0xffff, snap = now - d;
                const uint32_t ax = rg.aux[r];
                if (d != 0) {
                    int x0, y0, x1, y1;
                    if (st == R_LIGHT) {
                        const uint32_t sp = seedpos[f];
                        const int sxp = (int)(sp % (uint32_t)w), syp = (int)(sp / (uint32_t)w);
                        x0 = sxp + (int)(ax & 63u) - 32; y0 = syp + (int)((ax >> 6) & 63u) - 32;
                        x1 = sxp + (int)((ax >> 12) & 63u) - 32; y1 = syp + (int)((ax >> 18) & 63u) - 32;
                    } else if (st == R_LIGHTL) { x0 = stab.box[ax][0]; y0 = stab.box[ax][1]; x1 = stab.box[ax][2]; y1 = stab.box[ax][3]; }
                    else {
                        const uint32_t b0 = ld_l2(&xr[3 * kXReq + ax * 8 + 3]), b1 = ld_l2(&xr[3 * kXReq + ax * 8 + 4]);
                        x0 = (int)(b0 & 0xffffu); y0 = (int)(b0 >> 16); x1 = (int)(b1 & 0xffffu); y1 = (int)(b1 >> 16);
                    }
                    if (hit_since(snap, now, x0, y0, x1, y1) && (st != R_LIGHT || box_tile_hit(snap, x0, y0, x1, y1))) {
                        bool conflict = true;
                        const uint32_t lc = st == R_LIGHTL ? stab.lcnt[ax] : st == R_XLIGHTL ? ld_l2(&xr[3 * kXReq + ax * 8 + 5]) : 0u;
                        if ((lc & 0xffffu) != 0u) {   // the lists are still in their slot: look at the pixels themselves
                            const size_t gs = st == R_LIGHTL ? img * (size_t)(NW * NS) + ax : (size_t)ld_l2(&xr[3 * kXReq + ax * 8 + 1]);
                            wg_fence();
                            conflict = examined_hit(b.slist + gs * b.gcap, (int)(lc & 0xffffu) - 1 + (int)(lc >> 16), snap);
                        }
                        if (conflict) {
                            STAT(ST_REDO, 1);
                            if (lane == 0) {
                                if (st == R_XLIGHTL) s_xk[ax] = -1;
                                st_st(&rg.state[r], R_REDO); lds_st(&s_depth, max(depth_min, lds_ld(&s_depth) - kDepthDown));
                            }
                            break;
                        }
                    }
                }
                if (st == R_XLIGHTL && lane == 0) s_xk[ax] = -1;
                bool used_now = false;
                if (trace) used_now = (c.pw[seedpos[f]] & 3u) != 0u;          // the reference skips it then (:222): no record
                if (trace && !used_now) {
                    const int no = rnum[r * 2 + 1];
                    write_trace(f, rnum[r * 2], no >> 2, no & 3, 0.0);
                } else if (!trace) write_trace(f, 0, 0, 0, 0.0);
                if (lane == 0) { rg.state[r] = (uint8_t)R_EMPTY; lds_st(&s_commit, f + 1); }
                continue;
            }
            if (st == R_SETL) {
                if (ld_l2(&c.sets[rg.aux[r]]) == 0u) {                             // the set ended before the seed's turn: evaluate in full, here
                    STAT(ST_REDO, 1);
                    if (lane == 0) { st_st(&rg.state[r], R_REDO); lds_st(&s_depth, max(depth_min, lds_ld(&s_depth) - kDepthDown)); }
                    break;
                }
                bool used_now = false;
                if (trace) used_now = (c.pw[seedpos[f]] & 3u) != 0u;               // the reference skips it then (:222): no record
                if (trace && !used_now) {
                    const int no = rnum[r * 2 + 1];
                    write_trace(f, rnum[r * 2], no >> 2, no & 3, 0.0);
                } else if (!trace) write_trace(f, 0, 0, 0, 0.0);
                if (lane == 0) { rg.state[r] = (uint8_t)R_EMPTY; lds_st(&s_commit, f + 1); }
                continue;
            }
            if (st == R_STASH || st == R_XSTASH) {
                // ---- a stashed result at the cursor: is it still what the sequential run would get? ----
                const uint32_t ax = rg.aux[r];
                const size_t lr = st == R_STASH ? img * (size_t)(NW * NS) + ax : (size_t)ld_l2(&xr[3 * kXReq + ax * 8 + 1]);   // the result slot, of this image's waves or of a helper's
                wg_fence();
                if (st == R_XSTASH && lane == 0) s_xk[ax] = -1;                   // (whatever happens below, the request is over)
                const double pv = b.pend[lr * 24 + (lane < 24 ? lane : 0)];
                const double logNFA = rl(pv, 12);
                const int outcome = (int)rl(pv, 14), num0 = (int)rl(pv, 15), num = (int)rl(pv, 16), m_cnt = (int)rl(pv, 17);
                const int x0 = (int)rl(pv, 18), y0 = (int)rl(pv, 19), x1 = (int)rl(pv, 20), y1 = (int)rl(pv, 21);
                const int snap = (int)rl(pv, 23);
                const long long pk3 = (long long)rl(pv, 22);
                const int st_n1 = (int)(pk3 % 32768ll) - 1, st_n2 = (int)((pk3 / 32768ll) % 32768ll);
                const uint32_t* st_list = b.slist + lr * b.gcap;
                const uint32_t* m_src = st_list + (int)(pk3 / (32768ll * 32768ll));
                if ((c.pw[seedpos[f]] & 3u) != 0u) {       // an earlier seed marked the pixel meanwhile: the reference skips it (:222)
                    STAT(ST_DISCARD, 1);
                    if (lane == 0) { rg.state[r] = (uint8_t)R_EMPTY; lds_st(&s_commit, f + 1); lds_st(&s_depth, max(depth_min, lds_ld(&s_depth) - kDepthDown)); }
                    continue;
                }
                const int now = lds_ld(&s_epoch);
                if (now != snap && hit_since(snap, now, x0, y0, x1, y1)) {
                    bool conflict = true;
                    if (st_n1 >= 0) conflict = examined_hit(st_list, st_n1 + st_n2, snap);   // the pixels themselves
                    if (conflict) {
                        STAT(ST_REDO, 1);
                        if (lane == 0) { st_st(&rg.state[r], R_REDO); lds_st(&s_depth, max(depth_min, lds_ld(&s_depth) - kDepthDown)); }     // evaluate again; everything earlier is committed now
                        break;
                    }
                }
                commit_marks(f, num0, num, outcome, logNFA, pv, m_src, m_cnt);
                if (lane == 0) { rg.state[r] = (uint8_t)R_EMPTY; lds_st(&s_commit, f + 1); }
                continue;
            }
            break;
        }
        if (xr && lane == 0) st_l2(&xr[1], (uint32_t)lds_ld(&s_commit));     // (helpers free their result slots behind the cursor)
#ifdef LSD_REGION_MILESTONES
        if (lane == 0 && b.stats) {   // developer experiment: when the cursor passed 1/16, 1/8, 1/4, 1/2 of the seeds
            const int fc = lds_ld(&s_commit);
            for (int m = 0; m < 4; m++)
                if (fc >= (nseeds >> (4 - m)) && b.stats[img * kStatWords + 24 + m] == 0) b.stats[img * kStatWords + 24 + m] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
        }
#endif
        if (lane == 0) lds_st(&s_lock, 0);
    };

    // ---- Help from other workgroups ----
    // The batch ends with its heaviest images: long sparse structures regrown from hundreds of seeds, millisecond evaluations that
    // are independent of one another, on one CU each while the CUs that have finished their images idle.  So a workgroup whose
    // image is done -- once every workgroup of the launch has started -- turns its wavefronts into HELPERS (the loop below, second
    // half): each attaches itself to an image that has asked for help, takes seeds that wait for a full evaluation from that
    // image's request table in HBM, evaluates them speculatively against that image's arrays with its own workspace, and answers
    // with what a local evaluation publishes: lists and record in the helper's own result slot, the rest in an 8-word message.
    // The owner's cursor validates and commits such an answer exactly like a local result, so nothing of the order of decisions
    // changes.  What crosses CUs goes through L2 (agent-scope atomics) behind release / acquire fences: bans and their epoch from
    // the owner to the helper, lists and records back.  A request nobody has taken when the cursor reaches it is taken back.
    auto xlock = [&]() -> bool {
        int got = 0;
        if (lane == 0) got = atomicCAS(&s_xlock, 0, 1) == 0 ? 1 : 0;
        return __builtin_amdgcn_readfirstlane(got) != 0;
    };
    auto xunlock = [&]() { wg_fence(); if (lane == 0) lds_st(&s_xlock, 0); };
    // takes the answers that have arrived into the ring; learns how many helpers the image has and tells them its backlog
    auto xpoll = [&]() -> bool {
        if (!xlock()) return false;
        const int kx = lane < kXReq ? s_xk[lane] : -1;
        const uint32_t fl = kx >= 0 ? ld_l2(&xr[2 * kXReq + lane]) : 0u;
        const bool ans = kx >= 0 && fl == (uint32_t)(kx + 1);
        const unsigned long long am = ballot64(ans);
        if (am) {
            agent_acquire();                               // the answer, and the lists and the record behind it
            if (ans) {
                const uint32_t* m = xr + 3 * kXReq + lane * 8;         // (slot, box and list sizes stay there until the cursor comes by)
                const uint32_t res = ld_l2(&m[0]), sn = ld_l2(&m[2]);
                const int r = kx & (RW - 1);
                rg.snap[r] = (uint16_t)sn; rg.aux[r] = (uint32_t)lane;
                st_l2(&xr[2 * kXReq + lane], 0u);
                s_xk[lane] = (res == (uint32_t)R_SKIP || res == (uint32_t)R_REDO) ? -1 : -2;
                st_st(&rg.state[r], (int)res);             // (a release: after the table entries)
            }
            if (lane == 0) atomicSub(&s_xout, __builtin_popcountll(am));
        }
        if (lane == 0) {
            // Help pays where the waves are busy evaluating (an image whose waves wait for the cursor gains nothing from more
            // evaluators): the share of the last ~100 us the waves spent in the idle path below decides whether the image asks
            const int tn = (int)__builtin_amdgcn_s_memtime(), dt = tn - s_xt_last;
            if (dt > 200000) {
                const int ic = lds_ld(&s_idlecnt);
                // ... and only an image that has been running for a while asks at all: at least tun_gate (x 1024 clocks), and at least
                // tun_share percent of the time since the launch began -- the typical image is through before help could pay for the
                // traffic it causes; the ones that have been running for most of the launch are the ones it will end on
                const uint32_t rt_now = (uint32_t)__builtin_amdgcn_s_memrealtime();
                const uint32_t since_launch = rt_now - (ld_l2(&xhdr[4]) & ~1u), mine = rt_now - (uint32_t)rt_begin;
                s_workbound = ((long long)(ic - s_xc_last) * (64 * LSD_REGION_WAIT_SLEEP + 1000) * 100 < (long long)dt * NW * b.tun_wb &&
                               (long long)__builtin_amdgcn_s_memtime() - t_begin > (long long)b.tun_gate * 1024 &&
                               (unsigned long long)mine * 100ull >= (unsigned long long)since_launch * (unsigned)b.tun_share) ? 1 : 0;   // idle < tun_wb %
                s_xc_last = ic; s_xt_last = tn;
            }
            const int nbg = s_workbound ? max(lds_ld(&s_nbig), 0) : 0;
            s_nhelp = (int)ld_l2(&xr[3]);
            st_l2(&xr[4], (uint32_t)nbg);
            s_xpub = nbg;
            if (!s_xreg && nbg >= NW) {                    // more seeds wait for an evaluation than this workgroup has waves: ask for help
                s_xreg = 1;
                const uint32_t i = atomicAdd(&xhdr[2], 1u);
                st_l2(&xhdr[kXHdr + i], (uint32_t)img + 1u);
            }
        }
        xunlock();
        return am != 0ull;
    };
    // the cursor stands on a seed that was given to the helpers: take the answers in; if nobody has taken the request, take it back
    auto xservice = [&](int f0) -> bool {
        bool moved = false;
        const int r = f0 & (RW - 1);
        if (st_ld(&rg.state[r]) != R_REMOTE || lds_ld(&s_commit) != f0) return true;
        const int j = (int)rg.aux[r];
        int won = 0;
        if (lane == 0) won = atomicCAS(&xr[kXReq + j], (uint32_t)(f0 + 1), 0u) == (uint32_t)(f0 + 1) ? 1 : 0;
        if (__builtin_amdgcn_readfirstlane(won)) {
            if (lane == 0) { s_xk[j] = -1; atomicSub(&s_xout, 1); st_st(&rg.state[r], R_REDO); }
            moved = true;
        }
        return moved;
    };
    // seeds that wait for a full evaluation, the youngest first (the oldest are the local waves'), go to the helpers: at most two
    // requests per helper wavefront outstanding
    auto xexport = [&](int f, int lim) {
        const int nh = lds_ld(&s_nhelp);
        const int cap = min(kXReq - 4, 2 * nh);
        if (nh <= 0 || lim <= f || lds_ld(&s_nbig) <= 0 || lds_ld(&s_xout) >= cap) return;
        if (!xlock()) return;
        for (int it = 0; it < 4; it++) {
            if (lds_ld(&s_nbig) <= 0 || lds_ld(&s_xout) >= cap) break;
            const unsigned long long fm = ballot64(lane < kXReq && s_xk[lane] == -1);
            if (!fm) break;
            const int j = __builtin_ctzll(fm);
            int kb = -1;
            const int b0 = f & ~3;
            for (int base = b0 + ((lim - 1 - b0) & ~255); base >= b0 && kb < 0; base -= 256) {
                const int i0 = base + 4 * lane;
                const uint32_t x = i0 < lim ? __hip_atomic_load(reinterpret_cast<uint32_t*>(rg.state) + ((i0 & (RW - 1)) >> 2), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                int last = -1;
                #pragma unroll
                for (int t = 0; t < 4; t++)
                    if (((x >> (8 * t)) & 0xffu) == (uint32_t)R_BIG && i0 + t >= f && i0 + t < lim) last = i0 + t;
                const unsigned long long bm = ballot64(last >= 0);
                if (bm) kb = __builtin_amdgcn_readlane(last, 63 - __builtin_clzll(bm));
            }
            if (kb < 0) break;
            int won = 0;
            if (lane == 0) {
                const int r = kb & (RW - 1);
                won = st_cas(rg.state, r, R_BIG, R_EVAL) ? 1 : 0;       // (nobody reads aux of a record in R_EVAL)
                if (won) {
                    atomicSub(&s_nbig, 1); atomicAdd(&s_xout, 1);
                    rg.aux[r] = (uint32_t)j; s_xk[j] = kb;
                    st_st(&rg.state[r], R_REMOTE);
                    st_l2(&xr[kXReq + j], (uint32_t)(kb + 1));
                }
            }
            if (!__builtin_amdgcn_readfirstlane(won)) break;
            STAT(ST_XEXP, 1);
        }
        xunlock();
    };

    // Result slots of this wave: slot s holds the lists (and, for a result that marks usedMap, the record in pend[]) of
    // the speculative result of seed slot_k (lane s of slot_k_l); it is free again once the cursor has passed that seed.
    double* const wave_pend = b.pend + (img * NW + wave) * (size_t)NS * 24;
    int slot_k_l = -1;                                     // lane s < NS: seed whose result sits in slot s
    [[maybe_unused]] long long tl = NOW();
    // coarse accounting of this wave's time (s_memtime ticks since the last stamp go to slot i)
#define LT(i) do { const long long t_ = NOW(); DSTAT((i), t_ - tl); tl = t_; } while (0)
    bool adv = false;                                      // a record has been published since the cursor was last looked at

    // ---- The small-region grower: eight seeds side by side ----
    // Nine seeds in ten grow a region of fewer than regThre (12..16) pixels and are dropped at once (:228), and such a region
    // keeps 8 of the 64 lanes of grow() busy.  So the wave's eight 8-lane GROUPS each grow the region of one seed, one list
    // entry x its 8 neighbours per step, out of a private 16x16-pixel window of packed pixel words around the seed (LDS,
    // loaded once per seed: a small region cannot leave it without being given up first).  The candidates of an entry are
    // decided in reference order against the estimated sum vector exactly as grow() decides a batch pixel by pixel; a test too
    // close to call, a neighbour outside the window or a region that reaches regThre pixels hands the seed over to a full
    // evaluation (R_BIG), which starts from scratch.  A region that reaches its fixpoint first is what RegionGrower returns
    // for that seed, decision for decision; nothing of usedMap is written, so the result is published as R_LIGHT with the
    // box of what it examined.  Groups take the next seed of the wave's chunk as they finish.
    const int grp = lane >> 3, kq = lane & 7;
    const int kk8 = kq + (kq >= 4);                         // 3x3 neighbourhood, row-major, centre skipped (:533-534)
    const int ox = kk8 % 3 - 1, oy = kk8 / 3 - 1;
    int ncap = 1;                                           // a group gives its seed up when the region reaches ncap pixels
    if (g.regThre > 1.0 && g.degThre < 1.5) ncap = g.regThre >= (double)SCAP ? SCAP : (int)ceil(g.regThre);
    const float cos_tol_s = (float)g_tol0[2];
    uint32_t* const swin = G_ARENA(wave) + grp * 256;       // this group's window (the arena holds no full evaluation meanwhile)
    uint32_t* const slst = G_ARENA(wave) + 8 * 256 + grp * SCAP;   // this group's list: ly << 4 | lx
    int gk = -1;                                            // seed of this lane's group, -1: idle
    int gn = 0, gi = 0, gex = 0, gsnap = 0;
    float gC = 0.0f, gS = 0.0f;                             // estimated sum vector of the group's region
    int ch_k0 = 0, ch_sx = 0, ch_sy = 0;                    // the wave's chunk: lane j < CH holds seed ch_k0 + j
    unsigned long long ch_pend = 0ull;                      // seeds of the chunk not handed to a group yet
    bool tw_small = false;                                  // the arena holds windows and small lists (not tiles / a region list)
    bool gld = false;                                       // this lane's group waits for its window
    unsigned long long gldm = 0ull;                         // bit 8g: group g waits for its window (wave-uniform)
    int gld_age = 0;                                        // steps since the fetch
// Steps of the small-region groups a wave takes in a row before it goes back to the top of its loop (the cursor, the help protocol, the
// hand-out: ~100 instructions that find nothing new while no group has finished and no window is awaited).  1 / 4 / 16: 31.2 / 30.9 /
// 30.8 ms per step with eight steps in flight (profiles/r06n_inner_steps.log).
#ifndef LSD_REGION_INNER
#define LSD_REGION_INNER 16
#endif
#ifndef LSD_REGION_WINAGE
#define LSD_REGION_WINAGE 2
#endif
    constexpr int kWinAge = LSD_REGION_WINAGE;              // steps the other groups take before a wave waits for the windows it fetched
    constexpr float kEpsS = 1.0e-5f;                        // kEpsU + the fp32 running sums of up to SCAP unit vectors

    int pend_k = -1, pend_slot = 0;                         // a full evaluation this wave has claimed and starts once its groups are done
    bool pend_spec = false;
    int nwait = 0, wd_f = -1;                               // looks that found nothing to do since the cursor was last seen to move (watchdog)
    long long xlast = 0;                                    // when this wave last looked at the help protocol
    int xwant = 0;                                          // 1: look at the help protocol, 2: ... and the cursor stands on a seed given away
    // The windows fetched at the last refill are in LDS: seeds used meanwhile (:222) are skipped, the others start with their own pixel (:515-520)
    auto window_arrived = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t sw = swin[7 * 16 + 7];              // (row 7, column 7 of the window: the seed's own word)
        const bool used = gld && (sw & 3u) != 0u;
        if (gld && !used) {
            float s0, c0;
            fast_sincos(__uint_as_float(sw & ~3u), s0, c0);
            gC = c0; gS = s0; gn = 1; gi = 0; gex = 1;
            if (kq == 0) { swin[7 * 16 + 7] = sw | 1u; slst[0] = (7u << 4) | 7u; }
        }
        if (used) {
            if (kq == 0) st_st(&rg.state[gk & (RW - 1)], R_SKIP);
            gk = -1;
        }
        if (ballot64(used)) adv = true;
        gld = false;
        gldm = 0ull;
        LT(ST_TREFILL);
    };
    while (true) {
        int k = -1, slot = 0;
        bool spec = false;
        if (xwant) {                                       // (the one place the protocol is looked at from: it is a lot of code)
            bool moved = xpoll();
            if (xwant == 2) moved = xservice(lds_ld(&s_commit)) || moved;
            xwant = 0;
            if (moved) adv = true;
        }
        if (adv) {
            // (the cursor is worth a look only when the record it stands on is finished)
            const int f0 = lds_ld(&s_commit);
            const int s0 = f0 < nseeds ? st_ld(&rg.state[f0 & (RW - 1)]) : R_EMPTY;
            LT(ST_TSELECT);
            adv = false;
            if (s0 == R_SKIP || s0 == R_LIGHT || s0 == R_LIGHTL || s0 == R_STASH || s0 == R_XLIGHTL || s0 == R_XSTASH || s0 == R_SETL) advance();
            else if (s0 == R_REMOTE) xwant = 2;
            LT(ST_TCOMMIT);
        }
        LT(ST_TSELECT);
        const unsigned long long actm = ballot64(gk >= 0);
        const int nidle = 8 - __builtin_popcountll(actm & 0x0101010101010101ull);
        bool progress = false;
        int f = 0;
        if (nidle >= (actm ? kFeed : 1)) {
            // ---- feed the idle groups.  Between chunks (the seeds of a chunk are nobody else's) a wave first looks for a full
            //      evaluation to claim; it starts it when the groups still at work have finished ----
            f = lds_ld(&s_commit);
            if (xr && !ch_pend && (lds_ld(&s_nbig) >= NW / 2 || lds_ld(&s_xout) > 0 || lds_ld(&s_xpub) > 0)) {
                // (at most every ~10 us per wave: a look costs an L2 round trip)
                const long long tn = (long long)__builtin_amdgcn_s_memtime();
                if (tn - xlast > b.tun_xpoll) { xlast = tn; xwant = max(xwant, 1); }
            }
            if (!ch_pend && pend_k < 0) {
                const int stf = f < nseeds ? st_ld(&rg.state[f & (RW - 1)]) : R_EMPTY;
                if (stf == R_REDO || stf == R_BIG) {
                    // the record at the cursor: evaluated where everything earlier is committed, no result slot needed
                    int won = 0;
                    if (lane == 0) {
                        won = st_cas(rg.state, f & (RW - 1), stf, R_BUSY) ? 1 : 0;
                        if (won && stf == R_BIG) atomicSub(&s_nbig, 1);
                    }
                    won = __builtin_amdgcn_readfirstlane(won);
                    if (won) { pend_k = f; pend_spec = false; }
                }
                const unsigned long long freem = ballot64(lane < NS && slot_k_l < f);
                if (pend_k < 0 && freem != 0ull && __builtin_popcountll(freem) > NS - max(b.tun_big, min(NS, lds_ld(&s_depth) / (16 * NW))) && lds_ld(&s_nbig) > 0) {   // (tun_big: results a wave may have waiting for the cursor)
                    // the oldest seed waiting for a full evaluation: four records per lane and step
                    const int lim = min(min(lds_ld(&s_next), nseeds), f + lds_ld(&s_depth));
                    int kb = -1;
                    for (int base = f & ~3; base < lim && kb < 0; base += 256) {
                        const int i0 = base + 4 * lane;
                        uint32_t x = i0 < lim ? __hip_atomic_load(reinterpret_cast<uint32_t*>(rg.state) + ((i0 & (RW - 1)) >> 2), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                        int first = -1;
                        #pragma unroll
                        for (int t = 3; t >= 0; t--)
                            if (((x >> (8 * t)) & 0xffu) == (uint32_t)R_BIG && i0 + t >= f && i0 + t < lim) first = i0 + t;
                        const unsigned long long bm = ballot64(first >= 0);
                        if (bm) kb = __builtin_amdgcn_readlane(first, __builtin_ctzll(bm));
                    }
                    int won = 0;
                    if (lane == 0 && kb >= 0) {
                        const int tgt = kb == lds_ld(&s_commit) ? R_BUSY : R_EVAL;
                        won = st_cas(rg.state, kb & (RW - 1), R_BIG, tgt) ? (tgt == R_BUSY ? 2 : 1) : 0;
                        if (won) atomicSub(&s_nbig, 1);
                    }
                    won = __builtin_amdgcn_readfirstlane(won);
                    if (won) {
                        pend_k = kb; pend_spec = won == 1; pend_slot = __builtin_ctzll(freem);
                        if (pend_spec && lane == pend_slot) slot_k_l = kb;      // the slot is taken until the cursor has passed seed kb
                        progress = true;
                    }
                }
                if (xr && lds_ld(&s_nhelp) > 0 && lds_ld(&s_workbound)) xexport(f, min(min(lds_ld(&s_next), nseeds), f + lds_ld(&s_depth)));
                if (pend_k < 0) {
                    // reserve the next chunk of seeds
                    const int old = lds_ld(&s_next);
                    if (old < nseeds && old + CH - f <= lds_ld(&s_depth)) {
                        int got = 0;
                        if (lane == 0) got = atomicCAS(&s_next, old, old + CH) == old ? 1 : 0;
                        got = __builtin_amdgcn_readfirstlane(got);
                        progress = true;                       // (lost the race: somebody moved, look again)
                        if (got) {
                            ch_k0 = old;
                            const int kx = old + lane;
                            const bool valid = lane < CH && kx < nseeds;
                            uint32_t pp = 0u, code = 1u;
                            if (valid) { pp = seedpos[kx]; code = c.pw[pp] & 3u; }
                            ch_sx = (int)(pp % (uint32_t)w); ch_sy = (int)(pp / (uint32_t)w);
                            const bool used = valid && code != 0u;      // monotone: once used, always used (:222)
                            if (used) st_st(&rg.state[kx & (RW - 1)], R_SKIP);
                            if (ncap <= 1) {
                                // no region is small under these parameters: every seed goes to a full evaluation
                                if (valid && !used) st_st(&rg.state[kx & (RW - 1)], R_BIG);
                                const int nbg = __builtin_popcountll(ballot64(valid && !used));
                                if (lane == 0 && nbg) atomicAdd(&s_nbig, nbg);
                            } else ch_pend = ballot64(valid && !used);
                            adv = true;
                        }
                    }
                }
            }
            // idle groups take the next seeds of the chunk: their windows are FETCHED here (straight into LDS, nothing waits) and the
            // groups start at window_arrived() below, a step or two later -- the round trip to L2 / HBM runs beside the other groups' steps
            const unsigned long long idle0 = ~actm & 0x0101010101010101ull;     // bit 8g: group g is idle
            if (ch_pend && idle0) {
                if (gldm) window_arrived();                // (one fetch in flight at a time)
                unsigned long long idle = idle0;
                const int snap = lds_ld(&s_epoch);         // before anything of usedMap is read for these seeds
                wg_fence();
                tw_small = true;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the windows' last stores have landed before the fetch may)
                while (idle && ch_pend) {
                    const int j = __builtin_ctzll(ch_pend);
                    ch_pend &= ch_pend - 1ull;
                    const int gg = __builtin_ctzll(idle) >> 3;
                    idle &= idle - 1ull;
                    const int sxj = __builtin_amdgcn_readlane(ch_sx, j), syj = __builtin_amdgcn_readlane(ch_sy, j);
                    if (grp == gg) { gk = ch_k0 + j; gld = true; gsnap = snap; }
                    gldm |= 1ull << (8 * gg);
                    // the window: 16 rows of 16 packed pixel words around the seed, as they are in pw[]; lane l fetches quarter l & 3 of row l >> 2
                    const int wx = sxj - 7, wy = syj - 7;
                    uint32_t* const win = G_ARENA(wave) + gg * 256;
                    const int row = lane >> 2, q4 = (lane & 3) * 4;
                    if (wx >= 0 && wx + 15 < w && wy >= 0 && wy + 15 < h) {
                        // LDS-DMA: lane l's 16 bytes land at M0 + 16 l -- the window's layout
                        const uint32_t* src = c.pw + (size_t)(wy + row) * w + (wx + q4);
                        const uint32_t la = (uint32_t)uni((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)win);
                        // (M0 is the compiler's to manage and it says so; nothing else in this kernel uses it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(la), "v"(src) : "memory", "m0");
#pragma clang diagnostic pop
                    } else {
                        const int y = wy + row, x0 = wx + q4;
                        uint32_t v0 = kPwStatic, v1 = kPwStatic, v2 = kPwStatic, v3 = kPwStatic;   // outside the image: banned
                        if ((unsigned)y < (unsigned)h) {
                            const uint32_t* rowp = c.pw + (size_t)y * w;
                            if ((unsigned)(x0 + 0) < (unsigned)w) v0 = rowp[x0 + 0];
                            if ((unsigned)(x0 + 1) < (unsigned)w) v1 = rowp[x0 + 1];
                            if ((unsigned)(x0 + 2) < (unsigned)w) v2 = rowp[x0 + 2];
                            if ((unsigned)(x0 + 3) < (unsigned)w) v3 = rowp[x0 + 3];
                        }
                        *reinterpret_cast<uint4*>(&win[lane * 4]) = uint4{v0, v1, v2, v3};
                    }
                }
                gld_age = 0;
                progress = true;
                DSTAT(ST_SLOW, 1);                          // (refill rounds)
                LT(ST_TREFILL);
            }
        }
        if (gldm && (ballot64(gk >= 0 && !gld) == 0ull || ++gld_age >= kWinAge)) window_arrived();
        if (ballot64(gk >= 0)) {
          // (up to kInnerSteps steps in a row while no group finishes and no window is awaited: nothing the top of the loop looks at changes)
          for (int inner = 0; ; inner++) {
            // ---- one step: every active group tests the 8 neighbours of its next list entry ----
            const bool act = gk >= 0 && !gld;
            const uint32_t e = slst[act ? gi : 0];
            const int lx = (int)(e & 15u) + ox, ly = (int)(e >> 4) + oy;
            const bool inwin = ((unsigned)lx < 16u) & ((unsigned)ly < 16u);
            const int cell = ((ly & 15) << 4) | (lx & 15);
            const uint32_t word = swin[cell];
            // a neighbour outside the window: the region is not small enough for this grower
            bool bail = ((ballot64(act & !inwin) >> (lane & 56)) & 0xffull) != 0ull;
            bool todo = act & inwin & ((word & 1u) == 0u) & !bail;        // :536-537 (codes 0 and 2 are growable, Q5; a member gets bit 0)
            float sf, cf;
            fast_sincos(__uint_as_float(word & ~3u), sf, cf);
            while (ballot64(todo)) {
                // all candidates still to come, against the estimate as it stands: the ones that clearly fail before the first
                // one that does not are decided for good (nothing is accepted in between); that one must clearly pass
                const float Vg = __builtin_amdgcn_sqrtf(gC * gC + gS * gS) * 1.000001f;
                const float nr = (float)gn * inv_ub(fmaxf(Vg, 1e-3f));
                const float ec = kEpsS * (1.0f + 2.1f * nr) + 5e-6f;
                const float d1 = cf * gC + sf * gS;
                const unsigned long long nfm = ballot64(todo & !(d1 < (cos_tol_s - ec) * Vg));
                const unsigned long long pm = ballot64(d1 > (cos_tol_s + ec) * Vg);
                const uint32_t byte = (uint32_t)(nfm >> (lane & 56)) & 0xffu;
                const bool has = byte != 0u;
                const int l = has ? __builtin_ctz(byte) : 0;
                const int src = (lane & 56) + l;
                const bool acc = has & (((pm >> src) & 1ull) != 0ull);
                // (cos, sin) of the group's lane l in all its lanes: three DPP steps each instead of a round trip through the LDS crossbar
                const float cl = sum8(kq == l ? cf : 0.0f);
                const float sl = sum8(kq == l ? sf : 0.0f);
                if (acc & (kq == l)) {
                    swin[cell] = word | 1u;                                   // :549
                    slst[gn] = (uint32_t)((ly << 4) | lx);                   // :551-556
                }
                gC += acc ? cl : 0.0f; gS += acc ? sl : 0.0f;                // :545-546 (estimate)
                gn += acc ? 1 : 0;
                bail = bail | (has & !acc) | (acc & (gn >= ncap));            // too close to call / not a small region
                todo = todo & acc & !bail & (kq > l);
            }
            const int ni = gi + 1;
            const bool sweep_end = act & !bail & (ni >= gn);                  // :529 (the list is live)
            const bool done = sweep_end & (gn == gex);                       // :525 a sweep that added nothing
            gex = sweep_end ? gn : gex;
            gi = sweep_end ? 0 : ni;
            const bool fin = act & (done | bail);
            const unsigned long long finm = ballot64(fin & (kq == 0));
            if (finm) {
                // box of the list's pixels, relative to the seed (window cell 7, 7): what a line accepted before the seed's turn must not touch
                const uint32_t e0 = slst[kq < gn ? kq : 0], e1 = slst[kq + 8 < gn ? kq + 8 : 0];
                const int bx0 = imin8(min((int)(e0 & 15u), (int)(e1 & 15u))), bx1 = imax8(max((int)(e0 & 15u), (int)(e1 & 15u)));
                const int by0 = imin8(min((int)(e0 >> 4), (int)(e1 >> 4))), by1 = imax8(max((int)(e0 >> 4), (int)(e1 >> 4)));
                const bool light = fin & !bail;
                if (fin & (kq == 0)) {
                    const int r = gk & (RW - 1);
                    if (light) {
                        rg.snap[r] = (uint16_t)gsnap;
                        rg.aux[r] = (uint32_t)(bx0 - 7 + 32) | ((uint32_t)(by0 - 7 + 32) << 6) | ((uint32_t)(bx1 - 7 + 32) << 12) | ((uint32_t)(by1 - 7 + 32) << 18);
                        if (trace) { rnum[r * 2] = gn; rnum[r * 2 + 1] = gn << 2; }
                    }
                    st_st(&rg.state[r], light ? R_LIGHT : R_BIG);
                }
                const unsigned long long lightm = ballot64(light & (kq == 0));
                const int nbg = __builtin_popcountll(finm & ~lightm);
                if (lane == 0 && nbg) atomicAdd(&s_nbig, nbg);
                int gsum = 0;
                for (unsigned long long t = lightm; t; t &= t - 1ull) gsum += __builtin_amdgcn_readlane(gn, __builtin_ctzll(t));
                STAT(ST_GROW, __builtin_popcountll(lightm)); STAT(ST_GROWN, gsum); DSTAT(ST_SMALLBAIL, nbg);
                gk = fin ? -1 : gk;
                adv = true;
            }
            DSTAT(ST_SMALLSTEPS, 1);
            if (finm || gldm || inner >= LSD_REGION_INNER - 1) break;
          }
            LT(ST_TSMALL);
            nwait = 0;
            continue;
        }
        if (pend_k >= 0) { k = pend_k; spec = pend_spec; slot = pend_slot; pend_k = -1; nwait = 0; }
        else {
            if (progress) { nwait = 0; continue; }
            if (lds_ld(&s_next) >= nseeds && lds_ld(&s_commit) >= nseeds) break;   // everything is committed
            // Watchdog: the protocol has no state in which every wave waits; should one arise all the same (a defect), the image
            // is given up after seconds of nobody moving instead of hanging the device: counts[img] = -1, the state goes to stats.
            {   // (the cursor as this wave last saw it is kept across looks: movement during the sleep below, or by this wave's own
                //  advance() at the top of the next look, counts)
                const int fc = lds_ld(&s_commit);
                if (fc != wd_f) { wd_f = fc; nwait = 0; }
            }
            if (++nwait > LSD_REGION_WATCHDOG || lds_ld(&s_abort)) {
                if (b.stats && lane == 0) {
                    long long* st = b.stats + img * kStatWords;
                    if (!lds_ld(&s_abort)) {
                        const int fc = lds_ld(&s_commit);
                        st[40] = fc; st[41] = lds_ld(&s_next); st[42] = nseeds; st[43] = fc < nseeds ? st_ld(&rg.state[fc & (RW - 1)]) : -1;
                        st[44] = lds_ld(&s_nbig); st[45] = lds_ld(&s_lock); st[46] = pend_k; st[47] = wave;
                    }
                    st[24 + 2 * wave] = (long long)ch_k0 | ((long long)(pend_k + 1) << 32);     // what this wave holds (developer record)
                    st[25 + 2 * wave] = (long long)ch_pend;
                }
                if (lane == 0) lds_st(&s_abort, 1);
                break;
            }
            // within the look-ahead there is nothing for this wave (no seed to hand out, no evaluation to claim): look further
            if (nwait == 1 && lane == 0 && lds_ld(&s_next) < nseeds) {
                const int d = lds_ld(&s_depth);
                if (d < depth_max) lds_st(&s_depth, min(d + kDepthUp, depth_max));
            }
            // nothing to do: every slot waits for the cursor, the ring is full, or nothing is left to hand out.  Sleep long
            // enough that the polling of the waiting waves does not take issue slots from the evaluation the cursor waits for
            if (lds_ld(&s_commit) == f) __builtin_amdgcn_s_sleep(LSD_REGION_WAIT_SLEEP);
            if (xr) { if (lane == 0) atomicAdd(&s_idlecnt, 1); if (lds_ld(&s_xout) > 0) xwant = max(xwant, 1); }   // answers of the helpers
            adv = true;                                    // (look at the cursor again before asking for a job)
#ifdef LSD_REGION_STATS
            {   // why this wave had nothing to do: no result slot for a waiting seed / the ring is full / no seed is left
                const int old = lds_ld(&s_next);
                const long long t_ = NOW();
                const int why = old >= nseeds ? ST_WNOSEED : (lds_ld(&s_nbig) > 0 ? ST_WNOSLOT : ST_WAIT);
                DSTAT(why, t_ - tl); tl = t_;
            }
#else
            LT(ST_WAIT);
#endif
            continue;
        }
        // ---- a full evaluation of seed k (RegionGrower ... RectangleImprover with all 64 lanes) ----
        const uint32_t pp = seedpos[k];

        // ---- evaluate ----
        const int epoch_snap = lds_ld(&s_epoch);           // before anything of usedMap is read for this seed
        wg_fence();
        if (tw_small || !spec || epoch_snap != g_ws[wave].cache_epoch) {   // tiles fetched before the last accept may miss its bans
            invalidate_tiles(c);
            g_ws[wave].cache_epoch = epoch_snap;
            tw_small = false;
        }
        eval_seed(c.wave, pp, spec ? 1 : 0, slot);
#ifdef LSD_REGION_INJECT_STALL
        if (img & 1) continue;                             // test build (make wdtest): odd images never publish a full evaluation -> the watchdog has to end them
#endif
        const EvalOut& eo = g_eo[wave];
        const bool skip = eo.skip != 0;
        const int outcome = eo.outcome, num = eo.num, num0 = eo.num0, rec_pk = eo.rec_pk;
        const double logNFA = eo.logNFA;
        const double pv = lane < 12 ? reinterpret_cast<const double*>(&g_ws[wave].rec)[lane] : 0.0;   // lane j < 12: field j of the result's rectangle (structRec order)

        // ---- hand the result over ----
#ifdef LSD_REGION_STATS
        if (!spec) DSTAT(ST_DEPTHDN, NOW() - tl);          // (time of the evaluations AT the cursor: everybody else may be waiting for them)
#endif
        LT(ST_TEVAL);
        if (!spec) {
            // ---- evaluated at the cursor (k == s_commit, record R_BUSY: nobody else can commit): commit right away ----
            if (!skip && outcome >= 2) {
                // (under the cursor lock although nobody else can commit here: certify_set() relies on no ban appearing while it holds it)
                while (true) {
                    int got = 0;
                    if (lane == 0) got = atomicCAS(&s_lock, 0, 1) == 0 ? 1 : 0;
                    if (__builtin_amdgcn_readfirstlane(got)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                commit_marks(k, num0, num, outcome, logNFA, pv, nullptr, 0);
                if (lane == 0) lds_st(&s_lock, 0);
            }
            else if (!skip) write_trace(k, num0, outcome == 0 ? num0 : num, outcome, logNFA);
            wg_fence();
            if (lane == 0) { rg.state[k & (RW - 1)] = (uint8_t)R_EMPTY; lds_st(&s_commit, k + 1); }
            LT(ST_TCOMMIT);
            adv = true;
            continue;
        }
        if (skip) {
            if (lane == slot) slot_k_l = -1;               // nothing kept in the slot
            if (lane == 0) st_st(&rg.state[k & (RW - 1)], R_SKIP);
            adv = true;
            continue;
        }
        if (eo.setid) {                                    // answered by a certified set: nothing is kept in the slot
            if (lane == slot) slot_k_l = -1;
            if (lane == 0) {
                const int r = k & (RW - 1);
                rg.snap[r] = (uint16_t)epoch_snap;
                rg.aux[r] = (uint32_t)eo.setid;
                if (trace) { rnum[r * 2] = num0; rnum[r * 2 + 1] = (num << 2) | outcome; }
                st_st(&rg.state[r], R_SETL);
            }
            adv = true;
            continue;
        }
        const int x0 = eo.x0, y0 = eo.y0, x1 = eo.x1, y1 = eo.y1, n1 = eo.n1, n2 = eo.n2, m_off = eo.m_off, mcnt = eo.mcnt;
        const bool precise = eo.precise != 0, redo = eo.redo != 0;
        if (outcome <= 1) {                                // nothing to mark: publish and move on
            if (lane == 0) {
                const int r = k & (RW - 1), si = wave * NS + slot;
                rg.snap[r] = (uint16_t)epoch_snap;
                rg.aux[r] = (uint32_t)si;
                stab.box[si][0] = (short)x0; stab.box[si][1] = (short)y0; stab.box[si][2] = (short)x1; stab.box[si][3] = (short)y1;
                stab.lcnt[si] = precise ? ((uint32_t)(n1 + 1) | ((uint32_t)n2 << 16)) : 0u;
                if (trace) { rnum[r * 2] = num0; rnum[r * 2 + 1] = (num << 2) | outcome; }
            }
            wg_fence();                                    // the lists are in the slot before the record says so
            if (lane == 0) st_st(&rg.state[k & (RW - 1)], R_LIGHTL);
            adv = true;
            // an evaluation that went the way of a uniform set offers its first list (still in the slot) as a certified set
            if (eo.cert && certify_set(c.wave, pp, slot, num0, &s_lock, &s_nsets)) STAT(ST_SETNEW, 1);
            continue;
        }
        // marks to make: the result is stashed (record in pend[], the pixels to mark in the list slot); whoever moves the
        // cursor over it commits it
        if (redo) {
            if (lane == 0) st_st(&rg.state[k & (RW - 1)], R_REDO);
            STAT(ST_REDO, 1);
            adv = true;
            continue;
        }
        if (lane < 12) wave_pend[slot * 24 + lane] = pv;
        if (lane == 0) {
            double* P = wave_pend + slot * 24;
            P[12] = logNFA;
            P[13] = (double)rec_pk; P[14] = (double)outcome; P[15] = (double)num0; P[16] = (double)num; P[17] = (double)mcnt;
            P[18] = (double)x0; P[19] = (double)y0; P[20] = (double)x1; P[21] = (double)y1;
            P[22] = (double)((long long)(precise ? n1 + 1 : 0) + 32768ll * n2 + 32768ll * 32768ll * m_off);
            P[23] = (double)epoch_snap;
            rg.aux[k & (RW - 1)] = (uint32_t)(wave * NS + slot);
        }
        wg_fence();                                        // record and lists are in the slot before the ring says so
        if (lane == 0) st_st(&rg.state[k & (RW - 1)], R_STASH);
        adv = true;
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (an image given up by the watchdog may leave a window fetch in flight: nothing lands in LDS after the wave is gone)
    // ---- this image is finished (every seed committed, or given up): results out ----
    if (wave == 0 && lane == 0 && !pool) {
        b.counts[img] = lds_ld(&s_abort) ? -1 : s_lines;
        if (b.nseed) b.nseed[img] = s_ntrace;
    }
    if (b.stats && !pool) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(b.stats + img * kStatWords);
        if (lane == 0 && wave == 0 && !lds_ld(&s_abort)) { b.stats[img * kStatWords + 45] = (long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15); b.stats[img * kStatWords + 46] = rt_begin; b.stats[img * kStatWords + 47] = (long long)__builtin_amdgcn_s_memrealtime(); }   // (developer record: when the image ran)
        if (lane == 0 && wave == 0) {
            // shader clocks of this workgroup on the image.  s_memtime is a counter of the XCD the wavefront runs on: a workgroup that was
            // preempted (more hardware queues in use than the device has -- e.g. two processes with 16 each -- make the scheduler time-slice
            // them, saving and restoring wavefronts) and resumed on another XCD reads another counter, and the difference comes out negative or
            // absurd (seen in the test suite, which starts bench.py beside its own process).  Then the constant 100 MHz clock stands in, at the
            // nominal 24 shader clocks per tick.
            long long tot = (long long)__builtin_amdgcn_s_memtime() - t_begin;
            const long long rt = ((long long)__builtin_amdgcn_s_memrealtime() - rt_begin) * 24;
            if (tot <= 0 || tot > 4 * rt + 1000000) tot = rt > 0 ? rt : 1;
            g_stat[c.wave][sslot(ST_TOTAL)] = (unsigned long long)tot; g_stat[c.wave][sslot(ST_SEEDS)] = (unsigned long long)nseeds; DSTAT(ST_DEPTHEND, lds_ld(&s_depth));
        }
        if (lane < ST_COUNT && !lds_ld(&s_abort) && (sslot(lane) != 11 || lane == ST_NFASLOW || kStatSlots == ST_COUNT)) {
            if (lane == ST_MINNFA || lane == ST_MINGAP) atomicMax(&st[lane], g_stat[c.wave][sslot(lane)]);
            else atomicAdd(&st[lane], g_stat[c.wave][sslot(lane)]);
        }
        g_stat[c.wave][sslot(ST_XHELP)] = 0ull;
        // (what a helper adds to ANOTHER image's record below starts from zero: not from this image's own margins and ties)
        g_stat[c.wave][sslot(ST_MINNFA)] = 0ull; g_stat[c.wave][sslot(ST_MINGAP)] = 0ull; g_stat[c.wave][sslot(ST_NFASLOW)] = 0ull; g_stat[c.wave][sslot(ST_TIES)] = 0ull;
    }
    if (!b.xq) return;
    if (wave == 0 && !pool) {
        agent_release();
        if (lane == 0) { st_l2(&xr[2], 1u); atomicAdd(&xhdr[1], 1u); }
    }
    // While workgroups still wait for a CU this one should make room -- unless an image is so far behind with its evaluations
    // that a few wavefronts are better spent there (EARLY helpers: at most tun_early workgroups' worth at a time, only for images
    // with two backlogged seeds per wave of their own, and gone as soon as there is nothing of that kind)
    bool early = !pool && ld_l2(&xhdr[0]) < (uint32_t)nimg;
    if (early) {
        if (b.tun_early <= 0) return;
        int ok = 0;
        if (lane == 0) {
            if (atomicAdd(&xhdr[3], 1u) >= (uint32_t)(b.tun_early * NW)) atomicSub(&xhdr[3], 1u);
            else ok = 1;
        }
        if (!__builtin_amdgcn_readfirstlane(ok)) return;
    }

    // ---- ... then help elsewhere: every wavefront on its own (see "Help from other workgroups" above) ----
    if (lane < NS) slot_k_l = -1;
    int hx = -1, hj = 0, hidle = 0, hscan = 0;              // image this wave helps (-1: none), request it is answering, looks without work / for an image
    uint32_t* hrec = nullptr;                               // that image's record of the help protocol
    const uint32_t* cur_seedpos = seedpos;
    // the work counters of an evaluation done for another image (grows, grown pixels, NFA and RegionRadiusReducer calls: slots 0..6)
    // go to THAT image's record, as if its own waves had done it
    unsigned long long hprev = lane < 7 ? g_stat[c.wave][lane] : 0ull;
    while (true) {
        int k = -1, slot = 0;
        if (hx < 0) {
            // ---- not attached: the image with the largest backlog per wave already working on it ----
            if (ld_l2(&xhdr[1]) >= (uint32_t)nimg) break;   // every image is finished
            if (early && ld_l2(&xhdr[0]) >= (uint32_t)nimg) {   // every workgroup has its CU by now
                early = false;
                if (lane == 0) atomicSub(&xhdr[3], 1u);
            }
            const int wt = min(min((int)ld_l2(&xhdr[2]), nimg), 4096);
            uint32_t best = 0u;
            for (int base = 0; base < wt; base += 64) {
                const int i = base + lane;
                uint32_t key = 0u;
                if (i < wt) {
                    const uint32_t im1 = ld_l2(&xhdr[kXHdr + i]);
                    if (im1) {
                        const uint32_t* rc = b.xq + (size_t)(im1 - 1u) * kXStride;
                        if (!ld_l2(&rc[2])) {
                            const uint32_t bl = ld_l2(&rc[4]), nh = ld_l2(&rc[3]);
                            if (bl >= (early ? 2u * NW : 1u) && nh < (uint32_t)(early ? NW : b.tun_help)) key = ((min(bl * 16u / (nh + NW), 0xffffeu) + 1u) << 12) | (uint32_t)i;
                        }
                    }
                }
                #pragma unroll
                for (int o = 32; o; o >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, o));
                best = max(best, key);
            }
            best = (uint32_t)uni((int)best);
            if (!best) {
                if (early) { if (lane == 0) atomicSub(&xhdr[3], 1u); break; }
                if (++hscan > b.tun_linger) break;         // nobody has asked for a while: give the CU back (another launch may be waiting for it)
                for (int t = 0; t < 8; t++) __builtin_amdgcn_s_sleep(127);
                continue;
            }
            const uint32_t im = ld_l2(&xhdr[kXHdr + (best & 4095u)]) - 1u;
            uint32_t* rc = b.xq + (size_t)im * kXStride;
            int ok = 0;
            if (lane == 0) {
                if (atomicAdd(&rc[3], 1u) >= (uint32_t)(early ? NW : b.tun_help)) atomicSub(&rc[3], 1u);
                else ok = 1;
            }
            if (!__builtin_amdgcn_readfirstlane(ok)) continue;
            hx = (int)im; hrec = rc; hidle = 0; hscan = 0;
            cur_seedpos = b.seedpos + (size_t)im * npx;
            c.mag = b.mag + (size_t)im * npx; c.deg = b.deg + (size_t)im * npx; c.pw = b.pw + (size_t)im * npx;
            c.epochmap = b.epochmap + (size_t)im * npx; c.sc = b.sc + (size_t)im * npx; c.sets = nullptr;
            c.tep = b.tepoch + (size_t)im * (size_t)(((w + 7) >> 3) * ((h + 7) >> 3));
            if (lane == 0) {
                RCtx& gc = g_ctx[wave];
                gc.mag = c.mag; gc.deg = c.deg; gc.pw = c.pw; gc.epochmap = c.epochmap; gc.sc = c.sc; gc.tep = c.tep; gc.sets = nullptr;
                g_ws[wave].cache_epoch = -1;
            }
            wg_fence();
            continue;
        }
        // ---- attached to image hx: a request to answer ----
        const uint32_t cm = ld_l2(&hrec[1]);
        const bool xdone = ld_l2(&hrec[2]) != 0u;
        if (lane < NS && (xdone || slot_k_l < (int)cm)) slot_k_l = -1;          // the cursor has passed these results
        const unsigned long long freem = ballot64(lane < NS && slot_k_l < 0);
        if (xdone || (hidle > kHelpIdle && freem == (1ull << NS) - 1ull)) {
            if (lane == 0) atomicSub(&hrec[3], 1u);
            hx = -1;
            continue;
        }
        if (__builtin_popcountll(freem) <= NS - b.tun_big) { __builtin_amdgcn_s_sleep(127); continue; }   // (as for local waves)
        const uint32_t rq = lane < kXReq ? ld_l2(&hrec[kXReq + lane]) : 0u;
        uint32_t key = (rq != 0u && (rq >> 31) == 0u) ? rq : 0xffffffffu;          // the oldest seed first
        #pragma unroll
        for (int o = 32; o; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o));
        key = (uint32_t)uni((int)key);
        if (key == 0xffffffffu) { hidle++; __builtin_amdgcn_s_sleep(127); continue; }
        const int jj = __builtin_ctzll(ballot64(rq == key));
        int ok = 0;
        if (lane == 0) ok = atomicCAS(&hrec[kXReq + jj], key, key | 0x80000000u) == key ? 1 : 0;
        if (!__builtin_amdgcn_readfirstlane(ok)) continue;
        k = (int)key - 1; slot = __builtin_ctzll(freem); hj = jj; hidle = 0;
        if (lane == slot) slot_k_l = k;
        const uint32_t pp = cur_seedpos[k];
        const int epoch_snap = (int)ld_l2(&hrec[0]);       // before anything of usedMap is read for this seed ...
        agent_acquire();                                   // ... (the acquire empties this CU's L1: the bans made up to that epoch are seen)
        if (epoch_snap != g_ws[wave].cache_epoch) {         // tiles fetched before the last accept may miss its bans
            invalidate_tiles(c);
            g_ws[wave].cache_epoch = epoch_snap;
        }
        eval_seed(c.wave, pp, 1, slot);
        if (b.stats && lane < 7) {
            const unsigned long long v = g_stat[c.wave][lane];
            if (v != hprev) atomicAdd(reinterpret_cast<unsigned long long*>(b.stats + (size_t)hx * kStatWords) + lane, v - hprev);
            hprev = v;
        }
        if (b.stats && lane == 0) {                        // ... and how close its NFA comparisons came to a tie
            unsigned long long* const hs = reinterpret_cast<unsigned long long*>(b.stats + (size_t)hx * kStatWords);
            atomicMax(&hs[ST_MINNFA], g_stat[c.wave][sslot(ST_MINNFA)]); atomicMax(&hs[ST_MINGAP], g_stat[c.wave][sslot(ST_MINGAP)]);
            atomicAdd(&hs[ST_NFASLOW], g_stat[c.wave][sslot(ST_NFASLOW)]);
            atomicAdd(&hs[ST_TIES], g_stat[c.wave][sslot(ST_TIES)]);
        }
        if (b.stats) { g_stat[c.wave][sslot(ST_MINNFA)] = 0ull; g_stat[c.wave][sslot(ST_MINGAP)] = 0ull; g_stat[c.wave][sslot(ST_NFASLOW)] = 0ull; g_stat[c.wave][sslot(ST_TIES)] = 0ull; }
        const EvalOut& eo = g_eo[wave];
        const bool skip = eo.skip != 0;
        const int outcome = eo.outcome, num = eo.num, num0 = eo.num0, rec_pk = eo.rec_pk;
        const double logNFA = eo.logNFA;
        const double pv = lane < 12 ? reinterpret_cast<const double*>(&g_ws[wave].rec)[lane] : 0.0;
        // the answer: what a local evaluation would have published, with the lists (and the record) in this wave's slot
        const bool xredo = !skip && outcome >= 2 && eo.redo != 0;
        const int res = skip ? R_SKIP : outcome <= 1 ? R_XLIGHTL : xredo ? R_REDO : R_XSTASH;
        const size_t my_slot = (img * NW + wave) * (size_t)NS + slot;
        if (res == R_XSTASH) {
            double* P = b.pend + my_slot * 24;
            if (lane < 12) P[lane] = pv;
            if (lane == 0) {
                P[12] = logNFA;
                P[13] = (double)rec_pk; P[14] = (double)outcome; P[15] = (double)num0; P[16] = (double)num; P[17] = (double)eo.mcnt;
                P[18] = (double)eo.x0; P[19] = (double)eo.y0; P[20] = (double)eo.x1; P[21] = (double)eo.y1;
                P[22] = (double)((long long)(eo.precise ? eo.n1 + 1 : 0) + 32768ll * eo.n2 + 32768ll * 32768ll * eo.m_off);
                P[23] = (double)epoch_snap;
            }
        }
        if (res == R_SKIP || res == R_REDO) { if (lane == slot) slot_k_l = -1; }      // nothing kept in the slot
        if (lane == 0) {
            uint32_t* m = hrec + 3 * kXReq + hj * 8;
            st_l2(&m[0], (uint32_t)res); st_l2(&m[1], (uint32_t)my_slot); st_l2(&m[2], (uint32_t)epoch_snap);
            if (res == R_XLIGHTL) {
                st_l2(&m[3], (uint32_t)(eo.x0 & 0xffff) | ((uint32_t)(eo.y0 & 0xffff) << 16));
                st_l2(&m[4], (uint32_t)(eo.x1 & 0xffff) | ((uint32_t)(eo.y1 & 0xffff) << 16));
                st_l2(&m[5], eo.precise ? ((uint32_t)(eo.n1 + 1) | ((uint32_t)eo.n2 << 16)) : 0u);
            }
        }
        agent_release();                               // lists, record and message before the flag
        if (lane == 0) st_l2(&hrec[2 * kXReq + hj], (uint32_t)(k + 1));
        STAT(ST_XHELP, 1);
        continue;
    }
    // (a helper's count of evaluations done for others goes to its own image's record)
    if (b.stats && lane == 0 && !pool) atomicAdd(reinterpret_cast<unsigned long long*>(b.stats + img * kStatWords) + ST_XHELP, g_stat[c.wave][sslot(ST_XHELP)]);
}

// The launch.  One workgroup per image (+ b.npool helper-only workgroups), dispatched in the order of k_order -- or, on the 8-wave
// build, PERSISTENT workgroups (b.pcount, lsd_ctx.hip: a batch of more images than the device has CUs, help off): as many workgroups
// as CUs, each taking the next image of that order from the launch's counter until none is left.  The hardware deals the workgroups
// of a launch onto the eight XCDs round-robin and never moves them, so with one workgroup per image an XCD whose 64 images are heavy
// finishes last while CUs of the others idle; the counter balances across XCDs: 512 maps as one step 78.6 -> 73.4 ms
// (profiles/r06d_scheduling_probes.log).  Not for the 4-wave build: there the loop around the image's code costs the kernel body's register
// allocation 16 % (31.3 -> 36.4 ms per step with eight steps in flight, same log) and buys nothing (persistent workgroups let the
// front ends of the other steps flow -- their dispatches no longer wait behind a launch that does not fit the device -- but the steps'
// tails leave the slots of finished workgroups empty: 36.6-45.7 ms against 36.8).
__global__ __launch_bounds__(64 * NW, LSD_REGION_WAVES_PER_SIMD) LSD_REGION_KATTR void k_region(Geom g, Buffers b, uint32_t id_base) {
#if LSD_REGION_NW == 8
    __shared__ int s_take;
    while (true) {                                          // (one call site: the image's code exists once)
        int bid = (int)blockIdx.x, nimg = (int)gridDim.x - b.npool;
        if (b.pcount) {
            if (threadIdx.x == 0) s_take = atomicAdd(b.pcount, 1);
            __syncthreads();
            bid = s_take; nimg = b.nimg;
            __syncthreads();                                // (everybody has read it before the next round overwrites it)
            if (bid >= nimg) return;
        }
        region_image(g, b, id_base, bid, nimg);
        if (!b.pcount) return;
        __syncthreads();                                    // every wavefront is out of the image before its shared state is set up again
    }
#else
    region_image(g, b, id_base, (int)blockIdx.x, (int)gridDim.x - b.npool);
#endif
}

}  // namespace RVAR

#if LSD_REGION_NW == 8
void launch_region_w8(const Geom& g, const Buffers& b, int n, uint32_t id_base, hipStream_t s) {
    hipLaunchKernelGGL(w8::k_region, dim3(n + b.npool), dim3(64 * w8::NW), w8::kDynLds, s, g, b, id_base);
}
// workspace is sized for the wider variant
int region_slots() { return w8::NS; }
int region_waves() { return w8::NW; }
int region_ring() { return w8::RW; }
#else
void launch_region_w4(const Geom& g, const Buffers& b, int n, uint32_t id_base, hipStream_t s) {
    hipLaunchKernelGGL(w4::k_region, dim3(n + b.npool), dim3(64 * w4::NW), w4::kDynLds, s, g, b, id_base);
}
#endif

}  // namespace lsdhip
