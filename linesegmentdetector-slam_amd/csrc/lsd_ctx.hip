// lsd_ctx.hip -- host side of liblsdhip.so: context, HBM workspace, host-computed tables, launch
// sequencing and the C ABI declared in include/lsd_hip.h.
//
// Everything numerical on the hot path runs in the HIP kernels (k_*.hip).  The host computes only
// what the reference also computes once per call on scalars: the Gaussian taps (myLSD.cpp:398-417),
// the thresholds (myLSD.cpp:148-149, :207-209) and two small lookup tables (log-gamma of integers,
// logs of p = aliPro/2^k) that the kernels index instead of evaluating libm on the device.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "lsd_internal.h"

using namespace lsdhip;

struct lsd_ctx {
    int device = 0;
    int num_cus = 256;                 // compute units of the device
    uint32_t id_budget = 0xFFFF0u;     // curMap stamp ids a wave may use per run before it clears its stamps (lsd_debug_set_stamp_budget)
    int tun_soft = 0, tun_claim = 0, tun_feed = 3, tun_big = 0;   // region-stage schedule (0: default), see k_region.hip
    int tun_help = -1;                                             // helper wavefronts per image (-1: default, 0: none)
    // developer experiments (environment variables read once, when the context is created; DESIGN_NOTES.md says what each was for)
    int tun_gate = 12000;                                          // an image asks for help once it has run for this long (x 1024 clocks: ~5 ms)
    int tun_share = 0;                                             // ... and for at least this share (%) of the time since the launch began (LSD_REGION_SHARE)
    int pool_max_images = 4;                                       // calls with at most this many images get a pool of helper workgroups (LSD_REGION_POOL)
    int tun_early = 0, tun_wb = 10, tun_up = 32, tun_down = 96, tun_requeue = 1, tun_xpoll = 20000, tun_linger = 1000000, tun_stop = 0;
    uint32_t* xq = nullptr;
    int region_waves_mode = 0;         // 0: choose per batch; 4 / 8: force that region-stage variant (lsd_set_region_waves)
    bool prefer4 = false;              // the 8-wave workspace did not fit this device's memory once: batches run on 4 waves per image
    hipStream_t stream = nullptr;      // the context's own stream
    hipStream_t last_stream = nullptr; // stream of the last enqueue
    std::string err;
    // workspace capacity
    size_t cap_gpx = 0;                                      // Gaussian elements per image (rows padded to Geom::gp)
    size_t cap_n = 0, cap_npx = 0, cap_wh = 0, cap_ws = 0;   // images, scaled pixels per image, input pixels per image, wave slots
    int cap_max_lines = 0;
    bool cap_trace = false;
    // workspace
    double *gauss = nullptr, *mag = nullptr, *deg = nullptr, *recs = nullptr, *recs_scaled = nullptr;
    double2* sc = nullptr;
    uint32_t* order = nullptr;
    uint32_t *sets = nullptr;            // n x 256: certified sets of the region stage (k_region.hip)
    uint32_t *pw = nullptr, *epochmap = nullptr, *ord = nullptr, *spill = nullptr, *gcopy = nullptr, *stamps = nullptr, *seedidx = nullptr, *seedpos = nullptr, *tepoch = nullptr;
    uint32_t run_id = 0;   // curMap stamps are unique per run: (run_id << 20) + grow number (a wave that uses up its 2^20 clears its stamps)
    uint32_t* slist = nullptr;
    double* pend = nullptr;
    float4* wmeta = nullptr;
    int* rnum = nullptr;
    int mcap = 16384;
    int gcap = 8192;
    unsigned long long* maxbits = nullptr;
    int32_t *nb = nullptr, *nseed = nullptr;
    long long* stats = nullptr;
    void* seeds = nullptr;
    // host-API staging (device side), and the pinned host buffers every host <-> device copy goes through
    uint8_t *h_in = nullptr, *h_lineim = nullptr;
    lsd_line *h_lines = nullptr, *h_flat = nullptr;
    int32_t *h_counts = nullptr, *h_offs = nullptr;
    uint8_t* pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_used[2] = {false, false};  // a DMA through the buffer has been queued: its event must be waited for before the buffer is written again
    hipStream_t copy_stream = nullptr;  // second stream: the remapped maps travel back while the rest of the pipeline runs
    size_t hcap_n = 0, hcap_wh = 0;
    int hcap_max_lines = 0;
    bool hcap_lineim = false;
    // tables
    double *d_taps = nullptr, *d_lgamma = nullptr, *d_ptab = nullptr;
    int* d_centres = nullptr;
    int lg_count = 0;                   // entries of d_lgamma
    bool cost_history = false;          // lsd_set_cost_history: the region stage takes the images in the order of their cost in the last launch
    int hist_n = 0;                     // images of the launch whose counter records are in `stats` (0: none)
    lsd_params tab_params{};
    bool tab_valid = false;
    int tapR = 0;
    // createMapCache workspace
    unsigned long long* mc_claim = nullptr;
    uint32_t *mc_fa = nullptr, *mc_fb = nullptr;
    int* mc_ctl = nullptr;                              // spread flood: frontier sizes + per-chunk counts
    size_t mc_ctl_n = 0;
    uint8_t* mc_in = nullptr;
    double* mc_out = nullptr;
    size_t mc_cap = 0, mc_hcap = 0;
    uint8_t *oc_in = nullptr, *oc_out = nullptr;        // occupancy-grid staging of the host entry point
    size_t oc_cap = 0;
    uint8_t* mt_buf = nullptr;                          // staging of the host scan-to-map matching entry point
    size_t mt_cap = 0;
    // lsd_gather_lines: this rank's padded counts + offsets, and its slab of packed line records
    int32_t* ga_cnt = nullptr;
    lsd_line* ga_slab = nullptr;
    size_t ga_cnt_cap = 0, ga_slab_cap = 0;
    hipEvent_t ga_ev = nullptr;         // recorded behind the collectives of the last lsd_gather_lines (they read ga_cnt / ga_slab)
    bool ga_ev_valid = false;
    // options
    int stop_after = 0;
    int tun_groups = -1;                // the 8-wave region stage as persistent workgroups (k_region.hip: k_region): -1 = as many as CUs when the batch has more images than that, 0 = never
    int* pcount = nullptr;              // ... and the launch's image counter
    bool trace = false;
    int host_max_lines = 8192;
    // last run
    Geom geom{};
    int last_n = 0;
    int last_max_lines = 0;
    int32_t* last_counts = nullptr;
    hipEvent_t ev[7]{};
    bool ev_valid = false;
    hipEvent_t ev_done = nullptr;      // end of the last enqueue: a later enqueue on ANOTHER stream waits for it (shared workspace)
    bool done_valid = false;
};

constexpr size_t kPinBytes = 32u << 20;   // two pinned staging buffers of this size per context

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return e_ == hipErrorOutOfMemory ? LSD_ERR_NOMEM : LSD_ERR_HIP;                       \
        }                                                                                         \
    } while (0)

// ---------------------------------------------------------------------------------------------
// host-computed scalars and tables
// ---------------------------------------------------------------------------------------------
static int tap_radius(double sca, double sig) {                     // myLSD.cpp:390-393
    const int prec = 3;
    if (sca < 1) sig = sig / sca;
    return cvt_x86(ceil(sig * sqrt(2 * prec * log(10))));
}

static void gauss_taps(double sca, double sig, int h, std::vector<double>& t) {   // myLSD.cpp:398-417
    if (sca < 1) sig = sig / sca;
    const int hSize = 1 + 2 * h;
    t.assign((size_t)3 * hSize, 0.0);
    double s1 = 0, s2 = 0, s3 = 0;
    for (int k = 0; k < hSize; k++) {
        const double a = (k - h) / sig, b = (k - h - 1.0 / 3) / sig, c = (k - h + 1.0 / 3) / sig;
        t[0 * hSize + k] = exp(-0.5 * (a * a));
        t[1 * hSize + k] = exp(-0.5 * (b * b));
        t[2 * hSize + k] = exp(-0.5 * (c * c));
        s1 += t[0 * hSize + k]; s2 += t[1 * hSize + k]; s3 += t[2 * hSize + k];
    }
    for (int k = 0; k < hSize; k++) {
        t[0 * hSize + k] /= s1; t[1 * hSize + k] /= s2; t[2 * hSize + k] /= s3;
    }
}

static double log_gamma_host(int x) {                               // LogGammaCalculator, myLSD.cpp:882-924
    if (x > 15)
        return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1.0 / x) + 1.0 / (810 * pow(x, 6)));
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                                1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5), b = 0;
    for (int i = 0; i < 7; i++) { a -= log(x + i); b += q[i] * pow(x, i); }
    return a + log(b);
}

// LogGammaCalculator(0 .. count - 1) from the host libm, computed once per process (a 2048 x 2048 map needs 377 k entries: ~40 ms)
static std::mutex g_lg_mu;                                          // held while the table is grown and while a context copies out of it
static const double* log_gamma_table(int count) {
    static std::vector<double> tab;
    if ((int)tab.size() < count) {
        const int from = (int)tab.size();
        tab.resize(count);
        for (int i = from; i < count; i++) tab[i] = i >= 1 ? log_gamma_host(i) : 0.0;
    }
    return tab.data();
}

static int make_geom(const lsd_params* p, int cols, int rows, Geom* g) {
    if (!p || cols <= 0 || rows <= 0) return LSD_ERR_INVALID;
    if (!(p->sca > 0) || !(p->sig > 0) || !(p->angThre > 0) || p->pseBin < 1) return LSD_ERR_INVALID;
    if (p->pseBin > 1024) return LSD_ERR_UNSUPPORTED;
    if (cols > 65535 || rows > 65535) return LSD_ERR_UNSUPPORTED;
    g->W = cols; g->H = rows;
    g->w = cvt_x86(floor(cols * p->sca));                           // myLSD.cpp:132
    g->h = cvt_x86(floor(rows * p->sca));                           // :133
    if (g->w < 2 || g->h < 2) return LSD_ERR_INVALID;
    // the region stage packs scaled coordinates as (y<<16 | x) and keeps bounding boxes (+-1 margin) in 16-bit signed fields
    if (g->w > 32766 || g->h > 32766) return LSD_ERR_UNSUPPORTED;
    if ((long long)g->w * g->h > (1ll << 30)) return LSD_ERR_UNSUPPORTED;
    g->npx = g->w * g->h;
    g->gp = (g->w + 15) & ~15;
    g->sca = p->sca;
    g->tapR = tap_radius(p->sca, p->sig);
    if (g->tapR < 0 || g->tapR > kMaxTapRadius) return LSD_ERR_UNSUPPORTED;
    g->pseBin = p->pseBin;
    g->degThre = p->angThre / 180.0 * kPi;                          // :148
    g->gradThre = 2.0 / sin(g->degThre);                            // :149
    g->logNT = 5 * (log10(g->h) + log10(g->w)) / 2.0;               // :207
    g->regThre = -g->logNT / log10(p->angThre / 180.0);             // :208
    g->aliPro = p->angThre / 180.0;                                 // :209
    g->denThre = p->denThre;
    return LSD_OK;
}

static int ensure_tables(lsd_ctx* c, const lsd_params* p, const Geom& g, hipStream_t s) {
    // log-gamma of every pixel count a rectangle of this geometry can have (all + 1 <= w*h + 1, myLSD.cpp:1030): host libm values, as
    // the reference computes them; the device only looks them up
    const int lg_need = (int)std::min<long long>(std::max<long long>((long long)g.npx + 2, kLgTable), kLgTableMax);
    if (c->lg_count < lg_need) {
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->done_valid) HIPCHK(c, hipEventSynchronize(c->ev_done));
        std::lock_guard<std::mutex> lk(g_lg_mu);
        const double* tab = log_gamma_table(lg_need);
        if (c->d_lgamma) { HIPCHK(c, hipFree(c->d_lgamma)); c->d_lgamma = nullptr; c->lg_count = 0; }
        HIPCHK(c, hipMalloc(&c->d_lgamma, sizeof(double) * lg_need));
        HIPCHK(c, hipMemcpy(c->d_lgamma, tab, sizeof(double) * lg_need, hipMemcpyHostToDevice));
        c->lg_count = lg_need;
    }
    if (c->tab_valid && memcmp(&c->tab_params, p, sizeof(lsd_params)) == 0) return LSD_OK;
    if (!c->d_ptab) {
        HIPCHK(c, hipMalloc(&c->d_ptab, sizeof(double) * kPTable * 3));
        HIPCHK(c, hipMalloc(&c->d_taps, sizeof(double) * 3 * (2 * kMaxTapRadius + 1)));
        HIPCHK(c, hipMalloc(&c->d_centres, sizeof(int) * kCentreCount));
    }
    std::vector<double> taps;
    gauss_taps(p->sca, p->sig, g.tapR, taps);
    std::vector<int> centres(kCentreCount);                         // myLSD.cpp:428 / :460, the host's own division and rounding
    for (int x = 0; x < kCentreCount; x++) centres[x] = cvt_x86(floor(x / p->sca + 0.5));
    double pt[kPTable * 3];
    double pr = g.aliPro;
    for (int k = 0; k < kPTable; k++) {
        pt[k * 3 + 0] = log(pr); pt[k * 3 + 1] = log10(pr); pt[k * 3 + 2] = log(1 - pr);   // myLSD.cpp:1024,:1033
        pr /= 2.0;                                                                          // :1085,:1149
    }
    HIPCHK(c, hipStreamSynchronize(s));
    // kernels of an earlier enqueue (on whatever stream, which may be gone by now) may still read the tables: wait for the event
    // that enqueue recorded, never for its stream
    if (c->done_valid) HIPCHK(c, hipEventSynchronize(c->ev_done));
    HIPCHK(c, hipMemcpy(c->d_taps, taps.data(), sizeof(double) * taps.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_ptab, pt, sizeof(pt), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_centres, centres.data(), sizeof(int) * kCentreCount, hipMemcpyHostToDevice));
    c->tab_params = *p;
    c->tab_valid = true;
    c->tapR = g.tapR;
    return LSD_OK;
}

template <class T>
static hipError_t re_alloc(T** p, size_t count) {
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; if (e != hipSuccess) return e; }
    if (count == 0) return hipSuccess;
    return hipMalloc((void**)p, count * sizeof(T));
}

// wavefronts per image of the region-stage build a batch of n images runs on (see launch below)
static int waves_for(const lsd_ctx* c, int n) {
    // 8 wavefronts per image (one image per CU) finish an image ~1.5x sooner; 4 (two images per CU) have the higher throughput.
    // While the batch is only a few images per CU its time is that of its heaviest images: 8.  Long batches: 4.
    // (the per-wave workspace -- stamps / spill / gcopy, 12 B per scaled pixel and wave, + result slots -- doubles with 8 waves:
    //  the whole workspace is 56 MB per 2048^2 map on 8 waves and 40 MB on 4 (tools/workspace_size.py: 28.6 / 20.3 GB for the bench
    //  batch of 512); if the 8-wave workspace cannot be allocated the context falls back to 4)
    if (c->region_waves_mode == 8 || (c->region_waves_mode == 0 && !c->prefer4 && n <= 4 * c->num_cus)) return 8;
    return 4;
}

// Workgroups that own no image and help from the start (k_region.hip): as many as the images leave workgroup slots of the device
// free -- one 8-wave workgroup per CU, three 4-wave ones -- and the images' helper wavefronts (tun_help each) can use.
constexpr int kPoolHelpMax = 64;        // helper wavefronts per image the pool is sized for at most (workspace: a wave slot each)
static int pool_for(const lsd_ctx* c, int n, int help) {
    if (help < 0) help = 24;
    if (help > kPoolHelpMax) help = kPoolHelpMax;
    // Measured (tools/single_step_probe.py): the heaviest bench image alone 77.5 -> 30 ms, typical single images unchanged (they never
    // ask: tun_gate); a 64-image shard 49 -> 55 ms and the 512-image batch on 4 waves 87 -> 115 ms -- helpers that are there from the
    // start serve many images that merely look busy, and every remote evaluation costs its owner an export, a poll and a validation
    // out of HBM.  So the pool exists for calls with a handful of images (the reference's usage: one map per call); batches get their
    // helpers from the workgroups that finish first, as before.
    if (help <= 0 || c->trace || n > c->pool_max_images) return 0;
    const int nw = waves_for(c, n);
    const long long free_slots = (long long)(nw == 8 ? 1 : 3) * c->num_cus - n;
    const long long want = ((long long)n * help + nw - 1) / nw;
    const long long p = free_slots < want ? free_slots : want;
    return p > 0 ? (int)p : 0;
}
static int pool_for(const lsd_ctx* c, int n) { return pool_for(c, n, c->tun_help); }
// ... and what the workspace is sized for: the pool of the default help setting even while help is switched off, so that switching
// it on later (lsd_set_region_help) never makes the never-allocating entry point allocate
static int pool_reserve_for(const lsd_ctx* c, int n) {
    const int now = c->tun_help < 0 ? 24 : c->tun_help;
    return pool_for(c, n, now > 24 ? now : 24);
}

// Words per wave of the member-mask array (4 per 8x8 tile) for any image of up to npx scaled pixels: tiles <= npx / 64 + (w + h) / 8 + 1,
// and w, h <= 32766 (make_geom).
static size_t tm_words(size_t npx) { return npx / 16 + 4 * 8200; }

static int ensure_workspace_impl(lsd_ctx* c, size_t n, size_t npx, size_t gpx, int max_lines, bool trace) {
    const size_t need_ws = (n + (size_t)pool_reserve_for(c, (int)n)) * (size_t)waves_for(c, (int)n);   // per-wave arrays: wave slots of the images and of the helper pool
    const bool grow_main = n > c->cap_n || npx > c->cap_npx || gpx > c->cap_gpx || need_ws > c->cap_ws;
    if (grow_main) {
        const size_t nn = n > c->cap_n ? n : c->cap_n, pp = npx > c->cap_npx ? npx : c->cap_npx;
        const size_t gg = gpx > c->cap_gpx ? gpx : c->cap_gpx;
        const size_t ws = need_ws > c->cap_ws ? need_ws : c->cap_ws;
        HIPCHK(c, hipDeviceSynchronize());
        const size_t tot = nn * pp;
        HIPCHK(c, re_alloc(&c->gauss, nn * gg)); HIPCHK(c, re_alloc(&c->mag, tot)); HIPCHK(c, re_alloc(&c->deg, tot));
        HIPCHK(c, re_alloc(&c->sc, tot));
        HIPCHK(c, re_alloc(&c->pw, tot)); HIPCHK(c, re_alloc(&c->epochmap, tot)); HIPCHK(c, re_alloc(&c->ord, tot));
        HIPCHK(c, re_alloc(&c->spill, ws * pp)); HIPCHK(c, re_alloc(&c->gcopy, ws * pp)); HIPCHK(c, re_alloc(&c->wmeta, ws * (size_t)c->mcap));
        HIPCHK(c, re_alloc(&c->stamps, ws * tm_words(pp))); HIPCHK(c, re_alloc(&c->seedidx, tot)); HIPCHK(c, re_alloc(&c->seedpos, tot)); HIPCHK(c, re_alloc(&c->tepoch, nn * (pp / 16 + 4096)));
        HIPCHK(c, hipMemset(c->stamps, 0, ws * tm_words(pp) * sizeof(uint32_t)));
        HIPCHK(c, hipMemset(c->epochmap, 0, tot * sizeof(uint32_t)));
        c->run_id = 0;
        const size_t gs = ws * (size_t)region_slots();                          // result slots: NS per wave slot
        HIPCHK(c, re_alloc(&c->slist, gs * (size_t)c->gcap));
        HIPCHK(c, re_alloc(&c->pend, gs * 24));
        HIPCHK(c, re_alloc(&c->order, nn));
        HIPCHK(c, re_alloc(&c->sets, nn * 256));
        HIPCHK(c, re_alloc(&c->xq, nn * (size_t)(kXStride + 1) + kXHdr));
        HIPCHK(c, re_alloc(&c->maxbits, nn + (nn + 1) / 2)); HIPCHK(c, re_alloc(&c->nb, nn)); HIPCHK(c, re_alloc(&c->nseed, nn));   // (maxbits: [0, nn) the maxima, then nn int32: the gradient pass's near-tie counts -- one memset clears both)
        HIPCHK(c, re_alloc(&c->stats, nn * kStatWords)); HIPCHK(c, re_alloc(&c->rnum, nn * (size_t)region_ring() * 2));
        if (c->seeds) { HIPCHK(c, hipFree(c->seeds)); c->seeds = nullptr; c->cap_trace = false; }
        if (nn != c->cap_n) { c->cap_max_lines = 0; }
        c->cap_n = nn; c->cap_npx = pp; c->cap_gpx = gg; c->cap_ws = ws;
    }
    if (max_lines > c->cap_max_lines) {
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->recs, c->cap_n * (size_t)max_lines * 12));
        HIPCHK(c, re_alloc(&c->recs_scaled, c->cap_n * (size_t)max_lines * 4));
        c->cap_max_lines = max_lines;
    }
    if (trace && !c->cap_trace) {
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMalloc(&c->seeds, c->cap_n * c->cap_npx * sizeof(SeedRec)));
        c->cap_trace = true;
    }
    return LSD_OK;
}

// A failed (re)allocation leaves some arrays freed and others at their old size: forget the whole workspace, so that the next
// call starts from nothing instead of trusting stale capacities.
static int ensure_workspace(lsd_ctx* c, size_t n, size_t npx, size_t gpx, int max_lines, bool trace) {
    int st = ensure_workspace_impl(c, n, npx, gpx, max_lines, trace);
    if (st != LSD_OK) {
        (void)hipGetLastError();                                      // the failed hipMalloc is sticky otherwise
        void** ptrs[] = {(void**)&c->gauss, (void**)&c->mag, (void**)&c->deg, (void**)&c->sc, (void**)&c->pw, (void**)&c->epochmap,
                         (void**)&c->ord, (void**)&c->spill, (void**)&c->gcopy, (void**)&c->wmeta, (void**)&c->stamps,
                         (void**)&c->seedidx, (void**)&c->seedpos, (void**)&c->tepoch, (void**)&c->slist, (void**)&c->pend, (void**)&c->order, (void**)&c->xq, (void**)&c->sets,
                         (void**)&c->maxbits, (void**)&c->nb, (void**)&c->nseed, (void**)&c->stats, (void**)&c->rnum, (void**)&c->seeds,
                         (void**)&c->recs, (void**)&c->recs_scaled};
        for (void** pp : ptrs) if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
        c->cap_n = c->cap_npx = c->cap_gpx = c->cap_ws = 0; c->cap_max_lines = 0; c->cap_trace = false;
        if (st == LSD_ERR_NOMEM && c->region_waves_mode == 0 && !c->prefer4 && waves_for(c, (int)n) == 8) {
            // the 8-wave variant's workspace does not fit: once more with 4 wavefronts per image (half the per-wave arrays)
            c->prefer4 = true;
            return ensure_workspace(c, n, npx, gpx, max_lines, trace);
        }
    }
    return st;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
// The schedule settings of the region stage by name (see lsd_create / lsd_debug_set_tuning)
struct Tuning { const char* name; int lo, hi; int lsd_ctx::*field; bool shipped; };
static const Tuning kTunings[] = {
    {"HELP", -1, 4096, &lsd_ctx::tun_help, true},          // helper wavefronts per image (as lsd_set_region_help)
    {"POOL", 0, 16, &lsd_ctx::pool_max_images, true},      // calls with at most this many images get helper-only workgroups
    {"SOFT", 0, 1 << 20, &lsd_ctx::tun_soft, false},       // look-ahead of the seed hand-out, shallow / deep end (seeds; 0: default)
    {"CLAIM", 0, 1 << 20, &lsd_ctx::tun_claim, false},
    {"FEED", 1, 8, &lsd_ctx::tun_feed, false},             // idle lane groups per refill
    {"BIG", 0, 16, &lsd_ctx::tun_big, false},              // results a wave may have waiting for the cursor (0: default)
    {"EARLY", 0, 4096, &lsd_ctx::tun_early, false},        // helpers before every workgroup has its CU (measured: a loss)
    {"WB", 0, 100, &lsd_ctx::tun_wb, false},               // idle share (%) below which an image asks for help
    {"GATE", 0, 1 << 22, &lsd_ctx::tun_gate, false},       // ... once it has been running for this long (x 1024 clocks)
    {"SHARE", 0, 100, &lsd_ctx::tun_share, false},
    {"UP", 0, 1 << 16, &lsd_ctx::tun_up, false},           // steps of the adaptive look-ahead
    {"DOWN", 0, 1 << 16, &lsd_ctx::tun_down, false},
    {"REQUEUE", 0, 1, &lsd_ctx::tun_requeue, false},       // 0: invalidated results are found at the cursor only
    {"XPOLL", 100, 1 << 30, &lsd_ctx::tun_xpoll, false},   // clocks between two looks of a wave at the help protocol
    {"LINGER", 1, 1 << 30, &lsd_ctx::tun_linger, false},   // looks (~27 us each) a helper takes for an image that asks before it gives its CU back
    {"GROUPS", -1, 1 << 20, &lsd_ctx::tun_groups, false},  // 8-wave region stage as persistent workgroups: -1 (default) as many as CUs when the batch has more images, 0 never, n that many
    {"STOP", 0, 1 << 30, &lsd_ctx::tun_stop, false},       // developer build of the kernel: the seed loop ends after this many seeds (probe experiment)
};

extern "C" {

int lsd_abi_version(void) { return LSD_ABI_VERSION; }

const char* lsd_strerror(int st) {
    switch (st) {
        case LSD_OK: return "ok";
        case LSD_ERR_INVALID: return "invalid argument";
        case LSD_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case LSD_ERR_HIP: return "HIP runtime error";
        case LSD_ERR_UNSUPPORTED: return "parameter outside the implemented range";
        case LSD_ERR_CAPACITY: return "line capacity exceeded";
        case LSD_ERR_NOMEM: return "out of memory (host or device)";
        case LSD_ERR_INTERNAL: return "the region stage gave an image up (its watchdog; see lsd_last_error)";
        default: return "unknown status";
    }
}

void lsd_default_params(lsd_params* p) {                            // LSD/baseFunc.h:64-68
    if (!p) return;
    p->sca = 0.3; p->sig = 0.6; p->angThre = 22.5; p->denThre = 0.7; p->pseBin = 1024;
}

void lsd_scaled_size(int cols, int rows, double sca, int* w, int* h) {
    if (w) *w = cvt_x86(floor(cols * sca));
    if (h) *h = cvt_x86(floor(rows * sca));
}

int lsd_create(lsd_ctx** out, int device) {
    if (!out) return LSD_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LSD_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return LSD_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LSD_ERR_NO_DEVICE;
    lsd_ctx* c = new (std::nothrow) lsd_ctx();
    if (!c) return LSD_ERR_NOMEM;
    c->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return LSD_ERR_HIP; }
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return LSD_ERR_HIP; }
    if (hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) { delete c; return LSD_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return LSD_ERR_HIP; }
    for (int k = 0; k < 2; k++)
        if (hipHostMalloc((void**)&c->pin[k], kPinBytes, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming) != hipSuccess) { delete c; return LSD_ERR_NOMEM; }
    c->last_stream = c->stream;
    // Schedule settings of the region stage.  The shipped library reads two from the environment, once, here: LSD_REGION_HELP (as
    // lsd_set_region_help) and LSD_REGION_POOL (calls with at most this many images get helper-only workgroups).  The others are
    // developer settings: lsd_debug_set_tuning() by name, and -- in the developer builds only (make stats / exp: -DLSD_DEVELOPER_KNOBS)
    // -- LSD_REGION_<NAME> from the environment.  None changes a result (tests/test_parity_gpu.py::
    // test_schedule_of_the_region_stage_changes_nothing); every value is clamped to the range the kernel assumes.
    for (const Tuning& t : kTunings) {
        const std::string name = std::string("LSD_REGION_") + t.name;
        const char* e = getenv(name.c_str());
        if (!e || !*e) continue;
#ifndef LSD_DEVELOPER_KNOBS
        if (!t.shipped) {                               // say so once: a sweep script that drives the shipped library by these would measure one configuration n times
            static bool warned = false;
            if (!warned) fprintf(stderr, "liblsdhip: %s is set but only read by the developer builds (make stats / exp, LSD_HIP_LIB=...); "
                                         "use lsd_debug_set_tuning() with the shipped library\n", name.c_str());
            warned = true;
            continue;
        }
#endif
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end == e) continue;
        c->*(t.field) = (int)(v < t.lo ? t.lo : v > t.hi ? t.hi : v);
    }
    *out = c;
    return LSD_OK;
}

void lsd_destroy(lsd_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {c->gauss, c->mag, c->deg, c->sc, c->recs, c->recs_scaled, c->pw, c->epochmap, c->ord, c->spill, c->gcopy, c->stamps, c->seedidx, c->seedpos, c->tepoch, c->slist, c->pend, c->order, c->xq, c->sets, c->wmeta, c->rnum,
                    c->maxbits, c->nb, c->nseed, c->stats, c->seeds, c->h_in, c->h_lineim, c->h_lines, c->h_counts,
                    c->d_taps, c->d_lgamma, c->d_ptab, c->d_centres, c->mc_claim, c->mc_fa, c->mc_fb, c->mc_ctl, c->mc_in, c->mc_out, c->oc_in, c->oc_out, c->mt_buf, c->ga_cnt, c->ga_slab};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->ga_ev) (void)hipEventDestroy(c->ga_ev);
    if (c->pcount) (void)hipFree(c->pcount);
    for (int k = 0; k < 2; k++) { if (c->pin[k]) (void)hipHostFree(c->pin[k]); if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]); }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->h_flat) (void)hipFree(c->h_flat);
    if (c->h_offs) (void)hipFree(c->h_offs);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* lsd_last_error(const lsd_ctx* c) { return c ? c->err.c_str() : ""; }
void lsd_free(void* p) { free(p); }

int lsd_set_stop_after(lsd_ctx* c, int stage) {
    if (!c || stage < 0 || stage > LSD_STAGE_REGION) return LSD_ERR_INVALID;
    c->stop_after = stage;
    return LSD_OK;
}
int lsd_set_region_waves(lsd_ctx* c, int waves) {
    if (!c || (waves != 0 && waves != 4 && waves != 8)) return LSD_ERR_INVALID;
    c->region_waves_mode = waves;
    return LSD_OK;
}

int lsd_set_region_help(lsd_ctx* c, int waves) {
    if (!c || waves < -1 || waves > 4096) return LSD_ERR_INVALID;
    c->tun_help = waves;
    return LSD_OK;
}

int lsd_set_cost_history(lsd_ctx* c, int on) {
    if (!c) return LSD_ERR_INVALID;
    c->cost_history = on != 0;
    return LSD_OK;
}

int lsd_debug_set_tuning(lsd_ctx* c, const char* name, int value) {
    if (!c || !name) return LSD_ERR_INVALID;
    for (const Tuning& t : kTunings)
        if (strcmp(name, t.name) == 0) {
            c->*(t.field) = value < t.lo ? t.lo : value > t.hi ? t.hi : value;
            return LSD_OK;
        }
    return LSD_ERR_INVALID;
}

int lsd_debug_set_stamp_budget(lsd_ctx* c, unsigned grows) {
    if (!c || grows < 2u || grows > 0xFFFF0u) return LSD_ERR_INVALID;
    c->id_budget = grows;
    return LSD_OK;
}

int lsd_set_trace(lsd_ctx* c, int on) {
    if (!c) return LSD_ERR_INVALID;
    c->trace = on != 0;
    return LSD_OK;
}

int lsd_reserve(lsd_ctx* c, int n, int cols, int rows) {
    if (!c || n <= 0) return LSD_ERR_INVALID;
    lsd_params p; lsd_default_params(&p);
    Geom g;
    int st = make_geom(&p, cols, rows, &g);
    if (st != LSD_OK) return st;
    HIPCHK(c, hipSetDevice(c->device));
    st = ensure_workspace(c, (size_t)n, (size_t)g.npx, (size_t)g.gp * g.h, c->cap_max_lines, c->trace);
    if (st != LSD_OK) return st;
    // ... and the tables: the log-gamma table is sized by the geometry (w*h + 2 host-libm values: ~40 ms of host work and a blocking
    // copy for a 2048^2 map), taps / centres / log p for the default parameters.  Done here, the first enqueue after a reserve neither
    // allocates nor synchronises; ensure_tables' own grow path stays as the fallback for a LARGER geometry or other parameters later
    // (other parameters: three small blocking copies, no allocation).
    return ensure_tables(c, &p, g, nullptr);
}

int lsd_enqueue_batch_device(lsd_ctx* c, uint8_t* d_maps, int n, int cols, int rows, const lsd_params* p,
                             unsigned flags, uint8_t* d_line_ims, lsd_line* d_lines, int max_lines, int32_t* d_counts,
                             void* stream) {
    if (!c || !d_maps || n <= 0 || !d_lines || !d_counts || max_lines <= 0) return LSD_ERR_INVALID;
    Geom g;
    int st = make_geom(p, cols, rows, &g);
    if (st != LSD_OK) return st;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;              // NULL: the default (null) stream, as everywhere in HIP
    st = ensure_workspace(c, (size_t)n, (size_t)g.npx, (size_t)g.gp * g.h, max_lines, c->trace);
    if (st != LSD_OK) return st;
    st = ensure_tables(c, p, g, s);
    if (st != LSD_OK) return st;
    // the workspace is shared by every enqueue of this context: work queued on another stream must be over first
    if (c->done_valid && c->last_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->ev_done, 0));

    Buffers b{};
    b.in = d_maps;
    b.in_rw = (flags & LSD_FLAG_WRITEBACK_MAP) ? d_maps : nullptr;
    b.gauss = c->gauss; b.mag = c->mag; b.deg = c->deg; b.sc = c->sc; b.pw = c->pw; b.epochmap = c->epochmap; b.sets = c->sets; b.maxbits = c->maxbits; b.nb = c->nb; b.ties = reinterpret_cast<int32_t*>(c->maxbits + c->cap_n);
    b.ord = c->ord; b.spill = c->spill; b.gcopy = c->gcopy; b.wmeta = c->wmeta; b.mcap = c->mcap; b.stamps = c->stamps; b.seedidx = c->seedidx; b.seedpos = c->seedpos; b.tepoch = c->tepoch;
    b.tm_stride = 4 * ((g.w + 7) >> 3) * ((g.h + 7) >> 3);
    b.order = c->order; b.slist = c->slist; b.gcap = c->gcap; b.id_budget = c->id_budget; b.pend = c->pend; b.rnum = c->rnum;
    b.recs = c->recs; b.recs_scaled = c->recs_scaled; b.counts = d_counts; b.lines = d_lines; b.line_im = d_line_ims;
    b.max_lines = max_lines;
    {   // the region stage's schedule (k_region.hip): look-ahead of the seed hand-out and of the full evaluations (seeds ahead of the
        // cursor; it adapts between the two values), idle lane groups per refill, results a wave may have waiting for the cursor
        const int nw = waves_for(c, n);
        b.tun_soft = c->tun_soft > 0 ? c->tun_soft : 48 * nw;
        b.tun_claim = c->tun_claim > 0 ? c->tun_claim : 192 * nw;
        b.tun_feed = c->tun_feed;
        b.tun_big = c->tun_big > 0 ? c->tun_big : 3;
        // Help across workgroups is OFF unless lsd_set_region_help asks for it (round 6).  Round 5 had already taken it away from calls of more
        // than 64 images (every image of a large batch pays for the protocol and few are helped: 512 maps 83 ms with, 78 without).  Since the
        // certified sets answer the structures that used to make single images heavy, it loses or ties everywhere else too
        // (profiles/r06q_help_small_probe.log, 2048^2 bench images, with / without): 1 image 8.9 / 8.9 ms, a heavy one 50.5 / 49.9; 2 heavy
        // images 59.2 / 49.7; 4: 55.9 / 48.5; 16: 39.1 / 37.2; 64: 51.4 / 50.2; the reference's maps one per call 1.01 / 0.96 ... 6.37 / 6.36 ms.
        b.tun_help = c->tun_help >= 0 ? c->tun_help : 0;
        b.tun_early = c->tun_early; b.tun_wb = c->tun_wb; b.tun_gate = c->tun_gate; b.tun_share = c->tun_share; b.tun_up = c->tun_up; b.tun_down = c->tun_down; b.tun_requeue = c->tun_requeue;
        b.tun_xpoll = c->tun_xpoll; b.tun_linger = c->tun_linger; b.tun_stop = c->tun_stop;
        b.xq = (b.tun_help > 0 && !c->trace) ? c->xq : nullptr;
        b.npool = b.xq ? pool_for(c, n) : 0;
    }
    b.taps = c->d_taps; b.centres = c->d_centres; b.lgamma = c->d_lgamma; b.lg_count = c->lg_count; b.ptab = c->d_ptab;
    b.seeds = c->trace ? c->seeds : nullptr; b.nseed = c->nseed; b.stats = c->stats;

    // Every dispatch of a batch in flight has to get onto a hardware pipe that the launches of other batches may be holding (a region launch
    // waits for workgroup slots for tens of milliseconds), so a batch costs as few dispatches as it can: ONE fill in front of the kernels (the
    // gradient maxima and, behind them, the gradient pass's near-tie counts); the region stage clears its own counter records and tile epochs
    // image by image; the line counts and list lengths are written by the kernels that own them unless the pipeline is cut short.
    HIPCHK(c, hipMemsetAsync(c->maxbits, 0, sizeof(unsigned long long) * c->cap_n + sizeof(int32_t) * (size_t)n, s));
    const bool full_region = c->stop_after == 0 || c->stop_after >= LSD_STAGE_REGION;
    if (!full_region) HIPCHK(c, hipMemsetAsync(d_counts, 0, sizeof(int32_t) * n, s));            // (else k_region writes every image's count)
    if (c->stop_after != 0 && c->stop_after < LSD_STAGE_SORT) HIPCHK(c, hipMemsetAsync(c->nb, 0, sizeof(int32_t) * (size_t)n, s));   // (else k_sort writes every image's length)

    HIPCHK(c, hipEventRecord(c->ev[0], s));
    // Mat::zeros, myLSD.cpp:215: the Gaussian's tiles clear lineIm on the way where the raster is made of whole 16-byte words; else a
    // kernel of its own does (inside the event window either way: part of "gauss")
    const bool fused_clear = d_line_ims && (((size_t)cols * rows) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_line_ims) & 15) == 0;
    if (d_line_ims && !fused_clear) launch_clear(d_line_ims, (size_t)n * cols * rows, c->num_cus, s);
    launch_gauss(g, b, n, fused_clear ? d_line_ims : nullptr, s);
    if (b.in_rw) launch_remap_writeback(g, b, n, s);
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    if (c->stop_after == 0 || c->stop_after >= LSD_STAGE_GRAD) launch_gradient(g, b, n, s);
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    if (c->stop_after == 0 || c->stop_after >= LSD_STAGE_SORT) { launch_sort(g, b, n, s); launch_order(b, n, g.npx, (c->cost_history && c->hist_n == n && !c->trace) ? c->stats : nullptr, s); }
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    hipStream_t sr = s;
    if (c->stop_after == 0 || c->stop_after >= LSD_STAGE_REGION) {
        // stamps of earlier runs must never look current: every run gets its own 2^20-wide id range
        if (++c->run_id >= 1023u) {
            HIPCHK(c, hipMemsetAsync(c->stamps, 0, c->cap_ws * tm_words(c->cap_npx) * sizeof(uint32_t), sr));
            HIPCHK(c, hipMemsetAsync(c->epochmap, 0, c->cap_n * c->cap_npx * sizeof(uint32_t), sr));   // (the set labels carry the run number too)
            c->run_id = 1;
        }
        // (the counter records and the tile epochs are cleared by the region stage itself, image by image: region_image)
        if (b.xq) HIPCHK(c, hipMemsetAsync(c->xq, 0, sizeof(uint32_t) * ((size_t)n * (kXStride + 1) + kXHdr), sr));
        // 8 wavefronts per image take a whole CU each: worth it up to four images per CU (waves_for); the per-wave workspace
        // (stamps / spill / gcopy: 12 B per scaled pixel and wave) was sized for it by ensure_workspace
        const bool wide = waves_for(c, n) == 8;
        int grid = n;
        const int groups = c->tun_groups >= 0 ? c->tun_groups : c->num_cus;
        if (wide && groups > 0 && groups < n && !b.xq && !c->trace) {           // persistent workgroups (k_region.hip: k_region): fewer workgroups than images
            if (!c->pcount) HIPCHK(c, hipMalloc(&c->pcount, 64));
            HIPCHK(c, hipMemsetAsync(c->pcount, 0, sizeof(int), sr));
            b.pcount = c->pcount; b.nimg = n; b.npool = 0;
            grid = groups;
        }
        if (wide) launch_region_w8(g, b, grid, c->run_id << 20, sr);
        else launch_region_w4(g, b, grid, c->run_id << 20, sr);
    }
    HIPCHK(c, hipEventRecord(c->ev[4], s));
    if (c->stop_after == 0) launch_lines(g, b, n, s);
    HIPCHK(c, hipEventRecord(c->ev[5], s));
    HIPCHK(c, hipEventRecord(c->ev_done, s));
    c->done_valid = true;
    HIPCHK(c, hipGetLastError());
    c->ev_valid = true;
    c->hist_n = (c->stop_after == 0 || c->stop_after >= LSD_STAGE_REGION) ? n : 0;     // (the counter records of this launch: the next one's cost history)
    c->geom = g; c->last_n = n; c->last_max_lines = max_lines; c->last_counts = d_counts; c->last_stream = s;
    return LSD_OK;
}

int lsd_synchronize(lsd_ctx* c) {
    if (!c) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    return LSD_OK;
}

int lsd_last_timings(lsd_ctx* c, float ms[6]) {
    if (!c || !ms || !c->ev_valid) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev[5]));
    for (int i = 0; i < 5; i++) HIPCHK(c, hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    HIPCHK(c, hipEventElapsedTime(&ms[5], c->ev[0], c->ev[5]));
    return LSD_OK;
}

// Host -> device through the two pinned buffers: the memcpy into one overlaps the DMA out of the other.
static int h2d_staged(lsd_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t s) {
    size_t off = 0;
    for (int k = 0; off < bytes; k++) {
        const int b = k & 1;
        const size_t len = bytes - off < kPinBytes ? bytes - off : kPinBytes;
        if (c->pin_used[b]) HIPCHK(c, hipEventSynchronize(c->pin_ev[b]));   // (also a DMA left behind by a call that returned with an error)
        memcpy(c->pin[b], (const uint8_t*)src + off, len);
        HIPCHK(c, hipMemcpyAsync((uint8_t*)dst + off, c->pin[b], len, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(c->pin_ev[b], s));
        c->pin_used[b] = true;
        off += len;
    }
    return LSD_OK;
}
// Device -> host the same way (returns when the bytes are in dst).
static int d2h_staged(lsd_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t s) {
    size_t off = 0, prev_off = 0, prev_len = 0;
    for (int k = 0; off < bytes; k++) {
        const int b = k & 1;
        const size_t len = bytes - off < kPinBytes ? bytes - off : kPinBytes;
        if (c->pin_used[b] && k < 2) HIPCHK(c, hipEventSynchronize(c->pin_ev[b]));
        HIPCHK(c, hipMemcpyAsync(c->pin[b], (const uint8_t*)src + off, len, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->pin_ev[b], s));
        c->pin_used[b] = true;
        if (prev_len) {
            HIPCHK(c, hipEventSynchronize(c->pin_ev[b ^ 1]));
            memcpy((uint8_t*)dst + prev_off, c->pin[b ^ 1], prev_len);
        }
        prev_off = off; prev_len = len;
        off += len;
        if (off >= bytes) {
            HIPCHK(c, hipEventSynchronize(c->pin_ev[b]));
            memcpy((uint8_t*)dst + prev_off, c->pin[b], prev_len);
        }
    }
    return LSD_OK;
}

static int ensure_host_staging(lsd_ctx* c, size_t n, size_t wh, int max_lines, bool lineim) {
    if (n > c->hcap_n || wh > c->hcap_wh || max_lines > c->hcap_max_lines || (lineim && !c->hcap_lineim)) {
        const size_t nn = n > c->hcap_n ? n : c->hcap_n, ww = wh > c->hcap_wh ? wh : c->hcap_wh;
        const int ml = max_lines > c->hcap_max_lines ? max_lines : c->hcap_max_lines;
        const bool li = lineim || c->hcap_lineim;
        c->hcap_n = 0; c->hcap_wh = 0; c->hcap_max_lines = 0; c->hcap_lineim = false;   // (stay 0 if an allocation below fails)
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->h_in, nn * ww));
        HIPCHK(c, re_alloc(&c->h_lineim, li ? nn * ww : 0));
        HIPCHK(c, re_alloc(&c->h_lines, nn * (size_t)ml)); HIPCHK(c, re_alloc(&c->h_flat, nn * (size_t)ml));
        HIPCHK(c, re_alloc(&c->h_counts, nn)); HIPCHK(c, re_alloc(&c->h_offs, nn + 3));
        c->hcap_n = nn; c->hcap_wh = ww; c->hcap_max_lines = ml; c->hcap_lineim = li;
    }
    return LSD_OK;
}

int lsd_run_batch(lsd_ctx* c, uint8_t* maps, int n, int cols, int rows, const lsd_params* p, uint8_t* line_ims,
                  lsd_line** lines_out, int* offsets_out) {
    if (!c || !maps || n <= 0 || !lines_out || !offsets_out) return LSD_ERR_INVALID;
    *lines_out = nullptr;
    Geom g;
    int st = make_geom(p, cols, rows, &g);
    if (st != LSD_OK) return st;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t wh = (size_t)cols * rows;
    const int ml = c->host_max_lines;
    if ((long long)n * ml >= (1ll << 31) / 10) return LSD_ERR_UNSUPPORTED;   // (int32 offsets into the flat record array)
    st = ensure_host_staging(c, (size_t)n, wh, ml, line_ims != nullptr);
    if (st != LSD_OK) return st;
    hipStream_t s = c->stream;
    st = h2d_staged(c, c->h_in, maps, (size_t)n * wh, s);
    if (st != LSD_OK) return st;
    st = lsd_enqueue_batch_device(c, c->h_in, n, cols, rows, p, LSD_FLAG_WRITEBACK_MAP, line_ims ? c->h_lineim : nullptr,
                                  c->h_lines, ml, c->h_counts, s);
    if (st != LSD_OK) return st;
    launch_compact_lines(c->h_lines, c->h_counts, ml, n, c->h_flat, c->h_offs, s);
    // the observable in-place remap is final after K1 (event 1 of the enqueue): it travels back on the second stream while
    // the gradient / sort / region / line kernels run
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev[1], 0));
    st = d2h_staged(c, maps, c->h_in, (size_t)n * wh, c->copy_stream);
    if (st != LSD_OK) return st;
    std::vector<int32_t> offs((size_t)n + 3);                                                  // offsets[n + 1], the overflow count, 1 + first image given up
    st = d2h_staged(c, offs.data(), c->h_offs, sizeof(int32_t) * (size_t)(n + 3), s);          // (waits for the pipeline)
    if (st != LSD_OK) return st;
    memcpy(offsets_out, offs.data(), sizeof(int32_t) * (size_t)(n + 1));
    const int total = offsets_out[n];
    lsd_line* out = (lsd_line*)calloc(total > 0 ? total : 1, sizeof(lsd_line));
    if (!out) return LSD_ERR_NOMEM;
    st = total > 0 ? d2h_staged(c, out, c->h_flat, sizeof(lsd_line) * (size_t)total, s) : LSD_OK;
    if (st == LSD_OK && line_ims) st = d2h_staged(c, line_ims, c->h_lineim, (size_t)n * wh, s);
    if (st != LSD_OK) { free(out); return st; }
    *lines_out = out;
    if (offs[(size_t)n + 2] > 0) {
        // the region stage's watchdog gave an image up (a defect of its protocol, never seen on a released build): that image has
        // no lines in the result, everything else is valid
        c->err = "region stage gave up image " + std::to_string(offs[(size_t)n + 2] - 1) + " of the batch (watchdog; counts = -1)";
        return LSD_ERR_INTERNAL;
    }
    // more lines than host_max_lines in some image: the first host_max_lines of it are returned, and the status says so
    return offs[(size_t)n + 1] > 0 ? LSD_ERR_CAPACITY : LSD_OK;
}

int lsd_set_host_max_lines(lsd_ctx* c, int max_lines) {
    if (!c || max_lines < 1 || max_lines > (1 << 20)) return LSD_ERR_INVALID;
    c->host_max_lines = max_lines;
    return LSD_OK;
}

int lsd_run(lsd_ctx* c, uint8_t* map, int cols, int rows, size_t stride, const lsd_params* p, uint8_t* line_im,
            size_t line_im_stride, lsd_line** lines_out, int* n_lines) {
    if (!c || !map || !lines_out || !n_lines || cols <= 0 || rows <= 0) return LSD_ERR_INVALID;
    if (stride < (size_t)cols || (line_im && line_im_stride < (size_t)cols)) return LSD_ERR_INVALID;
    *n_lines = 0;
    const bool packed = stride == (size_t)cols && (!line_im || line_im_stride == (size_t)cols);
    int offs[2] = {0, 0};
    if (packed) {
        int st = lsd_run_batch(c, map, 1, cols, rows, p, line_im, lines_out, offs);
        *n_lines = offs[1];
        return st;
    }
    std::vector<uint8_t> tmp((size_t)cols * rows), tim(line_im ? (size_t)cols * rows : 0);
    for (int y = 0; y < rows; y++) memcpy(&tmp[(size_t)y * cols], map + (size_t)y * stride, cols);
    int st = lsd_run_batch(c, tmp.data(), 1, cols, rows, p, line_im ? tim.data() : nullptr, lines_out, offs);
    for (int y = 0; y < rows; y++) memcpy(map + (size_t)y * stride, &tmp[(size_t)y * cols], cols);
    if (line_im)
        for (int y = 0; y < rows; y++) memcpy(line_im + (size_t)y * line_im_stride, &tim[(size_t)y * cols], cols);
    *n_lines = offs[1];
    return st;
}

int lsd_last_region_cycles(lsd_ctx* c, int n, long long* cycles_out) {
    // (hist_n: images of the last call whose REGION stage ran -- 0 after a call that lsd_set_stop_after ended earlier, whose counter
    //  records would be stale or zero)
    if (!c || !cycles_out || n <= 0 || n > c->last_n || n > c->hist_n) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy2D(cycles_out, sizeof(long long), c->stats + kStatTotalWord, sizeof(long long) * kStatWords, sizeof(long long), (size_t)n,
                          hipMemcpyDeviceToHost));
    return LSD_OK;
}

int lsd_last_sensitivity(lsd_ctx* c, int n, int* near_ties) {
    if (!c || !near_ties || n <= 0 || n > c->last_n || n > c->hist_n) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    std::vector<long long> reg((size_t)n);
    std::vector<int32_t> grad((size_t)n);
    HIPCHK(c, hipMemcpy2D(reg.data(), sizeof(long long), c->stats + kStatTiesWord, sizeof(long long) * kStatWords, sizeof(long long), (size_t)n,
                          hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(grad.data(), reinterpret_cast<int32_t*>(c->maxbits + c->cap_n), sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        const long long v = reg[(size_t)i] + grad[(size_t)i];
        near_ties[i] = v > 0x7fffffffll ? 0x7fffffff : (int)v;
    }
    return LSD_OK;
}

int lsd_debug_fetch(lsd_ctx* c, int image, int what, void* out, size_t bytes) {
    if (!c || !out || image < 0 || image >= c->last_n) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    const size_t npx = (size_t)c->geom.npx, off = (size_t)image * npx;
    const void* src = nullptr;
    size_t need = 0;
    int32_t nbv = 0, cnt = 0, nseed = 0;
    switch (what) {
        case LSD_DBG_GAUSS:                                           // (rows are padded to gp doubles on the device)
            if (bytes < npx * 8) return LSD_ERR_INVALID;
            HIPCHK(c, hipMemcpy2D(out, (size_t)c->geom.w * 8, c->gauss + (size_t)image * c->geom.gp * c->geom.h, (size_t)c->geom.gp * 8,
                                  (size_t)c->geom.w * 8, (size_t)c->geom.h, hipMemcpyDeviceToHost));
            return LSD_OK;
        case LSD_DBG_MAG: src = c->mag + off; need = npx * 8; break;
        case LSD_DBG_DEG: src = c->deg + off; need = npx * 8; break;
        case LSD_DBG_STATE: src = c->pw + off; need = npx * 4; break;
        case LSD_DBG_ORDER:
        case LSD_DBG_ORDER_VAL:
            HIPCHK(c, hipMemcpy(&nbv, c->nb + image, 4, hipMemcpyDeviceToHost));
            if (what == LSD_DBG_ORDER) { src = c->ord + off; need = (size_t)nbv * 4; }
            else {                                                    // the bin values are recomputed on demand (k_sort.hip: k_ordv)
                if (bytes < (size_t)nbv * 2) return LSD_ERR_INVALID;
                uint16_t* tmp = nullptr;
                HIPCHK(c, hipMalloc(&tmp, (size_t)(nbv > 0 ? nbv : 1) * 2));
                launch_ordv(c->mag + off, c->maxbits + image, c->ord + off, tmp, nbv, c->geom.pseBin, c->last_stream);
                hipError_t e = hipStreamSynchronize(c->last_stream);
                if (e == hipSuccess && nbv > 0) e = hipMemcpy(out, tmp, (size_t)nbv * 2, hipMemcpyDeviceToHost);
                (void)hipFree(tmp);
                HIPCHK(c, e);
                return LSD_OK;
            }
            break;
        case LSD_DBG_NB: src = c->nb + image; need = 4; break;
        case LSD_DBG_MAXGRAD: src = c->maxbits + image; need = 8; break;
        case LSD_DBG_RECS:
            HIPCHK(c, hipMemcpy(&cnt, c->last_counts + image, 4, hipMemcpyDeviceToHost));
            if (cnt > c->last_max_lines) cnt = c->last_max_lines;
            src = c->recs + (size_t)image * c->last_max_lines * 12; need = (size_t)cnt * 12 * 8;
            break;
        case LSD_DBG_NSEED: src = c->nseed + image; need = 4; break;
        case LSD_DBG_SEEDS:
            if (!c->seeds) return LSD_ERR_INVALID;
            HIPCHK(c, hipMemcpy(&nseed, c->nseed + image, 4, hipMemcpyDeviceToHost));
            src = (const SeedRec*)c->seeds + off; need = (size_t)nseed * sizeof(SeedRec);
            break;
        case LSD_DBG_STATS:                                           // (a larger `bytes` reads the records of the following images too)
            src = c->stats + (size_t)image * kStatWords; need = 8 * kStatWords;
            if (bytes > need) { const size_t all = (size_t)(c->last_n - image) * need; need = bytes < all ? bytes / need * need : all; }
            break;
        default: return LSD_ERR_INVALID;
    }
    if (bytes < need) return LSD_ERR_INVALID;
    if (need) HIPCHK(c, hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
    if (what == LSD_DBG_STATE) {                                   // packed pixel words -> usedMap values (lsd_internal.h)
        uint32_t* o = static_cast<uint32_t*>(out);
        for (size_t i = 0; i < npx; i++) o[i] = pw_used(o[i]);
    }
    return LSD_OK;
}

int lsd_enqueue_map_cache_device(lsd_ctx* c, const uint8_t* d_maps, int n, int cols, int rows, double res,
                                 double z_occ_max_dis, double* d_out, void* stream) {
    if (!c || !d_maps || !d_out || n <= 0 || cols <= 0 || rows <= 0 || !(res > 0) || !(z_occ_max_dis >= 0)) return LSD_ERR_INVALID;
    if ((long long)cols * rows >= (1ll << 31)) return LSD_ERR_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;              // NULL: the default (null) stream, as everywhere in HIP
    const size_t need = (size_t)n * cols * rows;
    if (need > c->mc_cap) {
        c->mc_cap = 0;                                   // (stays 0 if an allocation below fails)
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->mc_claim, need)); HIPCHK(c, re_alloc(&c->mc_fa, need * 2)); HIPCHK(c, re_alloc(&c->mc_fb, need * 2));
        c->mc_cap = need;
    }
    if ((size_t)n > c->mc_ctl_n) {                                                     // frontier sizes + up to 64 chunk counts per map
        c->mc_ctl_n = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->mc_ctl, (size_t)n * (2 + 64)));
        c->mc_ctl_n = (size_t)n;
    }
    const int cell_radius = cvt_x86(floor(z_occ_max_dis / res));           // myLSD.cpp:13
    // few maps: spread each over G workgroups (kernel per level phase); many maps: one workgroup per map, one launch
    int G = (2 * c->num_cus) / n;
    if (G > 64) G = 64;
    if (G >= 4)                                                    // measured crossover: 128 maps 33 vs 42 ms, 256 maps 67 vs 56 ms
        launch_mapcache_spread(d_maps, d_out, c->mc_claim, c->mc_fa, c->mc_fb, c->mc_ctl, c->mc_ctl + 2 * (size_t)n, n, G, cols, rows, res,
                               z_occ_max_dis, cell_radius, s);
    else
        launch_mapcache(d_maps, d_out, c->mc_claim, c->mc_fa, c->mc_fb, n, cols, rows, res, z_occ_max_dis, cell_radius, s);
    HIPCHK(c, hipGetLastError());
    c->last_stream = s;
    return LSD_OK;
}

int lsd_map_cache(lsd_ctx* c, const uint8_t* map, int cols, int rows, size_t stride, double res, double z_occ_max_dis,
                  double* out) {
    if (!c || !map || !out || cols <= 0 || rows <= 0 || stride < (size_t)cols) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t wh = (size_t)cols * rows;
    if (wh > c->mc_hcap) {
        c->mc_hcap = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->mc_in, wh)); HIPCHK(c, re_alloc(&c->mc_out, wh));
        c->mc_hcap = wh;
    }
    int st;
    if (stride == (size_t)cols) { st = h2d_staged(c, c->mc_in, map, wh, c->stream); if (st != LSD_OK) return st; }
    else HIPCHK(c, hipMemcpy2DAsync(c->mc_in, cols, map, stride, cols, rows, hipMemcpyHostToDevice, c->stream));
    st = lsd_enqueue_map_cache_device(c, c->mc_in, 1, cols, rows, res, z_occ_max_dis, c->mc_out, c->stream);
    if (st != LSD_OK) return st;
    return d2h_staged(c, out, c->mc_out, wh * sizeof(double), c->stream);
}

int lsd_enqueue_occupancy_to_map_device(lsd_ctx* c, const int8_t* d_grid, size_t n_cells, uint8_t* d_map, void* stream) {
    if (!c || !d_grid || !d_map || n_cells == 0) return LSD_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(d_grid) | reinterpret_cast<uintptr_t>(d_map)) & 15u) return LSD_ERR_INVALID;   // 16-byte accesses
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;              // NULL: the default (null) stream, as everywhere in HIP
    launch_occ_to_map(reinterpret_cast<const uint8_t*>(d_grid), d_map, n_cells, s);
    HIPCHK(c, hipGetLastError());
    c->last_stream = s;
    return LSD_OK;
}

int lsd_occupancy_to_map(lsd_ctx* c, const int8_t* grid, int cols, int rows, uint8_t* map_out, size_t map_stride) {
    if (!c || !grid || !map_out || cols <= 0 || rows <= 0 || map_stride < (size_t)cols) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t wh = (size_t)cols * rows;
    if (wh > c->oc_cap) {
        c->oc_cap = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->oc_in, wh)); HIPCHK(c, re_alloc(&c->oc_out, wh));
        c->oc_cap = wh;
    }
    HIPCHK(c, hipMemcpyAsync(c->oc_in, grid, wh, hipMemcpyHostToDevice, c->stream));
    const int st = lsd_enqueue_occupancy_to_map_device(c, reinterpret_cast<const int8_t*>(c->oc_in), wh, c->oc_out, c->stream);
    if (st != LSD_OK) return st;
    HIPCHK(c, hipMemcpy2DAsync(map_out, map_stride, c->oc_out, cols, cols, rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LSD_OK;
}

int lsd_enqueue_scan_to_map_match_device(lsd_ctx* c, const double* d_map_cache, int cols, int rows, const lsd_line* d_map_lines,
                                         const lsd_line* d_scan_lines, const lsd_position* d_pts, int n_points,
                                         lsd_position lidar, lsd_position last, const int* d_pairs, int n_pairs,
                                         double z_occ_max_dis, double max_esti_dist, lsd_match_score* d_out, void* stream) {
    if (!c || !d_map_cache || !d_map_lines || !d_scan_lines || !d_pairs || !d_out || cols <= 0 || rows <= 0 || n_pairs <= 0 ||
        n_points < 0 || (n_points > 0 && !d_pts))
        return LSD_ERR_INVALID;
    if (n_pairs > (1 << 28)) return LSD_ERR_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;              // NULL: the default (null) stream, as everywhere in HIP
    launch_match(d_map_cache, cols, rows, d_map_lines, d_scan_lines, reinterpret_cast<const double*>(d_pts), n_points, lidar.x,
                 lidar.y, last.x, last.y, d_pairs, n_pairs, z_occ_max_dis, max_esti_dist, reinterpret_cast<double*>(d_out), s);
    HIPCHK(c, hipGetLastError());
    c->last_stream = s;
    return LSD_OK;
}

int lsd_scan_to_map_match(lsd_ctx* c, const double* map_cache, int cols, int rows, const lsd_line* map_lines, int n_map,
                          const lsd_line* scan_lines, int n_scan, const lsd_position* pts, int n_points, lsd_position lidar,
                          lsd_position last, const int* pairs, int n_pairs, double z_occ_max_dis, double max_esti_dist,
                          lsd_match_score* out) {
    if (!c || !map_cache || !map_lines || !scan_lines || !pairs || !out || cols <= 0 || rows <= 0 || n_map <= 0 || n_scan <= 0 ||
        n_pairs <= 0 || n_points < 0 || (n_points > 0 && !pts))
        return LSD_ERR_INVALID;
    for (int p = 0; p < n_pairs; p++)
        if (pairs[2 * p] < 0 || pairs[2 * p] >= n_map || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_scan) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b_mc = (size_t)cols * rows * sizeof(double), b_ml = (size_t)n_map * sizeof(lsd_line), b_sl = (size_t)n_scan * sizeof(lsd_line);
    const size_t b_pt = (size_t)(n_points > 0 ? n_points : 1) * sizeof(lsd_position), b_pr = (size_t)n_pairs * 2 * sizeof(int);
    const size_t b_out = (size_t)n_pairs * 4 * sizeof(lsd_match_score);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t total = up(b_mc) + up(b_ml) + up(b_sl) + up(b_pt) + up(b_pr) + up(b_out);
    if (total > c->mt_cap) {
        c->mt_cap = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->mt_buf, total));
        c->mt_cap = total;
    }
    uint8_t* base = c->mt_buf;
    double* d_mc = reinterpret_cast<double*>(base); base += up(b_mc);
    lsd_line* d_ml = reinterpret_cast<lsd_line*>(base); base += up(b_ml);
    lsd_line* d_sl = reinterpret_cast<lsd_line*>(base); base += up(b_sl);
    lsd_position* d_pt = reinterpret_cast<lsd_position*>(base); base += up(b_pt);
    int* d_pr = reinterpret_cast<int*>(base); base += up(b_pr);
    lsd_match_score* d_out = reinterpret_cast<lsd_match_score*>(base);
    HIPCHK(c, hipMemcpyAsync(d_mc, map_cache, b_mc, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_ml, map_lines, b_ml, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_sl, scan_lines, b_sl, hipMemcpyHostToDevice, c->stream));
    if (n_points > 0) HIPCHK(c, hipMemcpyAsync(d_pt, pts, (size_t)n_points * sizeof(lsd_position), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_pr, pairs, b_pr, hipMemcpyHostToDevice, c->stream));
    const int st = lsd_enqueue_scan_to_map_match_device(c, d_mc, cols, rows, d_ml, d_sl, d_pt, n_points, lidar, last, d_pr, n_pairs,
                                                        z_occ_max_dis, max_esti_dist, d_out, c->stream);
    if (st != LSD_OK) return st;
    HIPCHK(c, hipMemcpyAsync(out, d_out, b_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LSD_OK;
}

int lsd_enqueue_feature_scan_batch_device(lsd_ctx* c, const lsd_polar* d_scans, const int* d_lens, int n_scans, int stride,
                                          lsd_map_param mp, int region_point_limit, double thre_line, double line_dist_thre_m,
                                          lsd_line* d_lines_out, int* d_n_lines, lsd_position* d_pts_out, int pts_cap, int* d_n_pts,
                                          double* d_lidar_pos, int* d_im_size, void* stream) {
    if (!c || !d_scans || !d_lens || n_scans <= 0 || stride <= 0 || !d_lines_out || !d_n_lines || !d_n_pts || !d_lidar_pos || !d_im_size ||
        pts_cap < 0 || (pts_cap > 0 && !d_pts_out) || !(mp.mapResol > 0))
        return LSD_ERR_INVALID;
    if (stride > rdp_max_len()) return LSD_ERR_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;              // NULL: the default (null) stream, as everywhere in HIP
    launch_rdp(reinterpret_cast<const double*>(d_scans), d_lens, n_scans, stride, mp.oriMapCol, mp.oriMapRow, mp.mapResol, mp.mapOriX, mp.mapOriY,
               region_point_limit, thre_line, line_dist_thre_m, d_lines_out, d_n_lines, reinterpret_cast<double*>(d_pts_out), pts_cap, d_n_pts,
               d_lidar_pos, d_im_size, s);
    HIPCHK(c, hipGetLastError());
    c->last_stream = s;
    return LSD_OK;
}

int lsd_feature_scan_batch(lsd_ctx* c, const lsd_polar* scans, const int* lens, int n_scans, int stride, lsd_map_param mp,
                           int region_point_limit, double thre_line, double line_dist_thre_m, lsd_line* lines_out, int* n_lines,
                           lsd_position* pts_out, int pts_cap, int* n_pts, double* lidar_pos, int* im_size) {
    if (!c || !scans || !lens || n_scans <= 0 || stride <= 0 || !lines_out || !n_lines || !n_pts || !lidar_pos || !im_size || pts_cap < 0 ||
        (pts_cap > 0 && !pts_out))
        return LSD_ERR_INVALID;
    for (int i = 0; i < n_scans; i++) if (lens[i] < 0 || lens[i] > stride) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b_sc = (size_t)n_scans * stride * sizeof(lsd_polar), b_len = (size_t)n_scans * sizeof(int);
    const size_t b_li = (size_t)n_scans * LSD_RDP_MAX_LINES * sizeof(lsd_line), b_pt = (size_t)n_scans * (pts_cap > 0 ? pts_cap : 1) * sizeof(lsd_position);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t total = up(b_sc) + up(b_len) + up(b_li) + up(b_pt) + 4 * up((size_t)n_scans * 16);
    if (total > c->mt_cap) {                           // (shares the staging buffer of the matching entry point)
        c->mt_cap = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->mt_buf, total));
        c->mt_cap = total;
    }
    uint8_t* base = c->mt_buf;
    lsd_polar* d_sc = reinterpret_cast<lsd_polar*>(base); base += up(b_sc);
    int* d_len = reinterpret_cast<int*>(base); base += up(b_len);
    lsd_line* d_li = reinterpret_cast<lsd_line*>(base); base += up(b_li);
    lsd_position* d_pt = reinterpret_cast<lsd_position*>(base); base += up(b_pt);
    int* d_nl = reinterpret_cast<int*>(base); base += up((size_t)n_scans * 16);
    int* d_np = reinterpret_cast<int*>(base); base += up((size_t)n_scans * 16);
    double* d_lp = reinterpret_cast<double*>(base); base += up((size_t)n_scans * 16);
    int* d_sz = reinterpret_cast<int*>(base);
    HIPCHK(c, hipMemcpyAsync(d_sc, scans, b_sc, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_len, lens, b_len, hipMemcpyHostToDevice, c->stream));
    const int st = lsd_enqueue_feature_scan_batch_device(c, d_sc, d_len, n_scans, stride, mp, region_point_limit, thre_line, line_dist_thre_m,
                                                         d_li, d_nl, d_pt, pts_cap, d_np, d_lp, d_sz, c->stream);
    if (st != LSD_OK) return st;
    HIPCHK(c, hipMemcpyAsync(lines_out, d_li, b_li, hipMemcpyDeviceToHost, c->stream));
    if (pts_cap > 0) HIPCHK(c, hipMemcpyAsync(pts_out, d_pt, (size_t)n_scans * pts_cap * sizeof(lsd_position), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(n_lines, d_nl, b_len, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(n_pts, d_np, b_len, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lidar_pos, d_lp, (size_t)n_scans * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(im_size, d_sz, (size_t)n_scans * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n_scans; i++)
        if (n_lines[i] > LSD_RDP_MAX_LINES) return LSD_ERR_CAPACITY;      // more chords than the 360 records per scan hold (the first 360 are valid)
    return LSD_OK;
}

int lsd_gather_lines(lsd_ctx* c, const lsd_comm* comm, const lsd_line* d_lines, const int32_t* d_counts, int n_local, int max_lines,
                     int n_total, int cap_rows, int32_t* d_counts_all, lsd_line* d_slabs_all, void* stream) {
    if (!c || !comm || !comm->all_gather || comm->world <= 0 || comm->rank < 0 || comm->rank >= comm->world || n_total <= 0 || n_local < 0 ||
        max_lines <= 0 || cap_rows <= 0 || !d_counts_all || !d_slabs_all || (n_local > 0 && (!d_lines || !d_counts)))
        return LSD_ERR_INVALID;
    int lo, hi, per;
    lsd_shard_range(n_total, comm->world, comm->rank, &lo, &hi);
    if (hi - lo != n_local) return LSD_ERR_INVALID;                      // the caller's shard is not the one lsd_shard_range gives this rank
    lsd_gather_layout(n_total, comm->world, &per, nullptr);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t need_cnt = (size_t)(per + 2) + (size_t)(n_local > 0 ? n_local : 1);
    if (need_cnt > c->ga_cnt_cap) {
        c->ga_cnt_cap = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->ga_cnt, need_cnt));
        c->ga_cnt_cap = need_cnt;
    }
    if ((size_t)cap_rows > c->ga_slab_cap) {
        c->ga_slab_cap = 0;
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, re_alloc(&c->ga_slab, (size_t)cap_rows));
        c->ga_slab_cap = (size_t)cap_rows;
    }
    // the staging buffers belong to the context: an earlier hand-off's collectives (on whatever stream) must have read them
    if (c->ga_ev_valid) HIPCHK(c, hipStreamWaitEvent(s, c->ga_ev, 0));
    // rows past this rank's lines are zero (nothing stale travels)
    HIPCHK(c, hipMemsetAsync(c->ga_slab, 0, sizeof(lsd_line) * (size_t)cap_rows, s));
    launch_pack_lines(d_lines, d_counts, n_local, max_lines, per, cap_rows, c->ga_cnt, c->ga_cnt + (per + 2), c->ga_slab, s);
    HIPCHK(c, hipGetLastError());
    if (comm->all_gather(comm->user, c->ga_cnt, d_counts_all, sizeof(int32_t) * (size_t)(per + 2), s) != 0 ||
        comm->all_gather(comm->user, c->ga_slab, d_slabs_all, sizeof(lsd_line) * (size_t)cap_rows, s) != 0) {
        c->err = "lsd_gather_lines: the communicator's all_gather failed";
        return LSD_ERR_HIP;
    }
    if (!c->ga_ev) HIPCHK(c, hipEventCreateWithFlags(&c->ga_ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ga_ev, s));
    c->ga_ev_valid = true;
    c->last_stream = s;
    return LSD_OK;
}

int lsd_debug_calibrate(lsd_ctx* c, size_t bytes) {
    if (!c || bytes < 8) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    double* buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, bytes));
    launch_calib(buf, bytes / 8, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(buf);
    return LSD_OK;
}

int lsd_debug_eval_math(lsd_ctx* c, int fn, const double* a, const double* b, double* out0, double* out1, size_t n) {
    if (!c || !a || !out0 || !out1 || n == 0 || fn < 0 || fn > 6 || ((fn == 1 || fn == 6) && !b)) return LSD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    double *da = nullptr, *db = nullptr, *d0 = nullptr, *d1 = nullptr;
    HIPCHK(c, hipMalloc(&da, n * 8)); HIPCHK(c, hipMalloc(&db, n * 8)); HIPCHK(c, hipMalloc(&d0, n * 8)); HIPCHK(c, hipMalloc(&d1, n * 8));
    HIPCHK(c, hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
    if (b) HIPCHK(c, hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice));
    launch_dbgmath(fn, da, db, d0, d1, n, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out0, d0, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out1, d1, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(d0); (void)hipFree(d1);
    return LSD_OK;
}

}  // extern "C"
