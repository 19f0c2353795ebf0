// lsd_internal.h -- shared between the host side (lsd_ctx.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lsd_hip.h"

namespace lsdhip {

constexpr double kPi = 3.14159265358979323846;  // == 4.0*atan(1.0), myLSD.cpp:9
constexpr int kCentreCount = 32768;             // scaled coordinates are below 32766 (make_geom)
constexpr int kMaxTapRadius = 40;               // hSize = 2*h+1 <= 81 taps per phase kernel
constexpr int kLgTable = 16384;                 // host-tabulated log-gamma entries every context starts with; it grows to w*h + 2 (the largest
constexpr int kLgTableMax = 1 << 23;            //    pixel count a rectangle can have, + 1), up to this many
constexpr int kStatWords = 48;                  // counters per image of the region stage (lsd_debug_fetch LSD_DBG_STATS)
constexpr int kStatTiesWord = 39;               // ... and this one its decisions within the libm's noise (k_region.hip: ST_TIES; lsd_last_sensitivity)
constexpr int kStatTotalWord = 8;               // ... of which this one holds the shader clocks the stage spent on the image (k_region.hip: ST_TOTAL)
constexpr int kPTable = 16;                     // host-tabulated log(p), log10(p), log(1-p) for p = aliPro/2^k
// Help across workgroups in the region stage (k_region.hip): a control block of 32-bit words per launch, cleared before it.
//   per image (kXStride words): [0] accept epoch [1] commit cursor [2] image finished [3] helper wavefronts attached
//   [4] seeds waiting for a full evaluation; [R, 2R) requests (seed + 1, bit 31: taken by a helper); [2R, 3R) flags of the
//   answers (seed + 1); [3R, 11R) the answers, 8 words each: result state, result slot, accept epoch of the snapshot, box (2),
//   list sizes (R = kXReq).  After the n image records (kXHdr words): [0] workgroups started [1] images finished [2] length
//   of the list of images that ask for help [3] early helper wavefronts [4] s_memrealtime of the launch's start | 1; the list (image + 1) follows.
constexpr int kXStride = 768, kXReq = 64, kXHdr = 16;

// Geometry + thresholds of one (cols, rows, params) configuration; computed on the host with the
// host libm so that they are the very numbers the reference computes (myLSD.cpp:132-133,148-149,207-209).
struct Geom {
    int W, H;            // original size
    int w, h;            // scaled size
    int npx;             // w*h
    int gp;              // row pitch (doubles) of the Gaussian image: w rounded up to 16, so that every row starts on a 128-byte line
    double sca;
    int tapR;            // Gaussian tap radius h (myLSD.cpp:393)
    int pseBin;
    double degThre;      // angThre/180*pi
    double gradThre;     // 2/sin(degThre)
    double logNT;        // 5*(log10(h)+log10(w))/2
    double regThre;      // -logNT/log10(angThre/180)
    double aliPro;       // angThre/180
    double denThre;
};

// The packed pixel word pw[q] of the scaled image (written by K2, read by the region stage):
//   bits 31..2  the level-line angle degMap[q] as fp32 with the two lowest mantissa bits cleared (|error| < 1e-6 rad)
//   bits  1..0  usedMap code: 0 free, 1 banned by the gradient threshold (myLSD.cpp:165-166), 2 marked by a rejected
//               region (usedMap == 2: growable, not seedable), 3 banned by an accepted line (usedMap == 1; the line's
//               epoch is in epochmap[q], written before the code).  usedMap value = code == 3 ? 1 : code.
constexpr uint32_t kPwFree = 0u, kPwStatic = 1u, kPwRejected = 2u, kPwLine = 3u;
__host__ __device__ inline uint32_t pw_used(uint32_t w) { return (w & 3u) == 3u ? 1u : (w & 3u); }

// Per-batch device buffers (image i lives at base + i*stride of each array).
struct Buffers {
    const uint8_t* in;     // n x H x W (pitch W)
    uint8_t* in_rw;        // same pointer when write-back of the remap is requested, else null
    double* gauss;         // n x h x gp (row pitch gp: see Geom)
    double* mag;           // n x npx
    double* deg;           // n x npx
    double2* sc;           // n x npx : (sin, cos)(deg), written where usedMap == 0 after the gradient pass
    uint32_t* pw;          // n x npx : packed pixel word (see above)
    uint32_t* epochmap;    // n x npx : accept epoch of pixels with code 3; for growable pixels the label of their certified set, tagged with the run number (k_region.hip; cleared with the stamps when the run numbers wrap)
    uint32_t* sets;        // n x 256 : sizes of the certified sets of the launch by label, 0 = none / ended (cleared by the region stage itself)
    uint32_t* tepoch;      // n x ceil(w/8) x ceil(h/8) : per tile, epoch + 1 of the latest accepted line with a pixel in it (cleared per run)
    unsigned long long* maxbits;  // n : bit pattern of max gradient (non-negative double)
    int32_t* nb;           // n : sorted-list length
    int32_t* ties;         // n : decisions of the gradient pass within the libm's noise (lsd_last_sensitivity; the region stage counts its own in the counter records)
    uint32_t* ord;         // n x npx : sorted seed list (y*w+x)
    uint32_t* stamps;      // n x NW x tm_stride : per wave, the member masks of the tiles its cache has evicted (4 words per 8x8 tile: grow id, -, 64 bits)
    int tm_stride;         // words per wave of the above: 4 x tiles of the scaled image
    uint32_t* spill;       // n x NW x npx : region list beyond the LDS part
    uint32_t* gcopy;       // n x NW x npx : grow-order copy used when RegionRadiusReducer reorders the list
    float4* wmeta;         // n x NW x mcap : per list entry (unit sum vector, sin of the smallest slack) of its last neighbourhood test
    int mcap;
    int* rnum;             // n x RW x 2 : sizes/outcome of published records (seed trace only)
    uint32_t* order;       // n : image indices, heaviest (largest nb) first: the region stage's workgroup -> image map
    uint32_t* seedidx;     // n x npx : sorted-list indices of the potential seeds (usedMap == 0 after the gradient pass)
    uint32_t* seedpos;     // n x npx : their pixels (y*w+x), same order
    uint32_t* slist;       // n x NW x NS x gcap : lists of the speculative results in flight (examined pixels, pixels to mark)
    int gcap;
    uint32_t id_budget;    // curMap stamp ids per wave and run (k_region.hip: grow())
    int tun_soft, tun_claim, tun_feed, tun_big;   // schedule of the region stage (k_region.hip; lsd_ctx.hip has the defaults)
    uint32_t* xq;          // n x kXStride + kXHdr + n : control block of the help across workgroups (see above), null: no help
    int tun_help;          // helper wavefronts an image may have attached
    int* pcount;           // persistent workgroups of the 8-wave region stage (k_region.hip: k_region; lsd_ctx.hip: tun_groups): the launch's image counter (zeroed before the launch), else null
    int nimg;              // ... and the number of images its workgroups share out
    int npool;             // workgroups at the end of the region stage's launch that own no image and help from the start (workspace slots n .. n + npool - 1)
    int tun_early;         // workgroups that may help while others still wait for a CU
    int tun_wb;            // an image asks for help while its waves idle less than this share of the time (percent)
    int tun_gate;          // ... and once it has been running for this long (1024-clock units)
    int tun_share;         // ... and for at least this share (percent) of the time since the launch began
    int tun_up, tun_down;  // steps of the adaptive look-ahead (seeds): up when a wave finds nothing to do, down on a redo / discard
    int tun_requeue;       // results invalidated by a line are queued for another evaluation when the line is accepted (1) or found at the cursor (0)
    int tun_xpoll;         // shader clocks between two looks of a wave at the help protocol
    int tun_linger;        // looks (~27 us each) a helper wavefront takes for an image that asks before it gives its CU back
    int tun_stop;          // experiments: the seed loop ends after this many potential seeds (0: all)
    double* pend;          // n x NW x NS x 24 : finished results that mark usedMap, waiting for their turn to commit
    double* recs;          // n x max_lines x 12 (structRec before rescale)
    double* recs_scaled;   // n x max_lines x 4 (x1 y1 x2 y2 after the 1/sca rescale)
    int32_t* counts;       // n
    lsd_line* lines;       // n x max_lines
    uint8_t* line_im;      // n x H x W or null
    int max_lines;
    // tables
    const double* taps;    // 3 x (2*tapR+1)
    const int* centres;    // kCentreCount: (int)floor(x / sca + 0.5) for every scaled coordinate x (K1's window centres, myLSD.cpp:428 / :460)
    const double* lgamma;  // lg_count entries: LogGammaCalculator(i) from the host libm
    int lg_count;
    const double* ptab;    // kPTable x 3 : log(p), log10(p), log(1-p)
    // debug
    void* seeds;           // n x npx trace records or null
    int32_t* nseed;        // n
    long long* stats;      // n x kStatWords
};

struct SeedRec {  // mirrors oracle's orc_seed
    int order_idx, x, y, num, outcome, final_num;
    double logNFA;
};

// launchers (each enqueues on `s`)
void launch_gauss(const Geom& g, const Buffers& b, int n, uint8_t* clr, hipStream_t s);   // clr: lineIm to be cleared on the way, or null
void launch_remap_writeback(const Geom& g, const Buffers& b, int n, hipStream_t s);
void launch_gradient(const Geom& g, const Buffers& b, int n, hipStream_t s);
void launch_sort(const Geom& g, const Buffers& b, int n, hipStream_t s);
void launch_ordv(const double* mag, const unsigned long long* maxbits, const uint32_t* ord, uint16_t* out, int count, int pseBin, hipStream_t s);   // debug fetch of the bin values
void launch_order(const Buffers& b, int n, int npx, const long long* hist, hipStream_t s);   // hist: the last launch's counter records (cost history) or null
// the region stage with 4 resp. 8 wavefronts per image (k_region.hip is compiled twice)
void launch_region_w4(const Geom& g, const Buffers& b, int n, uint32_t id_base, hipStream_t s);
void launch_region_w8(const Geom& g, const Buffers& b, int n, uint32_t id_base, hipStream_t s);
int region_slots();    // result slots per wavefront (slist and pend are sized x this)
int region_waves();    // wavefronts per image of the wider variant: what the workspace is sized for
int region_ring();     // commit-ring records per image (rnum is sized x this x 2)
void launch_calib(double* buf, size_t n, hipStream_t s);
void launch_match(const double* map_cache, int cols, int rows, const lsd_line* map_lines, const lsd_line* scan_lines,
                  const double* pts, int n_points, double lidx, double lidy, double lastx, double lasty, const int* pairs,
                  int n_pairs, double zmax, double max_esti_dist, double* out, hipStream_t s);
void launch_occ_to_map(const uint8_t* in, uint8_t* out, size_t n, hipStream_t s);
void launch_mapcache_spread(const uint8_t* maps, double* out, unsigned long long* claim, uint32_t* fr_a, uint32_t* fr_b, int* ctl,
                            int* cnt, int n, int G, int W, int H, double res, double zmax, int cell_radius, hipStream_t s);
void launch_mapcache(const uint8_t* maps, double* out, unsigned long long* claim, uint32_t* fr_a, uint32_t* fr_b, int n,
                     int W, int H, double res, double zmax, int cell_radius, hipStream_t s);
void launch_lines(const Geom& g, const Buffers& b, int n, hipStream_t s);
void launch_clear(uint8_t* p, size_t bytes, int num_cus, hipStream_t s);   // lineIm = zeros (k_lines.hip)
void launch_compact_lines(const lsd_line* lines, const int32_t* counts, int max_lines, int n, lsd_line* flat, int32_t* offsets, hipStream_t s);
void launch_rdp(const double* scans, const int* lens, int n, int stride, int oriMapCol, int oriMapRow, double mapResol, double mapOriX,
                double mapOriY, int region_point_limit, double thre_line, double line_dist_thre_m, lsd_line* lines_out, int* n_lines,
                double* pts_out, int pts_cap, int* n_pts, double* lidar_pos, int* im_size, hipStream_t s);
int rdp_max_len();
void launch_pack_lines(const lsd_line* lines, const int32_t* counts, int n_local, int max_lines, int per, int cap_rows, int32_t* cpad,
                       int32_t* offs, lsd_line* slab, hipStream_t s);
void launch_dbgmath(int fn, const double* a, const double* b, double* o0, double* o1, size_t n, hipStream_t s);

// x86-64 cvttsd2si semantics of the reference's (int) casts: NaN, +-inf and
// out-of-range values give INT_MIN.  v_cvt_i32_f64 would give 0 / saturate instead.
__host__ __device__ inline int cvt_x86(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

}  // namespace lsdhip
