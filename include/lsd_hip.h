/*
 * lsd_hip.h -- C ABI of liblsdhip.so, the MI355X (gfx950) implementation of the LSD hot path of
 * Pyrokine/LineSegmentDetector-SLAM.
 *
 * The reference has no FFI layer: its boundary for this path is the C++ free function
 *     structLSD mylsd::myLineSegmentDetector(Mat MapGray, int oriMapCol, int oriMapRow,
 *                                            double sca, double sig, double angThre,
 *                                            double denThre, int pseBin);      LSD/myLSD.h:132
 * (definition LSD/myLSD.cpp:129-376), called from LSD/main_on_windows.cpp:70 and
 * LSD/main_on_linux.cpp:132.  Every entry point below states the reference interface it
 * replaces.  Plain pointers and sizes only; no C++/torch types; status codes, no exceptions.
 * The C++ adapter with the reference's own names lives in include/myLSD.h.
 *
 * There is NO CPU fallback: lsd_create() fails with LSD_ERR_NO_DEVICE when no gfx950 GPU is
 * visible, and nothing in this library computes the path on the host.
 */
#ifndef LSD_HIP_H
#define LSD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSD_ABI_VERSION 1

/* status codes */
enum {
    LSD_OK = 0,
    LSD_ERR_INVALID = 1,        /* bad argument */
    LSD_ERR_NO_DEVICE = 2,      /* no usable HIP device (the library never falls back to the CPU) */
    LSD_ERR_HIP = 3,            /* a HIP runtime call failed; see lsd_last_error() */
    LSD_ERR_UNSUPPORTED = 4,    /* parameter outside the implemented range (e.g. pseBin > 1024) */
    LSD_ERR_CAPACITY = 5,       /* more lines than max_lines in at least one image */
    LSD_ERR_NOMEM = 6,
    LSD_ERR_INTERNAL = 7        /* the region stage's watchdog gave an image up (a defect detector; lsd_last_error() names the image) */
};

/* The five LSD knobs of myLineSegmentDetector (LSD/myLSD.h:132); defaults LSD/baseFunc.h:64-68. */
typedef struct lsd_params {
    double sca;      /* 0.3  */
    double sig;      /* 0.6  */
    double angThre;  /* 22.5 */
    double denThre;  /* 0.7  */
    int pseBin;      /* 1024 */
} lsd_params;

/* Layout-identical to structLinesInfo (LSD/baseFunc.h:33-44): 9 doubles + int, sizeof == 80. */
typedef struct lsd_line {
    double k, b, dx, dy, x1, y1, x2, y2, len;
    int orient;
} lsd_line;

typedef struct lsd_ctx lsd_ctx;

/* --- lifetime ------------------------------------------------------------------------ */
/* Creates a context bound to HIP device `device` (one context per GPU / per process rank). */
int lsd_create(lsd_ctx **out, int device);
void lsd_destroy(lsd_ctx *ctx);
const char *lsd_strerror(int status);
/* Text of the last HIP failure seen by this context ("" if none). */
const char *lsd_last_error(const lsd_ctx *ctx);
/* Fills the reference defaults (LSD/baseFunc.h:64-68). */
void lsd_default_params(lsd_params *p);
int lsd_abi_version(void);
/* Frees buffers returned through lines_out. */
void lsd_free(void *p);

/* --- the hot path, host buffers -------------------------------------------------------- */
/* Replaces mylsd::myLineSegmentDetector (LSD/myLSD.h:132, LSD/myLSD.cpp:129).
 *   map        IN-OUT rows x cols uint8, row pitch `stride` bytes.  As in the reference the
 *              caller's image is rewritten (1 -> 255, 255 -> 0 for y >= 1, x >= 1; myLSD.cpp:135-142).
 *   line_im    rows x cols uint8 (pitch line_im_stride), receives structLSD.lineIm (0/255), or NULL.
 *   lines_out  receives a malloc'ed array of *n_lines lsd_line (structLSD.linesInfo; free with lsd_free).
 * Blocking; uses the context's own streams.  Host buffers travel through pinned double buffers; the rewritten map comes
 * back while the rest of the pipeline still runs; the lines arrive compacted in one copy. */
int lsd_run(lsd_ctx *ctx, uint8_t *map, int cols, int rows, size_t stride, const lsd_params *p,
            uint8_t *line_im, size_t line_im_stride, lsd_line **lines_out, int *n_lines);

/* Batch of n equally sized images packed back to back (image i at maps + i*rows*cols, pitch cols).
 * offsets_out[n+1] receives the prefix sums of the per-image line counts; *lines_out the
 * concatenated lines (malloc'ed).  maps are rewritten like lsd_run does.  line_ims may be NULL. */
int lsd_run_batch(lsd_ctx *ctx, uint8_t *maps, int n, int cols, int rows, const lsd_params *p,
                  uint8_t *line_ims, lsd_line **lines_out, int *offsets_out);

/* Line capacity per image of the host entry points above (default 8192; the device entry point takes it as an argument).
 * An image with more lines returns its first max_lines and the call reports LSD_ERR_CAPACITY (lines_out is still valid). */
int lsd_set_host_max_lines(lsd_ctx *ctx, int max_lines);

/* --- the hot path, device-resident batch ------------------------------------------------ */
/* Same computation on buffers that already live in HBM (e.g. torch tensors' data_ptr()).
 *   d_maps     n x rows x cols uint8, read-only unless LSD_FLAG_WRITEBACK_MAP is set
 *   d_line_ims n x rows x cols uint8 or NULL
 *   d_lines    n x max_lines lsd_line (image i's lines start at d_lines + i*max_lines)
 *   d_counts   n int32 line counts.  A count > max_lines means that image overflowed (its first max_lines lines are valid; the
 *              host entry points report LSD_ERR_CAPACITY).  A count of -1 means the region stage's watchdog gave the image up
 *              (no wavefront of its workgroup found anything to do for seconds: a defect of the commit protocol, never seen
 *              on a released build): the image has no valid lines, its lineIm is blank, the other images are unaffected;
 *              the host entry points report LSD_ERR_INTERNAL and treat the image as having no lines.
 *   stream     hipStream_t on which to enqueue (NULL = the default stream, as everywhere in HIP).  Asynchronous:
 *              returns after enqueueing; workspace is (re)allocated before the first launch only
 *              when (n, cols, rows) grew. */
#define LSD_FLAG_WRITEBACK_MAP 1u
/* Known cliff: the wavefronts of an image's workgroup evaluate different seeds; ONE region's frontier is walked by one wavefront.  An
 * image made of regions of ten thousand pixels and more (sawtooth test images) returns the reference's result but takes 35-93 ms,
 * 2.8-3.3 x ONE host thread of the algorithm (measured: profiles/r06g_cliff_probe.log), where occupancy maps (largest region of the
 * reference's maps: 680 pixels) run hundreds of times faster than that thread in batches.  lsd_last_region_cycles shows it: occupancy
 * maps cost 57-350 cycles per scaled pixel, such images 850-1000. */
int lsd_enqueue_batch_device(lsd_ctx *ctx, uint8_t *d_maps, int n, int cols, int rows,
                             const lsd_params *p, unsigned flags, uint8_t *d_line_ims,
                             lsd_line *d_lines, int max_lines, int32_t *d_counts, void *stream);
/* Pre-sizes the workspace so that lsd_enqueue_batch_device never allocates: 103 B per scaled pixel on 4 wavefronts per image, 146 B on
 * 8 (lsd_set_region_waves, below; call it first), i.e. 39 / 55 MB per 2048 x 2048 map at sca 0.3 -- 20.0 / 28.2 GB for 512 of them
 * (tools/workspace_size.py).  The workspace is sized for the helper pool of the default help setting even while help is off, so that
 * lsd_set_region_help never makes a later enqueue allocate.  The lookup tables are sized here as well (log-gamma of every pixel count of
 * this geometry, the Gaussian taps of the DEFAULT parameters): an enqueue of a larger geometry than any reserved one still grows them
 * (one synchronisation, one allocation), an enqueue with other parameters rewrites the small ones (blocking copies, no allocation). */
int lsd_reserve(lsd_ctx *ctx, int n, int cols, int rows);
/* Blocks until the stream used by the last enqueue is idle. */
int lsd_synchronize(lsd_ctx *ctx);

/* --- multi-GPU: image shards and the hand-off of the line lists (BASELINE configs[4]) ------------------------- */
/* The reference runs one map on one host (LSD/main_on_windows.cpp:67-70, LSD/main_on_linux.cpp:130-132); a batch of independent
 * maps is sharded over the GPUs of a node, one process (or thread) and one lsd_ctx per GPU, with no collective on the data path.
 * The only exchange is the result hand-off below.
 * lsd_shard_range: the contiguous shard [*lo, *hi) of n_items images that rank `rank` of `world` takes. */
void lsd_shard_range(int n_items, int world, int rank, int *lo, int *hi);
/* Cost-aware deal of the same shards: perm[n_items] orders the images such that rank r, taking perm[lo_r .. hi_r) (lsd_shard_range),
 * carries about the same total cost as every other rank (longest-processing-time-first, deterministic; inside a shard ascending image
 * index).  costs[i] >= 0: e.g. the region stage's cycles of image i in the previous step (lsd_last_region_cycles) -- maps of one site
 * cost about the same from step to step.  The gathered lists then arrive in perm order: image perm[g] at position g. */
int lsd_shard_balanced(const long long *costs, int n_items, int world, int *perm);
/* Shader clocks the region stage spent on each of the first n images of the context's last batch (synchronises): the cost
 * lsd_shard_balanced deals by.  An image the region stage gave up reports 0.  LSD_ERR_INVALID if the context's last call did not run
 * the region stage on at least n images (lsd_set_stop_after). */
int lsd_last_region_cycles(lsd_ctx *ctx, int n, long long *cycles_out);

/* How much of the last batch's result hangs on the last place of the libm.  The reference decides with glibc's sin / cos / atan2 / exp /
 * log10 / pow (growth test LSD/myLSD.cpp:540-543, orientation flip :655-665, density :829 / :869, Refiner's width test :845, Reducer's
 * radius test :780, rectangle edges against pixel rows :973-1004, aligned count :1009-1013, RectangleImprover's comparisons of logNFA
 * :1075-1156, the "pi -> 0" rule :170); this library evaluates the same functions correctly rounded, and glibc's results are within
 * one ulp of those.  near_ties[i] = the number of decisions image i's evaluations took with their operands closer than that ulp can
 * move them (speculative evaluations included: an upper bound).  0: any libm within one ulp of correct rounding gives the same
 * decisions -- the lines, usedMap and lineIm of image i are the reference's on every such platform.  > 0: on this image a platform's
 * libm may decide (about one random map in 4 000 differs from the glibc build in one decision; each of those has
 * a count > 0 -- the property tools/campaign.py checks on every image).  Synchronises; LSD_ERR_INVALID if the last call did not run the
 * region stage on at least n images. */
int lsd_last_sensitivity(lsd_ctx *ctx, int n, int *near_ties);

/* A communicator as this library sees it: who am I, how many are we, and ONE operation -- an all-gather of equally sized device
 * buffers (d_recv holds world x bytes_per_rank, rank r's bytes at r * bytes_per_rank), enqueued on `stream`, 0 on success. */
typedef struct lsd_comm {
    int rank, world;
    int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
    void *user;
} lsd_comm;
/* Binds *out to an RCCL communicator (an ncclComm_t, passed as void*): rank / world from ncclCommUserRank / ncclCommCount,
 * all_gather = ncclAllGather(..., ncclInt8, comm, stream) -- RCCL over xGMI between the GPUs of a node.  The RCCL symbols are taken
 * from the RCCL the calling process has loaded (the library the communicator belongs to; a second copy is never loaded);
 * LSD_ERR_UNSUPPORTED if the process has none. */
int lsd_comm_from_rccl(void *nccl_comm, lsd_comm *out);

/* Sizes of the gathered arrays: *per_rank = images of the largest shard; *counts_words = world * (per_rank + 2) int32. */
int lsd_gather_layout(int n_total, int world, int *per_rank, size_t *counts_words);

/* Hands this rank's line lists to every rank.  d_lines / d_counts: the outputs of lsd_enqueue_batch_device for this rank's shard
 * (n_local == the size lsd_shard_range gives comm->rank, else LSD_ERR_INVALID).  On `stream`, without host synchronisation:
 *   1. the records are packed on the device, image-major, into a slab of cap_rows records (rows past the rank's lines are zero);
 *   2. all-gather of the padded counts:  d_counts_all [world][per_rank + 2] int32 -- rank r's per-image counts (clamped to
 *      max_lines, zero-padded), then [per_rank] = rows in its slab, [per_rank + 1] = flags: bit 0 it had to drop rows (more than
 *      cap_rows lines or an image over max_lines), bit 1 an image the region stage gave up (lsd_gather_unpack: LSD_ERR_CAPACITY
 *      resp. LSD_ERR_INTERNAL, with everything that did arrive in its outputs);
 *   3. all-gather of the slabs:          d_slabs_all [world][cap_rows] lsd_line.
 * Image g of the batch (rank r = its shard, local index j) has its lines at d_slabs_all[r][sum of counts[r][0..j)].
 * ~22 MB per GPU and step for the 512 x 2048^2 bench batch at cap_rows = 512 per image.
 * The packing stages through buffers of the CONTEXT: like lsd_enqueue_batch_device, one context serves one stream at a time -- a
 * second hand-off from the same context is ordered behind the first one's collectives (an event), whatever its stream. */
int lsd_gather_lines(lsd_ctx *ctx, const lsd_comm *comm, const lsd_line *d_lines, const int32_t *d_counts, int n_local, int max_lines,
                     int n_total, int cap_rows, int32_t *d_counts_all, lsd_line *d_slabs_all, void *stream);
/* Host side of the hand-off: turns HOST copies of the two gathered arrays into offsets_out[n_total + 1] and (if lines_out is
 * not NULL) the lines of all images in global image order (lines_cap records available).  LSD_ERR_CAPACITY if a rank flagged
 * dropped rows (what arrived is still unpacked). */
int lsd_gather_unpack(const int32_t *counts_all, const lsd_line *slabs_all, int n_total, int world, int cap_rows, int32_t *offsets_out,
                      lsd_line *lines_out, size_t lines_cap);

/* --- createMapCache (SURVEY 8f "next" #1) ------------------------------------------------------ */
/* Replaces mylsd::createMapCache(Mat MapGray, double res) (LSD/myLSD.h:131, LSD/myLSD.cpp:11-127): distance (metres)
 * from every cell to the occupied cell (value 1) whose breadth-first flood reaches it first, capped at
 * z_occ_max_dis (LSD/baseFunc.h:60; unreachable cells hold z_occ_max_dis, occupied cells 0).
 *   map  rows x cols uint8 (pitch `stride`), read-only -- call it BEFORE lsd_run, which rewrites the map (Q2);
 *   out  rows x cols doubles, packed (the CV_64FC1 mapCache FeatureAssociation reads, LSD/myFA.cpp:371-380).
 * Bit-identical to the reference (integer flood order + one exactly rounded sqrt and multiply per cell). */
int lsd_map_cache(lsd_ctx *ctx, const uint8_t *map, int cols, int rows, size_t stride, double res,
                  double z_occ_max_dis, double *out);
/* The same for n packed device-resident maps (d_out: n x rows x cols doubles), asynchronous on `stream`. */
int lsd_enqueue_map_cache_device(lsd_ctx *ctx, const uint8_t *d_maps, int n, int cols, int rows, double res,
                                 double z_occ_max_dis, double *d_out, void *stream);

/* --- scan-to-map matching batch (SURVEY 8f "next" #2) ------------------------------------------- */
/* Replaces the body of myfa::thread_ScanToMapMatch (LSD/myFA.cpp:197-270) with NormalizedLineDirection (:272-305),
 * rotateScanIm (:307-357) and CalcScore (:359-396), which the reference runs once per (map line, scan line) pair on a
 * 30-thread pool.  For pair p = {cntMapLine, cntScanLine} and matching i = 1..4 (:205-249)
 *   out[4 * p + i - 1] = { rotated lidar pose (x, y, angDiff in (-180, 180]), score }
 * with score = INFINITY where rotateScanIm rejects the candidate (farther than max_esti_dist from last_pose, :330) or
 * CalcScore does (:388).  The caller keeps score < 3 and sorts, as FeatureAssociation does (:60-100).
 * lsd_position == structPosition (LSD/baseFunc.h:46-50); lines are structLinesInfo; map_cache is the rows x cols
 * CV_64FC1 array of lsd_map_cache.  Scores agree with the reference arithmetic up to the libm caveat of lsd_run. */
typedef struct lsd_position { double x, y, ang; } lsd_position;
typedef struct lsd_match_score { lsd_position pos; double score; } lsd_match_score;
int lsd_scan_to_map_match(lsd_ctx *ctx, const double *map_cache, int cols, int rows,
                          const lsd_line *map_lines, int n_map, const lsd_line *scan_lines, int n_scan,
                          const lsd_position *scan_im_points, int n_points, lsd_position lidar_pose, lsd_position last_pose,
                          const int *pairs, int n_pairs, double z_occ_max_dis, double max_esti_dist, lsd_match_score *out);
/* The same with every array resident on the device (d_map_cache as written by lsd_enqueue_map_cache_device), asynchronous
 * on `stream`; pair indices are NOT range-checked here. */
int lsd_enqueue_scan_to_map_match_device(lsd_ctx *ctx, const double *d_map_cache, int cols, int rows,
                                         const lsd_line *d_map_lines, const lsd_line *d_scan_lines,
                                         const lsd_position *d_scan_im_points, int n_points, lsd_position lidar_pose,
                                         lsd_position last_pose, const int *d_pairs, int n_pairs, double z_occ_max_dis,
                                         double max_esti_dist, lsd_match_score *d_out, void *stream);

/* --- scan-line extraction batch (SURVEY 8f "next" #4) ------------------------------------------- */
/* Replaces myrdp::FeatureScan (LSD/myRDP.cpp:9-185; RegionSegmentation :274-345, SplitMerge / SplitMergeAssistant :187-272,
 * getThresholdDeltaDist :347-368; caller LSD/main_on_windows.cpp:127) for a BATCH of lidar scans: n_scans scans at a pitch of
 * `stride` readings, scan i holding lens[i] <= stride finite readings (range, angle) -- what the caller's read loop leaves
 * after dropping the infinite ranges (:115-121).  Per scan i:
 *   lines_out[i * 360 .. ]   the line records (structLinesInfo) in the reference's order; n_lines[i] is their number, of which the
 *                            first 360 are stored (the reference's malloc, :39, which it would overrun: the host entry point returns
 *                            LSD_ERR_CAPACITY when a scan has more, the stored records are valid)
 *   pts_out[i * pts_cap .. ] scanImPoint: the pixels of the lines' rasters (x, y, 0) in the reference's order; n_pts[i] is
 *                            their number, of which the first pts_cap are stored
 *   lidar_pos[2 i .. ]       structLidarPointRec lidarPos (x, y), :33-36;  im_size[2 i ..] = (cols, rows) of FS.lineIm, :31
 * FS.lineIm itself is im_size zeros with 255 at the listed pixels.  region_point_limit / thre_line / line_dist_thre_m are
 * rdp_leastPoint / rdp_threLine / rdp_leastDist (LSD/baseFunc.h:70-72: 3, 0.08, 0.5).  lens[i] <= 1024.
 * lsd_polar is structLidarPointPolar (LSD/myRDP.h:34-38) without its `split` work flag; lsd_map_param is structMapParam
 * (LSD/baseFunc.h:25-31). */
typedef struct lsd_polar { double range, angle; } lsd_polar;
typedef struct lsd_map_param { int oriMapCol, oriMapRow; double mapResol, mapOriX, mapOriY; } lsd_map_param;
#define LSD_RDP_MAX_LINES 360
int lsd_feature_scan_batch(lsd_ctx *ctx, const lsd_polar *scans, const int *lens, int n_scans, int stride, lsd_map_param map_param,
                           int region_point_limit, double thre_line, double line_dist_thre_m, lsd_line *lines_out, int *n_lines,
                           lsd_position *pts_out, int pts_cap, int *n_pts, double *lidar_pos, int *im_size);
/* The same with every array resident on the device, asynchronous on `stream`. */
int lsd_enqueue_feature_scan_batch_device(lsd_ctx *ctx, const lsd_polar *d_scans, const int *d_lens, int n_scans, int stride,
                                          lsd_map_param map_param, int region_point_limit, double thre_line, double line_dist_thre_m,
                                          lsd_line *d_lines_out, int *d_n_lines, lsd_position *d_pts_out, int pts_cap, int *d_n_pts,
                                          double *d_lidar_pos, int *d_im_size, void *stream);

/* --- wire format (SURVEY 8f "next" #3) ---------------------------------------------------------- */
/* Replaces the cell loop of the ROS map callback (LSD/main_on_linux.cpp:108-124): nav_msgs/OccupancyGrid cells
 * (int8: -1 unknown, 0 free, 1..100 occupied) become the loader's map values (0 unknown, 255 free, 1 occupied), the
 * input of lsd_map_cache and lsd_run.  Host buffers: grid rows x cols packed, map_out with pitch map_stride. */
int lsd_occupancy_to_map(lsd_ctx *ctx, const int8_t *grid, int cols, int rows, uint8_t *map_out, size_t map_stride);
/* The same for n_cells device-resident cells (any number of equally sized grids back to back), asynchronous on
 * `stream`; both pointers 16-byte aligned.  Lets a map that arrives on the device never touch the host. */
int lsd_enqueue_occupancy_to_map_device(lsd_ctx *ctx, const int8_t *d_grid, size_t n_cells, uint8_t *d_map, void *stream);

/* --- introspection used by the parity tests and the bench ------------------------------- */
/* Scaled size of a cols x rows map: w = floor(cols*sca), h = floor(rows*sca) (myLSD.cpp:132-133). */
void lsd_scaled_size(int cols, int rows, double sca, int *w, int *h);

/* Stops the pipeline after a stage (parity tests of intermediate maps); 0 = run everything. */
enum { LSD_STAGE_ALL = 0, LSD_STAGE_GAUSS = 1, LSD_STAGE_GRAD = 2, LSD_STAGE_SORT = 3, LSD_STAGE_REGION = 4 };
int lsd_set_stop_after(lsd_ctx *ctx, int stage);
/* Enables the per-seed trace buffer (LSD_DBG_SEEDS); costs one record store per grown seed. */
int lsd_set_trace(lsd_ctx *ctx, int on);
/* Region stage variant: 4 wavefronts per image (three images per CU: the throughput build) or 8 (one image per CU, ~1.5x lower
 * latency per image, images taken heaviest first).  0 (default) picks 8 while the batch has at most four images per CU (the step is
 * bounded by its heaviest image until then) and 4 beyond.  Results do not depend on the choice.  Set it before lsd_reserve: the
 * per-wave workspace (8 B per scaled pixel and wavefront + result slots) is sized for the variant. */
int lsd_set_region_waves(lsd_ctx *ctx, int waves);
/* Help across workgroups in the region stage: wavefronts of workgroups whose image is finished evaluate seeds of the images still
 * running (up to `waves` helper wavefronts per image; 0 or -1, the default since round 6: none); an image asks once it has run for
 * ~5 ms with its own wavefronts busy, and a call with up to four images also gets helper-only workgroups from the start.  OFF by
 * default: it paid while single images were dominated by long uniform structures grown again and again (round 4: a heavy 2048 x 2048
 * map 77 -> 30 ms); since those are answered from certified sets it ties or loses at every call size (one heavy map 50.5 ms with, 49.9
 * without; two 59.2 / 49.7; 64 maps 51.4 / 50.2; 512 maps 83 / 78; the reference's maps one per call 1.01 / 0.96 ... 6.37 / 6.36 ms:
 * profiles/r06q_help_small_probe.log), and a caller that keeps several batches in flight needs none either.  The machinery stays
 * available and tested (lsd_set_region_help(ctx, 24)); results do not depend on it. */
int lsd_set_region_help(lsd_ctx *ctx, int waves);
/* Scheduling hint: "the batches this context gets hold the same maps from call to call" (a site's maps, re-extracted as they are
 * updated).  The region stage then starts the images in descending order of the time each one took in the context's previous call with
 * the same number of images, instead of by their count of gradient pixels (which predicts the cost poorly): a batch run alone ends with
 * its heaviest image, and that image should start first.  Off by default; results never depend on it. */
int lsd_set_cost_history(lsd_ctx *ctx, int on);
/* Test hook: the region stage marks the pixels of the region it is growing with a fresh 32-bit id per grow; a wavefront that
 * uses up its 2^20 ids within one run clears its stamp array and starts over.  That takes more than a million grows by one
 * wavefront on one image; this lowers the budget (2 .. 0xFFFF0 grows) so that tests reach the path.  Results do not change. */
int lsd_debug_set_stamp_budget(lsd_ctx *ctx, unsigned grows);
/* Test / developer hook: a schedule setting of the region stage by name ("SOFT", "CLAIM", "FEED", "BIG", "EARLY", "WB", "GATE", "SHARE",
 * "UP", "DOWN", "REQUEUE", "XPOLL", "LINGER", "HELP", "POOL"; csrc/lsd_ctx.hip: kTunings), clamped to its range.  None changes a
 * result.  From the ENVIRONMENT the shipped library takes two settings only, when a context is created: LSD_REGION_HELP (as
 * lsd_set_region_help) and LSD_REGION_POOL (calls with at most that many images, 0..16, default 4, get helper-only workgroups). */
int lsd_debug_set_tuning(lsd_ctx *ctx, const char *name, int value);

/* Copies an intermediate of image `image` of the LAST run/enqueue to host memory (synchronises).
 *   GAUSS/MAG/DEG  h*w doubles      (GaussImage / magMap / degMap, myLSD.cpp:143-147)
 *   STATE          h*w uint32       usedMap value 0 / 1 / 2 (myLSD.cpp:145)
 *   ORDER          nb uint32        sorted seed list, element = y*w + x (binCell after qsort, :204)
 *   ORDER_VAL      nb uint16        its bin values
 *   NB             1 int32          len_binCell
 *   MAXGRAD        1 double
 *   RECS           count*12 doubles accepted structRec before rescale (x1 y1 x2 y2 wid cX cY deg dx dy p prec)
 *   SEEDS          n_seed records {int order_idx, x, y, num, outcome, final_num; double logNFA}
 *   NSEED          1 int32
 *   STATS          32 int64         grow_calls, grown_px, nfa_calls, rrr_calls, rrr_passes, rrr_sentinel, rrr_oob, cycles_rrr*,
 *                                   cycles_total, cycles_grow*, cycles_rect*, cycles_nfa*, cycles_mark*, max_region*, nfa_px*, seeds,
 *                                   exact_angle_evals*, tile_fetches*, batches*, cycles_tiles*, spec_redos, spec_discards, cycles_wait*,
 *                                   resweep_batches*, slow_batches*, cycles_eval*, cycles_sums*, cycles_refine*, cycles_idle*,
 *                                   cycles_select*, cycles_commit*, filter_skips*
 *                                   (summed over the wavefronts that share an image; * = counted by the developer build only,
 *                                   `make stats` -> liblsdhip_stats.so, and 0 in the product build)
 * Returns LSD_ERR_INVALID if `bytes` is smaller than the item. */
enum { LSD_DBG_GAUSS = 1, LSD_DBG_MAG, LSD_DBG_DEG, LSD_DBG_STATE, LSD_DBG_ORDER, LSD_DBG_ORDER_VAL,
       LSD_DBG_NB, LSD_DBG_MAXGRAD, LSD_DBG_RECS, LSD_DBG_SEEDS, LSD_DBG_NSEED, LSD_DBG_STATS };
int lsd_debug_fetch(lsd_ctx *ctx, int image, int what, void *out, size_t bytes);

/* Test hook: evaluates the DEVICE build of the path's transcendental functions on host arrays of n
 * doubles: fn 0 = sin/cos(a) -> out0,out1; fn 1 = atan2(a, b) -> out0; fn 2 = atan(a) -> out0;
 * fn 3 = the region stage's fp32 ESTIMATE of sin/cos of the packed angle of a (its error bound is a test). */
int lsd_debug_eval_math(lsd_ctx *ctx, int fn, const double *a, const double *b, double *out0, double *out1, size_t n);

/* Profiling hook: streams `bytes` once with 8-B-per-lane stores (k_calib_write8) and once with 8-B-per-lane loads
 * (k_calib_read8) so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be calibrated for the front end's access shape. */
int lsd_debug_calibrate(lsd_ctx *ctx, size_t bytes);

/* Per-kernel device time (ms) of the last lsd_run/lsd_run_batch, measured with HIP events on the
 * context's stream: [0] gauss, [1] gradient, [2] sort, [3] region, [4] lines, [5] total. */
int lsd_last_timings(lsd_ctx *ctx, float ms_out[6]);

#ifdef __cplusplus
}
#endif
#endif /* LSD_HIP_H */
