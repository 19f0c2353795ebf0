/*
 * lsd_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's LSD hot path
 * (Pyrokine/LineSegmentDetector-SLAM, LSD/myLSD.cpp) used ONLY as the checker for the HIP
 * implementation: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * it; nothing under linesegmentdetector-slam_amd/ links, imports or executes it.
 *
 * Pinning: the reference itself cannot be rebuilt in this image (it needs OpenCV and Eigen
 * headers, both absent; hand-written stand-ins are not allowed), so the oracle is pinned
 * against the reference outputs recorded in SURVEY.md section 8c / Appendix A (line counts,
 * lit-pixel counts, mapCache sums for 8 maps and the %.17g map1 line list -- all produced by
 * the unmodified reference) and, loosely, against the reference's own MATLAB-era golden
 * files data/MaplinesInfo.txt / data/MaplineIm.txt.  See tests/test_oracle.py.
 */
#ifndef LSD_ORACLE_H
#define LSD_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* layout-identical to structLinesInfo (LSD/baseFunc.h:33-44), sizeof == 80 */
typedef struct {
    double k, b, dx, dy, x1, y1, x2, y2, len;
    int orient;
} orc_line;

/* per-seed trace record (one per RegionGrower call made from the seed loop, myLSD.cpp:225) */
typedef struct {
    int order_idx;   /* index into the sorted list */
    int x, y;        /* seed */
    int num;         /* region size returned by the first grow */
    int outcome;     /* 0 small (myLSD.cpp:228), 1 refine failed (:237), 2 NFA reject (:242), 3 accepted */
    int final_num;   /* region size after Refiner */
    double logNFA;   /* after RectangleImprover (0 if not reached) */
} orc_seed;

typedef struct {
    int w, h;             /* scaled size (myLSD.cpp:132-133) */
    double *gauss;        /* [h*w]  GaussianSampler output */
    double *mag;          /* [h*w]  myLSD.cpp:164 */
    double *deg;          /* [h*w]  myLSD.cpp:172 */
    uint8_t *used0;       /* [h*w]  usedMap right after the gradient loop (:166) */
    uint8_t *used;        /* [h*w]  usedMap at the end of the seed loop */
    int nb;               /* sorted-list length (cnt_binCell) */
    int *ord_v, *ord_x, *ord_y;  /* [nb] sorted list (value, x, y) after qsort (:204) */
    double maxGrad;
    int n_seed;           /* number of trace records */
    orc_seed *seeds;
    double *recs;         /* [n_lines*12] accepted structRec before rescale: x1 y1 x2 y2 wid cX cY deg dx dy p prec */
    /* counters */
    long grow_calls, grown_px, nfa_calls, rrr_calls, rrr_passes, rrr_sentinel_drops, rrr_oob_reads;
    /* how close RectangleImprover's comparisons came to a tie, as margins (>= 1: two libms within an ulp of the correctly rounded
     * functions cannot decide differently): the smallest |logNFA| / noise of a value compared with 0 (:1075, :242) and the smallest
     * |v - best| / (noise(v) + noise(best)) of two different NFA values compared (:1086 ...); HUGE_VAL: none */
    double nfa_min_abs, nfa_min_gap;
} orc_debug;

/* Restates mylsd::myLineSegmentDetector (LSD/myLSD.h:132, LSD/myLSD.cpp:129-376).
 * map is IN-OUT (the reference mutates the caller's image, myLSD.cpp:135-142).
 * lineIm (rows*cols, may be NULL) receives the 0/255 raster (:215, :343-355).
 * *lines is malloc'ed (caller frees with orc_free), *n = len_linesInfo.
 * dbg may be NULL; if not, arrays are malloc'ed and released with orc_debug_free. */
int orc_lsd(uint8_t *map, int cols, int rows, size_t stride,
            double sca, double sig, double angThre, double denThre, int pseBin,
            uint8_t *lineIm, orc_line **lines, int *n, orc_debug *dbg);

/* Restates mylsd::createMapCache (LSD/myLSD.cpp:11-127); out is rows*cols doubles. */
int orc_map_cache(const uint8_t *map, int cols, int rows, size_t stride,
                  double res, double z_occ_max_dis, double *out);

/* Pins SURVEY 8a-Q4: glibc qsort + the reference comparator (myLSD.cpp:486-489) must equal a
 * stable descending sort.  Returns 0 when the two orders agree on a pseudo-random list. */
int orc_selftest_qsort_stable(int n, unsigned seed);

/* the reference's log-gamma (myLSD.cpp:882-924) exposed for table checks */
double orc_log_gamma(int x);

void orc_free(void *p);
void orc_debug_free(orc_debug *dbg);


/* Restates one batch of myfa::thread_ScanToMapMatch (LSD/myFA.cpp:197-270): for every (map line, scan line) pair the
 * four start/end matchings (:205-249), NormalizedLineDirection (:272-305), rotateScanIm (:307-357) and CalcScore
 * (:359-396).  out[4 * pair + i - 1] = {rotated lidar pose, score}; score = INFINITY where rotateScanIm rejects the
 * candidate (:327).  PARITY UNPINNED: the reference holds no input/output pair for this loop and cannot be built here;
 * the GPU path is compared with this restatement only. */
typedef struct { double x, y, ang; } orc_position;                 /* structPosition, LSD/baseFunc.h:46-50 */
typedef struct { orc_position pos; double score; } orc_match_score;
int orc_scan_to_map_match(const double *map_cache, int cols, int rows,
                          const orc_line *map_lines, int n_map, const orc_line *scan_lines, int n_scan,
                          const orc_position *scan_im_points, int n_points, orc_position lidar_pose, orc_position last_pose,
                          const int *pairs, int n_pairs, double z_occ_max_dis, double max_esti_dist, orc_match_score *out);

#ifdef __cplusplus
}
#endif

/* ---- myrdp::FeatureScan (LSD/myRDP.cpp:9-185 with RegionSegmentation :274-345, SplitMerge :187-217, SplitMergeAssistant
 * :219-272, getThresholdDeltaDist :347-368): clusters one lidar scan, splits the clusters by Ramer-Douglas-Peucker and turns the
 * chords of at least lineDistThreM metres into line records + the pixels of their raster.  SURVEY 8f #4.
 * PARITY: pinned only loosely -- data/ScanlinesInfo.txt holds the MATLAB prototype's 11 lines for one frame of data/Lidar.txt
 * (tests/test_oracle.py::test_feature_scan_against_the_matlab_golden); the reference cannot be built here.
 * The reference reads one element past its point array when the last reading closes a cluster (:318-322); the value is never
 * used, so the restatement does not read it.  A vertical chord makes its slope infinite (:245): IEEE arithmetic then gives NaN
 * distances and no split, which is restated as it is. */
typedef struct { double range, angle; } orc_polar;                 /* structLidarPointPolar (myRDP.h:34-38) without its work flag */
typedef struct { int oriMapCol, oriMapRow; double mapResol, mapOriX, mapOriY; } orc_map_param;   /* structMapParam, baseFunc.h:25-31 */
/* lines_out[360], pts_out[pts_cap] (x, y, 0): returns 0 and sets *n_lines, *n_pts (the number of raster pixels; only the first
 * pts_cap are stored), lidar_pos[2], im_size[2] = (oriXLim, oriYLim). */
int orc_feature_scan(orc_map_param mp, const orc_polar *scan, int len_lp, int region_point_limit, double thre_line,
                     double line_dist_thre_m, orc_line *lines_out, int *n_lines, orc_position *pts_out, int pts_cap, int *n_pts,
                     double *lidar_pos, int *im_size);

#endif
