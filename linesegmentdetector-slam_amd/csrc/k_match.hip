// k_match.hip -- the scan-to-map matching batch of FeatureAssociation on the device (SURVEY 8f "next" #2), gfx950.
//
// Replaces the body of myfa::thread_ScanToMapMatch (LSD/myFA.cpp:197-270) and its callees NormalizedLineDirection
// (:272-305), rotateScanIm (:307-357) and CalcScore (:359-396), which the reference runs on a 30-thread pool: for every
// (map line, scan line) pair, the four start/end matchings each give a candidate pose; the scan's image points are
// rotated to it and scored against mapCache.
// One lane per candidate: the sums of CalcScore are accumulated in point order (as the reference does), the scan
// points are read by all lanes at the same address (broadcast), mapCache is gathered.  sin/cos/atan are the correctly
// rounded ones of crmath.h.  Bound: L2/HBM gather latency; ~30 flops per point.
#include "lsd_internal.h"
#include "devmath.h"

namespace lsdhip {

__device__ __forceinline__ double deg2rad_ref(double x) { return x / 180.0 * kPi; }     // baseFunc.cpp:6-12 (pi = 4*atan(1))

__device__ __forceinline__ double line_direction(double staX, double staY, double endX, double endY) {   // myFA.cpp:272-305
    double angle;
    if (staX == endX && staY != endY) angle = staY < endY ? 90 : -90;
    else if (staX != endX && staY == endY) angle = staX < endX ? 0 : 180;
    else angle = atan_g((endY - staY) / (endX - staX)) * 180.0 / kPi;                    // atand, baseFunc.cpp:14-16
    if (angle < 0 && staX > endX) return angle + 180;
    if (angle > 0 && staX > endX) return angle - 180;
    return angle;
}

__global__ __launch_bounds__(64) void k_match(const double* __restrict__ map_cache, int cols, int rows,
                                              const lsd_line* __restrict__ map_lines, const lsd_line* __restrict__ scan_lines,
                                              const double* __restrict__ pts /* n x {x, y, ang} */, int n_points,
                                              double lidx, double lidy, double lastx, double lasty,
                                              const int* __restrict__ pairs, int n_cand, double zmax, double max_esti_dist,
                                              double* __restrict__ out /* n_cand x {x, y, ang, score} */) {
    const int cidx = blockIdx.x * 64 + threadIdx.x;
    const bool act = cidx < n_cand;
    const int p = act ? cidx >> 2 : 0, i = (cidx & 3) + 1;                   // :205-249
    const lsd_line ml = map_lines[pairs[2 * p]], sl = scan_lines[pairs[2 * p + 1]];
    const bool mrev = i >= 3, srev = (i == 2 || i == 4);
    const double msx = mrev ? ml.x2 : ml.x1, msy = mrev ? ml.y2 : ml.y1, mex = mrev ? ml.x1 : ml.x2, mey = mrev ? ml.y1 : ml.y2;
    const double ssx = srev ? sl.x2 : sl.x1, ssy = srev ? sl.y2 : sl.y1, sex = srev ? sl.x1 : sl.x2, sey = srev ? sl.y1 : sl.y2;
    double angDiff = line_direction(msx, msy, mex, mey) - line_direction(ssx, ssy, sex, sey);   // :252-258, :310
    double sd, cd;
    sincos_g(deg2rad_ref(angDiff), sd, cd);                                   // sind / cosd
    const double rlx = (lidx - ssx) * cd - (lidy - ssy) * sd + msx;           // :323-324
    const double rly = (lidx - ssx) * sd + (lidy - ssy) * cd + msy;
    const double ddx = rlx - lastx, ddy = rly - lasty;
    const bool near_ = sqrt(ddx * ddx + ddy * ddy) < max_esti_dist || lastx == -1;   // :330
    double sumValidDist = 0, sumMaxDist = 0, numValidPoint = 0;               // CalcScore
    if (__ballot(act && near_)) {
        for (int c = 0; c < n_points; c++) {
            const double px = pts[3 * c], py = pts[3 * c + 1];                // same address in every lane
            const double ox = px - ssx, oy = py - ssy;                        // :317-320
            const double rx = ox * cd - oy * sd + msx;                        // :333-336
            const double ry = ox * sd + oy * cd + msy;
            const int x = cvt_x86(round(rx)), y = cvt_x86(round(ry));         // :368-369
            if (act && near_ && y >= 0 && y < rows && x >= 0 && x < cols) {
                numValidPoint += 1;
                const double v = map_cache[(size_t)y * cols + x];
                if (v >= zmax) sumMaxDist += 10;                              // :374-378
                else sumValidDist += v;
            }
        }
    }
    if (!act) return;
    double ang = 0, score = HUGE_VAL;
    if (near_) {
        while (angDiff <= -180) angDiff += 360;                               // :339-342
        while (angDiff > 180) angDiff -= 360;
        ang = angDiff;
        const double numAllPoint = n_points;
        if (n_points != 0 && !(numValidPoint < 0.7 * numAllPoint))            // :248-263 (no scan points: the score stays infinite), :388-392
            score = (sumValidDist + sumMaxDist) / (numValidPoint) + 10 * (numAllPoint - numValidPoint) / numAllPoint;
    }
    double* o = out + (size_t)cidx * 4;
    o[0] = rlx; o[1] = rly; o[2] = ang; o[3] = score;
}

void launch_match(const double* map_cache, int cols, int rows, const lsd_line* map_lines, const lsd_line* scan_lines,
                  const double* pts, int n_points, double lidx, double lidy, double lastx, double lasty, const int* pairs,
                  int n_pairs, double zmax, double max_esti_dist, double* out, hipStream_t s) {
    const int n_cand = 4 * n_pairs;
    hipLaunchKernelGGL(k_match, dim3((n_cand + 63) / 64), dim3(64), 0, s, map_cache, cols, rows, map_lines, scan_lines, pts,
                       n_points, lidx, lidy, lastx, lasty, pairs, n_cand, zmax, max_esti_dist, out);
}

}  // namespace lsdhip
