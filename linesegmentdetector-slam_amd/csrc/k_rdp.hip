// k_rdp.hip -- myrdp::FeatureScan for a BATCH of lidar scans (gfx950): one wavefront per scan.
//
// Replaces LSD/myRDP.cpp:9-185 (FeatureScan) with its callees RegionSegmentation (:274-345), SplitMerge (:187-217),
// SplitMergeAssistant (:219-272) and getThresholdDeltaDist (:347-368), SURVEY 8f #4.  The reference handles one scan of at most
// 360 readings per frame on the host (LSD/main_on_windows.cpp:127); this is the batched form: N scans of len[i] readings each,
// results at fixed strides.  Per scan:
//   1. metric coordinates of the readings (lanes), the gaps between neighbours against the range-dependent threshold (lanes),
//      then the clusters -- a serial walk over the gaps, as in the reference, by one lane (a few hundred steps);
//   2. Ramer-Douglas-Peucker per cluster with an explicit stack: the point of a chord's span farthest from it is found by the
//      64 lanes (first maximum in scan order, like the reference's strict '>'), the spans to split further go on the stack;
//   3. pixel coordinates, their extent (wave min / max), the chords of every cluster in order -> the ones long enough become
//      line records (one lane per line; atand / cosd / sind are the correctly rounded ones of the LSD path), and the pixels of
//      their rasters are listed in the reference's order (line by line; a prefix sum over the lines' pixel counts).
// IEEE corner cases are kept as they are: a vertical chord has an infinite slope, its distances are NaN and nothing is split
// (:245-257); (int) casts of non-finite values follow x86 (cvt_x86).  The reference's read one past its point array (:318-322)
// is of a value it never uses and is not made.
#include "lsd_internal.h"
#include "devmath.h"

namespace lsdhip {

constexpr int kRdpMaxLen = 1024;                    // readings per scan this kernel takes (the reference: 360)
constexpr int kRdpMaxLines = 360;                   // line records per scan (the reference's malloc, :39)

__device__ __forceinline__ double rdp_thre_delta(double val) {                      // getThresholdDeltaDist :347-368
    if (val <= 0.3) return 0.02;
    if (val <= 0.5) return 0.05;
    if (val <= 0.8) return 0.11;
    if (val <= 1) return 0.17;
    if (val <= 2) return 0.6;
    if (val <= 3) return 0.7;
    if (val <= 4) return 0.85;
    if (val <= 5) return 0.9;
    if (val <= 6) return 1;
    return 1.1;
}

__global__ __launch_bounds__(64) void k_rdp(const double* __restrict__ scans /* n x stride x {range, angle} */, const int* __restrict__ lens,
                                            int stride, int oriMapCol, int oriMapRow, double mapResol, double mapOriX, double mapOriY,
                                            int region_point_limit, double thre_line, double line_dist_thre_m,
                                            lsd_line* __restrict__ lines_out, int* __restrict__ n_lines, double* __restrict__ pts_out, int pts_cap,
                                            int* __restrict__ n_pts, double* __restrict__ lidar_pos, int* __restrict__ im_size) {
    __shared__ double px[kRdpMaxLen], py[kRdpMaxLen];
    __shared__ unsigned char brk[kRdpMaxLen], split[kRdpMaxLen];
    __shared__ short cs[kRdpMaxLen], ce[kRdpMaxLen];           // clusters: first and last reading
    __shared__ short stk[2 * kRdpMaxLen];                       // spans still to look at (RDP), later: the chords (a, b) in order
    __shared__ int s_cells, s_nch;
    (void)oriMapCol; (void)oriMapRow;                          // (carried by structMapParam, unused by FeatureScan)
    const size_t scan = blockIdx.x;
    const int lane = threadIdx.x;
    const int len_lp = min(lens[scan], min(stride, kRdpMaxLen));
    const double* sc = scans + scan * (size_t)stride * 2;
    lsd_line* lout = lines_out + scan * (size_t)kRdpMaxLines;
    double* pout = pts_out + scan * (size_t)pts_cap * 3;
    if (len_lp < 1) {
        if (lane == 0) { n_lines[scan] = 0; n_pts[scan] = 0; lidar_pos[scan * 2] = 0; lidar_pos[scan * 2 + 1] = 0; im_size[scan * 2] = 0; im_size[scan * 2 + 1] = 0; }
        return;
    }
    // 1. metric coordinates (scanPose = 0, :11) and the gaps
    for (int i = lane; i < len_lp; i += 64) {
        double s, c;
        sincos_g(sc[2 * i + 1] + 0.0, s, c);
        px[i] = sc[2 * i] * c + 0.0;                                               // :286-287
        py[i] = sc[2 * i] * s + 0.0;
        split[i] = 0;
    }
    __syncthreads();
    for (int i = lane; i < len_lp; i += 64) {
        const int nx = i == len_lp - 1 ? 0 : i + 1;                                // :299-306
        const double dX = px[i] - px[nx], dY = py[i] - py[nx];
        brk[i] = sqrt(dX * dX + dY * dY) > rdp_thre_delta(sc[2 * i]) ? 1 : 0;      // :307-309
    }
    __syncthreads();
    if (lane == 0) {                                                               // RegionSegmentation's walk :297-330
        int cellNumber = 0, startNum = 0;
        for (int i = 0; i < len_lp; i++) {
            if (brk[i]) {
                cs[cellNumber] = (short)startNum; ce[cellNumber] = (short)i;
                if (abs(i - startNum) >= region_point_limit) cellNumber++;
                startNum = i + 1;
            }
            if (!brk[i] && i == len_lp - 1) cs[0] = (short)startNum;               // the last cluster joins the first
        }
        s_cells = cellNumber;
    }
    __syncthreads();
    const int cells = s_cells;
    // 2. SplitMerge :187-217 / SplitMergeAssistant :219-272
    for (int cidx = 0; cidx < cells; cidx++) {
        int sp_top = 0;
        if (lane == 0) { stk[0] = cs[cidx]; stk[1] = ce[cidx]; }
        sp_top = 1;
        __syncthreads();
        while (sp_top > 0) {
            sp_top--;
            const int sp = stk[2 * sp_top], ep = stk[2 * sp_top + 1];
            __syncthreads();
            const int len = ep > sp ? ep - sp + 1 : len_lp + ep - sp + 1;          // :223-239
            if (len <= 2) continue;
            const double k = (py[ep] - py[sp]) / (px[ep] - px[sp]);                // :245-246
            const double d = py[ep] - k * px[ep];
            const double den = sqrt(k * k + 1);
            double best = 0.0;                                                     // dist_max = 0: only a distance > 0 is taken (:257)
            int besto = 0x7fffffff;                                                // its position in the span (first maximum wins)
            for (int i = 1 + lane; i < len - 1; i += 64) {
                int a = sp + i;
                if (a >= len_lp) a -= len_lp;
                const double dist = fabs(k * px[a] - py[a] + d) / den;             // :256
                if (dist > best) { best = dist; besto = i; }                       // (ascending i per lane: the first of equal ones stays)
            }
            for (int off = 32; off >= 1; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oo = __shfl_xor(besto, off);
                if (ob > best || (ob == best && oo < besto)) { best = ob; besto = oo; }
            }
            int i_max = 0;                                                         // :252 (reading 0 when nothing was farther than 0)
            if (besto != 0x7fffffff) { i_max = sp + besto; if (i_max >= len_lp) i_max -= len_lp; }
            const double r = sc[2 * i_max];
            const double threDist = r > 9 ? r * thre_line : thre_line;             // :259-263
            if (best > threDist) {
                if (lane == 0) {
                    stk[2 * sp_top] = (short)sp; stk[2 * sp_top + 1] = (short)i_max;
                    stk[2 * sp_top + 2] = (short)i_max; stk[2 * sp_top + 3] = (short)ep;
                    split[i_max] = 1;
                }
                sp_top += 2;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // 3. pixel coordinates and the size of the image :16-37
    double minX = INFINITY, minY = INFINITY, maxX = 0, maxY = 0;
    for (int i = lane; i < len_lp; i += 64) {
        const double X = floor((px[i] - mapOriX) / mapResol), Y = floor((py[i] - mapOriY) / mapResol);
        px[i] = X; py[i] = Y;
        minX = fmin(minX, X); maxX = fmax(maxX, X); minY = fmin(minY, Y); maxY = fmax(maxY, Y);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        minX = fmin(minX, __shfl_xor(minX, off)); maxX = fmax(maxX, __shfl_xor(maxX, off));
        minY = fmin(minY, __shfl_xor(minY, off)); maxY = fmax(maxY, __shfl_xor(maxY, off));
    }
    const int oriXLim = cvt_x86(ceil(maxX - minX)), oriYLim = cvt_x86(ceil(maxY - minY));
    if (lane == 0) {
        lidar_pos[scan * 2] = floor((0.0 - mapOriX) / mapResol - minX);            // :35-36
        lidar_pos[scan * 2 + 1] = floor((0.0 - mapOriY) / mapResol - minY);
        im_size[scan * 2] = oriXLim; im_size[scan * 2 + 1] = oriYLim;
    }
    __syncthreads();
    // the chords of every cluster, in order (:45-71): its first reading, its split points, its last reading
    if (lane == 0) {
        int nch = 0;
        for (int cidx = 0; cidx < cells; cidx++) {
            const int sp = cs[cidx], ep = ce[cidx];
            const int len_axis = ep > sp ? ep - sp + 1 : len_lp + ep - sp + 1;
            int prev = sp;
            for (int j = 0; j < len_axis; j++) {
                int v = sp + j;
                if (v >= len_lp) v -= len_lp;
                if (split[v] && nch < kRdpMaxLen - 1) { stk[2 * nch] = (short)prev; stk[2 * nch + 1] = (short)v; nch++; prev = v; }
            }
            if (nch < kRdpMaxLen) { stk[2 * nch] = (short)prev; stk[2 * nch + 1] = (short)ep; nch++; }
        }
        s_nch = nch;
    }
    __syncthreads();
    // (the reference puts the first reading in front of the collected split points and the last one behind, :68-69: a flag on one
    //  of the two -- flags sit strictly inside a span, so there is none -- would give a chord of length 0 here as there)
    const int nch = s_nch;
    const double lineDistThre = line_dist_thre_m / mapResol;
    int nl = 0, np = 0;                                                            // lines / pixels so far (wave-uniform)
    for (int base = 0; base < nch; base += 64) {
        const int ci = base + lane;
        bool keep = false;
        double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        if (ci < nch) {
            const int a = stk[2 * ci], b = stk[2 * ci + 1];
            const double ax = px[a], ay = py[a], bx = px[b], by = py[b];
            const double ex = ax - bx, ey = ay - by;
            keep = sqrt(ex * ex + ey * ey) >= lineDistThre;                        // :78-79
            x1 = ax - minX; y1 = ay - minY; x2 = bx - minX; y2 = by - minY;        // :81-84
        }
        const unsigned long long km = __ballot(keep);
        const int li = nl + __builtin_popcountll(km & ((1ull << lane) - 1ull));
        const double k = (y2 - y1) / (x2 - x1);                                    // :86
        const int xLow = cvt_x86(floor(x1 > x2 ? x2 : x1)), xHigh = cvt_x86(ceil(x1 > x2 ? x1 : x2));   // :94-109
        const int yLow = cvt_x86(floor(y1 > y2 ? y2 : y1)), yHigh = cvt_x86(ceil(y1 > y2 ? y1 : y2));
        const bool along_x = fabs(x2 - x1) > fabs(y2 - y1);                        // :110-115 (integer coordinates: the same as xx_len > yy_len)
        const int cnt = keep ? (along_x ? xHigh - xLow + 1 : yHigh - yLow + 1) : 0;
        // the pixels this line marks (:116-153): count, then place behind the earlier lines' pixels
        int mine = 0;
        for (int m = 0; m < cnt; m++) {
            int xx, yy;
            if (along_x) { xx = m + xLow; yy = cvt_x86(round((xx - x1) * k + y1)); }
            else { yy = m + yLow; xx = cvt_x86(round((yy - y1) / k + x1)); }
            if (!(xx < 0 || xx >= oriXLim || yy < 0 || yy >= oriYLim) && xx != 0 && yy != 0) mine++;   // 0 doubles as "invalid"
        }
        int inc = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        int wr = np + inc - mine;
        for (int m = 0; m < cnt; m++) {
            int xx, yy;
            if (along_x) { xx = m + xLow; yy = cvt_x86(round((xx - x1) * k + y1)); }
            else { yy = m + yLow; xx = cvt_x86(round((yy - y1) / k + x1)); }
            if (!(xx < 0 || xx >= oriXLim || yy < 0 || yy >= oriYLim) && xx != 0 && yy != 0) {
                if (wr < pts_cap) { pout[3 * (size_t)wr] = xx; pout[3 * (size_t)wr + 1] = yy; pout[3 * (size_t)wr + 2] = 0; }
                wr++;
            }
        }
        if (keep && li < kRdpMaxLines) {
            double ang = atan_g(k) * 180.0 / kPi;                                  // atand, baseFunc.cpp:14-16
            int orient = 1;
            if (ang < 0) { ang += 180; orient = -1; }                              // :89-92
            lsd_line L;
            L.k = k;
            L.b = (y1 + y2) / 2.0 - k * (x1 + x2) / 2.0;                           // :164
            sincos_g(ang / 180.0 * kPi, L.dy, L.dx);                               // sind / cosd
            L.x1 = x1; L.y1 = y1; L.x2 = x2; L.y2 = y2;
            const double ey = y2 - y1, ex = x2 - x1;
            L.len = sqrt(ey * ey + ex * ex);                                       // :171
            L.orient = orient;
            lout[li] = L;
            reinterpret_cast<uint32_t*>(&lout[li])[19] = 0u;                       // the tail padding: defined bytes
        }
        nl += __builtin_popcountll(km);
        np += __shfl(inc, 63);
    }
    if (lane == 0) { n_lines[scan] = nl; n_pts[scan] = np; }
}

void launch_rdp(const double* scans, const int* lens, int n, int stride, int oriMapCol, int oriMapRow, double mapResol, double mapOriX,
                double mapOriY, int region_point_limit, double thre_line, double line_dist_thre_m, lsd_line* lines_out, int* n_lines,
                double* pts_out, int pts_cap, int* n_pts, double* lidar_pos, int* im_size, hipStream_t s) {
    hipLaunchKernelGGL(k_rdp, dim3(n), dim3(64), 0, s, scans, lens, stride, oriMapCol, oriMapRow, mapResol, mapOriX, mapOriY,
                       region_point_limit, thre_line, line_dist_thre_m, lines_out, n_lines, pts_out, pts_cap, n_pts, lidar_pos, im_size);
}
int rdp_max_len() { return kRdpMaxLen; }
int rdp_max_lines() { return kRdpMaxLines; }

}  // namespace lsdhip
