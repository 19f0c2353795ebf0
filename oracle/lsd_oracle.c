/*
 * lsd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See lsd_oracle.h.
 *
 * Plain-C restatement of the reference's LSD hot path.  It is organised around flat arrays
 * and index lists (no linked lists, no per-call image allocation) but keeps the reference's
 * floating-point operation ORDER and every behaviour listed in SURVEY.md section 8a-Q, each
 * marked with the reference file:line it follows.  "myLSD.cpp" = /root/reference/LSD/myLSD.cpp.
 *
 * Build flags that matter (oracle/Makefile): -O2 -ffp-contract=off, no -ffast-math, x86-64
 * baseline (no FMA), glibc libm -- the reference's floating-point environment (Q12).
 */
#include "lsd_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static const double PI = 3.14159265358979323846; /* == 4.0*atan(1.0), myLSD.cpp:9 */

/* x86-64 cvttsd2si semantics of the reference's (int) casts: NaN, +-inf and out-of-range
 * values become INT_MIN ("integer indefinite") -- SURVEY 8a-Q8. */
static int cvt_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

/* ------------------------------------------------------------------------------------ */
/* a2: GaussianSampler, myLSD.cpp:378-484                                                */
/* ------------------------------------------------------------------------------------ */
static int reflect_idx(int j, int lim)
{
    int dou = 2 * lim;                        /* :395-396 */
    while (j < 0) j += dou;                   /* :436-438 */
    while (j >= dou) j -= dou;                /* :439-441 */
    if (j >= lim) j = dou - j - 1;            /* :442-443 */
    return j;
}

/* taps[3][hSize]; returns h.  Kernel values :398-417 */
static int gauss_taps(double sca, double sig, double **taps_out)
{
    int prec = 3;
    if (sca < 1) sig = sig / sca;                                 /* :390-391 */
    int h = cvt_int(ceil(sig * sqrt(2 * prec * log(10))));        /* :393 */
    int hSize = 1 + 2 * h;
    double *t = (double *)malloc(sizeof(double) * 3 * hSize);
    double s1 = 0, s2 = 0, s3 = 0;
    for (int k = 0; k < hSize; k++) {
        double a = (k - h) / sig;
        double b = (k - h - 1.0 / 3) / sig;
        double c = (k - h + 1.0 / 3) / sig;
        t[0 * hSize + k] = exp(-0.5 * (a * a));                   /* pow(.,2) == x*x (Q12) */
        t[1 * hSize + k] = exp(-0.5 * (b * b));
        t[2 * hSize + k] = exp(-0.5 * (c * c));
        s1 += t[0 * hSize + k];
        s2 += t[1 * hSize + k];
        s3 += t[2 * hSize + k];
    }
    for (int k = 0; k < hSize; k++) {                             /* :413-417 */
        t[0 * hSize + k] /= s1;
        t[1 * hSize + k] /= s2;
        t[2 * hSize + k] /= s3;
    }
    *taps_out = t;
    return h;
}

static double *gaussian_sampler(const uint8_t *img, int xLim, int yLim, size_t stride,
                                double sca, double sig, int newX, int newY)
{
    double *taps;
    int h = gauss_taps(sca, sig, &taps);
    int hSize = 1 + 2 * h;
    double *aux = (double *)calloc((size_t)yLim * newX + 1, sizeof(double));
    double *out = (double *)calloc((size_t)newY * newX + 1, sizeof(double));
    for (int x = 0; x < newX; x++) {                              /* :420-448 */
        const double *ker = taps + (x % 3) * hSize;              /* :422-427 */
        int xc = cvt_int(floor(x / sca + 0.5));                   /* :428 */
        for (int y = 0; y < yLim; y++) {
            double v = 0;
            for (int i = 0; i < hSize; i++) {
                int j = reflect_idx(xc - h + i, xLim);
                v += img[(size_t)y * stride + j] * ker[i];        /* :444 */
            }
            aux[(size_t)y * newX + x] = v;
        }
    }
    for (int y = 0; y < newY; y++) {                              /* :452-482 */
        const double *ker = taps + (y % 3) * hSize;
        int yc = cvt_int(floor(y / sca + 0.5));                   /* :460 */
        for (int x = 0; x < newX; x++) {
            double v = 0;
            for (int i = 0; i < hSize; i++) {
                int j = reflect_idx(yc - h + i, yLim);
                v += aux[(size_t)j * newX + x] * ker[i];          /* :478 */
            }
            out[(size_t)y * newX + x] = v;
        }
    }
    free(aux);
    free(taps);
    return out;
}

/* ------------------------------------------------------------------------------------ */
/* working state of one image                                                            */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int w, h;
    const double *mag, *deg;
    uint8_t *used;     /* usedMap / banMap */
    int *cur;          /* curMap as stamps: curMap(p)==1  <=>  cur[p]==cur_id */
    int cur_id;
    /* region of the most recent grow */
    int *rx, *ry;      /* working list (regPts_x / regPts_y); capacity w*h+1 */
    int *gx, *gy;      /* grow-order copy (never reordered) for the marking loops */
    int gnum;
    double logNT;
    int nfa_host_only; /* (debug counters) the last NFA value was -logNT [- n log10 p]: no exp / log10 / pow of the tail behind it */
    orc_debug *dbg;
} orc_state;

typedef struct { int x, y, num; double deg; } orc_reg;              /* structReg, myLSD.h:65-73 */
typedef struct { double x1, y1, x2, y2, wid, cX, cY, deg, dx, dy, p, prec; } orc_rec; /* structRec myLSD.h:80-93 */

/* a8: RegionGrower, myLSD.cpp:491-590.  Fills st->rx/ry (and the gx/gy copy). */
static orc_reg region_grower(orc_state *st, int x, int y, double regDeg, double degThre)
{
    const int xLim = st->w, yLim = st->h;
    const int id = ++st->cur_id;              /* fresh curMap (:519) */
    int *rx = st->rx, *ry = st->ry;
    rx[0] = x; ry[0] = y;
    double sinDeg = sin(regDeg);              /* :515 */
    double cosDeg = cos(regDeg);              /* :516 */
    st->cur[(size_t)y * xLim + x] = id;       /* :520 */
    int growNum = 1, exNum = 0;
    while (exNum != growNum) {                /* :525 sweeps to fixpoint (Q7) */
        exNum = growNum;
        for (int i = 0; i < growNum; i++) {   /* growNum is live (:529) */
            int roi_x = rx[i], roi_y = ry[i];
            for (int m = roi_y - 1; m <= roi_y + 1; m++) {
                for (int n = roi_x - 1; n <= roi_x + 1; n++) {
                    if (m >= 0 && n >= 0 && m < yLim && n < xLim) {
                        size_t p = (size_t)m * xLim + n;
                        if (st->cur[p] != id && st->used[p] != 1) {     /* :537 (2 is growable, Q5) */
                            double curDeg = st->deg[p];
                            double degDif = fabs(regDeg - curDeg);      /* :540 */
                            if (degDif > PI * 3 / 2.0)                  /* :541 */
                                degDif = fabs(degDif - 2.0 * PI);
                            if (degDif < degThre) {                     /* :543 */
                                cosDeg += cos(curDeg);                  /* :545 */
                                sinDeg += sin(curDeg);                  /* :546 */
                                regDeg = atan2(sinDeg, cosDeg);         /* :547 */
                                st->cur[p] = id;
                                rx[growNum] = n;
                                ry[growNum] = m;
                                growNum++;
                            }
                        }
                    }
                }
            }
        }
    }
    memcpy(st->gx, rx, sizeof(int) * growNum);
    memcpy(st->gy, ry, sizeof(int) * growNum);
    st->gnum = growNum;
    if (st->dbg) { st->dbg->grow_calls++; st->dbg->grown_px += growNum; }
    orc_reg reg = { x, y, growNum, regDeg };
    return reg;
}

/* a9: CenterGetter :592-619, OrientationGetter :621-667, RectangleConverter :669-734 */
static orc_rec rectangle_converter(const orc_state *st, const orc_reg *reg, double aliPro, double degThre)
{
    const int *rx = st->rx, *ry = st->ry;
    const int W = st->w;
    double cenX = 0, cenY = 0, weiSum = 0;
    for (int k = 0; k < reg->num; k++) {                         /* :608-613 */
        double pixWei = st->mag[(size_t)ry[k] * W + rx[k]];
        cenX += pixWei * rx[k];
        cenY += pixWei * ry[k];
        weiSum += pixWei;
    }
    cenX = cenX / weiSum;
    cenY = cenY / weiSum;

    double Ixx = 0, Iyy = 0, Ixy = 0;
    weiSum = 0;
    for (int k = 0; k < reg->num; k++) {                         /* :637-643 */
        double pixWei = st->mag[(size_t)ry[k] * W + rx[k]];
        double ddy = ry[k] - cenY, ddx = rx[k] - cenX;
        Ixx += pixWei * (ddy * ddy);
        Iyy += pixWei * (ddx * ddx);
        Ixy -= pixWei * ddx * ddy;
        weiSum += pixWei;
    }
    Ixx /= weiSum; Iyy /= weiSum; Ixy /= weiSum;
    double dI = Ixx - Iyy;
    double lamb = (Ixx + Iyy - sqrt(dI * dI + 4 * Ixy * Ixy)) / 2.0;   /* :647 */
    double inertiaDeg;
    if (fabs(Ixx) > fabs(Iyy)) inertiaDeg = atan2(lamb - Ixx, Ixy);    /* :649-652 */
    else inertiaDeg = atan2(Ixy, lamb - Iyy);
    double regDif = inertiaDeg - reg->deg;                             /* :655-665 */
    while (regDif <= -PI) regDif += 2 * PI;
    while (regDif > PI) regDif -= 2 * PI;
    if (regDif < 0) regDif = -regDif;
    if (regDif > degThre) inertiaDeg += PI;

    double dx = cos(inertiaDeg), dy = sin(inertiaDeg);                 /* :699-700 */
    double lenMin = 0, lenMax = 0, widMin = 0, widMax = 0;             /* Q9: initialised to 0 (:701) */
    for (int m = 0; m < reg->num; m++) {
        double len = (rx[m] - cenX) * dx + (ry[m] - cenY) * dy;       /* :704 */
        double wid = -(rx[m] - cenX) * dy + (ry[m] - cenY) * dx;      /* :705 */
        if (len < lenMin) lenMin = len;
        if (len > lenMax) lenMax = len;
        if (wid < widMin) widMin = wid;
        if (wid > widMax) widMax = wid;
    }
    orc_rec rec;
    rec.x1 = cenX + lenMin * dx; rec.y1 = cenY + lenMin * dy;          /* :717-720 */
    rec.x2 = cenX + lenMax * dx; rec.y2 = cenY + lenMax * dy;
    rec.wid = widMax - widMin;
    rec.cX = cenX; rec.cY = cenY; rec.deg = inertiaDeg; rec.dx = dx; rec.dy = dy;
    rec.p = aliPro; rec.prec = degThre;
    if (rec.wid < 1) rec.wid = 1;                                      /* :730 */
    return rec;
}

static double rec_density(const orc_reg *reg, const orc_rec *rec)
{
    double ex = rec->x1 - rec->x2, ey = rec->y1 - rec->y2;
    return reg->num / (sqrt(ex * ex + ey * ey) * rec->wid);            /* :757, :798, :827, :867 */
}

/* a11: RegionRadiusReducer, myLSD.cpp:736-802 */
static int region_radius_reducer(orc_state *st, orc_reg *reg, orc_rec *rec, double denThre)
{
    int *rx = st->rx, *ry = st->ry;
    const int W = st->w;
    if (st->dbg) st->dbg->rrr_calls++;
    double den = rec_density(reg, rec);
    if (den > denThre) return 1;                                       /* :760 */
    int oriX = reg->x, oriY = reg->y;
    double ax = oriX - rec->x1, ay = oriY - rec->y1;
    double bx = oriX - rec->x2, by = oriY - rec->y2;
    double rad1 = sqrt(ax * ax + ay * ay);                             /* :768 */
    double rad2 = sqrt(bx * bx + by * by);
    double rad = rad1 > rad2 ? rad1 : rad2;
    int removed_any = 0;   /* true once slot [num] is known to hold the (0,0) "NULL" written at :784-785 */
    while (den < denThre) {                                            /* :775 */
        rad *= 0.75;
        if (st->dbg) st->dbg->rrr_passes++;
        int i = 0;
        while (i <= reg->num) {                                        /* :779 -- `<=`, SURVEY 8a-Q6 */
            int px, py;
            if (i == reg->num) {
                if (!removed_any) {
                    /* out-of-bounds read in the reference (UB); restated as "no extra removal" */
                    if (st->dbg) st->dbg->rrr_oob_reads++;
                    break;
                }
                px = 0; py = 0;                                        /* slot holds NULL == 0 */
            } else { px = rx[i]; py = ry[i]; }
            double ddx = oriX - px, ddy = oriY - py;
            if (sqrt(ddx * ddx + ddy * ddy) > rad) {                   /* :780 */
                st->cur[(size_t)py * W + px] = 0;                      /* :781 */
                if (i == reg->num) {
                    /* sentinel slot "removed": the last valid point leaves the list, its curMap bit stays */
                    if (st->dbg) st->dbg->rrr_sentinel_drops++;
                    /* rx[num] = rx[num-1] (out of range afterwards); rx[num-1] = 0 */
                    rx[reg->num - 1] = 0; ry[reg->num - 1] = 0;
                } else {
                    rx[i] = rx[reg->num - 1]; ry[i] = ry[reg->num - 1]; /* :782-783 */
                    rx[reg->num - 1] = 0; ry[reg->num - 1] = 0;         /* :784-785 */
                }
                removed_any = 1;
                i--;
                reg->num--;
            }
            i++;
        }
        if (reg->num < 2) return 0;                                    /* :792 */
        *rec = rectangle_converter(st, reg, rec->p, rec->prec);        /* :797 */
        den = rec_density(reg, rec);
    }
    return 1;
}

/* a10: Refiner, myLSD.cpp:804-880 */
static int refiner(orc_state *st, orc_reg *reg, orc_rec *rec, double denThre)
{
    const int W = st->w;
    double den = rec_density(reg, rec);
    if (den >= denThre) return 1;                                      /* :829 */
    int oriX = reg->x, oriY = reg->y;
    double cenDeg = st->deg[(size_t)oriY * W + oriX];
    double difSum = 0, squSum = 0;
    int ptNum = 0;
    for (int i = 0; i < reg->num; i++) {                               /* :839-853 */
        double ddx = oriX - st->rx[i], ddy = oriY - st->ry[i];
        if (sqrt(ddx * ddx + ddy * ddy) < rec->wid) {
            double curDeg = st->deg[(size_t)st->ry[i] * W + st->rx[i]];
            double degDif = curDeg - cenDeg;
            while (degDif <= -PI) degDif += 2 * PI;
            while (degDif > PI) degDif -= 2 * PI;
            difSum += degDif;
            squSum += degDif * degDif;
            ptNum++;
        }
    }
    double meanDif = difSum / (ptNum * 1.0);
    double degThre = 2.0 * sqrt((squSum - 2 * meanDif * difSum) / (ptNum * 1.0) + meanDif * meanDif); /* :855 */
    *reg = region_grower(st, oriX, oriY, cenDeg, degThre);             /* :857 */
    if (reg->num < 2) return 0;                                        /* :861 */
    *rec = rectangle_converter(st, reg, rec->p, rec->prec);            /* :866 */
    den = rec_density(reg, rec);
    if (den < denThre)                                                 /* :869 */
        return region_radius_reducer(st, reg, rec, denThre);
    return 1;
}

/* a13: LogGammaCalculator, myLSD.cpp:882-924 */
double orc_log_gamma(int x)
{
    if (x > 15) {
        return 0.918938533204673 + (x - 0.5) * log(x) - x +
               0.5 * x * log(x * sinh(1.0 / x) + 1.0 / (810 * pow(x, 6)));   /* :908-909 */
    }
    static const double q[7] = { 75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                                 1168.92649479, 83.8676043424, 2.50662827511 };
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int i = 0; i < 7; i++) {
        a -= log(x + i);
        b += q[i] * pow(x, i);
    }
    return a + log(b);
}

/* The libm calls of RectangleNFACalculator that the HIP path evaluates ON THE DEVICE (everything else on this page the host
 * computes with its libm and hands over as tables): plain libm here; the diagnostic build liblsd_oracle_cr.so (cr_shim.h)
 * replaces them by correctly rounded ones. */
#ifndef NFA_EXP
#define NFA_EXP exp
#define NFA_LOG10 log10
#define NFA_POW pow
#endif

/* a12: RectangleNFACalculator, myLSD.cpp:926-1059.
 * The full-image "deg > pi -> -= pi" pass (:940-945) is a no-op (atan2 <= pi and values within
 * 1e-6 of pi were zeroed at :170-171) and is not restated. */
static double rectangle_nfa(orc_state *st, const orc_rec *rec)
{
    const int xLim = st->w, yLim = st->h;
    const double logNT = st->logNT;
    if (st->dbg) st->dbg->nfa_calls++;
    st->nfa_host_only = 1;
    double verX[4], verY[4], vx[4], vy[4];
    verX[0] = rec->x1 - rec->dy * rec->wid / 2.0;                      /* :949-956 */
    verX[1] = rec->x2 - rec->dy * rec->wid / 2.0;
    verX[2] = rec->x2 + rec->dy * rec->wid / 2.0;
    verX[3] = rec->x1 + rec->dy * rec->wid / 2.0;
    verY[0] = rec->y1 + rec->dx * rec->wid / 2.0;
    verY[1] = rec->y2 + rec->dx * rec->wid / 2.0;
    verY[2] = rec->y2 - rec->dx * rec->wid / 2.0;
    verY[3] = rec->y1 - rec->dx * rec->wid / 2.0;
    int offset;
    if ((rec->x1 < rec->x2) && (rec->y1 <= rec->y2)) offset = 0;       /* :959-966 */
    else if ((rec->x1 >= rec->x2) && (rec->y1 < rec->y2)) offset = 1;
    else if ((rec->x1 > rec->x2) && (rec->y1 >= rec->y2)) offset = 2;
    else offset = 3;
    for (int i = 0; i < 4; i++) { vx[i] = verX[(offset + i) % 4]; vy[i] = verY[(offset + i) % 4]; }

    double cx0 = ceil(vx[0]);
    int xRang_len = cvt_int(cx0 - floor(vx[2]));                       /* :973 */
    if (xRang_len < 0 && xRang_len != INT_MIN) xRang_len = -xRang_len;
    xRang_len = (int)((unsigned)xRang_len + 1u);                       /* abs(INT_MIN)+1 wraps like x86 */
    double lineK[4];
    lineK[0] = (vy[1] - vy[0]) / (vx[1] - vx[0]);                      /* :979-982 */
    lineK[1] = (vy[2] - vy[1]) / (vx[2] - vx[1]);
    lineK[2] = (vy[2] - vy[3]) / (vx[2] - vx[3]);
    lineK[3] = (vy[3] - vy[0]) / (vx[3] - vx[0]);
    int allPixNum = 0, aliPixNum = 0;
    for (int i = 0; i < xRang_len; i++) {
        int xr = cvt_int(i + cx0);                                     /* :976 */
        int yLow, yHigh;
        if (xr < vx[3]) yLow = cvt_int(ceil(vy[0] + (xr - vx[0]) * lineK[3]));     /* :988-989 */
        else            yLow = cvt_int(ceil(vy[3] + (xr - vx[3]) * lineK[2]));     /* :992-993 */
        if (xr < vx[1]) yHigh = cvt_int(floor(vy[0] + (xr - vx[0]) * lineK[0]));   /* :998-999 */
        else            yHigh = cvt_int(floor(vy[1] + (xr - vx[1]) * lineK[1]));   /* :1002-1003 */
        if (xr < 0 || xr >= xLim) continue;                            /* :1007 */
        int lo = yLow < 0 ? 0 : yLow;
        int hi = yHigh > yLim - 1 ? yLim - 1 : yHigh;
        for (int j = lo; j <= hi; j++) {                               /* :1006-1015 */
            allPixNum++;
            double degDif = fabs(rec->deg - st->deg[(size_t)j * xLim + xr]);
            if (degDif > PI * 3 / 2.0) degDif = fabs(degDif - 2 * PI);
            if (degDif < rec->prec) aliPixNum++;
        }
    }
    if (allPixNum == 0 || aliPixNum == 0) return -logNT;               /* :1019-1022 */
    if (allPixNum == aliPixNum) return -logNT - allPixNum * log10(rec->p);  /* :1023-1026 */
    double proTerm = rec->p / (1.0 - rec->p);
    double log1Coef = orc_log_gamma(allPixNum + 1) - orc_log_gamma(aliPixNum + 1)
                    - orc_log_gamma(allPixNum - aliPixNum + 1);        /* :1029-1030 */
    double log1Term = log1Coef + aliPixNum * log(rec->p) + (allPixNum - aliPixNum) * log(1 - rec->p);
    double term = NFA_EXP(log1Term);
    double eps = 2.2204e-16;
    if (fabs(term) < 100 * eps) {                                      /* :1037-1043 */
        if (aliPixNum > allPixNum * rec->p) { st->nfa_host_only = 0; return -NFA_LOG10(term) - logNT; }
        return -logNT;
    }
    st->nfa_host_only = 0;
    double binTail = term, tole = 0.1;
    for (int i = aliPixNum + 1; i <= allPixNum; i++) {                 /* :1046-1056 */
        double binTerm = (allPixNum - i + 1) / (i * 1.0);
        double multTerm = binTerm * proTerm;
        term *= multTerm;
        binTail += term;
        if (binTerm < 1) {
            double err = term * ((1 - NFA_POW(multTerm, allPixNum - i + 1)) / (1.0 - multTerm) - 1);
            if (err < tole * fabs(-NFA_LOG10(binTail) - logNT) * binTail) break;
        }
    }
    return -NFA_LOG10(binTail) - logNT;
}

/* (debug counters only) a candidate v is about to be compared with 0 and, unless it is the first, with the best so far */
static double note_nfa(orc_state *st, double v, double best, int first)
{
    if (st->dbg && fabs(v) <= DBL_MAX) {
        /* margins: distance over the most two libms (each within an ulp of the correctly rounded exp / log10 / pow) can move the
         * operands apart -- noise(v) = 2^-51 |v + logNT| + 2^-52 (1 + max(|v|, logNT)), see k_region.hip: improve() */
        const double nv = 0x1p-51 * fabs(v + st->logNT) + 0x1p-52 * (1.0 + fmax(fabs(v), st->logNT));
        const double a = fabs(v) / nv;
        if (!st->nfa_host_only && a < st->dbg->nfa_min_abs) st->dbg->nfa_min_abs = a;   /* (-logNT - n log10 p: host numbers on the HIP path too) */
        if (!first && v != best) {
            const double nb = 0x1p-51 * fabs(best + st->logNT) + 0x1p-52 * (1.0 + fmax(fabs(best), st->logNT));
            const double g = fabs(v - best) / (nv + nb);
            if (g < st->dbg->nfa_min_gap) st->dbg->nfa_min_gap = g;
        }
    }
    return v;
}

/* a14: RectangleImprover, myLSD.cpp:1061-1158 */
static double rectangle_improver(orc_state *st, orc_rec *rec_io)
{
    const double delt = 0.5, delt2 = delt / 2.0;
    orc_rec best = *rec_io;
    double bestNFA = note_nfa(st, rectangle_nfa(st, &best), 0, 1);
    if (bestNFA > 0) return bestNFA;
    orc_rec r = best;
    for (int i = 0; i < 5; i++) {                                      /* :1084-1092 */
        r.p /= 2.0; r.prec = r.p * PI;
        double v = note_nfa(st, rectangle_nfa(st, &r), bestNFA, 0);
        if (v > bestNFA) { bestNFA = v; best = r; }
    }
    if (bestNFA > 0) { *rec_io = best; return bestNFA; }
    r = best;
    for (int i = 0; i < 5; i++) {                                      /* :1097-1107 */
        if (r.wid - delt >= 0.5) {
            r.wid -= delt;
            double v = note_nfa(st, rectangle_nfa(st, &r), bestNFA, 0);
            if (v > bestNFA) { bestNFA = v; best = r; }
        }
    }
    if (bestNFA > 0) { *rec_io = best; return bestNFA; }
    r = best;
    for (int i = 0; i < 5; i++) {                                      /* :1112-1125 */
        if (r.wid - delt >= 0.5) {
            r.x1 -= r.dy * delt2; r.y1 += r.dx * delt2;
            r.x2 -= r.dy * delt2; r.y2 += r.dx * delt2;
            r.wid -= delt;
            double v = note_nfa(st, rectangle_nfa(st, &r), bestNFA, 0);
            if (v > bestNFA) { bestNFA = v; best = r; }
        }
    }
    if (bestNFA > 0) { *rec_io = best; return bestNFA; }
    r = best;
    for (int i = 0; i < 5; i++) {                                      /* :1130-1143 */
        if (r.wid - delt >= 0.5) {
            r.x1 += r.dy * delt2; r.y1 -= r.dx * delt2;
            r.x2 += r.dy * delt2; r.y2 -= r.dx * delt2;
            r.wid -= delt;
            double v = note_nfa(st, rectangle_nfa(st, &r), bestNFA, 0);
            if (v > bestNFA) { bestNFA = v; best = r; }
        }
    }
    if (bestNFA > 0) { *rec_io = best; return bestNFA; }
    r = best;
    for (int i = 0; i < 5; i++) {                                      /* :1148-1156 */
        r.p /= 2.0; r.prec = r.p * PI;
        double v = note_nfa(st, rectangle_nfa(st, &r), bestNFA, 0);
        if (v > bestNFA) { bestNFA = v; best = r; }
    }
    *rec_io = best;
    return bestNFA;
}

/* a15: rec -> structLinesInfo + lineIm raster, myLSD.cpp:282-368 (baseFunc.cpp:6-16 for sind/cosd/atand) */
static void line_from_rec(const orc_rec *rc, int oriMapCol, int oriMapRow, uint8_t *lineIm, orc_line *out)
{
    double x1 = rc->x1, y1 = rc->y1, x2 = rc->x2, y2 = rc->y2;
    double k = (y2 - y1) / (x2 - x1);                                  /* :289 */
    double ang = atan(k) * 180.0 / PI;                                 /* atand */
    int orient = 1;
    if (ang < 0) { ang += 180; orient = -1; }
    int xLow, xHigh, yLow, yHigh;
    if (x1 > x2) { xLow = cvt_int(floor(x2)); xHigh = cvt_int(ceil(x1)); }
    else         { xLow = cvt_int(floor(x1)); xHigh = cvt_int(ceil(x2)); }
    if (y1 > y2) { yLow = cvt_int(floor(y2)); yHigh = cvt_int(ceil(y1)); }
    else         { yLow = cvt_int(floor(y1)); yHigh = cvt_int(ceil(y2)); }
    double xRang = fabs(x2 - x1), yRang = fabs(y2 - y1);
    int xx_len = xHigh - xLow + 1, yy_len = yHigh - yLow + 1;
    if (lineIm) {
        /* Q10: the sampled array has `n` entries; the marking loop may run longer (UB) -- only the
         * sampled part is restated. */
        if (xRang > yRang) {                                           /* :319-330 */
            for (int j = 0; j < xx_len; j++) {
                int xx = j + xLow;
                int yy = cvt_int(round((xx - x1) * k + y1));
                if (xx < 0 || xx >= oriMapCol || yy < 0 || yy >= oriMapRow) continue;
                if (xx != 0 && yy != 0) lineIm[(size_t)yy * oriMapCol + xx] = 255;   /* :346 */
            }
        } else {                                                       /* :331-342 */
            for (int j = 0; j < yy_len; j++) {
                int yy = j + yLow;
                int xx = cvt_int(round((yy - y1) / k + x1));
                if (xx < 0 || xx >= oriMapCol || yy < 0 || yy >= oriMapRow) continue;
                if (xx != 0 && yy != 0) lineIm[(size_t)yy * oriMapCol + xx] = 255;   /* :352 */
            }
        }
    }
    out->k = k;
    out->b = (y1 + y2) / 2.0 - k * (x1 + x2) / 2.0;                    /* :359 */
    out->dx = cos(ang / 180.0 * PI);                                   /* cosd */
    out->dy = sin(ang / 180.0 * PI);                                   /* sind */
    out->x1 = x1; out->y1 = y1; out->x2 = x2; out->y2 = y2;
    double ey = y2 - y1, ex = x2 - x1;
    out->len = sqrt(ey * ey + ex * ex);                                /* :366 */
    out->orient = orient;
}

/* ------------------------------------------------------------------------------------ */
/* myLineSegmentDetector, myLSD.cpp:129-376                                              */
/* ------------------------------------------------------------------------------------ */
int orc_lsd(uint8_t *map, int cols, int rows, size_t stride,
            double sca, double sig, double angThre, double denThre, int pseBin,
            uint8_t *lineIm, orc_line **lines, int *n, orc_debug *dbg)
{
    if (!map || cols <= 0 || rows <= 0 || !lines || !n) return -1;
    if (dbg) { memset(dbg, 0, sizeof(*dbg)); dbg->nfa_min_abs = HUGE_VAL; dbg->nfa_min_gap = HUGE_VAL; }
    const int w = cvt_int(floor(cols * sca));                          /* :132 */
    const int h = cvt_int(floor(rows * sca));                          /* :133 */
    if (w < 2 || h < 2) { *lines = NULL; *n = 0; return 0; }
    /* a1: in-place remap, rows/cols >= 1 only (:135-142, Q2) */
    for (int y = 1; y < rows; y++)
        for (int x = 1; x < cols; x++) {
            uint8_t *p = &map[(size_t)y * stride + x];
            if (*p == 1) *p = 255; else if (*p == 255) *p = 0;
        }
    double *gauss = gaussian_sampler(map, cols, rows, stride, sca, sig, w, h);   /* :143 */

    const size_t npx = (size_t)w * h;
    uint8_t *used = (uint8_t *)calloc(npx + 1, 1);
    double *deg = (double *)calloc(npx + 1, sizeof(double));
    double *mag = (double *)calloc(npx + 1, sizeof(double));
    const double degThre = angThre / 180.0 * PI;                       /* :148 */
    const double gradThre = 2.0 / sin(degThre);                        /* :149 */
    double maxGrad = 0;
    for (int y = 1; y < h; y++)                                        /* :153-174, Q3 */
        for (int x = 1; x < w; x++) {
            double A = gauss[(size_t)y * w + x], B = gauss[(size_t)y * w + x - 1];
            double C = gauss[(size_t)(y - 1) * w + x], D = gauss[(size_t)(y - 1) * w + x - 1];
            double gradX = (B + D - A - C) / 2.0;
            double gradY = (C + D - A - B) / 2.0;
            double m = sqrt(gradX * gradX + gradY * gradY);
            mag[(size_t)y * w + x] = m;
            if (m < gradThre) used[(size_t)y * w + x] = 1;
            if (maxGrad < m) maxGrad = m;
            double d = atan2(gradX, -gradY);
            if (fabs(d - PI) < 0.000001) d = 0;
            deg[(size_t)y * w + x] = d;
        }
    if (dbg) {
        dbg->w = w; dbg->h = h; dbg->maxGrad = maxGrad;
        dbg->gauss = (double *)malloc(npx * sizeof(double)); memcpy(dbg->gauss, gauss, npx * sizeof(double));
        dbg->mag = (double *)malloc(npx * sizeof(double));   memcpy(dbg->mag, mag, npx * sizeof(double));
        dbg->deg = (double *)malloc(npx * sizeof(double));   memcpy(dbg->deg, deg, npx * sizeof(double));
        dbg->used0 = (uint8_t *)malloc(npx);                 memcpy(dbg->used0, used, npx);
    }

    /* a4: pseudo-bin + raster-order compaction (:177-201).  A blank image (maxGrad == 0) makes the
     * reference read uninitialised memory; restated as "no sortable pixels". */
    int *binv = (int *)malloc((npx + 1) * sizeof(int));
    int nb = 0;
    if (maxGrad > 0) {
        double zoom = 1.0 * pseBin / maxGrad;                          /* :179, Q4 */
        for (size_t p = 0; p < npx; p++) {
            int v = cvt_int(floor(mag[p] * zoom));
            if (v > pseBin) v = pseBin;
            binv[p] = v;
            if ((uint16_t)v != 0) nb++;
        }
    }
    /* a5: qsort with Comp (:204, :486-489) on glibc == stable descending order (Q4; pinned by
     * orc_selftest_qsort_stable).  Restated as a counting sort: bins high -> low, raster order inside. */
    int *ov = (int *)malloc((nb + 1) * sizeof(int));
    int *ox = (int *)malloc((nb + 1) * sizeof(int));
    int *oy = (int *)malloc((nb + 1) * sizeof(int));
    if (nb > 0) {
        size_t *start = (size_t *)calloc(65537, sizeof(size_t));
        for (size_t p = 0; p < npx; p++) { uint16_t v = (uint16_t)binv[p]; if (v) start[v]++; }
        size_t acc = 0;
        for (int v = 65535; v >= 1; v--) { size_t c = start[v]; start[v] = acc; acc += c; }
        for (size_t p = 0; p < npx; p++) {
            uint16_t v = (uint16_t)binv[p];
            if (v) { size_t r = start[v]++; ov[r] = v; ox[r] = (int)(p % w); oy[r] = (int)(p / w); }
        }
        free(start);
    }
    free(binv);
    if (dbg) {
        dbg->nb = nb;
        dbg->ord_v = (int *)malloc((nb + 1) * sizeof(int)); memcpy(dbg->ord_v, ov, nb * sizeof(int));
        dbg->ord_x = (int *)malloc((nb + 1) * sizeof(int)); memcpy(dbg->ord_x, ox, nb * sizeof(int));
        dbg->ord_y = (int *)malloc((nb + 1) * sizeof(int)); memcpy(dbg->ord_y, oy, nb * sizeof(int));
        dbg->seeds = (orc_seed *)malloc((nb + 1) * sizeof(orc_seed));
    }

    /* a6: thresholds (:207-209) */
    const double logNT = 5 * (log10(h) + log10(w)) / 2.0;
    const double regThre = -logNT / log10(angThre / 180.0);
    const double aliPro = angThre / 180.0;

    orc_state st;
    st.w = w; st.h = h; st.mag = mag; st.deg = deg; st.used = used;
    st.cur = (int *)calloc(npx + 1, sizeof(int)); st.cur_id = 0;
    st.rx = (int *)malloc((npx + 2) * sizeof(int)); st.ry = (int *)malloc((npx + 2) * sizeof(int));
    st.gx = (int *)malloc((npx + 2) * sizeof(int)); st.gy = (int *)malloc((npx + 2) * sizeof(int));
    st.gnum = 0; st.logNT = logNT; st.dbg = dbg;

    int cap = 64, regCnt = 0;
    orc_rec *recSave = (orc_rec *)malloc(cap * sizeof(orc_rec));
    double *recRaw = dbg ? (double *)malloc(cap * 12 * sizeof(double)) : NULL;

    /* a7: seed loop (:219-272) */
    for (int i = 0; i < nb; i++) {
        int yIdx = oy[i], xIdx = ox[i];
        if (used[(size_t)yIdx * w + xIdx] != 0) continue;              /* :222 */
        orc_reg reg = region_grower(&st, xIdx, yIdx, deg[(size_t)yIdx * w + xIdx], degThre);
        orc_seed *tr = NULL;
        if (dbg) {
            tr = &dbg->seeds[dbg->n_seed++];
            tr->order_idx = i; tr->x = xIdx; tr->y = yIdx; tr->num = reg.num;
            tr->outcome = 0; tr->final_num = reg.num; tr->logNFA = 0;
        }
        if (reg.num < regThre) continue;                               /* :228 -- dropped, NOT marked (Q5) */
        orc_rec rec = rectangle_converter(&st, &reg, aliPro, degThre); /* :232 */
        int ok = refiner(&st, &reg, &rec, denThre);                    /* :234 */
        if (tr) tr->final_num = reg.num;
        if (!ok) { if (tr) tr->outcome = 1; continue; }                /* :237 */
        double logNFA = rectangle_improver(&st, &rec);                 /* :240 */
        if (tr) tr->logNFA = logNFA;
        if (logNFA <= 0) {                                             /* :242-250: mark curMap==1 pixels 2 */
            for (int k = 0; k < st.gnum; k++) {
                size_t p = (size_t)st.gy[k] * w + st.gx[k];
                if (st.cur[p] == st.cur_id) used[p] = 2;
            }
            if (tr) tr->outcome = 2;
            continue;
        }
        if (regCnt >= cap) {
            cap *= 2;
            recSave = (orc_rec *)realloc(recSave, cap * sizeof(orc_rec));
            if (recRaw) recRaw = (double *)realloc(recRaw, cap * 12 * sizeof(double));
        }
        if (recRaw) memcpy(recRaw + (size_t)regCnt * 12, &rec, 12 * sizeof(double));
        if (sca != 1) {                                                /* :252-258 */
            rec.x1 = (rec.x1 - 1.0) / sca + 1;
            rec.y1 = (rec.y1 - 1.0) / sca + 1;
            rec.x2 = (rec.x2 - 1.0) / sca + 1;
            rec.y2 = (rec.y2 - 1.0) / sca + 1;
            rec.wid = (rec.wid - 1.0) / sca + 1;
        }
        for (int k = 0; k < st.gnum; k++) {                            /* :259-265: mark curMap==1 pixels 1 */
            size_t p = (size_t)st.gy[k] * w + st.gx[k];
            if (st.cur[p] == st.cur_id) used[p] = 1;
        }
        recSave[regCnt++] = rec;
        if (tr) tr->outcome = 3;
    }

    /* a15: line list + raster (:274-368); Q11: order = acceptance order */
    if (lineIm) memset(lineIm, 0, (size_t)rows * cols);
    orc_line *li = (orc_line *)calloc(regCnt > 0 ? regCnt : 1, sizeof(orc_line));
    for (int i = 0; i < regCnt; i++) line_from_rec(&recSave[i], cols, rows, lineIm, &li[i]);
    *lines = li; *n = regCnt;
    if (dbg) {
        dbg->used = (uint8_t *)malloc(npx); memcpy(dbg->used, used, npx);
        dbg->recs = recRaw;
    }
    free(recSave); free(st.cur); free(st.rx); free(st.ry); free(st.gx); free(st.gy);
    free(ov); free(ox); free(oy); free(used); free(deg); free(mag); free(gauss);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* createMapCache, myLSD.cpp:11-127                                                      */
/* ------------------------------------------------------------------------------------ */
int orc_map_cache(const uint8_t *map, int cols, int rows, size_t stride,
                  double res, double z_occ_max_dis, double *out)
{
    if (!map || !out || cols <= 0 || rows <= 0) return -1;
    const int cell_radius = cvt_int(floor(z_occ_max_dis / res));       /* :13 */
    const int height = rows, width = cols;
    const size_t npx = (size_t)rows * cols;
    uint8_t *flag = (uint8_t *)calloc(npx, 1);
    /* FIFO of (src, cur) cells; every cell enters at most once */
    int *q_si = (int *)malloc(npx * sizeof(int)), *q_sj = (int *)malloc(npx * sizeof(int));
    int *q_ci = (int *)malloc(npx * sizeof(int)), *q_cj = (int *)malloc(npx * sizeof(int));
    size_t head = 0, tail = 0;
    for (int i = 0; i < height; i++)                                   /* :22-40 */
        for (int j = 0; j < width; j++) {
            if (map[(size_t)i * stride + j] == 1) {
                q_si[tail] = i; q_sj[tail] = j; q_ci[tail] = i; q_cj[tail] = j; tail++;
                out[(size_t)i * width + j] = 0;
                flag[(size_t)i * width + j] = 1;
            } else out[(size_t)i * width + j] = z_occ_max_dis;
        }
    static const int di4[4] = { -1, 0, 1, 0 }, dj4[4] = { 0, -1, 0, 1 };  /* up, left, down, right :48,67,86,105 */
    while (head < tail) {                                              /* :44-124 */
        int src_i = q_si[head], src_j = q_sj[head], cur_i = q_ci[head], cur_j = q_cj[head];
        head++;
        for (int d = 0; d < 4; d++) {
            int ni = cur_i + di4[d], nj = cur_j + dj4[d];
            if (ni < 0 || nj < 0 || ni >= height || nj >= width) continue;
            if (flag[(size_t)ni * width + nj]) continue;
            double di = abs(cur_i - src_i), dj = abs(cur_j - src_j);   /* the PARENT's offset (:49-50) */
            double distance = sqrt(di * di + dj * dj);
            if (distance <= cell_radius) {
                out[(size_t)ni * width + nj] = distance * res;
                flag[(size_t)ni * width + nj] = 1;
                q_si[tail] = src_i; q_sj[tail] = src_j; q_ci[tail] = ni; q_cj[tail] = nj; tail++;
            }
        }
    }
    free(flag); free(q_si); free(q_sj); free(q_ci); free(q_cj);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
typedef struct { int value, x, y; } cell_t;                            /* nodeBinCell, myLSD.h:43-47 */
static int ref_comp(const void *p1, const void *p2)                    /* Comp, myLSD.cpp:486-489 */
{
    return ((const cell_t *)p2)->value > ((const cell_t *)p1)->value ? 1 : -1;
}
int orc_selftest_qsort_stable(int n, unsigned seed)
{
    cell_t *a = (cell_t *)malloc(sizeof(cell_t) * (n + 1));
    cell_t *b = (cell_t *)malloc(sizeof(cell_t) * (n + 1));
    unsigned s = seed ? seed : 1u;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        a[i].value = 1 + (int)((s >> 16) % 37u);   /* many ties */
        a[i].x = i; a[i].y = 0;
    }
    /* stable descending reference order */
    int m = 0;
    for (int v = 37; v >= 1; v--)
        for (int i = 0; i < n; i++) if (a[i].value == v) b[m++] = a[i];
    qsort(a, n, sizeof(cell_t), ref_comp);
    int bad = 0;
    for (int i = 0; i < n; i++) if (a[i].value != b[i].value || a[i].x != b[i].x) bad++;
    free(a); free(b);
    return bad;
}

void orc_free(void *p) { free(p); }
void orc_debug_free(orc_debug *d)
{
    if (!d) return;
    free(d->gauss); free(d->mag); free(d->deg); free(d->used0); free(d->used);
    free(d->ord_v); free(d->ord_x); free(d->ord_y); free(d->seeds); free(d->recs);
    memset(d, 0, sizeof(*d));
}


/* ------------------------------------------------------------------------------------------------------------
 * myfa::thread_ScanToMapMatch batch (LSD/myFA.cpp:197-396).  PARITY UNPINNED (see lsd_oracle.h).
 * ------------------------------------------------------------------------------------------------------------ */
static double orc_pi(void) { return 4.0 * atan(1.0); }                     /* LSD/baseFunc.cpp:4 */
static double orc_sind(double x) { return sin(x / 180.0 * orc_pi()); }     /* :6-8 */
static double orc_cosd(double x) { return cos(x / 180.0 * orc_pi()); }     /* :10-12 */
static double orc_atand(double x) { return atan(x) * 180.0 / orc_pi(); }   /* :14-16 */

static double orc_line_direction(double staX, double staY, double endX, double endY)   /* myFA.cpp:272-305 */
{
    double angle;
    if (staX == endX && staY != endY) angle = staY < endY ? 90 : -90;
    else if (staX != endX && staY == endY) angle = staX < endX ? 0 : 180;
    else angle = orc_atand((endY - staY) / (endX - staX));
    if (angle < 0 && staX > endX) { angle += 180; return angle; }
    if (angle > 0 && staX > endX) { angle -= 180; return angle; }
    return angle;
}

int orc_scan_to_map_match(const double *map_cache, int cols, int rows,
                          const orc_line *map_lines, int n_map, const orc_line *scan_lines, int n_scan,
                          const orc_position *pts, int n_points, orc_position lidar, orc_position last,
                          const int *pairs, int n_pairs, double z_occ_max_dis, double max_esti_dist, orc_match_score *out)
{
    if (!map_cache || !map_lines || !scan_lines || !pairs || !out || (n_points > 0 && !pts)) return -1;
    for (int p = 0; p < n_pairs; p++) {
        const int im = pairs[2 * p], is = pairs[2 * p + 1];
        if (im < 0 || im >= n_map || is < 0 || is >= n_scan) return -2;
        const orc_line *ml = &map_lines[im], *sl = &scan_lines[is];
        for (int i = 1; i <= 4; i++) {                                       /* :205-249 */
            const int mrev = i >= 3, srev = (i == 2 || i == 4);
            const double msx = mrev ? ml->x2 : ml->x1, msy = mrev ? ml->y2 : ml->y1;
            const double mex = mrev ? ml->x1 : ml->x2, mey = mrev ? ml->y1 : ml->y2;
            const double ssx = srev ? sl->x2 : sl->x1, ssy = srev ? sl->y2 : sl->y1;
            const double sex = srev ? sl->x1 : sl->x2, sey = srev ? sl->y1 : sl->y2;
            const double mapAng = orc_line_direction(msx, msy, mex, mey);    /* :252-258 */
            const double scanAng = orc_line_direction(ssx, ssy, sex, sey);
            double angDiff = mapAng - scanAng;                               /* :310 */
            orc_match_score *o = &out[4 * p + i - 1];
            /* rotateScanIm :323-326 */
            const double rlx = (lidar.x - ssx) * orc_cosd(angDiff) - (lidar.y - ssy) * orc_sind(angDiff) + msx;
            const double rly = (lidar.x - ssx) * orc_sind(angDiff) + (lidar.y - ssy) * orc_cosd(angDiff) + msy;
            o->pos.x = rlx; o->pos.y = rly; o->pos.ang = 0; o->score = INFINITY;
            if (!(sqrt(pow(rlx - last.x, 2) + pow(rly - last.y, 2)) < max_esti_dist || last.x == -1)) continue;   /* :330 */
            const double cd = orc_cosd(angDiff), sd = orc_sind(angDiff);
            double sumValidDist = 0, sumMaxDist = 0, numValidPoint = 0;      /* CalcScore :359-396 */
            for (int c = 0; c < n_points; c++) {
                const double ox = pts[c].x - ssx, oy = pts[c].y - ssy;       /* :317-320 */
                const double rx = ox * cd - oy * sd + msx;                   /* :333-336 */
                const double ry = ox * sd + oy * cd + msy;
                const int x = cvt_int(round(rx)), y = cvt_int(round(ry));    /* :368-369 */
                if (y >= 0 && y < rows && x >= 0 && x < cols) {
                    numValidPoint += 1;
                    const double v = map_cache[(size_t)y * cols + x];
                    if (v >= z_occ_max_dis) sumMaxDist += 10;                /* :374-378 */
                    else sumValidDist += v;
                }
            }
            while (angDiff <= -180) angDiff += 360;                          /* :339-342 */
            while (angDiff > 180) angDiff -= 360;
            o->pos.ang = angDiff;
            const double numAllPoint = n_points;
            if (n_points == 0) o->score = INFINITY;                          /* :248-263: RSI.numScanImPoint == 0, CalcScore is not called */
            else if (numValidPoint < 0.7 * numAllPoint) o->score = INFINITY; /* :388-392 */
            else o->score = (sumValidDist + sumMaxDist) / (numValidPoint) + 10 * (numAllPoint - numValidPoint) / numAllPoint;
        }
    }
    return 0;
}


/* ------------------------------------------------------------------------------------ */
/* myrdp::FeatureScan, LSD/myRDP.cpp (see lsd_oracle.h)                                  */
/* ------------------------------------------------------------------------------------ */
static double rdp_thre_delta(double val)                                  /* getThresholdDeltaDist :347-368 */
{
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

static void rdp_split(const orc_polar *sc, const double *px, const double *py, unsigned char *split, int len_lp, int sp, int ep,
                      double threLine)                                    /* SplitMergeAssistant :219-272 */
{
    const int len = ep > sp ? ep - sp + 1 : len_lp + ep - sp + 1;         /* :223-239 (the cluster may wrap around the scan) */
    if (len <= 2) return;
    const double k = (py[ep] - py[sp]) / (px[ep] - px[sp]);               /* :249 */
    const double d = py[ep] - k * px[ep];
    double dist_max = 0;
    int i_max = 0;
    for (int i = 1; i < len - 1; i++) {
        int a = sp + i;
        if (a >= len_lp) a -= len_lp;
        const double dist = fabs(k * px[a] - py[a] + d) / sqrt(pow(k, 2) + 1);   /* :256 */
        if (dist > dist_max) { dist_max = dist; i_max = a; }
    }
    const double threDist = sc[i_max].range > 9 ? sc[i_max].range * threLine : threLine;   /* :259-263 */
    if (dist_max > threDist) {
        rdp_split(sc, px, py, split, len_lp, sp, i_max, threLine);
        rdp_split(sc, px, py, split, len_lp, i_max, ep, threLine);
        split[i_max] = 1;
    }
}

int orc_feature_scan(orc_map_param mp, const orc_polar *scan, int len_lp, int region_point_limit, double thre_line,
                     double line_dist_thre_m, orc_line *lines_out, int *n_lines, orc_position *pts_out, int pts_cap, int *n_pts,
                     double *lidar_pos, int *im_size)
{
    if (!scan || len_lp < 1 || !lines_out || !n_lines || !n_pts || !lidar_pos || !im_size || (pts_cap > 0 && !pts_out)) return -1;
    double *px = (double *)malloc(sizeof(double) * (size_t)len_lp * 2), *py = px + len_lp;
    unsigned char *split = (unsigned char *)calloc((size_t)len_lp, 1);
    int *cs = (int *)malloc(sizeof(int) * (size_t)len_lp * 2), *ce = cs + len_lp;
    int *axis = (int *)malloc(sizeof(int) * ((size_t)len_lp + 2));
    if (!px || !split || !cs || !axis) { free(px); free(split); free(cs); free(axis); return -3; }
    /* scanPose = {0, 0, 0} (:11) */
    for (int i = 0; i < len_lp; i++) { px[i] = scan[i].range * cos(scan[i].angle + 0.0) + 0.0; py[i] = scan[i].range * sin(scan[i].angle + 0.0) + 0.0; }
    /* RegionSegmentation :274-345 */
    int cellNumber = 0, startNum = 0;
    for (int i = 0; i < len_lp; i++) {
        const int nx = i == len_lp - 1 ? 0 : i + 1;
        const double dX = px[i] - px[nx], dY = py[i] - py[nx];
        const double deltaDist = sqrt(dX * dX + dY * dY);
        const double thre = rdp_thre_delta(scan[i].range);
        if (deltaDist > thre) {
            cs[cellNumber] = startNum; ce[cellNumber] = i;                 /* :346-351 */
            if (abs(i - startNum) >= region_point_limit) cellNumber++;
            startNum = i + 1;                                              /* :318-322 (the point itself is never used) */
        }
        if (deltaDist <= thre && i == len_lp - 1) cs[0] = startNum;        /* :361-365 the last cluster joins the first */
    }
    /* SplitMerge :187-217 */
    for (int c = 0; c < cellNumber; c++) rdp_split(scan, px, py, split, len_lp, cs[c], ce[c], thre_line);
    /* pixel coordinates and the image size :16-37 */
    double minX = INFINITY, minY = INFINITY, maxX = 0, maxY = 0;
    double *gx = px, *gy = py;                                             /* (the metric coordinates are not needed any more) */
    for (int i = 0; i < len_lp; i++) {
        const double X = floor((scan[i].range * cos(scan[i].angle + 0.0) + 0.0 - mp.mapOriX) / mp.mapResol);
        const double Y = floor((scan[i].range * sin(scan[i].angle + 0.0) + 0.0 - mp.mapOriY) / mp.mapResol);
        gx[i] = X; gy[i] = Y;
        if (X < minX) minX = X;
        if (X > maxX) maxX = X;
        if (Y < minY) minY = Y;
        if (Y > maxY) maxY = Y;
    }
    const int oriXLim = cvt_int(ceil(maxX - minX)), oriYLim = cvt_int(ceil(maxY - minY));
    lidar_pos[0] = floor((0.0 - mp.mapOriX) / mp.mapResol - minX);
    lidar_pos[1] = floor((0.0 - mp.mapOriY) / mp.mapResol - minY);
    im_size[0] = oriXLim; im_size[1] = oriYLim;
    const double lineDistThre = line_dist_thre_m / mp.mapResol;
    int nl = 0, np = 0;
    for (int c = 0; c < cellNumber; c++) {                                 /* :45-177 */
        const int sp = cs[c], ep = ce[c];
        const int len_axis = ep > sp ? ep - sp + 1 : len_lp + ep - sp + 1;
        int num_split = 1;
        for (int j = 0; j < len_axis; j++) {
            int v = sp + j;
            if (v >= len_lp) v -= len_lp;
            if (split[v]) axis[num_split++] = v;
        }
        axis[0] = sp;
        axis[num_split++] = ep;
        for (int j = 0; j < num_split - 1; j++) {
            const double ax = gx[axis[j]], ay = gy[axis[j]], bx = gx[axis[j + 1]], by = gy[axis[j + 1]];
            const double lineDist = sqrt(pow(ax - bx, 2) + pow(ay - by, 2));
            if (!(lineDist >= lineDistThre)) continue;
            const double x1 = ax - minX, y1 = ay - minY, x2 = bx - minX, y2 = by - minY;
            const double k = (y2 - y1) / (x2 - x1);
            double ang = orc_atand(k);
            int orient = 1;
            if (ang < 0) { ang += 180; orient = -1; }
            const int xLow = cvt_int(floor(x1 > x2 ? x2 : x1)), xHigh = cvt_int(ceil(x1 > x2 ? x1 : x2));
            const int yLow = cvt_int(floor(y1 > y2 ? y2 : y1)), yHigh = cvt_int(ceil(y1 > y2 ? y1 : y2));
            const double xRang = fabs(x2 - x1), yRang = fabs(y2 - y1);
            const int xx_len = xHigh - xLow + 1, yy_len = yHigh - yLow + 1;
            /* the reference sizes its arrays by xRang > yRang and walks them by xx_len > yy_len (:109-153); the coordinates are
             * integers, so the two tests agree; the sampled length is walked here */
            const int along_x = xRang > yRang;
            const int cnt = along_x ? xx_len : yy_len;
            for (int m = 0; m < cnt; m++) {
                int xx, yy;
                if (along_x) { xx = m + xLow; yy = cvt_int(round((xx - x1) * k + y1)); }
                else { yy = m + yLow; xx = cvt_int(round((yy - y1) / k + x1)); }
                if (xx < 0 || xx >= oriXLim || yy < 0 || yy >= oriYLim) { xx = 0; yy = 0; }
                if (xx != 0 && yy != 0) {                                  /* 0 doubles as "invalid" (:140, :151) */
                    if (np < pts_cap) { pts_out[np].x = xx; pts_out[np].y = yy; pts_out[np].ang = 0; }
                    np++;
                }
            }
            if (nl < 360) {
                orc_line *L = &lines_out[nl];
                memset(L, 0, sizeof(*L));
                L->k = k; L->b = (y1 + y2) / 2.0 - k * (x1 + x2) / 2.0;
                L->dx = orc_cosd(ang); L->dy = orc_sind(ang);
                L->x1 = x1; L->y1 = y1; L->x2 = x2; L->y2 = y2;
                L->len = sqrt(pow(y2 - y1, 2) + pow(x2 - x1, 2));
                L->orient = orient;
            }
            nl++;
        }
    }
    *n_lines = nl; *n_pts = np;
    free(px); free(split); free(cs); free(axis);
    return 0;
}
