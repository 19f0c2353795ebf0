/*
 * myLSD.h -- C++ adapter with the reference's own names on top of the C ABI (lsd_hip.h).
 *
 * Gives host C++ code the call boundary of the reference's LSD/myLSD.h:
 *     mylsd::structLSD mylsd::myLineSegmentDetector(Mat MapGray, int oriMapCol, int oriMapRow,
 *             double sca, double sig, double angThre, double denThre, int pseBin);   (LSD/myLSD.h:132)
 * with the same argument meaning, the same in-place rewrite of MapGray (LSD/myLSD.cpp:135-142), the
 * same structLSD{lineIm, linesInfo, len_linesInfo} result (LSD/myLSD.h:123-127; linesInfo is malloc'ed
 * and owned by the caller exactly like the reference's), plus mylsd::runLSD -- the name BASELINE.json
 * uses; the reference has no such symbol, it is an alias with the baseFunc.h defaults.
 *
 * Image type: cv::Mat when OpenCV is available (define LSD_WITH_OPENCV or let __has_include find it),
 * otherwise the small ref-counted lsd::Image<T> below (same ptr<T>(row)/rows/cols/zeros surface).
 * Errors: the reference has no error handling (it would crash); the adapter throws mylsd::lsd_error
 * carrying the C-ABI status.  There is no CPU fallback behind this header.
 */
#ifndef LSD_MYLSD_ADAPTER_H
#define LSD_MYLSD_ADAPTER_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

#include "lsd_hip.h"

#if !defined(LSD_WITH_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define LSD_WITH_OPENCV 1
#endif
#endif
#ifdef LSD_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

/* Shared types and defaults of LSD/baseFunc.h.  The reference's own LSD/myLSD.h:37 does `#include <baseFunc.h>`, and its
 * callers include <myLSD.h> FIRST and <baseFunc.h> again afterwards (LSD/main_on_windows.cpp:5-8), so this header does the
 * same whenever the reference's baseFunc.h is on the include path (its include guard, _BASEFUNC_, makes the later include a
 * no-op).  Without the reference tree the few names this path needs are declared here under the SAME guard, so that a
 * baseFunc.h that turns up later in the translation unit (e.g. through a quoted include) is skipped instead of colliding. */
#if !defined(_BASEFUNC_) && defined(__has_include)
#if __has_include(<baseFunc.h>)
#include <baseFunc.h>
#endif
#endif
#ifndef _BASEFUNC_
#define _BASEFUNC_
#define LSD_ADAPTER_OWN_BASEFUNC 1
typedef struct _structMapParam { /* LSD/baseFunc.h:25-31 */
    int oriMapCol, oriMapRow;
    double mapResol, mapOriX, mapOriY;
} structMapParam;
typedef struct _structLinesInfo { /* LSD/baseFunc.h:33-44 */
    double k, b, dx, dy, x1, y1, x2, y2, len;
    int orient;
} structLinesInfo;
typedef struct _structPosition { /* LSD/baseFunc.h:46-50 */
    double x, y, ang;
} structPosition;
static const double z_occ_max_dis = 1; /* LSD/baseFunc.h:60 */
/* LSD defaults, LSD/baseFunc.h:64-68 */
static const double lsd_sca = 0.3, lsd_sig = 0.6, lsd_angThre = 22.5, lsd_denThre = 0.7;
static const int pseBin = 1024;
#endif
static_assert(sizeof(structLinesInfo) == sizeof(lsd_line), "structLinesInfo must be layout-identical to lsd_line");

namespace lsd {
/* Minimal ref-counted row-major image (shallow copies share pixels, like cv::Mat). */
template <class T>
class Image {
public:
    int rows = 0, cols = 0;
    size_t step = 0; /* bytes per row */
    Image() = default;
    static Image zeros(int r, int c) {
        Image m;
        m.rows = r; m.cols = c; m.step = sizeof(T) * (size_t)c;
        m.buf_.reset(static_cast<unsigned char*>(std::calloc((size_t)r * m.step + 16, 1)), std::free);
        m.data = m.buf_.get();
        return m;
    }
    template <class U> U* ptr(int row) { return reinterpret_cast<U*>(data + (size_t)row * step); }
    template <class U> const U* ptr(int row) const { return reinterpret_cast<const U*>(data + (size_t)row * step); }
    void release() { buf_.reset(); data = nullptr; rows = cols = 0; }
    unsigned char* data = nullptr;
private:
    std::shared_ptr<unsigned char> buf_;
};
}  // namespace lsd

namespace mylsd {

#ifdef LSD_WITH_OPENCV
typedef cv::Mat Mat;
inline Mat make_u8(int rows, int cols) { return cv::Mat::zeros(rows, cols, CV_8UC1); }
#else
typedef lsd::Image<unsigned char> Mat;
inline Mat make_u8(int rows, int cols) { return Mat::zeros(rows, cols); }
#endif

struct lsd_error : std::runtime_error {
    int status;
    lsd_error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

typedef struct _structLSD { /* LSD/myLSD.h:123-127 */
    Mat lineIm;
    structLinesInfo* linesInfo;
    int len_linesInfo;
} structLSD;

/* One context per GPU, created on first use and kept for the life of the process.  context() without an argument is the calling
 * thread's current device: set_device(d), else the environment variable LSD_DEVICE, else 0 -- so a host that drives several GPUs
 * (one thread or one process per GPU) calls mylsd::set_device(rank) once per thread and the reference's own call
 * sites stay as they are (LSD/main_on_windows.cpp:67-70). */
inline int& current_device_ref() {
    static thread_local int dev = -1;
    return dev;
}
inline void set_device(int device) { current_device_ref() = device; }
inline lsd_ctx* context(int device = -1) {
    enum { kMaxDevices = 64 };
    struct Table {
        std::mutex mu;
        lsd_ctx* c[kMaxDevices] = {};
        ~Table() { for (lsd_ctx* p : c) if (p) lsd_destroy(p); }
    };
    static Table t;
    if (device < 0) device = current_device_ref();
    if (device < 0) { const char* d = std::getenv("LSD_DEVICE"); device = d ? std::atoi(d) : 0; }
    if (device < 0 || device >= kMaxDevices) throw lsd_error(LSD_ERR_NO_DEVICE, lsd_strerror(LSD_ERR_NO_DEVICE));
    std::lock_guard<std::mutex> lock(t.mu);
    if (!t.c[device]) {
        const int st = lsd_create(&t.c[device], device);
        if (st != LSD_OK) { t.c[device] = nullptr; throw lsd_error(st, lsd_strerror(st)); }
    }
    return t.c[device];
}

inline structLSD myLineSegmentDetector(Mat MapGray, int oriMapCol, int oriMapRow, double sca, double sig,
                                       double angThre, double denThre, int pseBin_) {
    lsd_ctx* c = context();
    lsd_params p;
    p.sca = sca; p.sig = sig; p.angThre = angThre; p.denThre = denThre; p.pseBin = pseBin_;
    structLSD r;
    r.lineIm = make_u8(oriMapRow, oriMapCol);
    lsd_line* lines = nullptr;
    int n = 0;
    const int st = lsd_run(c, MapGray.template ptr<unsigned char>(0), oriMapCol, oriMapRow, (size_t)MapGray.step,
                           &p, r.lineIm.template ptr<unsigned char>(0), (size_t)r.lineIm.step, &lines, &n);
    if (st != LSD_OK) {
        lsd_free(lines);
        throw lsd_error(st, std::string(lsd_strerror(st)) + ": " + lsd_last_error(c));
    }
    r.linesInfo = reinterpret_cast<structLinesInfo*>(lines); /* malloc'ed; the caller owns it */
    r.len_linesInfo = n;
    return r;
}

/* mylsd::createMapCache (LSD/myLSD.h:131, LSD/myLSD.cpp:11-127): CV_64FC1 rows x cols, metres, capped at
 * z_occ_max_dis (LSD/baseFunc.h:60).  Must run before myLineSegmentDetector (which rewrites MapGray). */
#ifdef LSD_WITH_OPENCV
typedef cv::Mat MatF64;
inline MatF64 make_f64(int rows, int cols) { return cv::Mat::zeros(rows, cols, CV_64FC1); }
#else
typedef lsd::Image<double> MatF64;
inline MatF64 make_f64(int rows, int cols) { return MatF64::zeros(rows, cols); }
#endif
inline MatF64 createMapCache(Mat MapGray, double res, double z_occ_max_dis_ = z_occ_max_dis) {
    lsd_ctx* c = context();
    MatF64 out = make_f64(MapGray.rows, MapGray.cols);
    const int st = lsd_map_cache(c, MapGray.template ptr<unsigned char>(0), MapGray.cols, MapGray.rows, (size_t)MapGray.step,
                                 res, z_occ_max_dis_, out.template ptr<double>(0));
    if (st != LSD_OK) throw lsd_error(st, std::string(lsd_strerror(st)) + ": " + lsd_last_error(c));
    return out;
}

/* The replay driver's file formats (LSD/main_on_windows.cpp:27-46, main_on_linux.cpp's launch files point at the same data): mapParam.txt is
 * "cols rows resolution originX originY"; mapValue.txt holds rows x cols decimal cell values, row-major, cols-then-rows, which the driver
 * reads with fscanf("%d") straight into the uint8_t pixels -- i.e. every pixel keeps the LOW BYTE of its number (the three bytes the int
 * store spills over are rewritten by the reads that follow).  These two helpers give a host that binds the adapter the same maps without
 * the spill; false if the file is missing or short. */
inline bool loadMapParam(const char* path, structMapParam* mp) {
    FILE* fp = std::fopen(path, "r");
    if (!fp) return false;
    const int got = std::fscanf(fp, "%d %d %lf %lf %lf", &mp->oriMapCol, &mp->oriMapRow, &mp->mapResol, &mp->mapOriX, &mp->mapOriY);
    std::fclose(fp);
    return got == 5;
}
inline bool loadMapValue(const char* path, int oriMapCol, int oriMapRow, Mat* mapValue) {
    FILE* fp = std::fopen(path, "r");
    if (!fp) return false;
    Mat m = make_u8(oriMapRow, oriMapCol);
    bool ok = true;
    for (int r = 0; r < oriMapRow && ok; r++) {
        unsigned char* row = m.template ptr<unsigned char>(r);
        for (int c = 0; c < oriMapCol; c++) {
            int v;
            if (std::fscanf(fp, "%d", &v) != 1) { ok = false; break; }
            row[c] = (unsigned char)(v & 0xff);
        }
    }
    std::fclose(fp);
    if (ok) *mapValue = m;
    return ok;
}

/* north_star's name for the same call, with the LSD/baseFunc.h:64-68 defaults */
inline structLSD runLSD(Mat MapGray, double sca = 0.3, double sig = 0.6, double angThre = 22.5,
                        double denThre = 0.7, int pseBin_ = 1024) {
    return myLineSegmentDetector(MapGray, MapGray.cols, MapGray.rows, sca, sig, angThre, denThre, pseBin_);
}

}  // namespace mylsd

#endif /* LSD_MYLSD_ADAPTER_H */
