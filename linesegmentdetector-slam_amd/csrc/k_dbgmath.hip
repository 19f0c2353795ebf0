// k_dbgmath.hip -- test hook: evaluates the device build of the path's transcendental functions
// (devmath.h / crmath.h) on caller-supplied arrays so that tests can compare it with mpmath / glibc.
#include "lsd_internal.h"
#include "devmath.h"

namespace lsdhip {

__global__ void k_dbgmath(int fn, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o0,
                          double* __restrict__ o1, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == 0) { double s, c; sincos_g(a[i], s, c); o0[i] = s; o1[i] = c; }
    else if (fn == 1) { o0[i] = atan2_g(a[i], b[i]); o1[i] = 0; }
    else if (fn == 3) {          // what RegionGrower's classifier sees: fp32 angle with 2 mantissa bits dropped, hardware sin/cos
        const float af = __uint_as_float(__float_as_uint(__double2float_rn(a[i])) & ~3u);
        const float r = af * 0.15915494309189535f;
        o0[i] = (double)__builtin_amdgcn_sinf(r); o1[i] = (double)__builtin_amdgcn_cosf(r);
    }
    else if (fn == 4) { o0[i] = exp_g(a[i]); o1[i] = exp(a[i]); }                     // correctly rounded | the device math library's
    else if (fn == 5) { o0[i] = log10_g(a[i]); o1[i] = log10(a[i]); }
    else if (fn == 6) { o0[i] = pow_g(a[i], b[i]); o1[i] = pow(a[i], b[i]); }
    else { o0[i] = atan_g(a[i]); o1[i] = 0; }
}

// PMC calibration (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are only calibrated for 16-B-per-lane
// streams, while the LSD front end moves 8 B (fp64) and 1 B (u8) per lane.  These two kernels stream a known
// number of bytes with exactly those access shapes so that rocprofv3 --pmc readings can be scaled.
__global__ void k_calib_read8(const double* __restrict__ a, double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += stride) acc += a[i];
    if (acc == 123.456) out[0] = acc;
}
__global__ void k_calib_write8(double* __restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = (double)i;
}
void launch_calib(double* buf, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_calib_write8, dim3(8192), dim3(256), 0, s, buf, n);
    hipLaunchKernelGGL(k_calib_read8, dim3(8192), dim3(256), 0, s, buf, buf, n);
}

void launch_dbgmath(int fn, const double* a, const double* b, double* o0, double* o1, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_dbgmath, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, a, b, o0, o1, n);
}

}  // namespace lsdhip
