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
    else { o0[i] = atan_g(a[i]); o1[i] = 0; }
}

void launch_dbgmath(int fn, const double* a, const double* b, double* o0, double* o1, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_dbgmath, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, a, b, o0, o1, n);
}

}  // namespace lsdhip
