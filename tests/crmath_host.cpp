// Host build of crmath.h for tests/test_crmath.py (g++, libm fma): C entry points over arrays.
#include "../linesegmentdetector-slam_amd/csrc/crmath.h"
extern "C" {
int crm_sincos_n(const double* x, double* s, double* c, long n) {
    int bad = 0;
    for (long i = 0; i < n; i++) if (!crm::sincos_cr(x[i], s[i], c[i])) { s[i] = sin(x[i]); c[i] = cos(x[i]); bad++; }
    return bad;
}
int crm_atan2_n(const double* y, const double* x, double* out, long n) {
    int bad = 0;
    for (long i = 0; i < n; i++) if (!crm::atan2_cr(y[i], x[i], out[i])) { out[i] = atan2(y[i], x[i]); bad++; }
    return bad;
}
int crm_atan_n(const double* v, double* out, long n) {
    int bad = 0;
    for (long i = 0; i < n; i++) if (!crm::atan_cr(v[i], out[i])) { out[i] = atan(v[i]); bad++; }
    return bad;
}
void libm_sincos_n(const double* x, double* s, double* c, long n) { for (long i = 0; i < n; i++) { s[i] = sin(x[i]); c[i] = cos(x[i]); } }
void libm_atan2_n(const double* y, const double* x, double* out, long n) { for (long i = 0; i < n; i++) out[i] = atan2(y[i], x[i]); }
void libm_atan_n(const double* v, double* out, long n) { for (long i = 0; i < n; i++) out[i] = atan(v[i]); }
}
