// Diagnostic variant of the oracle (test infrastructure): the same restatement, but with sin / cos / atan2 / atan -- and the exp /
// log10 / pow of RectangleNFACalculator -- replaced by the correctly rounded double-double routines the HIP path uses
// (linesegmentdetector-slam_amd/csrc/crmath.h, host build; tests/test_crmath.py holds every one of them against mpmath).
// Purpose: to tell a genuine divergence of the HIP path from the one documented caveat -- glibc misrounds 0.1-0.2 % of
// sin/cos/atan2 calls by one ulp, and where that hits a structural tie (a rectangle edge exactly on a pixel row) the
// glibc-built reference and a correctly rounding implementation decide differently.  An image on which the HIP path
// differs from liblsd_oracle.so but equals liblsd_oracle_cr.so bit for bit is such a case (tests/golden/libm_ties.npz).
#include "../linesegmentdetector-slam_amd/csrc/crmath.h"
extern "C" {
double cr_sin(double x) { double s, c; if (!crm::sincos_cr(x, s, c)) return sin(x); return s; }
double cr_cos(double x) { double s, c; if (!crm::sincos_cr(x, s, c)) return cos(x); return c; }
double cr_atan2(double y, double x) { double o; if (!crm::atan2_cr(y, x, o)) return atan2(y, x); return o; }
double cr_atan(double v) { double o; if (!crm::atan_cr(v, o)) return atan(v); return o; }
double cr_exp(double x) { return crm::exp_cr(x); }
double cr_log10(double x) { return crm::log10_cr(x); }
double cr_pow(double x, double y) { return crm::pow_cr(x, y); }
}
