/* force-included when building liblsd_oracle_cr.so: see cr_shim.cpp */
#include <math.h>
double cr_sin(double), cr_cos(double), cr_atan2(double, double), cr_atan(double);
double cr_exp(double), cr_log10(double), cr_pow(double, double);
#define sin cr_sin
#define cos cr_cos
#define atan2 cr_atan2
#define atan cr_atan
/* RectangleNFACalculator's device-evaluated calls only (lsd_oracle.c): log-gamma and log(p) stay with the host libm, as on the HIP path */
#define NFA_EXP cr_exp
#define NFA_LOG10 cr_log10
#define NFA_POW cr_pow
