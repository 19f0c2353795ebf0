/* force-included when building liblsd_oracle_cr.so: see cr_shim.cpp */
#include <math.h>
double cr_sin(double), cr_cos(double), cr_atan2(double, double), cr_atan(double);
#define sin cr_sin
#define cos cr_cos
#define atan2 cr_atan2
#define atan cr_atan
