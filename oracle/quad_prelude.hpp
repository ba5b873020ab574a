// oracle/quad_prelude.hpp - TEST INFRASTRUCTURE ONLY.  Turns the oracle's scalar type into IEEE binary128 (__float128,
// libquadmath: 113-bit significand, unit roundoff 2^-113 ~ 1e-34) for ONE translation unit (avm_truth.cpp): every system
// header the oracle uses and include/avm.h (the ABI stays FP64) are included first, then `double` is re-defined for the
// text of the oracle's own headers.  The constants in those headers are double literals (0.5, 1e-8, the options): the
// extended-precision build evaluates the SAME formulas on the SAME double inputs, only without rounding at 2^-53.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <quadmath.h>
#include <vector>

#include "../include/avm.h"

typedef __float128 avmo_real;

namespace std {  // the math calls of the oracle's headers resolve here for the wide type
inline avmo_real sqrt(avmo_real x) { return sqrtq(x); }
inline avmo_real fabs(avmo_real x) { return fabsq(x); }
inline avmo_real log(avmo_real x) { return logq(x); }
inline avmo_real exp(avmo_real x) { return expq(x); }
inline avmo_real sin(avmo_real x) { return sinq(x); }
inline avmo_real cos(avmo_real x) { return cosq(x); }
inline avmo_real acos(avmo_real x) { return acosq(x); }
inline avmo_real asin(avmo_real x) { return asinq(x); }
inline avmo_real atan2(avmo_real y, avmo_real x) { return atan2q(y, x); }
inline avmo_real hypot(avmo_real x, avmo_real y) { return hypotq(x, y); }
inline avmo_real hypot(avmo_real x, double y) { return hypotq(x, y); }
inline avmo_real pow(avmo_real x, avmo_real y) { return powq(x, y); }
inline avmo_real pow(avmo_real x, double y) { return powq(x, y); }
inline avmo_real pow(avmo_real x, int y) { return powq(x, y); }
inline avmo_real floor(avmo_real x) { return floorq(x); }
inline avmo_real round(avmo_real x) { return roundq(x); }
inline bool isfinite(avmo_real x) { return finiteq(x) != 0; }
inline bool isnan(avmo_real x) { return isnanq(x) != 0; }
inline avmo_real fmax(avmo_real a, avmo_real b) { return fmaxq(a, b); }
inline avmo_real fmin(avmo_real a, avmo_real b) { return fminq(a, b); }
inline avmo_real max(avmo_real a, double b) { return a < b ? (avmo_real)b : a; }
inline avmo_real max(double a, avmo_real b) { return a < b ? b : (avmo_real)a; }
inline avmo_real min(avmo_real a, double b) { return b < a ? (avmo_real)b : a; }
inline avmo_real min(double a, avmo_real b) { return b < a ? b : (avmo_real)a; }
}  // namespace std

#define AVMO_EIG_EPS scalbnq((avmo_real)1, -112)
// (the scalar type's limits: see oracle/linalg.hpp.  The unit roundoff that Eigen's slerp compares with stays FP64's: the
//  branch `|d| >= 1 - eps` is part of the algorithm, the inputs are FP64 quaternions)
#define AVMO_NUM_MAX FLT128_MAX
#define AVMO_NUM_MIN FLT128_MIN
#define AVMO_NUM_EPSILON ((avmo_real)2.220446049250313e-16)
#define AVMO_NUM_INF ((avmo_real)HUGE_VAL)
#define AVMO_NUM_NAN nanq("")
#define double avmo_real
