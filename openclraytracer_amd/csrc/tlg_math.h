// tlg_math.h -- arithmetic of the Two-Level Grid shared by host and device code of librtmi.so.
//
// Same rules as device_math.h (no contraction, correctly rounded / and sqrt, ternary min/max).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#ifndef RMIN
#define RMIN(X, Y) (((X) < (Y)) ? (X) : (Y))
#define RMAX(X, Y) (((X) > (Y)) ? (X) : (Y))
#endif

// float -> uint as the reference's convert_uint3 / C casts do for values in range; NaN, negative and too large
// values (undefined in C, and the reference never produces them on valid grids) give 0.
__host__ __device__ inline uint32_t tlg_f2u(float x) { return (x >= 1.0f && x < 4294967296.0f) ? (uint32_t)x : 0u; }

// pow(x, oneThird) of TwoLevelGrid.h:190-198 and TwoLevelGridManager.cpp:578, oneThird = 1.0f / 3.0f.
// OpenCL allows pow 16 ulp and the host branch uses the C library, so the reference's value is platform defined;
// here it is DEFINED as the fp32 rounding of x^0.3333333432674408 evaluated by this fixed sequence of IEEE double
// operations (identical on host and device): x^y = cbrt(x) * exp(E ln x) with E = fp32(1/3) - 1/3.
__host__ __device__ inline float tlg_pow_third(float xf)
{
    if (xf != xf) return xf;
    if (xf == 0.0f) return 0.0f;
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf > FLT_MAX) return xf;
    const double x = (double)xf;                                  // always a normal double
    const uint64_t bits = __builtin_bit_cast(uint64_t, x);
    const int e = (int)((bits >> 52) & 0x7ff) - 1023;
    const double m = __builtin_bit_cast(double, (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);   // [1,2)
    // ln m = 2 atanh(s), s = (m-1)/(m+1) <= 1/3: 1 + s^2/3 + ... + s^26/27 by Horner
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 27.0;
    for (int k = 25; k >= 1; k -= 2) p = p * s2 + 1.0 / (double)k;
    const double lnx = (double)e * 0x1.62e42fefa39efp-1 + 2.0 * (s * p);
    // cbrt: e = 3q + r, a = m * 2^r in [1,8), seven Halley steps from 1.5
    const int q = (e >= 0) ? e / 3 : -((-e + 2) / 3);
    const int r = e - 3 * q;
    const double a = m * (double)(1 << r);
    double y = 1.5;
    for (int i = 0; i < 7; i++) { const double y3 = y * y * y; y = y * (y3 + 2.0 * a) / (2.0 * y3 + a); }
    const double scale = __builtin_bit_cast(double, (uint64_t)(q + 1023) << 52);
    const double z = 0x1.5555556000000p-27 * lnx;
    const double res = (y * scale) * (1.0 + z + 0.5 * z * z);
    return (float)res;
}
