// Counter-based Poisson sampling for the extractSlices / poissonProcess stage
// (replaces uncommons/PoissonGenerator.java:95-109 + java.util.Random, which cost ~lambda+1
// Math.log calls per voxel on one strictly sequential stream).
//
// Generator : Philox4x32-10, key = 64-bit seed, counter = (voxel index lo, hi, stream, attempt)
// Sampler   : lambda < 10  -> inversion by sequential search on one 53-bit uniform
//             lambda >= 10 -> Hoermann's PTRS transformed rejection (one Philox block per attempt)
// All accept/reject arithmetic is IEEE +,-,*,/,sqrt on doubles plus the bit-defined log/exp
// below (built with -ffp-contract=off), so a CPU implementation of the same recipe gives
// identical counts.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mvsim {

struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * 0x1.0p-53;
}

// log(x), x > 0: fdlibm-style reduction to [sqrt(1/2), sqrt(2)), degree-14 odd polynomial.
__device__ __forceinline__ double det_log(double x)
{
    if (!(x > 0.0)) return -1.0e300;
    uint64_t u = (uint64_t)__double_as_longlong(x);
    int e = (int)(u >> 52) - 1023;
    if (e == -1023) {
        x = x * 0x1.0p54;
        u = (uint64_t)__double_as_longlong(x);
        e = (int)(u >> 52) - 1023 - 54;
    }
    u = (u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// exp(x) for the range the sampler needs (x in (-10, 0]); general clamp kept for safety.
__device__ __forceinline__ double det_exp(double x)
{
    if (x < -745.0) return 0.0;
    if (x > 709.0) return 1.0e308;
    const double kf = floor(x * 1.44269504088896338700e+00 + 0.5);
    const double hi = x - kf * 6.93147180369123816490e-01;
    const double lo = kf * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 + t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k = (int)kf;
    if (k < -1000)
        return (y * __longlong_as_double((long long)((uint64_t)(k + 1000 + 1023) << 52))) * 0x1.0p-1000;
    return y * __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
}

// log(k!) : table to 16, Stirling series beyond.
__device__ __forceinline__ double det_lgamma_int(long long k)
{
    switch (k) {
        case 0: case 1: return 0.0;
        case 2: return 0.6931471805599453;
        case 3: return 1.791759469228055;
        case 4: return 3.1780538303479458;
        case 5: return 4.787491742782046;
        case 6: return 6.579251212010101;
        case 7: return 8.525161361065415;
        case 8: return 10.60460290274525;
        case 9: return 12.801827480081469;
        case 10: return 15.104412573075516;
        case 11: return 17.502307845873887;
        case 12: return 19.987214495661885;
        case 13: return 22.552163853123425;
        case 14: return 25.19122118273868;
        case 15: return 27.89927138384089;
        case 16: return 30.671860106080672;
        default: break;
    }
    if (k < 0) return 0.0;
    const double x = (double)k + 1.0;
    const double ix = 1.0 / x;
    const double ix2 = ix * ix;
    const double ser = ix * (8.3333333333333333e-02 + ix2 * (-2.7777777777777778e-03 + ix2 * (7.9365079365079365e-04 + ix2 * -5.9523809523809524e-04)));
    return ((x - 0.5) * det_log(x) - x) + 0.9189385332046727 + ser;
}

// Poisson(lambda) for voxel `index` of stream `stream`.  lambda <= 0 or NaN -> 0 (the
// reference's loop does not terminate there; documented deviation Q9).
__device__ __forceinline__ float poisson_counter(double lambda, uint32_t k0, uint32_t k1, uint32_t stream,
                                                 uint64_t index)
{
    if (!(lambda > 0.0)) return 0.0f;
    const uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32);
    if (lambda < 10.0) {
        const Philox4 r = philox4x32_10(c0, c1, stream, 0u, k0, k1);
        const double u = u53(r.x, r.y);
        double p = det_exp(-lambda);
        double F = p;
        int k = 0;
        while (u >= F && k < 1000) {
            k += 1;
            p = (p * lambda) / (double)k;
            F = F + p;
        }
        return (float)k;
    }
    const double slam = sqrt(lambda);
    const double loglam = det_log(lambda);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    const double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (uint32_t attempt = 0; attempt < 0xFFFFFFFFu; ++attempt) {
        const Philox4 r = philox4x32_10(c0, c1, stream, attempt, k0, k1);
        const double U = u53(r.x, r.y) - 0.5;
        const double V = u53(r.z, r.w);
        const double us = 0.5 - fabs(U);
        const double kd = floor((2.0 * a / us + b) * U + lambda + 0.43);
        if (us >= 0.07 && V <= vr) return (float)(long long)kd;
        if (kd < 0.0 || (us < 0.013 && V > us)) continue;
        const long long k = (long long)kd;
        const double lhs = det_log(V) + det_log(invalpha) - det_log(a / (us * us) + b);
        const double rhs = (-lambda + kd * loglam) - det_lgamma_int(k);
        if (lhs <= rhs) return (float)k;
    }
    return (float)lambda;
}

}  // namespace mvsim
