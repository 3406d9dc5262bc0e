// Counter-based Poisson sampling for the extractSlices / poissonProcess stage
// (replaces uncommons/PoissonGenerator.java:95-109 + java.util.Random, which cost ~lambda+1
// Math.log calls per voxel on one strictly sequential stream).
//
// Generator : Philox4x32-10, key = 64-bit seed
// Sampler   : lambda < 10  -> inversion by sequential search on a 32-bit uniform; the 4 voxels of
//                             group index>>2 share ONE Philox block, ctr = (index>>2, stream, 0)
//             lambda >= 10 -> Hoermann's PTRS transformed rejection on 32-bit uniforms; attempt 0 of the
//                             voxel pair index>>1 shares one block, ctr = (index>>1, stream, 1); retries
//                             take two attempts per block, ctr = (index, stream, 2 + (a-1)/2)
// All accept/reject arithmetic is IEEE +,-,*,/,sqrt,fma on doubles plus the bit-defined log/exp
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
        // one 32x32 -> 64 multiply per product (v_mad_u64_u32) instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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

// exp(-lambda), 0 < lambda < 10, division free: 2^k * sum_{n<=13} r^n / n!  (Horner, explicit FMAs)
__device__ __forceinline__ double det_exp_neg(double lambda)
{
    const double x = -lambda;
    const double kf = floor(x * 1.44269504088896338700e+00 + 0.5);
    const double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return p * __longlong_as_double((long long)((uint64_t)((int)kf + 1023) << 52));
}

// log(k!) : table to 16, Stirling series beyond.
__device__ const double kLogFact[17] = {
    0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479458, 4.787491742782046, 6.579251212010101,
    8.525161361065415, 10.60460290274525, 12.801827480081469, 15.104412573075516, 17.502307845873887,
    19.987214495661885, 22.552163853123425, 25.19122118273868, 27.89927138384089, 30.671860106080672};

// (`lf`: where the table is read from -- the resolver keeps a copy in LDS: a dependent GLOBAL load inside its divergent attempt costs every
// lane of the wave a trip to the cache)
__device__ __forceinline__ double det_lgamma_int(long long k, const double* lf = kLogFact)
{
    if (k <= 16) return lf[k < 0 ? 0 : k];
    const double x = (double)k + 1.0;
    const double ix = 1.0 / x;
    const double ix2 = ix * ix;
    const double ser = ix * (8.3333333333333333e-02 + ix2 * (-2.7777777777777778e-03 + ix2 * (7.9365079365079365e-04 + ix2 * -5.9523809523809524e-04)));
    return ((x - 0.5) * det_log(x) - x) + 0.9189385332046727 + ser;
}

// 1/k, k = 0..63 (k = 0 unused), correctly rounded at compile time
__device__ const double kInvK[64] = {
    0.0,       1.0 / 1,  1.0 / 2,  1.0 / 3,  1.0 / 4,  1.0 / 5,  1.0 / 6,  1.0 / 7,  1.0 / 8,  1.0 / 9,  1.0 / 10,
    1.0 / 11,  1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21,
    1.0 / 22,  1.0 / 23, 1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28, 1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32,
    1.0 / 33,  1.0 / 34, 1.0 / 35, 1.0 / 36, 1.0 / 37, 1.0 / 38, 1.0 / 39, 1.0 / 40, 1.0 / 41, 1.0 / 42, 1.0 / 43,
    1.0 / 44,  1.0 / 45, 1.0 / 46, 1.0 / 47, 1.0 / 48, 1.0 / 49, 1.0 / 50, 1.0 / 51, 1.0 / 52, 1.0 / 53, 1.0 / 54,
    1.0 / 55,  1.0 / 56, 1.0 / 57, 1.0 / 58, 1.0 / 59, 1.0 / 60, 1.0 / 61, 1.0 / 62, 1.0 / 63};

// Inversion by sequential search for 0 < lambda < 10 on the 32-bit word `w` of the group block.
__device__ __forceinline__ float poisson_small(double lambda, uint32_t w, const double* invk = kInvK)
{
    const double u = ((double)w + 0.5) * 0x1.0p-32;
    double p = det_exp_neg(lambda);
    double F = p;
    int k = 0;
    while (u >= F && k < 63) {
        k += 1;
        p = (p * lambda) * invk[k];
        F = F + p;
    }
    return (float)k;
}

__device__ __forceinline__ double u32_open(uint32_t w) { return ((double)w + 0.5) * 0x1.0p-32; }

// An integer-valued double (a floor: the count PTRS proposes) as float.  The recipe's (float)(long long)d rounds that integer to the nearest
// float, and so does (float)d -- the same value for every |d| < 2^63 -- in ONE conversion; f64 -> i64 -> f32 has no instruction of its own
// and costs ~16 per value (two per pair in every PTRS trip of phase 1: 4 % of its vector instructions).
__device__ __forceinline__ float count_as_float(double d) { return (float)d; }

constexpr uint32_t kPtrsMaxAttempts = 60000u;   // cap (never reached in practice: p ~ 0.1^k); then floor(lambda)

// ---- Hoermann PTRS for lambda >= 10 on 32-bit uniforms, split into pieces so that the one-voxel and the
// four-voxel drivers execute exactly the same arithmetic.
struct PtrsSetup {
    double b, a, bm2, vrq, bm34, ianum;
};

__device__ __forceinline__ PtrsSetup ptrs_setup(double lambda)
{
    PtrsSetup s;
    const double slam = sqrt(lambda);
    s.b = 0.931 + 2.53 * slam;
    s.a = -0.059 + 0.02483 * s.b;
    s.bm2 = s.b - 2.0;
    s.vrq = 0.9277 * s.bm2 - 3.6224;        // V <= vr  <=>  V*(b-2) <= 0.9277*(b-2) - 3.6224
    s.bm34 = s.b - 3.4;
    s.ianum = 1.1239 * s.bm34 + 1.1328;      // invalpha = ianum / bm34
    return s;
}

// One attempt's squeeze.  Returns 0: accepted (kd is the sample), 1: rejected, 2: needs the exact test.
__device__ __forceinline__ int ptrs_fast(const PtrsSetup& s, double lambda, uint32_t w0, uint32_t w1, double& us,
                                         double& V, double& kd)
{
    const double U = u32_open(w0) - 0.5;
    V = u32_open(w1);
    us = 0.5 - fabs(U);
    kd = floor((2.0 * s.a / us + s.b) * U + lambda + 0.43);
    if (us >= 0.07 && V * s.bm2 <= s.vrq) return 0;
    if (kd < 0.0 || (us < 0.013 && V > us)) return 1;
    return 2;
}

// The exact acceptance test of one attempt.
__device__ __forceinline__ bool ptrs_exact(const PtrsSetup& s, double lambda, double us, double V, double kd,
                                           double& loglam, bool& have_loglam, const double* lf = kLogFact)
{
    const double us2 = us * us;
    const double lhs = det_log((V * us2 * s.ianum) / (s.bm34 * (s.a + s.b * us2)));
    if (!have_loglam) { loglam = det_log(lambda); have_loglam = true; }
    const double rhs = (-lambda + kd * loglam) - det_lgamma_int((long long)kd, lf);
    return lhs <= rhs;
}

// Retry attempts a >= 1 until accepted: ctr = (index, stream, 2 + (a-1)/2), two attempts per block.
__device__ __forceinline__ double ptrs_retry(const PtrsSetup& s, double lambda, double& loglam, bool& have_loglam,
                                             uint32_t k0, uint32_t k1, uint32_t stream, uint64_t index)
{
    const uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32);
    Philox4 r = Philox4{0u, 0u, 0u, 0u};
    for (uint32_t attempt = 1; attempt < kPtrsMaxAttempts; ++attempt) {
        uint32_t w0, w1;
        if (((attempt - 1u) & 1u) == 0u) {
            r = philox4x32_10(c0, c1, stream, 2u + (attempt - 1u) / 2u, k0, k1);
            w0 = r.x; w1 = r.y;
        } else {
            w0 = r.z; w1 = r.w;
        }
        double us, V, kd;
        const int st = ptrs_fast(s, lambda, w0, w1, us, V, kd);
        if (st == 0) return kd;
        if (st == 2 && ptrs_exact(s, lambda, us, V, kd, loglam, have_loglam)) return kd;
    }
    return lambda;
}

// fp32 screening of the exact test.  D = lhs - rhs is re-derived in a cancellation-free form (valid for k >= 16,
// Stirling branch):  with t = (k+1-lambda)/lambda,
//     rhs = lambda*g(t) + log1p(t) - log(k+1)/2 - log(2 pi)/2 - ser(k+1),   g(t) = t - (1+t) log1p(t) = -t^2/2 + t^3/6 - ...
// so every term is O(1..50) and single precision is accurate to ~1e-5 absolute.  Returns +1 (surely accept),
// -1 (surely reject) or 0 (too close to call: run the bit-defined fp64 test).  Pure optimisation: whenever it
// answers, the answer equals the fp64 decision (margin >= 100x the error bound), so counts are unchanged.
__device__ __forceinline__ int ptrs_screen(const PtrsSetup& s, double lambda, double us, double V, double kd, const double* lf = kLogFact)
{
    const float lam = (float)lambda;
    if (kd <= 16.0) {
        // table branch of log(k!): -lambda + k log(lambda) - log(k!) has terms of magnitude <= ~100 here
        // (lambda < ~60 for such k to be proposed at all), so single precision is good to ~1e-5 absolute
        if (lambda > 64.0) return 0;
        const float usq = (float)us * (float)us;
        const float q0 = ((float)V * usq * (float)s.ianum) / ((float)s.bm34 * ((float)s.a + (float)s.b * usq));
        const float kf = (float)kd;
        const float ll = __logf(lam);
        const float d0 = __logf(q0) - ((kf * ll - lam) - (float)lf[(int)kd]);
        const float e0 = 1.0e-4f + 4.0e-6f * (kf * ll + lam);
        return d0 < -e0 ? 1 : (d0 > e0 ? -1 : 0);
    }
    const float x = (float)kd + 1.0f;
    const float t = (float)((kd + 1.0 - lambda) / lambda);          // formed in double: no cancellation error
    const float usf = (float)us;
    const float us2 = usf * usf;
    const float q = ((float)V * us2 * (float)s.ianum) / ((float)s.bm34 * ((float)s.a + (float)s.b * us2));
    const float lhs = __logf(q);
    float lg;                                                         // lambda * g(t) + log1p(t)
    if (fabsf(t) < 0.25f) {
        // g(t) = sum_{n>=2} (-1)^(n-1) t^n / (n (n-1));  log1p(t) = sum_{n>=1} (-1)^(n-1) t^n / n
        float g = 1.0f / 156.0f;                                      // n = 13
        g = fmaf(g, -t, 1.0f / 132.0f);
        g = fmaf(g, -t, 1.0f / 110.0f);
        g = fmaf(g, -t, 1.0f / 90.0f);
        g = fmaf(g, -t, 1.0f / 72.0f);
        g = fmaf(g, -t, 1.0f / 56.0f);
        g = fmaf(g, -t, 1.0f / 42.0f);
        g = fmaf(g, -t, 1.0f / 30.0f);
        g = fmaf(g, -t, 1.0f / 20.0f);
        g = fmaf(g, -t, 1.0f / 12.0f);
        g = fmaf(g, -t, 1.0f / 6.0f);
        g = fmaf(g, -t, 1.0f / 2.0f);
        g = -g * t * t;
        lg = lam * g + __logf(1.0f + t);
    } else {
        const float l1p = __logf(1.0f + t);
        lg = lam * (t - (1.0f + t) * l1p) + l1p;
    }
    const float ix = 1.0f / x;
    const float ix2 = ix * ix;
    const float ser = ix * (8.3333333e-02f + ix2 * (-2.7777778e-03f + ix2 * 7.9365079e-04f));
    const float lx = __logf(x);
    const float rhs = lg - 0.5f * lx - 0.9189385f - ser;
    const float d = lhs - rhs;
    // error budget: each term carries <= ~3 ulp of single precision relative to its own magnitude (q: 5 roundings,
    // series: fma chain, v_log_f32: 1 ulp), i.e. <= 4e-7 * (|lhs| + |lg| + lx + 1); the margin below is >= 10x that.
    const float eps = 2.0e-5f + 4.0e-6f * (fabsf(lhs) + fabsf(lg) + lx);
    if (d < -eps) return 1;
    if (d > eps) return -1;
    return 0;
}

// One attempt of one voxel from its two random words (used by the block-level work queue of
// k_extract4_noise): returns true when the voxel is resolved.
__device__ __forceinline__ bool ptrs_step_words(double lambda, uint32_t w0, uint32_t w1, float& res, const double* lf = kLogFact)
{
    const PtrsSetup s = ptrs_setup(lambda);
    double us, V, kd;
    const int st = ptrs_fast(s, lambda, w0, w1, us, V, kd);
    if (st == 1) return false;
#ifndef MVSIM_EXP_NOEXACT
    if (st == 2) {
        const int sc = ptrs_screen(s, lambda, us, V, kd, lf);
        if (sc < 0) return false;
        if (sc == 0) {
            double loglam = 0.0;
            bool have = false;
            if (!ptrs_exact(s, lambda, us, V, kd, loglam, have, lf)) return false;
        }
    }
#endif
    res = count_as_float(kd);
    return true;
}

// Random words of retry attempt a >= 1 of voxel `index`.
__device__ __forceinline__ void ptrs_retry_words(uint64_t index, uint32_t attempt, uint32_t k0, uint32_t k1, uint32_t stream,
                                                 uint32_t& w0, uint32_t& w1)
{
    const Philox4 r = philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), stream, 2u + (attempt - 1u) / 2u, k0, k1);
    const bool second = ((attempt - 1u) & 1u) != 0u;
    w0 = second ? r.z : r.x;
    w1 = second ? r.w : r.y;
}

// (w0, w1) are the attempt-0 words, taken by the caller from the block shared by the voxel pair index>>1.
__device__ __forceinline__ float poisson_ptrs(double lambda, uint32_t w0, uint32_t w1, uint32_t k0, uint32_t k1,
                                              uint32_t stream, uint64_t index)
{
    const PtrsSetup s = ptrs_setup(lambda);
    double loglam = 0.0;
    bool have_loglam = false;
    double us, V, kd;
    const int st = ptrs_fast(s, lambda, w0, w1, us, V, kd);
    if (st == 0) return (float)(long long)kd;
    if (st == 2 && ptrs_exact(s, lambda, us, V, kd, loglam, have_loglam)) return (float)(long long)kd;
    return (float)(long long)ptrs_retry(s, lambda, loglam, have_loglam, k0, k1, stream, index);
}

// Counter sampler v3 for one voxel (generic path: recomputes the shared blocks of index>>2 / index>>1).
// lambda <= 0 or NaN -> 0 (the reference's loop does not terminate there; documented deviation Q9).
__device__ __forceinline__ float poisson_counter(double lambda, uint32_t k0, uint32_t k1, uint32_t stream,
                                                 uint64_t index)
{
    if (!(lambda > 0.0)) return 0.0f;
    if (lambda < 10.0) {
        const uint64_t g = index >> 2;
        const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), stream, 0u, k0, k1);
        const uint32_t lane = (uint32_t)index & 3u;
        const uint32_t w = lane == 0 ? r.x : (lane == 1 ? r.y : (lane == 2 ? r.z : r.w));
        return poisson_small(lambda, w);
    }
    const uint64_t pr = index >> 1;
    const Philox4 r = philox4x32_10((uint32_t)pr, (uint32_t)(pr >> 32), stream, 1u, k0, k1);
    const bool odd = (index & 1) != 0;
    return poisson_ptrs(lambda, odd ? r.z : r.x, odd ? r.w : r.y, k0, k1, stream, index);
}

// Four voxels index4 .. index4+3 (index4 % 4 == 0) sharing their group / pair blocks.
__device__ __forceinline__ float4 poisson_counter4(double l0, double l1, double l2, double l3, uint32_t k0,
                                                   uint32_t k1, uint32_t stream, uint64_t index4)
{
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool s0 = l0 > 0.0 && l0 < 10.0, s1 = l1 > 0.0 && l1 < 10.0, s2 = l2 > 0.0 && l2 < 10.0,
               s3 = l3 > 0.0 && l3 < 10.0;
    if (s0 || s1 || s2 || s3) {
        const uint64_t g = index4 >> 2;
        const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), stream, 0u, k0, k1);
        if (s0) out.x = poisson_small(l0, r.x);
        if (s1) out.y = poisson_small(l1, r.y);
        if (s2) out.z = poisson_small(l2, r.z);
        if (s3) out.w = poisson_small(l3, r.w);
    }
    // Bright voxels.  Phase 1: the attempt-0 squeeze of every bright voxel of the lane (all lanes busy).
    // Phase 2: the unresolved voxels (exact test and/or retries, ~14 %) are processed one per lane per round, so
    // the expensive divergent code runs max-pending-per-lane times per wave instead of once per voxel slot.
    const bool b0 = l0 >= 10.0, b1 = l1 >= 10.0, b2 = l2 >= 10.0, b3 = l3 >= 10.0;
    uint32_t pend = 0u;                       // bit v: unresolved; bit 4+v: attempt 0 still needs its exact test
    double pus[4], pV[4], pkd[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { pus[v] = 0.0; pV[v] = 0.0; pkd[v] = 0.0; }
    if (b0 || b1) {
        const uint64_t pr = index4 >> 1;
        const Philox4 r = philox4x32_10((uint32_t)pr, (uint32_t)(pr >> 32), stream, 1u, k0, k1);
        if (b0) {
            const int st = ptrs_fast(ptrs_setup(l0), l0, r.x, r.y, pus[0], pV[0], pkd[0]);
            if (st == 0) out.x = (float)(long long)pkd[0]; else pend |= (st == 2 ? 0x11u : 0x01u);
        }
        if (b1) {
            const int st = ptrs_fast(ptrs_setup(l1), l1, r.z, r.w, pus[1], pV[1], pkd[1]);
            if (st == 0) out.y = (float)(long long)pkd[1]; else pend |= (st == 2 ? 0x22u : 0x02u);
        }
    }
    if (b2 || b3) {
        const uint64_t pr = (index4 >> 1) + 1;
        const Philox4 r = philox4x32_10((uint32_t)pr, (uint32_t)(pr >> 32), stream, 1u, k0, k1);
        if (b2) {
            const int st = ptrs_fast(ptrs_setup(l2), l2, r.x, r.y, pus[2], pV[2], pkd[2]);
            if (st == 0) out.z = (float)(long long)pkd[2]; else pend |= (st == 2 ? 0x44u : 0x04u);
        }
        if (b3) {
            const int st = ptrs_fast(ptrs_setup(l3), l3, r.z, r.w, pus[3], pV[3], pkd[3]);
            if (st == 0) out.w = (float)(long long)pkd[3]; else pend |= (st == 2 ? 0x88u : 0x08u);
        }
    }
    while (pend & 0xFu) {
        const int v = __ffs((int)(pend & 0xFu)) - 1;
        const double lam = v == 0 ? l0 : (v == 1 ? l1 : (v == 2 ? l2 : l3));
        const double us = v == 0 ? pus[0] : (v == 1 ? pus[1] : (v == 2 ? pus[2] : pus[3]));
        const double V = v == 0 ? pV[0] : (v == 1 ? pV[1] : (v == 2 ? pV[2] : pV[3]));
        const double kd0 = v == 0 ? pkd[0] : (v == 1 ? pkd[1] : (v == 2 ? pkd[2] : pkd[3]));
        const bool exact0 = ((pend >> (4 + v)) & 1u) != 0u;
        const PtrsSetup s = ptrs_setup(lam);
        double loglam = 0.0;
        bool have_loglam = false;
        double res;
        if (exact0 && ptrs_exact(s, lam, us, V, kd0, loglam, have_loglam)) res = kd0;
        else res = ptrs_retry(s, lam, loglam, have_loglam, k0, k1, stream, index4 + (uint64_t)v);
        const float fr = (float)(long long)res;
        if (v == 0) out.x = fr; else if (v == 1) out.y = fr; else if (v == 2) out.z = fr; else out.w = fr;
        pend &= ~(0x11u << v);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------------------
// Phase 1 of the production sampler, one wave at a time (used by k_extract4_noise and by the adjust + Poisson
// epilogue of the convolution's last pass).  Every lane brings 4 consecutive voxels (one Philox group).  What is
// cheap is decided here, everything else becomes a work item for k_poisson_resolve:
//   * lambda <= 0 / NaN                       -> 0
//   * 0 < lambda < 10 (inversion)             -> the "count is 0" shortcut; a possible count >= 1 is queued
//   * lambda >= 10 (PTRS)                     -> the bright voxel PAIRS of the wave (a pair shares the Philox block of its
//                                                attempt 0) are COMPACTED into a wave-private LDS list (ballot + prefix
//                                                count) and the attempt-0 squeeze then runs one pair per lane with all lanes
//                                                busy; what the squeeze does not accept is queued
// Same arithmetic per (voxel, attempt) as poisson_counter: bit-identical counts.  Three things are pure execution
// shortcuts that cannot change a decision:
//   - the regime (small / bright) is read off an fp32 product with a guard band; inside the band the fp64 product decides;
//   - the shortcut of the inversion regime, "u < exp(-lambda)" whenever u < 1 - lambda - 1e-12, is tested in fp32 with a
//     margin (1e-6) that covers every fp32 rounding involved, so it fires only where the fp64 statement holds;
//   - the squeeze of PTRS attempt 0 is evaluated in fp32 (ptrs_squeeze_f32) and only trusted with guard bands; whatever
//     it cannot certify goes to the resolver as "attempt 0 not evaluated", where the fp64 recipe runs.
// ------------------------------------------------------------------------------------------------------------
// 16 bytes per work item (round 3: 32): the RNG counter of the voxel follows from its output position (plane arithmetic, needed
// for retries only), the first attempt to evaluate is 0 for every bright item and implied by the segment end for inversion items.
struct __attribute__((aligned(16))) PItem {
    unsigned int out;             // element index in the output (< 2^32: larger outputs take the queue-less kernel)
    float v;                      // adjusted voxel value (lambda = v * mul)
    unsigned int w0, w1;          // random words of attempt 0 (inversion item: w0 = the voxel's word of its group block)
};

// wave-private LDS scratch of phase 1 (1.1 KB: a block of four waves fits beside the two resident 75 KB blocks of the
// convolution's y passes, which is what lets the hardware run the sampler on a second stream beside them -- DESIGN 4.5)
struct __attribute__((aligned(16))) P1Scratch {
    float vin[256];               // the slot's voxel values, [4 * lane + component]; the bright pairs' counts return in place
    unsigned char plist[128];     // bright pairs: 2 * lane + half
};

struct P1Args {                   // wave-uniform
    double mul;
    float mulf;
    uint32_t k0, k1, stream;
    PItem* seg;                   // this block's queue segment
    unsigned int segcap;          // items it holds: all of the block's voxels, or a share of them (poisson_queue_share)
    unsigned long long* ctr;      // LDS append counter of the block: low word bright items (front), high word inversion items (back)
    unsigned int* ovf;            // LDS: appends refused because the segment was full ([0] bright, [1] inversion)
};

// The count array in front of a queue's segments: per block three words -- items at the front, items at the back, voxels that found
// the segment full -- for up to POISSON_MAX_BLOCKS blocks, then a header the first block of phase 1 writes: {blocks, items per segment}
// (what mvsim_get_queue_stats reads) and the resolver completes: {.., .., 1 if any block refused a voxel} (what k_poisson_refused reads).
constexpr int QCOUNT_WORDS = 3;
constexpr int POISSON_MAX_BLOCKS = 256 * 64;
constexpr int QCOUNT_HEADER = QCOUNT_WORDS * POISSON_MAX_BLOCKS;        // word index of the header
// What phase 1 leaves in the output of a voxel its full segment refused: the NEGATED voxel value (v > 0 for every voxel that has
// anything to sample, counts are >= 0: a negative output cannot be a result).  k_poisson_refused walks the voxels of every block
// whose third count word is non-zero once more and samples those voxels where they stand (resolve_refused).

// One LDS atomic hands out the slot AND says whether the segment still has room.  Both words only grow, so once
// "front + back < segcap" fails for an append it fails for every later one: the accepted fronts and backs are prefixes
// [0, F) and (segcap - 1 - B, segcap - 1], and two accepted appends never meet (the later one saw the earlier one's word).
// CHECKED == false: the segment holds every voxel of the block, nothing can be refused, and the append is the one 32-bit LDS atomic on
// its half of the counter that it has always been (the check costs phase 1 ~310 vector instructions per wave, 11 %: the default queue
// is the full one).
template <bool CHECKED>
__device__ __forceinline__ bool p1_slot(const P1Args& a, bool back, unsigned int& pos)
{
    if (!CHECKED) {
        const unsigned int old = atomicAdd(reinterpret_cast<unsigned int*>(a.ctr) + (back ? 1 : 0), 1u);
        pos = back ? a.segcap - 1u - old : old;
        return true;
    }
    const unsigned long long old = atomicAdd(a.ctr, back ? (1ull << 32) : 1ull);
    const unsigned int f = (unsigned int)old, b = (unsigned int)(old >> 32);
    if (f + b >= a.segcap) {
        atomicAdd(&a.ovf[back ? 1 : 0], 1u);
        return false;
    }
    pos = back ? a.segcap - 1u - b : f;
    return true;
}

// end of a block's phase 1: what the resolver will find in the segment, and how many voxels were sampled in place
__device__ __forceinline__ void p1_publish_header(unsigned int* qcount_base, unsigned int blocks, unsigned int segcap)
{
    qcount_base[QCOUNT_HEADER] = blocks;
    qcount_base[QCOUNT_HEADER + 1] = segcap;
    qcount_base[QCOUNT_HEADER + 2] = 0u;                    // set by the resolver, the kernel after this one
}

__device__ __forceinline__ void p1_publish(unsigned int* qcount, unsigned long long ctr, const unsigned int* ovf)
{
    qcount[0] = (unsigned int)ctr - ovf[0];
    qcount[1] = (unsigned int)(ctr >> 32) - ovf[1];
    qcount[2] = ovf[0] + ovf[1];
}

__device__ __forceinline__ void p1_wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// regime of one voxel: 0: lambda <= 0 or NaN, 1: inversion, 2: PTRS
__device__ __forceinline__ int p1_class(float v, const P1Args& a)
{
    const float lf = v * a.mulf;
    if (!(v > 0.f) || !(a.mul > 0.0)) return 0;
    if (lf < 9.99f) return 1;
    if (lf > 10.01f) return 2;
    return ((double)v * a.mul >= 10.0) ? 2 : 1;
}

// The squeeze of PTRS attempt 0 (ptrs_setup + ptrs_fast, outcome 0) in single precision.  Returns true only when the
// fp64 recipe is CERTAIN to accept at the squeeze with the same k:
//   b, a:  v_sqrt_f32 / constants / two roundings          -> relative error <= 4e-7 (b), 1e-6 (a)
//   U, us: float(int32(w0 - 2^31)) * 2^-32                  -> absolute error <= 6e-8, so us >= 0.070001f implies us >= 0.07
//   t = 2a/us + b (v_rcp_f32, 1 ulp)                         -> relative error <= 2.3e-6
//   x = t U + 0.43                                           -> absolute error <= 1.1e-6 t + 5e-8
//   k = floor(lambda + x) with lambda in fp64: certain when the fraction stays 4e-6 t + 1e-5 away from an integer (3.6 x the bound);
//   V (b - 2) <= 0.9277 (b - 2) - 3.6224: both sides carry <= 1.2e-6 b; trusted with a margin of 4e-6 b + 1e-5.
// Everything else (about a quarter of the bright voxels: the squeeze rejects them anyway) is left to the resolver.
__device__ __forceinline__ bool ptrs_squeeze_f32(double lambda, uint32_t w0, uint32_t w1, float& res)
{
    const float slam = __builtin_amdgcn_sqrtf((float)lambda);
    const float b = 0.931f + 2.53f * slam;
    const float a = -0.059f + 0.02483f * b;
    const float U = (float)(int)(w0 ^ 0x80000000u) * 0x1.0p-32f;
    const float V = (float)w1 * 0x1.0p-32f;
    const float us = 0.5f - fabsf(U);
    const float bm2 = b - 2.0f;
    const float t = (2.0f * a) * __builtin_amdgcn_rcpf(us) + b;
    const float x = t * U + 0.43f;
    const double y = lambda + (double)x;
    const double fl = floor(y);
    const float frac = (float)(y - fl);
    const float g = 4.0e-6f * t + 1.0e-5f;
    res = count_as_float(fl);
    return us >= 0.070001f && V * bm2 <= (0.9277f * bm2 - 3.6224f) - (4.0e-6f * b + 1.0e-5f) && frac >= g && frac <= 1.0f - g &&
           lambda < 1.0e9;
}

__device__ __forceinline__ void p1_push(PItem* slot, unsigned long long out, float v, unsigned int w0, unsigned int w1)
{
    *reinterpret_cast<uint4*>(slot) = make_uint4((unsigned int)out, __float_as_uint(v), w0, w1);     // one 16-byte store
}

// (MVSIM_EXP_NOPHILOX / _NOSMALLPUSH / _NOBRIGHT: instruction-attribution builds of tools/attribute_valu.sh -- each removes one
// part of the work and with it the correctness of the counts; never defined in the product build.)
template <bool CHECKED>
__device__ __forceinline__ void poisson_phase1(const float vv[4], bool valid, unsigned long long index4, unsigned long long out4,
                                               const P1Args& a, P1Scratch* ws, int lane, float ov[4])
{
    int cls[4];
    bool small_any = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cls[c] = valid ? p1_class(vv[c], a) : 0;
        small_any |= cls[c] == 1;
        ov[c] = 0.f;
    }
    if (small_any) {
        const unsigned long long g = index4 >> 2;
#ifdef MVSIM_EXP_NOPHILOX
        const Philox4 r = Philox4{(uint32_t)g * 2654435761u, (uint32_t)g * 40503u + 77u, (uint32_t)g ^ 0x9E3779B9u, (uint32_t)(g >> 3)};
#else
        const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), a.stream, 0u, a.k0, a.k1);
#endif
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cls[c] == 1) {
                // count = 0  <=>  u = (w + 0.5) 2^-32 < exp(-lambda).  Sufficient, in integers: w < thr with
                // thr = floor((1 - lf - 2e-6) 2^32): lf is lambda to 1.2e-7 and the two fp32 subtractions round by <= 1.2e-7,
                // so w < thr implies u < 1 - lambda - 1.7e-6 < exp(-lambda).  (lf >= 1: thr = 0, never fires.)
                const float tf = (1.0f - vv[c] * a.mulf) - 2.0e-6f;
                const uint32_t thr = (uint32_t)(fmaxf(tf, 0.f) * 4294967296.0f);
#ifndef MVSIM_EXP_NOSMALLPUSH
                if (!(w[c] < thr)) {
                    unsigned int pos;
                    if (p1_slot<CHECKED>(a, true, pos)) p1_push(a.seg + pos, out4 + (unsigned long long)c, vv[c], w[c], 0u);
                    else ov[c] = -vv[c];
                }
#else
                if (!(w[c] < thr)) ov[c] = 1.f;
#endif
            }
    }
    // bright pairs of the wave -> dense list
    unsigned int nb = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bool b = cls[2 * h] == 2 || cls[2 * h + 1] == 2;
        const unsigned long long m = __ballot(b);
        if (m != 0ull) {
            const unsigned int pre = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (b) ws->plist[nb + pre] = (unsigned char)(2u * (unsigned int)lane + (unsigned int)h);
            nb += (unsigned int)__popcll(m);
        }
    }
#ifdef MVSIM_EXP_NOBRIGHT
    nb = 0u;
#endif
    if (nb == 0u) return;                                   // wave-uniform
    *reinterpret_cast<float4*>(&ws->vin[4 * lane]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    // a pair's owner lane is found through the list; what it knows about its voxels (RNG counter, output position) comes over
    // the lane crossbar instead of through LDS: the counters of a slot lie within 2^32 of lane 0's, the outputs are contiguous
    const unsigned long long index_base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)(index4 >> 32)) << 32) |
                                          (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)index4);
    const unsigned long long out_base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)(out4 >> 32)) << 32) |
                                        (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned int)out4);
    const unsigned int index_delta = (unsigned int)(index4 - index_base);
    p1_wave_order();
    for (unsigned int i0 = 0u; i0 < nb; i0 += 64u) {        // wave-uniform trip count: the crossbar reads need every lane
        const unsigned int i = i0 + (unsigned int)lane;
        const bool act = i < nb;
        const unsigned int pid = act ? (unsigned int)ws->plist[i] : 0u;
        const unsigned int owner = pid >> 1, first = (pid & 1u) * 2u;
        const unsigned int od = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)index_delta);
        if (!act) continue;
        const float2 pv = *reinterpret_cast<const float2*>(&ws->vin[4u * owner + first]);
        const unsigned long long idx = index_base + od + first;          // even: the pair's attempt-0 block is idx >> 1
        const unsigned long long pr = idx >> 1;
        const Philox4 r = philox4x32_10((uint32_t)pr, (uint32_t)(pr >> 32), a.stream, 1u, a.k0, a.k1);
        float2 res = make_float2(0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = e ? pv.y : pv.x;
            if (p1_class(v, a) == 2) {
                const uint32_t w0 = e ? r.z : r.x, w1 = e ? r.w : r.y;
                float k;
#ifdef MVSIM_P1_F64
                const double lam = (double)v * a.mul;
                double us, V, kd;
                const bool ok = ptrs_fast(ptrs_setup(lam), lam, w0, w1, us, V, kd) == 0;
                k = (float)(long long)kd;
#else
                const bool ok = ptrs_squeeze_f32((double)v * a.mul, w0, w1, k);
#endif
                if (ok) {
                    if (e) res.y = k; else res.x = k;
                } else {
                    // (Measured: one LDS atomic per lane is cheaper here than a ballot-aggregated append.)
                    unsigned int pos;
                    if (p1_slot<CHECKED>(a, false, pos)) p1_push(a.seg + pos, out_base + 4u * owner + first + (unsigned long long)e, v, w0, w1);
                    else if (e) res.y = -v;                   // refused: the resolver finds it by its sign
                    else res.x = -v;
                }
            }
        }
        *reinterpret_cast<float2*>(&ws->vin[4u * owner + first]) = res;  // this lane alone reads and writes the pair
    }
    p1_wave_order();
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (cls[c] == 2) ov[c] = ws->vin[4 * lane + c];
    p1_wave_order();                                        // the scratch is reused by the wave's next slot
}


__device__ __forceinline__ float adjust_one(float v, double corr, float min_value)
{
    const float t = (float)((double)v * corr);  // pass 1, Tools.java:150-151
    return t + min_value;                       // pass 2, Tools.java:154-155 (second rounding, Q6)
}

struct ResolveJob {
    float*              out;
    const PItem*        queue;
    const unsigned int* qcount;
    unsigned int        segcap;
    double              mul;
    uint32_t            k0, k1, stream;
    // RNG counter of output element o (retries only): k = o / plane; index = index_offset + k * idx_inc * plane + (o - k * plane)
    unsigned int        plane;
    unsigned int        idx_inc;
    unsigned long long  index_offset;
    // how phase 1 walked the volume, for the voxels a full segment refused: 0 = its segments hold every voxel of their blocks
    // (nothing is ever refused: the fused tail, share 16), 1 = k_extract4_noise2 (walk_n = float4 groups), 2 = k_extract_noise2_any
    // (walk_n = wave slots, walk_spp = slots per plane)
    int                 walk;
    long long           walk_n;
    long long           walk_spp;
};

// One voxel sampled where it stands, attempt by attempt from its own random words: what poisson_counter computes, in the
// resolver's formulation.
__device__ __forceinline__ float resolve_in_place(float v, const ResolveJob& j, unsigned long long index)
{
    const double lam = (double)v * j.mul;
    if (lam < 10.0) {
        const unsigned long long g = index >> 2;
        const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), j.stream, 0u, j.k0, j.k1);
        const uint32_t c = (uint32_t)index & 3u;
        return poisson_small(lam, c == 0u ? r.x : (c == 1u ? r.y : (c == 2u ? r.z : r.w)));
    }
    const unsigned long long pr = index >> 1;
    const Philox4 r = philox4x32_10((uint32_t)pr, (uint32_t)(pr >> 32), j.stream, 1u, j.k0, j.k1);
    uint32_t w0 = (index & 1ull) ? r.z : r.x, w1 = (index & 1ull) ? r.w : r.y;
    float val = 0.f;
#pragma unroll 1
    for (uint32_t att = 0u;; ++att) {
        if (att >= kPtrsMaxAttempts) return (float)(long long)lam;
        if (att != 0u) ptrs_retry_words(index, att, j.k0, j.k1, j.stream, w0, w1);
        if (ptrs_step_words(lam, w0, w1, val)) return val;
    }
}

// RNG counter of output element o (see ResolveJob)
__device__ __forceinline__ unsigned long long resolve_index_of(const ResolveJob& j, unsigned long long o)
{
    const unsigned long long kpl = j.idx_inc == 1u ? 0ull : o / j.plane;
    return j.index_offset + o + kpl * (unsigned long long)(j.idx_inc - 1u) * j.plane;
}

// The voxels phase-1 block `segment` was refused a queue slot for (its segment, sized for a share of its voxels, was full): the same
// walk as that block's, one trip of it (`trip`: 1024 candidate voxels), looking for the negated values it left behind.  Their output
// positions go to the block's LDS list (k_poisson_refused samples them from there, every lane busy).  Returns false past the walk's end.
__device__ __forceinline__ bool refused_collect(const ResolveJob& j, long long segment, long long trip, int t, long long grid,
                                                unsigned int* list, unsigned int* count)
{
    const bool vec = j.walk == 1;
    const long long first = vec ? segment * 256 + trip * grid * 256 : segment * 4 + trip * grid * 4;   // block-uniform
    if (first >= j.walk_n) return false;
    const long long u = first + (vec ? t : (t >> 6));
    if (u >= j.walk_n) return true;
    long long p0 = 4 * u, lo = 0, hi = 4;                                // output position of component 0, valid components [lo, hi)
    if (!vec) {
        const long long k = u / j.walk_spp, jb = u - k * j.walk_spp;
        const unsigned long long ibase = j.index_offset + (unsigned long long)(k * j.idx_inc) * (unsigned long long)j.plane;
        const unsigned long long g = (ibase >> 2) + (unsigned long long)(jb * 64 + (t & 63));
        const long long i0 = (long long)(4ull * g - ibase);
        lo = i0 < 0 ? -i0 : 0;
        hi = (long long)j.plane - i0 < 4 ? (long long)j.plane - i0 : 4;
        p0 = k * (long long)j.plane + i0;
    }
    for (long long c = lo; c < hi; ++c)
        if (j.out[p0 + c] < 0.f) list[atomicAdd(count, 1u)] = (unsigned int)(p0 + c);
    return true;
}

constexpr int RESOLVE_TAB = 17 + 64;                         // doubles of LDS: log(k!) for k <= 16, then 1 / k for k < 64

// One queue segment resolved by the 256 lanes of a block (k_poisson_resolve: kernels.hip).
__device__ __forceinline__ void resolve_segment_body(const ResolveJob& j, long long segment, int t, unsigned int* ticket, double* tab)
{
    const unsigned int n = j.qcount[QCOUNT_WORDS * segment], ns = j.qcount[QCOUNT_WORDS * segment + 1];
    // a block that was refused queue slots says so in the header: k_poisson_refused, the next kernel, reads that one word
    if (t == 0 && j.walk != 0 && j.qcount[QCOUNT_WORDS * segment + 2] != 0u) const_cast<unsigned int*>(j.qcount)[QCOUNT_HEADER + 2] = 1u;
    // the two small tables the recipe indexes, from LDS: a dependent GLOBAL load inside the divergent attempt costs every lane of the
    // wave a trip to the cache
    if (t < 17) tab[t] = kLogFact[t];
    else if (t < RESOLVE_TAB) tab[t] = kInvK[t - 17];
    if (t == 0) *ticket = 256u;
    __syncthreads();
    const double* lf = tab;
    const double* invk = tab + 17;
    const PItem* __restrict__ seg = j.queue + (unsigned long long)segment * j.segcap;
    // inversion items (0 < lambda < 10 that the shortcut of phase 1 could not settle), from the back
    for (unsigned int i = (unsigned int)t; i < ns; i += 256u) {
        const PItem it = seg[j.segcap - 1u - i];
        j.out[it.out] = poisson_small((double)it.v * j.mul, it.w0, invk);     // w0: the voxel's word of its group block
    }
    // PTRS items.  A lane works on ONE ATTEMPT per trip and, the moment its item is resolved, takes the next item of the
    // segment (LDS ticket): every trip has every lane on a live attempt, instead of the wave idling until its unluckiest
    // item -- retries come in geometrically distributed numbers -- has been accepted.  The loop ends when the tickets run
    // out: each attempt succeeds with probability > 0.6 and the attempt count is capped, so every lane gets there.
    unsigned int i = (unsigned int)t;
    PItem it;
    it.out = 0u; it.v = 0.f; it.w0 = 0u; it.w1 = 0u;
    bool have = i < n;
    if (have) it = seg[i];
    // The words of the attempt a lane is about to evaluate always sit in it.w0 / it.w1: phase 1's words for attempt 0, and for a retry the
    // words drawn at the END of the failed attempt.  (Rounds 2-5 chose between the two at the top of the loop, by `a == 0`; the compiler
    // threaded that test through the back edges, and what came out was an outer loop over items with an INNER loop over the retries of
    // one item -- lanes whose item was accepted waited for the wave's unluckiest item before any of them took a new one, the very thing
    // the ticket is there to avoid.  Found in the ISA in round 6: 1 706 -> 1 366 vector instructions per wave, the stage 0.510 -> 0.487 ms,
    // profiles/r06_resolver_ab.txt.  Measured on top of it and not kept: several segments per block as one list (fewer, longer-lived
    // blocks: fewer instructions, more waiting), the next item requested one attempt ahead (held items are items no other lane can take).)
    uint32_t a = 0u;
    if (have) do {
        const double lam = (double)it.v * j.mul;
        float val = 0.f;
        bool done;
        if (a >= kPtrsMaxAttempts) {
            val = (float)(long long)lam;
            done = true;
        } else {
            done = ptrs_step_words(lam, it.w0, it.w1, val, lf);
        }
        if (done) {
            j.out[it.out] = val;
            a = 0u;
            i = atomicAdd(ticket, 1u);
            have = i < n;
            if (have) it = seg[i];
        } else {
            a += 1u;
            const unsigned int kpl = j.idx_inc == 1u ? 0u : it.out / j.plane;
            const unsigned long long index = j.index_offset + (unsigned long long)it.out + (unsigned long long)kpl * (j.idx_inc - 1u) * j.plane;
            ptrs_retry_words(index, a, j.k0, j.k1, j.stream, it.w0, it.w1);
        }
        // ONE latch: a convergent no-op the optimiser may not duplicate, so that the retry path cannot get a back edge of its own
        // (on it `have` is known to be true, and the loop would be split into the nested form described above again)
        __builtin_amdgcn_wave_barrier();
    } while (have);
}

}  // namespace mvsim
