/*
 * mvsim_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See mvsim_oracle.h for scope, citations and the "PARITY UNPINNED" statement.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no -ffast-math: the
 * rounding points below are part of the specification being restated).
 */
#include "mvsim_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* java.util.Random -- JDK specification (48-bit LCG).                        */
/* ------------------------------------------------------------------------- */
#define JR_MULT 0x5DEECE66DULL
#define JR_ADD  0xBULL
#define JR_MASK ((1ULL << 48) - 1)

void orc_jrandom_seed(orc_jrandom* r, int64_t seed) { r->s = ((uint64_t)seed ^ JR_MULT) & JR_MASK; }

int32_t orc_jrandom_next(orc_jrandom* r, int bits)
{
    r->s = (r->s * JR_MULT + JR_ADD) & JR_MASK;
    /* (int)(seed >>> (48 - bits)) : arithmetic value is the low 32 bits, signed */
    return (int32_t)(uint32_t)(r->s >> (48 - bits));
}

int32_t orc_jrandom_next_int(orc_jrandom* r) { return orc_jrandom_next(r, 32); }

int32_t orc_jrandom_next_int_bound(orc_jrandom* r, int32_t bound)
{
    int32_t rr = orc_jrandom_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0)
        return (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    for (int32_t u = rr; ; u = orc_jrandom_next(r, 31)) {
        rr = u % bound;
        /* while (u - r + m < 0) with int overflow semantics */
        int32_t t = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
        if (t >= 0) break;
    }
    return rr;
}

int64_t orc_jrandom_next_long(orc_jrandom* r)
{
    int64_t hi = (int64_t)orc_jrandom_next(r, 32);
    int64_t lo = (int64_t)orc_jrandom_next(r, 32);
    return (int64_t)((uint64_t)hi << 32) + lo;
}

double orc_jrandom_next_double(orc_jrandom* r)
{
    int64_t a = (int64_t)orc_jrandom_next(r, 26);
    int64_t b = (int64_t)orc_jrandom_next(r, 27);
    return (double)((a << 27) + b) * 0x1.0p-53;
}

/* uncommons/PoissonGenerator.java:95-109: count exponential inter-arrival
 * times until their sum exceeds one unit of time. */
int32_t orc_poisson_interarrival(orc_jrandom* r, double mean)
{
    int32_t x = 0;
    double t = 0.0;
    for (;;) {
        t -= log(orc_jrandom_next_double(r)) / mean;
        if (t > 1.0) break;
        ++x;
    }
    return x;
}

double orc_poisson_mul(double snr) { return pow(snr / sqrt(5.0), 2.0); } /* Tools:76 */

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11; Random123).                           */
/* ------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Deterministic elementary functions: only IEEE +,-,*,/ and bit moves, so the
 * HIP kernel's own implementation of the same recipe gives identical bits.
 * Recipes follow the classic fdlibm argument reductions. */
static double bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

double orc_det_log(double x)
{
    if (!(x > 0.0)) return -1.0e300;              /* log(0) / invalid: "minus huge" */
    uint64_t u = d2bits(x);
    int e = (int)(u >> 52) - 1023;
    if (e == -1023) {                              /* subnormal: scale up */
        x = x * 0x1.0p54; u = d2bits(x); e = (int)(u >> 52) - 1023 - 54;
    }
    u = (u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m = bits2d(u);                          /* [1,2) */
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

double orc_det_exp(double x)
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
    int k = (int)kf;
    /* scale by 2^k in two exact steps so subnormal results stay well defined */
    if (k < -1000) return (y * bits2d((uint64_t)(k + 1000 + 1023) << 52)) * 0x1.0p-1000;
    return y * bits2d((uint64_t)(k + 1023) << 52);
}

static const double ORC_LOGFACT[17] = {
    0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479458,
    4.787491742782046, 6.579251212010101, 8.525161361065415, 10.60460290274525,
    12.801827480081469, 15.104412573075516, 17.502307845873887, 19.987214495661885,
    22.552163853123425, 25.19122118273868, 27.89927138384089, 30.671860106080672 };

double orc_det_lgamma_int(int64_t k)
{
    if (k <= 16) return ORC_LOGFACT[k < 0 ? 0 : k];
    const double x = (double)k + 1.0;
    const double ix = 1.0 / x;
    const double ix2 = ix * ix;
    /* Stirling series for lgamma(x) */
    const double ser = ix * (8.3333333333333333e-02 + ix2 * (-2.7777777777777778e-03 + ix2 * (7.9365079365079365e-04 + ix2 * -5.9523809523809524e-04)));
    return ((x - 0.5) * orc_det_log(x) - x) + 0.9189385332046727 + ser;
}

/* exp(-lambda) for 0 < lambda < 10, division free: 2^k * sum_{n<=13} r^n/n! by Horner with
 * explicit fused multiply-adds (correctly rounded on every platform). */
double orc_det_exp_neg(double lambda)
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
    return p * bits2d((uint64_t)((int)kf + 1023) << 52);
}

/* O(1) Poisson sampler keyed by (seed, stream, index)  -- "counter sampler v3".
 *   lambda <= 0 or NaN : 0 (Q9: the reference loops forever there)
 *   lambda < 10        : inversion by sequential search on a 32-bit uniform; the four voxels
 *                        index>>2 share one Philox block, ctr = (index>>2, stream, 0), voxel
 *                        index&3 takes word index&3; p_k = p_{k-1} * lambda * (1/k)
 *   lambda >= 10       : Hoermann's PTRS transformed rejection on 32-bit uniforms (U, V):
 *                        attempt 0 : block ctr = (index>>1, stream, 1), words 2*(index&1), +1
 *                        attempt a>=1 : block ctr = (index, stream, 2 + (a-1)/2), words 2*((a-1)&1), +1
 *                        divisions folded out of the squeeze, logs merged in the exact test. */
static double u32_open(uint32_t w) { return ((double)w + 0.5) * 0x1.0p-32; }

int64_t orc_poisson_counter(double lambda, uint64_t seed, uint32_t stream, uint64_t index)
{
    if (!(lambda > 0.0)) return 0;
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t r[4];
    if (lambda < 10.0) {
        const uint64_t g = index >> 2;
        uint32_t ctr[4] = { (uint32_t)g, (uint32_t)(g >> 32), stream, 0u };
        orc_philox4x32_10(ctr, key, r);
        const double u = u32_open(r[index & 3]);
        double p = orc_det_exp_neg(lambda);
        double F = p;
        int64_t k = 0;
        while (u >= F && k < 63) {
            k += 1;
            p = (p * lambda) * (1.0 / (double)k);
            F = F + p;
        }
        return k;
    }
    const double slam = sqrt(lambda);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double bm2 = b - 2.0;
    const double vrq = 0.9277 * bm2 - 3.6224;           /* V <= vr  <=>  V*(b-2) <= 0.9277*(b-2) - 3.6224 */
    const double bm34 = b - 3.4;
    const double ianum = 1.1239 * bm34 + 1.1328;         /* invalpha = ianum / bm34 */
    double loglam = 0.0;
    int have_loglam = 0;
    for (uint32_t attempt = 0; attempt < 60000u; ++attempt) {   /* cap: never reached in practice (p ~ 0.1^k) */
        uint32_t w0, w1;
        if (attempt == 0) {
            const uint64_t pr = index >> 1;
            uint32_t ctr[4] = { (uint32_t)pr, (uint32_t)(pr >> 32), stream, 1u };
            orc_philox4x32_10(ctr, key, r);
            w0 = r[2 * (index & 1)]; w1 = r[2 * (index & 1) + 1];
        } else {
            if ((attempt - 1) % 2 == 0) {
                uint32_t ctr[4] = { (uint32_t)index, (uint32_t)(index >> 32), stream, 2u + (attempt - 1) / 2 };
                orc_philox4x32_10(ctr, key, r);
            }
            w0 = r[2 * ((attempt - 1) & 1)]; w1 = r[2 * ((attempt - 1) & 1) + 1];
        }
        const double U = u32_open(w0) - 0.5;
        const double V = u32_open(w1);
        const double us = 0.5 - fabs(U);
        const double kd = floor((2.0 * a / us + b) * U + lambda + 0.43);
        if (us >= 0.07 && V * bm2 <= vrq) return (int64_t)kd;
        if (kd < 0.0 || (us < 0.013 && V > us)) continue;
        const int64_t k = (int64_t)kd;
        const double us2 = us * us;
        const double lhs = orc_det_log((V * us2 * ianum) / (bm34 * (a + b * us2)));
        if (!have_loglam) { loglam = orc_det_log(lambda); have_loglam = 1; }
        const double rhs = (-lambda + kd * loglam) - orc_det_lgamma_int(k);
        if (lhs <= rhs) return k;
    }
    return (int64_t)lambda;
}

/* ------------------------------------------------------------------------- */
/* Affine model (mpicbg.models.AffineModel3D semantics, row-major 3x4).        */
/* ------------------------------------------------------------------------- */
static void affine_set_identity(double m[12])
{
    memset(m, 0, 12 * sizeof(double));
    m[0] = m[5] = m[10] = 1.0;
}

/* a <- b o a  (AffineModel3D.preConcatenate) */
static void affine_pre_concatenate(double a[12], const double b[12])
{
    double r[12];
    for (int i = 0; i < 3; ++i) {
        const double b0 = b[4 * i], b1 = b[4 * i + 1], b2 = b[4 * i + 2], b3 = b[4 * i + 3];
        r[4 * i + 0] = b0 * a[0] + b1 * a[4] + b2 * a[8];
        r[4 * i + 1] = b0 * a[1] + b1 * a[5] + b2 * a[9];
        r[4 * i + 2] = b0 * a[2] + b1 * a[6] + b2 * a[10];
        r[4 * i + 3] = b0 * a[3] + b1 * a[7] + b2 * a[11] + b3;
    }
    memcpy(a, r, sizeof(r));
}

/* SMVD:80-102 */
void orc_axis_rotation(const int64_t dim[3], int axis, int degrees, double m[12])
{
    /* Q2: (max - min) / 2 in long arithmetic; zero-min interval => max = dim-1 */
    const double c0 = (double)((dim[0] - 1) / 2);
    const double c1 = (double)((dim[1] - 1) / 2);
    const double c2 = (double)((dim[2] - 1) / 2);
    double t1[12], rot[12], t2[12];
    affine_set_identity(t1); t1[3] = -c0; t1[7] = -c1; t1[11] = -c2;
    affine_set_identity(t2); t2[3] = c0; t2[7] = c1; t2[11] = c2;

    /* Q3: (float)Math.toRadians(degrees), then cos/sin in double */
    const double theta = (double)(float)((double)degrees * 0.017453292519943295);
    const double dc = cos(theta), ds = sin(theta);
    double dR[12];
    affine_set_identity(dR);
    if (axis == 0)      { dR[5] = dc; dR[6] = -ds; dR[9] = ds;  dR[10] = dc; }
    else if (axis == 1) { dR[0] = dc; dR[2] = ds;  dR[8] = -ds; dR[10] = dc; }
    else                { dR[0] = dc; dR[1] = -ds; dR[4] = ds;  dR[5] = dc; }
    affine_set_identity(rot);
    affine_pre_concatenate(rot, dR);              /* rot.rotate(axis, theta) */

    affine_pre_concatenate(t1, rot);              /* SMVD:98 */
    affine_pre_concatenate(t1, t2);               /* SMVD:99 */
    memcpy(m, t1, 12 * sizeof(double));
}

/* AffineModel3D.createInverse(): adjugate / determinant in double */
void orc_affine_invert(const double m[12], double v[12])
{
    const double m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    const double m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const double m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    const double det = m00 * m11 * m22 + m10 * m21 * m02 + m20 * m01 * m12
                     - m02 * m11 * m20 - m12 * m21 * m00 - m22 * m01 * m10;
    v[0] = (m11 * m22 - m12 * m21) / det;
    v[1] = (m02 * m21 - m01 * m22) / det;
    v[2] = (m01 * m12 - m02 * m11) / det;
    v[4] = (m12 * m20 - m10 * m22) / det;
    v[5] = (m00 * m22 - m02 * m20) / det;
    v[6] = (m02 * m10 - m00 * m12) / det;
    v[8] = (m10 * m21 - m11 * m20) / det;
    v[9] = (m01 * m20 - m00 * m21) / det;
    v[10] = (m00 * m11 - m01 * m10) / det;
    v[3]  = -v[0] * m03 - v[1] * m13 - v[2] * m23;
    v[7]  = -v[4] * m03 - v[5] * m13 - v[6] * m23;
    v[11] = -v[8] * m03 - v[9] * m13 - v[10] * m23;
}

/* ------------------------------------------------------------------------- */
/* ImgLib2 NLinearInterpolator (3-D): taps in Gray-code order, each tap
 * (float)(v * w_double), float accumulation.                                 */
/* ------------------------------------------------------------------------- */
typedef float (*tap_fn)(const float* in, const int64_t dim[3], int64_t x, int64_t y, int64_t z);

static float tap_zero(const float* in, const int64_t dim[3], int64_t x, int64_t y, int64_t z)
{
    if (x < 0 || y < 0 || z < 0 || x >= dim[0] || y >= dim[1] || z >= dim[2]) return 0.0f;
    return in[x + dim[0] * (y + dim[1] * z)];
}

static int64_t mirror_single(int64_t i, int64_t n)
{
    if (n == 1) return 0;
    const int64_t p = 2 * n - 2;
    i %= p; if (i < 0) i += p;
    return i < n ? i : p - i;
}

static float tap_mirror(const float* in, const int64_t dim[3], int64_t x, int64_t y, int64_t z)
{
    x = mirror_single(x, dim[0]); y = mirror_single(y, dim[1]); z = mirror_single(z, dim[2]);
    return in[x + dim[0] * (y + dim[1] * z)];
}

static float nlinear3(const float* in, const int64_t dim[3], tap_fn tap, double px, double py, double pz)
{
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    const int64_t x = (int64_t)fx, y = (int64_t)fy, z = (int64_t)fz;
    const double w0 = px - fx, w1 = py - fy, w2 = pz - fz;
    const double w0n = 1.0 - w0, w1n = 1.0 - w1, w2n = 1.0 - w2;
    float acc = (float)((double)tap(in, dim, x,     y,     z    ) * (w0n * w1n * w2n));
    acc += (float)((double)tap(in, dim, x + 1, y,     z    ) * (w0  * w1n * w2n));
    acc += (float)((double)tap(in, dim, x + 1, y + 1, z    ) * (w0  * w1  * w2n));
    acc += (float)((double)tap(in, dim, x,     y + 1, z    ) * (w0n * w1  * w2n));
    acc += (float)((double)tap(in, dim, x,     y + 1, z + 1) * (w0n * w1  * w2 ));
    acc += (float)((double)tap(in, dim, x + 1, y + 1, z + 1) * (w0  * w1  * w2 ));
    acc += (float)((double)tap(in, dim, x + 1, y,     z + 1) * (w0  * w1n * w2 ));
    acc += (float)((double)tap(in, dim, x,     y,     z + 1) * (w0n * w1n * w2 ));
    return acc;
}

/* Threading switch for the CPU-baseline leg of bench.py (mode "all_cores"): 0 (default) = every stage a single-threaded cursor
 * loop, as the reference runs them (only FFTConvolution gets the executor service, SMVD:257,527); 1 = OpenMP over planes /
 * columns.  rotate and attenuate are bit-identical either way (independent voxels / columns); adjustImage's sum then
 * cascades per 64 Ki chunk and over the chunk sums (not the serial RealSum order), so the parity tests keep the default. */
static int g_parallel = 0;
void orc_set_parallel(int on) { g_parallel = on ? 1 : 0; }
int  orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* SMVD:104-135, output planes z0 .. z0 + nzp - 1 only (out holds nzp planes): the cursor loop visits every output voxel
 * independently (SMVD:119-132), so a range of planes of the full result is that loop over those planes -- what the full-size
 * parity tests compare with planes downloaded from the device. */
int orc_rotate_around_axis_planes(const float* in, const int64_t dim[3], int axis, int degrees, int64_t z0, int64_t nzp, float* out)
{
    if (axis < 0 || axis > 2 || z0 < 0 || nzp < 0 || z0 + nzp > dim[2]) return -1;
    double m[12], a[12];
    orc_axis_rotation(dim, axis, degrees, m);
    orc_affine_invert(m, a);
#pragma omp parallel for schedule(static) if (g_parallel)
    for (int64_t z = z0; z < z0 + nzp; ++z)
        for (int64_t y = 0; y < dim[1]; ++y)
            for (int64_t x = 0; x < dim[0]; ++x) {
                const double l0 = (double)x, l1 = (double)y, l2 = (double)z;
                const double px = l0 * a[0] + l1 * a[1] + l2 * a[2] + a[3];
                const double py = l0 * a[4] + l1 * a[5] + l2 * a[6] + a[7];
                const double pz = l0 * a[8] + l1 * a[9] + l2 * a[10] + a[11];
                out[x + dim[0] * (y + dim[1] * (z - z0))] = nlinear3(in, dim, tap_zero, px, py, pz);
            }
    return 0;
}

/* SMVD:104-135 */
int orc_rotate_around_axis(const float* in, const int64_t dim[3], int axis, int degrees, float* out)
{
    return orc_rotate_around_axis_planes(in, dim, axis, degrees, 0, dim[2], out);
}

/* SMVD:318-364.  Q1: the loop runs dimension(0) steps along y. */
int orc_attenuate3d(const float* in, const int64_t dim[3], double delta, float* out)
{
    const int64_t nx = dim[0], ny = dim[1], nz = dim[2];
    if (nx > ny) return -1;                       /* reference walks out of the interval */
    memset(out, 0, (size_t)(nx * ny * nz) * sizeof(float));
#pragma omp parallel for schedule(static) if (g_parallel)
    for (int64_t z = 0; z < nz; ++z)
        for (int64_t x = 0; x < nx; ++x) {
            double n = 1.0;
            int64_t y = ny - 1;
            for (int64_t step = 0; step < nx; ++step, --y) {
                const int64_t i = x + nx * (y + ny * z);
                const double v = (double)in[i];
                const double phiN = v * delta * n;
                n = fmax(n - phiN, 0.0);
                out[i] = (float)(v * n);
            }
        }
    return 0;
}

/* mpicbg.util.RealSum: binary-counter cascade of partial sums. */
typedef struct { int flags[64]; double sums[64]; } realsum;

static void realsum_add(realsum* s, double a)
{
    int i = 0;
    double sum = a;
    while (i < 64 && s->flags[i]) {
        s->flags[i] = 0;
        sum += s->sums[i];
        s->sums[i] = 0.0;
        ++i;
    }
    if (i < 64) { s->flags[i] = 1; s->sums[i] = sum; }
}

static double realsum_get(const realsum* s)
{
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) sum += s->sums[i];
    return sum;
}

double orc_sum_image(const float* img, int64_t n)            /* Tools:124-132 */
{
    realsum s; memset(&s, 0, sizeof(s));
    for (int64_t i = 0; i < n; ++i) realsum_add(&s, (double)img[i]);
    return realsum_get(&s);
}

void orc_norm_image(float* img, int64_t n)                   /* Tools:112-118 */
{
    const double sum = orc_sum_image(img, n);
    for (int64_t i = 0; i < n; ++i) img[i] = (float)((double)img[i] / sum);
}

static double sum_image_chunked(const float* img, int64_t n)  /* g_parallel only: RealSum per chunk, RealSum of the chunk sums */
{
    const int64_t chunk = 65536, nc = (n + chunk - 1) / chunk;
    double* part = (double*)malloc((size_t)nc * sizeof(double));
    if (!part) return orc_sum_image(img, n);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        const int64_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
        part[c] = orc_sum_image(img + lo, hi - lo);
    }
    realsum s; memset(&s, 0, sizeof(s));
    for (int64_t c = 0; c < nc; ++c) realsum_add(&s, part[c]);
    free(part);
    return realsum_get(&s);
}

double orc_adjust_image(float* img, int64_t n, float min_value, float target_average) /* Tools:143-159 */
{
    const double avg = (g_parallel ? sum_image_chunked(img, n) : orc_sum_image(img, n)) / (double)n;
    const double correction = (double)(target_average - min_value) / avg;  /* float subtraction, Q6 */
#pragma omp parallel for schedule(static) if (g_parallel)
    for (int64_t i = 0; i < n; ++i) img[i] = (float)((double)img[i] * correction);
#pragma omp parallel for schedule(static) if (g_parallel)
    for (int64_t i = 0; i < n; ++i) img[i] = img[i] + min_value;
    return correction;
}

/* SMVD:253-264 with imglib2-algorithm FFTConvolution semantics. */
int orc_convolve_direct(const float* img, const int64_t dim[3], float* psf, const int64_t kdim[3], float* out)
{
    const int64_t nx = dim[0], ny = dim[1], nz = dim[2];
    const int64_t kx = kdim[0], ky = kdim[1], kz = kdim[2];
    orc_norm_image(psf, kx * ky * kz);                       /* SMVD:255, in place (Q5) */
    const int64_t cx = kx / 2, cy = ky / 2, cz = kz / 2;
    int64_t* mx = (int64_t*)malloc((size_t)(nx + kx) * sizeof(int64_t));
    if (!mx) return -2;
    /* mx[j] = mirror(j - (kx - 1 - cx))  for j in [0, nx + kx - 1) */
    for (int64_t j = 0; j < nx + kx - 1; ++j) mx[j] = mirror_single(j - (kx - 1 - cx), nx);
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t z = 0; z < nz; ++z)
        for (int64_t y = 0; y < ny; ++y) {
            double* acc = (double*)calloc((size_t)nx, sizeof(double));
            for (int64_t c = 0; c < kz; ++c) {
                const int64_t sz = mirror_single(z - (c - cz), nz);
                for (int64_t b = 0; b < ky; ++b) {
                    const int64_t sy = mirror_single(y - (b - cy), ny);
                    const float* row = img + nx * (sy + ny * sz);
                    const float* krow = psf + kx * (b + ky * c);
                    for (int64_t a = 0; a < kx; ++a) {
                        const double w = (double)krow[a];
                        if (w == 0.0) continue;
                        /* source x = x - (a - cx); index into mx: x - a + cx + (kx-1-cx) = x + (kx-1-a) */
                        const int64_t* mrow = mx + (kx - 1 - a);
                        for (int64_t x = 0; x < nx; ++x) acc[x] += w * (double)row[mrow[x]];
                    }
                }
            }
            float* o = out + nx * (y + ny * z);
            for (int64_t x = 0; x < nx; ++x) o[x] = (float)acc[x];
            free(acc);
        }
    free(mx);
    return 0;
}

/* The same sum (SMVD:253-264 / FFTConvolution semantics: mirror-single image boundary, centre K/2, no flip) at n LISTED
 * voxels only, psf taken AS GIVEN (already normalised by the caller): what lets a 63^3 PSF on a large volume be checked in
 * seconds.  idx[i] = x + Nx*(y + Ny*z).  Taps are summed in the order of orc_convolve_direct (c, b, a). */
int orc_convolve_direct_at(const float* img, const int64_t dim[3], const float* psf, const int64_t kdim[3],
                           const int64_t* idx, int64_t n, double* out)
{
    const int64_t nx = dim[0], ny = dim[1], nz = dim[2];
    const int64_t kx = kdim[0], ky = kdim[1], kz = kdim[2];
    const int64_t cx = kx / 2, cy = ky / 2, cz = kz / 2;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t x = idx[i] % nx, y = (idx[i] / nx) % ny, z = idx[i] / (nx * ny);
        double acc = 0.0;
        for (int64_t c = 0; c < kz; ++c) {
            const int64_t sz = mirror_single(z - (c - cz), nz);
            for (int64_t b = 0; b < ky; ++b) {
                const int64_t sy = mirror_single(y - (b - cy), ny);
                const float* row = img + nx * (sy + ny * sz);
                const float* krow = psf + kx * (b + ky * c);
                for (int64_t a = 0; a < kx; ++a) acc += (double)krow[a] * (double)row[mirror_single(x - (a - cx), nx)];
            }
        }
        out[i] = acc;
    }
    return 0;
}

/* SMVD:197 */
int64_t orc_extract_nz(int64_t nz, int inc) { return (nz - 1) / inc + 1; }

/* SMVD:195-251 + Tools:73-86, reference-exact RNG consumption (Q10). */
int orc_extract_slices_ref(const float* in, const int64_t dim[3], int inc, float snr,
                           orc_jrandom* rnd, float* out)
{
    if (inc < 1) return -1;
    const int64_t plane = dim[0] * dim[1];
    const double mul = orc_poisson_mul((double)snr);         /* float SNR widened to double, SMVD:248 */
    int64_t cz = 0;
    for (int64_t z = 0; z < dim[2]; z += inc, ++cz) {
        const float* src = in + plane * z;
        float* dst = out + plane * cz;
        if (snr >= 0.0f) {                                   /* SMVD:211 (Q8) */
            for (int64_t i = 0; i < plane; ++i)
                dst[i] = (float)orc_poisson_interarrival(rnd, (double)src[i] * mul);
        } else {
            memcpy(dst, src, (size_t)plane * sizeof(float));
        }
    }
    return 0;
}

/* The same on a WINDOW of the source volume: `in` holds the planes z0 .. z0 + dim[2] - 1 of a volume with planes of dim[0] x dim[1]
 * voxels, z0 a plane extractSlices reads (z0 % inc == 0); the counters are the source indices in the FULL volume, so the counts are the
 * planes z0 / inc ... of the whole view's acquisition (full-size parity tests download a window, not the volume). */
int orc_extract_slices_counter_window(const float* in, const int64_t dim[3], int inc, float snr,
                                      uint64_t seed, uint32_t stream, int64_t z0, float* out)
{
    if (inc < 1 || z0 < 0 || z0 % inc != 0) return -1;
    const int64_t plane = dim[0] * dim[1];
    const double mul = orc_poisson_mul((double)snr);
    int64_t cz = 0;
    for (int64_t z = 0; z < dim[2]; z += inc, ++cz) {
        const float* src = in + plane * z;
        float* dst = out + plane * cz;
        if (snr >= 0.0f) {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < plane; ++i)
                dst[i] = (float)orc_poisson_counter((double)src[i] * mul, seed, stream, (uint64_t)(plane * (z + z0) + i));
        } else {
            memcpy(dst, src, (size_t)plane * sizeof(float));
        }
    }
    return 0;
}

int orc_extract_slices_counter(const float* in, const int64_t dim[3], int inc, float snr,
                               uint64_t seed, uint32_t stream, float* out)
{
    return orc_extract_slices_counter_window(in, dim, inc, snr, seed, stream, 0, out);
}

/* SMVD:144-171 */
int64_t orc_isotropic_nz(int64_t nz_acq, int inc) { return (nz_acq - 1) * inc + 1; }

int orc_make_isotropic(const float* in, const int64_t dim[3], int inc, float* out)
{
    if (inc < 1) return -1;
    const int64_t onz = orc_isotropic_nz(dim[2], inc);
    for (int64_t z = 0; z < onz; ++z) {
        const double pz = (double)((float)z / (float)inc);   /* Q4: float division */
        for (int64_t y = 0; y < dim[1]; ++y)
            for (int64_t x = 0; x < dim[0]; ++x)
                out[x + dim[0] * (y + dim[1] * z)] = nlinear3(in, dim, tap_mirror, (double)x, (double)y, pz);
    }
    return 0;
}

/* SMVD:280-316 (Q11: delta unused) */
int orc_compute_weight_image(const int64_t dim[3], float* out)
{
    const int cosine_span = 40;
    const int size_y = (int)dim[1];
    for (int64_t z = 0; z < dim[2]; ++z)
        for (int64_t y = 0; y < dim[1]; ++y) {
            const int l = size_y - (int)y - 1;
            float value;
            if (l < size_y / 2) value = 1.0f;
            else if (l > size_y / 2 + cosine_span) value = 0.0f;
            else {
                const double pos = ((double)(l - size_y / 2) / (double)cosine_span) * 3.141592653589793;
                value = (float)((cos(pos) + 1.0) / 2.0);
            }
            float* o = out + dim[0] * (y + dim[1] * z);
            for (int64_t x = 0; x < dim[0]; ++x) o[x] = value;
        }
    return 0;
}

/* ---- phantom generator, SMVD:366-522 -------------------------------------------------------------------------
 * HyperSphereCursor as a literal state machine: position, per-dimension radius r[] and remaining steps s[];
 * fwd() moves the lowest dimension that still has steps and re-derives the radii of the dimensions below it
 * from the radius of the dimension above ((long)Math.sqrt, i.e. truncation).  reset() parks the cursor one step
 * in front of the first voxel of the outermost (last) dimension. */
typedef struct {
    int64_t c[3], pos[3], r[3], s[3], radius;
} hs_cursor;

static void hs_reset(hs_cursor* h, const int64_t c[3], int64_t radius)
{
    h->radius = radius;
    for (int d = 0; d < 3; ++d) { h->c[d] = c[d]; h->pos[d] = c[d]; h->r[d] = 0; h->s[d] = 0; }
    h->r[2] = radius;
    h->s[2] = 2 * radius + 1;
    h->pos[2] = c[2] - radius - 1;
}

static int hs_has_next(const hs_cursor* h) { return h->s[0] > 0 || h->s[1] > 0 || h->s[2] > 0; }

static void hs_fwd(hs_cursor* h)
{
    int d = 0;
    while (d < 3 && h->s[d] <= 0) ++d;
    h->s[d] -= 1;
    h->pos[d] += 1;
    for (int e = d - 1; e >= 0; --e) {
        const int64_t rad = h->r[e + 1];
        const int64_t off = h->pos[e + 1] - h->c[e + 1];
        const int64_t rad2 = (int64_t)sqrt((double)(rad * rad - off * off));
        h->r[e] = rad2;
        h->s[e] = 2 * rad2;
        h->pos[e] = h->c[e] - rad2;
    }
}

int64_t orc_hypersphere_size(int64_t radius)
{
    const int64_t c[3] = {0, 0, 0};
    hs_cursor h;
    hs_reset(&h, c, radius);
    int64_t n = 0;
    while (hs_has_next(&h)) { hs_fwd(&h); ++n; }
    return n;
}

static int64_t ipow(int64_t a, int b) { int64_t r = 1; while (b-- > 0) r *= a; return r; }

/* SMVD:436-522 */
int orc_draw_spheres(float* img, const int64_t dim[3], double min_value, double max_value, int scale,
                     int half_pixel_offset, orc_jrandom* rnd, int64_t* n_spheres)
{
    int64_t c[3], min_size = dim[0];
    for (int d = 0; d < 3; ++d) { c[d] = dim[d] / 2; if (dim[d] < min_size) min_size = dim[d]; }
    const int max_radius = 10 * scale;
    const int64_t radius_large = min_size / 2 - 47 * scale - 1;
    if (scale < 1 || radius_large < 0) return -1;
    const int64_t modulus = ipow(7 * scale, 3);            /* Util.pow( 7*scale, numDimensions ) */
    int64_t drawn = 0;
    hs_cursor big;
    hs_reset(&big, c, radius_large);
    while (hs_has_next(&big)) {
        hs_fwd(&big);
        const int radius = orc_jrandom_next_int_bound(rnd, max_radius) + 1;
        int64_t sc[3] = {big.pos[0], big.pos[1], big.pos[2]};
        if (half_pixel_offset) { sc[0] += 1; sc[1] += 1; }
        double random_value = orc_jrandom_next_double(rnd);
        const int64_t rounded = (int64_t)floor(random_value * 10000 + 0.5);      /* Math.round */
        if (rounded % modulus == 0) {
            random_value = orc_jrandom_next_double(rnd) * (max_value - min_value) + min_value;
            hs_cursor sm;
            hs_reset(&sm, sc, radius);
            while (hs_has_next(&sm)) {
                hs_fwd(&sm);
                if (sm.pos[0] < 0 || sm.pos[1] < 0 || sm.pos[2] < 0 || sm.pos[0] >= dim[0] || sm.pos[1] >= dim[1] ||
                    sm.pos[2] >= dim[2])
                    return -2;                                  /* the reference would throw */
                float* v = img + sm.pos[0] + dim[0] * (sm.pos[1] + dim[1] * sm.pos[2]);
                const double cur = (double)*v;
                *v = (float)(random_value > cur ? random_value : cur);          /* Math.max, setReal */
            }
            ++drawn;
        }
    }
    if (n_spheres) *n_spheres = drawn;
    return 0;
}

/* SMVD:394-424: samples at 2l + 0.5 with the n-linear interpolator over the mirror-single extension */
int orc_downsample2x(const float* in, const int64_t dim[3], float* out)
{
    const int64_t od[3] = {dim[0] / 2 - 1, dim[1] / 2 - 1, dim[2] / 2 - 1};
    if (od[0] < 1 || od[1] < 1 || od[2] < 1) return -1;
    for (int64_t z = 0; z < od[2]; ++z)
        for (int64_t y = 0; y < od[1]; ++y)
            for (int64_t x = 0; x < od[0]; ++x)
                out[x + od[0] * (y + od[1] * z)] =
                    nlinear3(in, dim, tap_mirror, (double)x * 2.0 + 0.5, (double)y * 2.0 + 0.5, (double)z * 2.0 + 0.5);
    return 0;
}

/* SMVD:615-640: float sum over the views in view order; zero sum -> zeros, else min(1, osem * (w / sum)). */
int orc_normalize_weights(float* const* weights, int n_views, int64_t n, float osem)
{
    if (n_views < 1) return -1;
    for (int64_t i = 0; i < n; ++i) {
        float sum = 0.0f;
        for (int v = 0; v < n_views; ++v) sum += weights[v][i];
        for (int v = 0; v < n_views; ++v) {
            if (sum == 0.0f) weights[v][i] = 0.0f;
            else {
                const float w = osem * (weights[v][i] / sum);
                weights[v][i] = w < 1.0f ? w : 1.0f;       /* Math.min(1, w) */
            }
        }
    }
    return 0;
}
