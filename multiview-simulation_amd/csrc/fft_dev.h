// Device-side building blocks of the hand-written FFT passes (fft_kernels.hip) shared with the fused
// rotate + attenuate + x-transform kernel (rotate_fft.hip): complex helpers, small DFTs, the wave-private Stockham
// passes in LDS, the plan / size table, and the index maps of the padded volume.
#pragma once

#include "common.h"

namespace mvsim {
namespace fft {

// ---------------------------------------------------------------------------------- complex helpers
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by -i (forward-direction quarter turn)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// ---------------------------------------------------------------------------------- small DFTs (forward)
template <int R> __device__ __forceinline__ void dft(float2* u);

template <> __device__ __forceinline__ void dft<2>(float2* u)
{
    const float2 a = u[0], b = u[1];
    u[0] = cadd(a, b);
    u[1] = csub(a, b);
}

template <> __device__ __forceinline__ void dft<3>(float2* u)
{
    const float c = -0.5f, s = -0.86602540378443864676f;   // w3 = c + i s
    const float2 t = cadd(u[1], u[2]);
    const float2 d = csub(u[1], u[2]);
    const float2 m = make_float2(fmaf(c, t.x, u[0].x), fmaf(c, t.y, u[0].y));
    const float2 r = make_float2(-s * d.y, s * d.x);        // i*s*d
    u[0] = cadd(u[0], t);
    u[1] = cadd(m, r);
    u[2] = csub(m, r);
}

template <> __device__ __forceinline__ void dft<4>(float2* u)
{
    const float2 a = cadd(u[0], u[2]), b = csub(u[0], u[2]);
    const float2 c = cadd(u[1], u[3]), d = mul_mi(csub(u[1], u[3]));
    u[0] = cadd(a, c);
    u[1] = cadd(b, d);
    u[2] = csub(a, c);
    u[3] = csub(b, d);
}

template <> __device__ __forceinline__ void dft<8>(float2* u)
{
    const float h = 0.70710678118654752440f;
    float2 e[4] = {u[0], u[2], u[4], u[6]};
    float2 o[4] = {u[1], u[3], u[5], u[7]};
    dft<4>(e);
    dft<4>(o);
    // o[k] *= w8^k
    o[1] = make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    o[2] = mul_mi(o[2]);
    o[3] = make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u[k] = cadd(e[k], o[k]);
        u[k + 4] = csub(e[k], o[k]);
    }
}

__device__ __forceinline__ float2 cscale(float c, float2 a) { return make_float2(c * a.x, c * a.y); }
__device__ __forceinline__ float2 cfma(float c, float2 a, float2 b) { return make_float2(fmaf(c, a.x, b.x), fmaf(c, a.y, b.y)); }

template <> __device__ __forceinline__ void dft<5>(float2* u)
{
    const float c1 = 0.30901699437494745f, c2 = -0.8090169943749473f;    // cos(2pi/5), cos(4pi/5)
    const float s1 = 0.9510565162951535f, s2 = 0.5877852522924732f;      // sin(2pi/5), sin(4pi/5)
    const float2 t1 = cadd(u[1], u[4]), t2 = cadd(u[2], u[3]);
    const float2 d1 = csub(u[1], u[4]), d2 = csub(u[2], u[3]);
    const float2 m1 = cfma(c2, t2, cfma(c1, t1, u[0]));
    const float2 m2 = cfma(c1, t2, cfma(c2, t1, u[0]));
    const float2 q1 = mul_mi(cfma(s2, d2, cscale(s1, d1)));             // -i (s1 d1 + s2 d2)
    const float2 q2 = mul_mi(cfma(-s1, d2, cscale(s2, d1)));            // -i (s2 d1 - s1 d2)
    u[0] = cadd(u[0], cadd(t1, t2));
    u[1] = cadd(m1, q1);
    u[4] = csub(m1, q1);
    u[2] = cadd(m2, q2);
    u[3] = csub(m2, q2);
}

template <> __device__ __forceinline__ void dft<7>(float2* u)
{
    const float c1 = 0.6234898018587336f, c2 = -0.22252093395631434f, c3 = -0.900968867902419f;
    const float s1 = 0.7818314824680298f, s2 = 0.9749279121818236f, s3 = 0.43388373911755823f;
    const float2 t1 = cadd(u[1], u[6]), t2 = cadd(u[2], u[5]), t3 = cadd(u[3], u[4]);
    const float2 d1 = csub(u[1], u[6]), d2 = csub(u[2], u[5]), d3 = csub(u[3], u[4]);
    const float2 m1 = cfma(c3, t3, cfma(c2, t2, cfma(c1, t1, u[0])));
    const float2 m2 = cfma(c1, t3, cfma(c3, t2, cfma(c2, t1, u[0])));
    const float2 m3 = cfma(c2, t3, cfma(c1, t2, cfma(c3, t1, u[0])));
    const float2 q1 = mul_mi(cfma(s3, d3, cfma(s2, d2, cscale(s1, d1))));
    const float2 q2 = mul_mi(cfma(-s1, d3, cfma(-s3, d2, cscale(s2, d1))));
    const float2 q3 = mul_mi(cfma(s2, d3, cfma(-s1, d2, cscale(s3, d1))));
    u[0] = cadd(cadd(u[0], t1), cadd(t2, t3));
    u[1] = cadd(m1, q1);
    u[6] = csub(m1, q1);
    u[2] = cadd(m2, q2);
    u[5] = csub(m2, q2);
    u[3] = cadd(m3, q3);
    u[4] = csub(m3, q3);
}

template <> __device__ __forceinline__ void dft<10>(float2* u)
{
    float2 e[5] = {u[0], u[2], u[4], u[6], u[8]};
    float2 o[5] = {u[1], u[3], u[5], u[7], u[9]};
    dft<5>(e);
    dft<5>(o);
    o[1] = cmul(o[1], make_float2(0.8090169943749475f, -0.5877852522924731f));     // w10^1
    o[2] = cmul(o[2], make_float2(0.30901699437494745f, -0.9510565162951535f));    // w10^2
    o[3] = cmul(o[3], make_float2(-0.30901699437494734f, -0.9510565162951536f));   // w10^3
    o[4] = cmul(o[4], make_float2(-0.8090169943749473f, -0.5877852522924732f));    // w10^4
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        u[k] = cadd(e[k], o[k]);
        u[k + 5] = csub(e[k], o[k]);
    }
}

template <> __device__ __forceinline__ void dft<9>(float2* u)
{
    // 9 = 3 x 3: columns n1 (stride 3), twiddle w9^(n1*k2), rows
    float2 a[3][3];
#pragma unroll
    for (int n1 = 0; n1 < 3; ++n1) {
        a[n1][0] = u[n1];
        a[n1][1] = u[n1 + 3];
        a[n1][2] = u[n1 + 6];
        dft<3>(a[n1]);
    }
    const float2 w1 = make_float2(0.76604444311897803520f, -0.64278760968653932632f);   // w9^1
    const float2 w2 = make_float2(0.17364817766693034885f, -0.98480775301220805937f);   // w9^2
    const float2 w4 = make_float2(-0.93969262078590838405f, -0.34202014332566873304f);  // w9^4
    a[1][1] = cmul(a[1][1], w1);
    a[1][2] = cmul(a[1][2], w2);
    a[2][1] = cmul(a[2][1], w2);
    a[2][2] = cmul(a[2][2], w4);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) {
        float2 b[3] = {a[0][k2], a[1][k2], a[2][k2]};
        dft<3>(b);
        u[k2] = b[0];
        u[k2 + 3] = b[1];
        u[k2 + 6] = b[2];
    }
}

// Tools.adjustImage on one voxel: (float)(v * corr), then + minValue as a second float rounding (Tools.java:150-155)
__device__ __forceinline__ float adjust_one_f(float v, double corr, float min_value)
{
    const float t = (float)((double)v * corr);
    return t + min_value;
}

// ---------------------------------------------------------------------------------- Stockham passes in LDS
// Compiler-level ordering point for LDS traffic of ONE wave (the hardware executes a wave's DS
// operations in issue order, so no s_barrier is needed between a wave's own writes and reads).
__device__ __forceinline__ void wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// wbuf: the LW lines owned by this wave (length L, pitch LP complex elements); tw: L twiddles
// exp(-2 pi i k / L) in LDS.  Per radix pass: read all butterflies of the wave into registers, then
// twiddle + DFT + write back in Stockham order.
// Identity / "multiply by the PSF spectrum and conjugate" operators applied to the inputs of the FIRST radix
// pass (the spectrum product of pass C rides on the second transform's loads: no separate LDS pass).
// (it, r: the butterfly's place in the first pass's register tile -- compile-time constants once the loops are unrolled)
struct LoadIdentity {
    __device__ __forceinline__ float2 operator()(float2 v, int, int, int, int) const { return v; }
};
struct LoadMulConj {
    const float2* g;        // spectrum lines of this wave, line-major: g[line * glen + n]
    int glen;
    __device__ __forceinline__ float2 operator()(float2 v, int line, int n, int, int) const
    {
        return cconj(cmul(v, g[line * glen + n]));
    }
};
// the same product with the spectrum held in REGISTERS, laid out as the first pass reads its operands (ps[it][r]: what
// capture_first_pass took from the transformed PSF lines of this wave)
template <int IT, int R> struct LoadMulConjReg {
    const float2 (&ps)[IT][R];
    __device__ __forceinline__ float2 operator()(float2 v, int, int, int it, int r) const { return cconj(cmul(v, ps[it][r])); }
};

// Twiddles: one table per transform length, laid out PER PASS and r-major -- pass (P, R) owns (R-1)*P entries at
// offset TOFF, entry (r-1)*P + k = exp(-2 pi i r k / (P R)).  Consecutive lanes (consecutive k) read consecutive
// LDS words: conflict-free, unlike indexing one exp(-2 pi i j / L) table at stride r*L/(P R).
template <int L, int LP, int LW, int P, int TOFF, class OP>
__device__ __forceinline__ void wpasses(float2*, const float2*, int, const OP&) {}

template <int L, int LP, int LW, int P, int TOFF, class OP, int R, int... Rest>
__device__ __forceinline__ void wpasses(float2* __restrict__ wbuf, const float2* __restrict__ tw, int lane, const OP& op)
{
    constexpr int STR = L / R;
    constexpr int NB = LW * STR;               // butterflies of this wave in this pass
    constexpr int IT = (NB + 63) / 64;
    float2 u[IT][R];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int b = lane + it * 64;
        if ((NB % 64 == 0) || b < NB) {
            const int line = b / STR, i = b - line * STR;
            const float2* src = wbuf + line * LP + i;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                u[it][r] = src[r * STR];
                if (P == 1) u[it][r] = op(u[it][r], line, i + r * STR, it, r);
            }
        }
    }
    wave_order();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int b = lane + it * 64;
        if ((NB % 64 == 0) || b < NB) {
            const int line = b / STR, i = b - line * STR;
            const int k = i % P;
            const int j = (i - k) * R + k;
            if (P > 1) {
#pragma unroll
                for (int r = 1; r < R; ++r) u[it][r] = cmul(u[it][r], tw[TOFF + (r - 1) * P + k]);
            }
            dft<R>(u[it]);
            float2* dst = wbuf + line * LP + j;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r * P] = u[it][r];
        }
    }
    wave_order();
    wpasses<L, LP, LW, P * R, TOFF + (P > 1 ? (R - 1) * P : 0), OP, Rest...>(wbuf, tw, lane, op);
}

// complex elements per wave in the x passes: 576 (2 rows of 288) keeps 8 blocks x 4 waves resident per CU
#ifndef MVSIM_X_ELEMS
#define MVSIM_X_ELEMS 576
#endif
#ifndef MVSIM_NL_BIG
#define MVSIM_NL_BIG ((L <= 576) ? 16 : 8)
#endif
// size traits ---------------------------------------------------------------------------------------
constexpr int pow2_floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }
constexpr int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// y / z passes: NL lines per tile (NL adjacent kx columns); NW waves, each owning LW = NL / NW lines
// (~1024 complex elements per wave).
template <int L> struct Cfg {
    static constexpr int NL = MVSIM_NL_BIG;
    static constexpr int LP = L + 1;                          // odd pitch: conflict-free transposed staging
    static constexpr int NW = clampi(pow2_floor((NL * L) / 1024 > 0 ? (NL * L) / 1024 : 1), 1, NL);
    static constexpr int LW = NL / NW;
    static constexpr int T = 64 * NW;
    static constexpr size_t LDS = (size_t)(NL * LP + L) * sizeof(float2) + 32 * sizeof(double);
};

// x passes: rows are contiguous in HBM; 4 waves per block, each owning LW rows (~MVSIM_X_ELEMS complex per wave).
template <int M> struct CfgX {
    static constexpr int LW = clampi(MVSIM_X_ELEMS / M, 1, 8);
    static constexpr int NW = 4;
    static constexpr int NL = NW * LW;
    static constexpr int LP = M + 1;
    static constexpr int T = 64 * NW;
    static constexpr size_t LDS = (size_t)(NL * LP + M) * sizeof(float2) + 32 * sizeof(double);
};

template <int A, int...> struct FirstOf { static constexpr int value = A; };

template <int L, int... Rs> struct Plan {
    static constexpr int len = L;
    static constexpr int R1 = FirstOf<Rs...>::value;            // radix of the first pass
    // the wave's transformed lines as the operands of the NEXT transform's first pass: ps[it][r] = wbuf[line][i + r * STR]
    template <int LW, int IT>
    static __device__ __forceinline__ void capture_first_pass(const float2* wbuf, int lane, float2 (&ps)[IT][R1])
    {
        constexpr int STR = L / R1, NB = LW * STR;
        static_assert(IT == (NB + 63) / 64, "register tile of the first pass");
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int b = lane + it * 64;
#pragma unroll
            for (int r = 0; r < R1; ++r) ps[it][r] = make_float2(0.f, 0.f);
            if ((NB % 64 == 0) || b < NB) {
                const int line = b / STR, i = b - line * STR;
#pragma unroll
                for (int r = 0; r < R1; ++r) ps[it][r] = wbuf[line * (L + 1) + i + r * STR];
            }
        }
    }
    // transform the LW lines starting at wbuf (owned by the calling wave)
    template <int LW>
    static __device__ __forceinline__ void run(float2* wbuf, const float2* tw, int lane)
    {
        wpasses<L, L + 1, LW, 1, 0, LoadIdentity, Rs...>(wbuf, tw, lane, LoadIdentity{});
    }
    template <int LW, class OP>
    static __device__ __forceinline__ void run_op(float2* wbuf, const float2* tw, int lane, const OP& op)
    {
        wpasses<L, L + 1, LW, 1, 0, OP, Rs...>(wbuf, tw, lane, op);
    }
};

// Three-zone index map of one padded dimension: j in [0,a) -> zone 1, j in [P-b,P) -> zone 3, else zero.
struct DimMap {
    int n;      // source extent
    int P;      // padded extent
    int a, b;   // zone lengths
    int mode;   // 0: mirror-single image padding; 1: PSF embedding (shift by c = K/2)
    int c;
};

__device__ __forceinline__ int map_src(const DimMap& m, int j)
{
    if (m.mode == 0) {
        int i;
        if (j < m.a) i = j;
        else if (j >= m.P - m.b) i = j - m.P;
        else return -1;
        if (i < 0) i = -i;                           // one reflection covers any halo shorter than the image
        if (i >= m.n) i = 2 * m.n - 2 - i;
        if (i >= 0 && i < m.n) return i;
        if (m.n == 1) return 0;
        const int p = 2 * m.n - 2;                   // general case: halo longer than the image
        i %= p;
        if (i < 0) i += p;
        return i < m.n ? i : p - i;
    }
    if (j < m.a) return j + m.c;
    if (j >= m.P - m.b) return j - m.P + m.c;
    return -1;
}

struct SrcMap {
    DimMap x, y, z;
    int    enum_y;          // pass A: rows enumerated per plane (0: y.P); rows [enum_y, y.P) of a plane are never visited
};

// 16 bytes at 8-byte alignment (two adjacent complex elements starting at an odd index): global memory takes a dwordx4 there
struct __attribute__((packed, aligned(8))) Pair16 {
    float a, b, c, d;
};

// ---------------------------------------------------------------------------------- size table
// (MVSIM_DEV_SIZES: experiment builds that only need the 512^3 / 31^3 workload compile in seconds; never defined in the product build)
#ifdef MVSIM_DEV_SIZES
#define MVSIM_FFT_SIZES(X) X(280, 7, 5, 8) X(560, 7, 8, 10)
#else
#define MVSIM_FFT_SIZES(X) \
    X(16, 4, 4)            \
    X(18, 9, 2)            \
    X(20, 5, 4)            \
    X(24, 3, 8)            \
    X(32, 4, 8)            \
    X(36, 9, 4)            \
    X(40, 5, 8)            \
    X(48, 3, 4, 4)         \
    X(56, 7, 8)            \
    X(64, 8, 8)            \
    X(72, 9, 8)            \
    X(80, 5, 4, 4)         \
    X(96, 3, 8, 4)         \
    X(112, 7, 4, 4)        \
    X(128, 4, 8, 4)        \
    X(140, 7, 5, 4)        \
    X(144, 9, 4, 4)        \
    X(160, 5, 8, 4)        \
    X(180, 9, 5, 4)        \
    X(192, 3, 8, 8)        \
    X(224, 7, 8, 4)        \
    X(256, 4, 8, 8)        \
    X(280, 7, 5, 8)        \
    X(288, 9, 8, 4)        \
    X(320, 5, 8, 8)        \
    X(350, 7, 10, 5)       \
    X(360, 9, 5, 8)        \
    X(384, 3, 8, 4, 4)     \
    X(448, 7, 8, 8)        \
    X(512, 8, 8, 8)        \
    X(540, 9, 5, 4, 3)     \
    X(560, 7, 8, 10)       \
    X(576, 9, 8, 8)        \
    X(640, 5, 8, 4, 4)     \
    X(720, 9, 8, 10)       \
    X(768, 3, 8, 8, 4)     \
    X(896, 7, 8, 4, 4)     \
    X(1024, 4, 8, 8, 4)    \
    X(1080, 9, 8, 5, 3)    \
    X(1120, 7, 8, 5, 4)    \
    X(1152, 9, 8, 4, 4)    \
    X(1280, 5, 8, 8, 4)    \
    X(1440, 9, 8, 5, 4)    \
    X(1536, 3, 8, 8, 8)    \
    X(1792, 7, 8, 8, 4)    \
    X(2048, 8, 8, 8, 4)    \
    X(2160, 10, 8, 9, 3)   \
    X(2240, 7, 8, 8, 5)
#endif

// fused rotate + attenuate + x transform (rotate_fft.hip)
struct RotFftArgs {
    const float* in;
    float*       rot_out;       // may be null
    float*       att_out;       // may be null
    float2*      dst;           // spectrum rows: row (z, y) at dst + (z * py + y) * hxp
    const float2* twg;          // plan twiddles of length M (see wpasses)
    const float2* twx;          // exp(-2 pi i k / Px), k = 0..Px
    int nx, ny, nz, steps;
    int z_first, nzl;           // planes z_first .. z_first + nzl - 1 of the rotated volume are computed (a z slab; whole view: 0, nz); outputs are indexed from 0
    int hxp, py;
    int halo_r, halo_l;         // mirrored positions right of the row (kx / 2) and at the end of the padded row (kx - 1 - kx / 2)
    Affine a;
    double delta;
    int*   plane_nz;         // [nz], zeroed by the caller: set to 1 for every plane that holds a non-zero attenuated voxel (null: not wanted)
};

bool rot_fftx_has_plan(int M);
int  launch_rot_fftx(mvsim_ctx* ctx, int M, const RotFftArgs& a, bool write_out);

}  // namespace fft
}  // namespace mvsim
