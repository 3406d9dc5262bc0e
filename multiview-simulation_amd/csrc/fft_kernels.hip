// Hand-written LDS-tiled FFT convolution for gfx950 (the production path of
// SimulateMultiViewDataset.convolve, :253-264, for padded sizes of the form 2^a 3^b).
//
// A 3-D real convolution is five streaming passes over one half-spectrum buffer F
// (complex float, [Pz][Py][Hxp], Hxp = Px/2+1 rounded up to the tile width):
//   A  x: mirror-pad on the fly + real->complex FFT along x          read 4N      write C
//   B  y: complex FFT along y, in place                               read C       write C
//   C  z: FFT along z, multiply by the PSF spectrum, inverse FFT z    read C (+G)  write C
//   D  y: inverse FFT along y, in place                               read C       write C
//   E  x: complex->real FFT along x + crop + 1/P^3 + block sums       read C       write 4N
// Each pass stages a tile of NL lines in LDS (line = the FFT axis, contiguous in LDS; NL adjacent
// kx columns = 128 B contiguous in HBM), runs a mixed-radix Stockham FFT with one register
// butterfly per lane per radix pass, and writes the tile back.  Twiddles come from a host-computed
// (double precision, rounded once) table staged in LDS.  Inverse transforms use conj(FFT(conj(.))).
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

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

// ---------------------------------------------------------------------------------- Stockham passes in LDS
// buf: NL lines of length L, line pitch LP (complex elements); tw: L twiddles exp(-2 pi i k / L).
// One register butterfly per lane-iteration; read phase, barrier, write phase, barrier.
template <int L, int LP, int NL, int T, int P>
__device__ __forceinline__ void passes(float2*, const float2*, int) {}

template <int L, int LP, int NL, int T, int P, int R, int... Rest>
__device__ __forceinline__ void passes(float2* __restrict__ buf, const float2* __restrict__ tw, int tid)
{
    constexpr int STR = L / R;
    constexpr int NB = NL * STR;
    constexpr int IT = (NB + T - 1) / T;
    float2 u[IT][R];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int b = tid + it * T;
        if ((NB % T == 0) || b < NB) {
            const int line = b / STR, i = b - line * STR;
            const float2* src = buf + line * LP + i;
#pragma unroll
            for (int r = 0; r < R; ++r) u[it][r] = src[r * STR];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int b = tid + it * T;
        if ((NB % T == 0) || b < NB) {
            const int line = b / STR, i = b - line * STR;
            const int k = i % P;
            const int j = (i - k) * R + k;
            if (P > 1) {
                const int idx = k * (L / (P * R));
#pragma unroll
                for (int r = 1; r < R; ++r) u[it][r] = cmul(u[it][r], tw[r * idx]);
            }
            dft<R>(u[it]);
            float2* dst = buf + line * LP + j;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r * P] = u[it][r];
        }
    }
    __syncthreads();
    passes<L, LP, NL, T, P * R, Rest...>(buf, tw, tid);
}

// size traits ---------------------------------------------------------------------------------------
constexpr int round64(int v) { return ((v + 63) / 64) * 64; }
template <int L> struct Cfg {
    static constexpr int NL = (L <= 576) ? 16 : 8;           // lines per tile (16 complex = 128 B in HBM)
    static constexpr int LP = L + 1;                         // odd pitch: conflict-free transposed staging
    static constexpr int T0 = round64(NL * L / 16);
    static constexpr int T = T0 < 64 ? 64 : (T0 > 1024 ? 1024 : T0);
    // tile + twiddles + 256 B of small per-block tables (row offsets, wave partial sums)
    static constexpr size_t LDS = (size_t)(NL * LP + L) * sizeof(float2) + 32 * sizeof(double);
};

template <int L, int... Rs> struct Plan {
    static constexpr int len = L;
    template <int NL, int T>
    static __device__ __forceinline__ void run(float2* buf, const float2* tw, int tid)
    {
        passes<L, L + 1, NL, T, 1, Rs...>(buf, tw, tid);
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
        if (i >= 0 && i < m.n) return i;
        if (m.n == 1) return 0;
        const int p = 2 * m.n - 2;
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
};

// ---------------------------------------------------------------------------------- y / z pass kernel
enum Mode { FWD = 0, INV = 1, CONV = 2 };

struct LinesArgs {
    const float2* src;      // element (line c, position n) of tile (bx, by): src[by*src_outer + n'*src_es + bx*NL + c]
    float2*       dst;
    const float2* spec;     // CONV: PSF spectrum, same tiling
    const float2* tw;
    long long     src_es, src_outer, dst_es, dst_outer, spec_es, spec_outer;
    DimMap        lmap;     // SPARSE: position n reads source position map_src(lmap, n) (or zero)
};

template <class PLAN, int MODE, bool SPARSE>
__global__ __launch_bounds__(Cfg<PLAN::len>::T) void k_fft_lines(LinesArgs p)
{
    constexpr int L = PLAN::len;
    using C = Cfg<L>;
    constexpr int NL = C::NL, LP = C::LP, T = C::T;
    constexpr int LPR = NL / 2;             // lanes per position: one float4 = two adjacent columns
    constexpr int ROWS = T / LPR;
    constexpr int NIT = (L + ROWS - 1) / ROWS;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw = lds + NL * LP;
    const int tid = threadIdx.x;
    const int c2 = (tid % LPR) * 2;
    const int r0 = tid / LPR;

    // all global loads of the tile (and of the spectrum tile) are issued before anything waits
    const float2* sbase = p.src + (long long)blockIdx.y * p.src_outer + (long long)blockIdx.x * NL + c2;
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((L % ROWS == 0) || n < L) {
            if (SPARSE) {
                const int sn = map_src(p.lmap, n);
                if (sn >= 0) v[it] = *reinterpret_cast<const float4*>(sbase + sn * p.src_es);
            } else {
                v[it] = *reinterpret_cast<const float4*>(sbase + n * p.src_es);
            }
        }
    }
    float4 g[MODE == CONV ? NIT : 1];
    if (MODE == CONV) {
        const float2* gbase = p.spec + (long long)blockIdx.y * p.spec_outer + (long long)blockIdx.x * NL + c2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = r0 + it * ROWS;
            if ((L % ROWS == 0) || n < L) g[it] = *reinterpret_cast<const float4*>(gbase + n * p.spec_es);
        }
    }
    for (int i = tid; i < L; i += T) tw[i] = p.tw[i];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        if ((L % ROWS == 0) || n < L) {
            float2 a = make_float2(v[it].x, v[it].y), b = make_float2(v[it].z, v[it].w);
            if (MODE == INV) { a = cconj(a); b = cconj(b); }
            buf[c2 * LP + n] = a;
            buf[(c2 + 1) * LP + n] = b;
        }
    }
    __syncthreads();
    PLAN::template run<NL, T>(buf, tw, tid);
    if (MODE == CONV) {
        // multiply by the PSF spectrum, conjugate, transform again (inverse = conj FFT conj)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = r0 + it * ROWS;
            if ((L % ROWS == 0) || n < L) {
                const float2 a = cmul(buf[c2 * LP + n], make_float2(g[it].x, g[it].y));
                const float2 b = cmul(buf[(c2 + 1) * LP + n], make_float2(g[it].z, g[it].w));
                buf[c2 * LP + n] = cconj(a);
                buf[(c2 + 1) * LP + n] = cconj(b);
            }
        }
        __syncthreads();
        PLAN::template run<NL, T>(buf, tw, tid);
    }
    float2* dbase = p.dst + (long long)blockIdx.y * p.dst_outer + (long long)blockIdx.x * NL + c2;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        if ((L % ROWS == 0) || n < L) {
            float2 a = buf[c2 * LP + n], b = buf[(c2 + 1) * LP + n];
            if (MODE != FWD) { a = cconj(a); b = cconj(b); }
            *reinterpret_cast<float4*>(dbase + n * p.dst_es) = make_float4(a.x, a.y, b.x, b.y);
        }
    }
}

// ---------------------------------------------------------------------------------- x passes
// A: rows of the (virtually) padded real volume -> half spectrum along x.  M = Px/2.
// NR rows per block; dst row pitch hxp (complex).
template <class PLAN>
__global__ __launch_bounds__(Cfg<PLAN::len>::T) void k_fft_x_r2c(const float* __restrict__ src, SrcMap map,
                                                                float2* __restrict__ dst,
                                                                const float2* __restrict__ twg,
                                                                const float2* __restrict__ twx, int hxp,
                                                                long long rows)
{
    constexpr int M = PLAN::len;
    using C = Cfg<M>;
    constexpr int NR = C::NL, LP = C::LP, T = C::T;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw = lds + NR * LP;
    long long* rowoff = reinterpret_cast<long long*>(lds + NR * LP + M);   // NR entries (<= 16)
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * NR;
    if (tid < NR) {
        const long long row = row0 + tid;
        long long off = -1;
        if (row < rows) {
            const int py = map.y.P;
            const int y = (int)(row % py), z = (int)(row / py);
            const int sy = map_src(map.y, y), sz = map_src(map.z, z);
            if (sy >= 0 && sz >= 0) off = (long long)map.x.n * (sy + (long long)map.y.n * sz);
        }
        rowoff[tid] = off;
    }
    for (int i = tid; i < M; i += T) tw[i] = twg[i];
    __syncthreads();
    const bool even_rows = (map.x.n & 1) == 0 && map.x.mode == 0;
    for (int e = tid; e < NR * M; e += T) {
        const int r = e / M, n = e - r * M;
        const long long off = rowoff[r];
        float2 v = make_float2(0.f, 0.f);
        if (off >= 0) {
            const float* __restrict__ srow = src + off;
            if (even_rows && 2 * n + 1 < map.x.n) {
                v = *reinterpret_cast<const float2*>(srow + 2 * n);      // interior: identity map, 8-B aligned
            } else {
                const int s0 = map_src(map.x, 2 * n), s1 = map_src(map.x, 2 * n + 1);
                v.x = s0 >= 0 ? srow[s0] : 0.f;
                v.y = s1 >= 0 ? srow[s1] : 0.f;
            }
        }
        buf[r * LP + n] = v;
    }
    __syncthreads();
    PLAN::template run<NR, T>(buf, tw, tid);
    // X[k] = (Z[k] + conj Z[M-k])/2 - i/2 * w_P^k * (Z[k] - conj Z[M-k]),  k = 0..M  (twx[k] = w_P^k)
    for (int e = tid; e < NR * hxp; e += T) {
        const int r = e / hxp, k = e - r * hxp;
        const long long row = row0 + r;
        if (row >= rows) continue;
        float2 out = make_float2(0.f, 0.f);
        if (k <= M) {
            const float2 zk = buf[r * LP + (k == M ? 0 : k)];
            const float2 zm = cconj(buf[r * LP + (k == 0 ? 0 : M - k)]);
            const float2 sm = cadd(zk, zm), d = csub(zk, zm);
            const float2 t = cmul(twx[k], d);            // w^k * d
            out = make_float2(0.5f * (sm.x + t.y), 0.5f * (sm.y - t.x));   // sm/2 - (i/2) t
        }
        dst[row * hxp + k] = out;
    }
}

// E: half spectrum rows -> real rows, cropped to nx, scaled; one partial sum (double) per block.
// Persistent blocks: each walks row groups blockIdx.x, blockIdx.x + gridDim.x, ...
template <class PLAN>
__global__ __launch_bounds__(Cfg<PLAN::len>::T) void k_fft_x_c2r(const float2* __restrict__ srcc,
                                                                float* __restrict__ out,
                                                                const float2* __restrict__ twg,
                                                                const float2* __restrict__ twx, int hxp, int py,
                                                                int nx, int ny, long long rows, float scale,
                                                                double* __restrict__ partial)
{
    constexpr int M = PLAN::len;
    using C = Cfg<M>;
    constexpr int NR = C::NL, LP = C::LP, T = C::T;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw = lds + NR * LP;
    double* red = reinterpret_cast<double*>(lds + NR * LP + M);              // T/64 <= 16 doubles
    long long* rowoff = reinterpret_cast<long long*>(red + 16);              // NR entries
    const int tid = threadIdx.x;
    for (int i = tid; i < M; i += T) tw[i] = twg[i];
    const long long ngroups = (rows + NR - 1) / NR;
    double acc = 0.0;
    for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long long row0 = grp * NR;                   // output rows (y < ny, z < nz)
        __syncthreads();                                   // previous group's buf / rowoff fully consumed
        if (tid < NR) {
            const long long row = row0 + tid;
            long long off = -1;
            if (row < rows) {
                const int y = (int)(row % ny), z = (int)(row / ny);
                off = ((long long)z * py + y) * hxp;
            }
            rowoff[tid] = off;
        }
        __syncthreads();
        // Z[k] = (X[k] + conj X[M-k]) + i w_P^{-k} (X[k] - conj X[M-k]); inverse FFT via conj trick: stage conj(Z)
        for (int e = tid; e < NR * M; e += T) {
            const int r = e / M, k = e - r * M;
            const long long off = rowoff[r];
            float2 v = make_float2(0.f, 0.f);
            if (off >= 0) {
                const float2* __restrict__ sp = srcc + off;
                const float2 a = sp[k], b = cconj(sp[M - k]);
                const float2 sm = cadd(a, b), d = csub(a, b);
                const float2 t = cmul(cconj(twx[k]), d);      // w^{-k} d
                v = make_float2(sm.x - t.y, -(sm.y + t.x));    // conj(sm + i t)
            }
            buf[r * LP + k] = v;
        }
        __syncthreads();
        PLAN::template run<NR, T>(buf, tw, tid);
        // z[n] = conj(buf[n]) = x[2n] + i x[2n+1]
        for (int e = tid; e < NR * M; e += T) {
            const int r = e / M, n = e - r * M;
            const long long row = row0 + r;
            if (row >= rows || 2 * n >= nx) continue;
            const float2 vv = buf[r * LP + n];
            const float x0 = vv.x * scale, x1 = -vv.y * scale;
            float* __restrict__ o = out + row * nx + 2 * n;
            if (2 * n + 1 < nx) {
                if ((nx & 1) == 0) *reinterpret_cast<float2*>(o) = make_float2(x0, x1);
                else { o[0] = x0; o[1] = x1; }
                acc += (double)x0 + (double)x1;
            } else {
                o[0] = x0;
                acc += (double)x0;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (int w = 0; w < T / 64; ++w) sum += red[w];
        partial[blockIdx.x] = sum;
    }
}

__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partial, long long count,
                                                         double* __restrict__ scal)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < count; i += 256) acc += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scal[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---------------------------------------------------------------------------------- size table
#define MVSIM_FFT_SIZES(X) \
    X(16, 4, 4)            \
    X(18, 9, 2)            \
    X(24, 3, 8)            \
    X(32, 4, 8)            \
    X(36, 9, 4)            \
    X(48, 3, 4, 4)         \
    X(64, 8, 8)            \
    X(72, 9, 8)            \
    X(96, 3, 8, 4)         \
    X(128, 4, 8, 4)        \
    X(144, 9, 4, 4)        \
    X(192, 3, 8, 8)        \
    X(256, 4, 8, 8)        \
    X(288, 9, 8, 4)        \
    X(384, 3, 8, 4, 4)     \
    X(512, 8, 8, 8)        \
    X(576, 9, 8, 8)        \
    X(768, 3, 8, 8, 4)     \
    X(1024, 4, 8, 8, 4)    \
    X(1152, 9, 8, 4, 4)

static const int kSizes[] = {
#define X(L, ...) L,
    MVSIM_FFT_SIZES(X)
#undef X
};

template <class K> static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024)
        MVSIM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MVSIM_OK;
}

template <class PLAN>
static int launch_lines_t(hipStream_t s, int mode, bool sparse, const LinesArgs& a, int tiles, int nouter)
{
    using C = Cfg<PLAN::len>;
    dim3 grid(tiles, nouter), block(C::T);
#define MVSIM_LL(MODE_, SP_)                                                                 \
    do {                                                                                     \
        MVSIM_TRY(set_lds(k_fft_lines<PLAN, MODE_, SP_>, C::LDS));                           \
        hipLaunchKernelGGL((k_fft_lines<PLAN, MODE_, SP_>), grid, block, C::LDS, s, a);      \
    } while (0)
    if (mode == FWD && sparse) MVSIM_LL(FWD, true);
    else if (mode == FWD) MVSIM_LL(FWD, false);
    else if (mode == INV) MVSIM_LL(INV, false);
    else MVSIM_LL(CONV, false);
#undef MVSIM_LL
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

template <class PLAN>
static int launch_r2c_t(hipStream_t s, const float* src, const SrcMap& map, float2* dst, const float2* tw,
                        const float2* twx, int hxp, long long rows)
{
    using C = Cfg<PLAN::len>;
    const long long blocks = (rows + C::NL - 1) / C::NL;
    MVSIM_TRY(set_lds(k_fft_x_r2c<PLAN>, C::LDS));
    hipLaunchKernelGGL((k_fft_x_r2c<PLAN>), dim3((unsigned)blocks), dim3(C::T), C::LDS, s, src, map, dst, tw, twx, hxp, rows);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

constexpr int C2R_MAX_BLOCKS = 2048;

template <class PLAN>
static int launch_c2r_t(hipStream_t s, const float2* srcc, float* out, const float2* tw, const float2* twx, int hxp,
                        int py, int nx, int ny, long long rows, float scale, double* partial, int* nblocks)
{
    using C = Cfg<PLAN::len>;
    const long long groups = (rows + C::NL - 1) / C::NL;
    const int blocks = (int)(groups < C2R_MAX_BLOCKS ? groups : C2R_MAX_BLOCKS);
    *nblocks = blocks;
    MVSIM_TRY(set_lds(k_fft_x_c2r<PLAN>, C::LDS));
    hipLaunchKernelGGL((k_fft_x_c2r<PLAN>), dim3((unsigned)blocks), dim3(C::T), C::LDS, s, srcc, out, tw, twx, hxp, py, nx, ny,
                       rows, scale, partial);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

static int launch_lines(hipStream_t s, int L, int mode, bool sparse, const LinesArgs& a, int tiles, int nouter)
{
    switch (L) {
#define X(LL, ...) \
    case LL: return launch_lines_t<Plan<LL, __VA_ARGS__>>(s, mode, sparse, a, tiles, nouter);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("custom FFT: unsupported length %d", L);
    return MVSIM_EINVAL;
}

static int launch_r2c(hipStream_t s, int M, const float* src, const SrcMap& map, float2* dst, const float2* tw,
                      const float2* twx, int hxp, long long rows)
{
    switch (M) {
#define X(LL, ...) \
    case LL: return launch_r2c_t<Plan<LL, __VA_ARGS__>>(s, src, map, dst, tw, twx, hxp, rows);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("custom FFT: unsupported half length %d", M);
    return MVSIM_EINVAL;
}

static int launch_c2r(hipStream_t s, int M, const float2* srcc, float* out, const float2* tw, const float2* twx, int hxp,
                      int py, int nx, int ny, long long rows, float scale, double* partial, int* nblocks)
{
    switch (M) {
#define X(LL, ...) \
    case LL: return launch_c2r_t<Plan<LL, __VA_ARGS__>>(s, srcc, out, tw, twx, hxp, py, nx, ny, rows, scale, partial, nblocks);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("custom FFT: unsupported half length %d", M);
    return MVSIM_EINVAL;
}

static int lines_per_tile(int L) { return L <= 576 ? 16 : 8; }

static int pick_size(int64_t need)
{
    for (int v : kSizes)
        if (v >= need) return v;
    return 0;
}

}  // namespace fft

// Padded sizes for the custom path, or false if some dimension has no supported size.
bool custom_fft_sizes(const int64_t dim[3], const int64_t kdim[3], int64_t P[3])
{
    if (const char* e = getenv("MVSIM_FFT_BACKEND"))
        if (std::strcmp(e, "rocfft") == 0) return false;
    for (int d = 0; d < 3; ++d) {
        const int64_t need = dim[d] + kdim[d] - 1;
        if (d == 0) {
            const int m = fft::pick_size((need + 1) / 2);
            if (!m) return false;
            P[0] = 2 * (int64_t)m;
        } else {
            const int v = fft::pick_size(need);
            if (!v) return false;
            P[d] = v;
        }
    }
    return true;
}

static int ensure_twiddles(mvsim_ctx* ctx, int L, const float2** out)
{
    auto it = ctx->twiddles.find(L);
    if (it != ctx->twiddles.end()) { *out = reinterpret_cast<const float2*>(it->second); return MVSIM_OK; }
    std::vector<float2> h((size_t)L + 1);
    for (int k = 0; k <= L; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)L;
        h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    void* d = nullptr;
    MVSIM_HIP(hipMalloc(&d, h.size() * sizeof(float2)));
    // synchronous copy from a temporary: happens once per length per context
    hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); set_error("twiddle upload failed: %s", hipGetErrorString(e)); return MVSIM_EHIP; }
    ctx->twiddles[L] = d;
    *out = reinterpret_cast<const float2*>(d);
    return MVSIM_OK;
}

void custom_fft_release(mvsim_ctx* ctx)
{
    for (auto& kv : ctx->twiddles) (void)hipFree(kv.second);
    ctx->twiddles.clear();
    ctx->cfft_f.release();
    ctx->cfft_g.release();
    ctx->cfft_g1.release();
    ctx->cfft_g2.release();
}

static void ev_begin(mvsim_ctx* ctx, int st)
{
    if (ctx->timing) (void)hipEventRecord(ctx->ev[st][0], ctx->stream);
}
static void ev_end(mvsim_ctx* ctx, int st)
{
    if (ctx->timing) { (void)hipEventRecord(ctx->ev[st][1], ctx->stream); ctx->ev_used[st] = true; }
}

int custom_fft_convolve(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                        const int64_t kdim[3], const int64_t P[3], float* out)
{
    using namespace fft;
    const int px = (int)P[0], py = (int)P[1], pz = (int)P[2];
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    const int M = px / 2;
    const int tile_y = lines_per_tile(py), tile_z = lines_per_tile(pz);
    const int tw_max = tile_y > tile_z ? tile_y : tile_z;
    const int hxp = ((M + 1 + tw_max - 1) / tw_max) * tw_max;
    const size_t cbytes = (size_t)hxp * py * pz * sizeof(float2);
    MVSIM_TRY(ctx->cfft_f.reserve(cbytes));
    MVSIM_TRY(ctx->cfft_g.reserve(cbytes));
    // compact PSF intermediates: G1 [kz][ky][hxp] (x transformed), G2 [kz][py][hxp] (x,y transformed)
    MVSIM_TRY(ctx->cfft_g1.reserve((size_t)hxp * ky * kz * sizeof(float2)));
    MVSIM_TRY(ctx->cfft_g2.reserve((size_t)hxp * py * kz * sizeof(float2)));
    MVSIM_TRY(ctx->partials.reserve((size_t)(SUM_BLOCKS + 8) * sizeof(double)));
    MVSIM_TRY(ctx->partials_e.reserve((size_t)C2R_MAX_BLOCKS * sizeof(double)));
    double* scal = ctx->partials.as<double>() + SUM_BLOCKS;

    const float2 *tw_m, *tw_px, *tw_py, *tw_pz;
    MVSIM_TRY(ensure_twiddles(ctx, M, &tw_m));
    MVSIM_TRY(ensure_twiddles(ctx, px, &tw_px));
    MVSIM_TRY(ensure_twiddles(ctx, py, &tw_py));
    MVSIM_TRY(ensure_twiddles(ctx, pz, &tw_pz));

    float2* F = ctx->cfft_f.as<float2>();
    float2* G = ctx->cfft_g.as<float2>();
    float2* G1 = ctx->cfft_g1.as<float2>();
    float2* G2 = ctx->cfft_g2.as<float2>();
    hipStream_t s = ctx->stream;
    const long long rows_all = (long long)py * pz;
    const long long rows_out = (long long)dim[1] * dim[2];
    const long long plane = (long long)hxp * py;
    const DimMap ident_none = DimMap{0, 0, 0, 0, 1, 0};

    // ---- PSF spectrum G.  The embedded kernel is zero outside Kx*Ky*Kz taps, so the x pass runs on the
    //      Ky*Kz non-zero rows only, the y pass on the Kz non-zero planes only (sparse loads), and the z
    //      pass expands Kz planes to the full spectrum.
    ev_begin(ctx, ST_PSF);
    {
        SrcMap m;
        m.x = DimMap{kx, px, kx - kx / 2, kx / 2, 1, kx / 2};   // embed along x with wrap-around
        m.y = DimMap{ky, ky, ky, 0, 1, 0};                       // compact: identity
        m.z = DimMap{kz, kz, kz, 0, 1, 0};
        MVSIM_TRY(launch_r2c(s, M, psf, m, G1, tw_m, tw_px, hxp, (long long)ky * kz));
        LinesArgs a{};
        a.src = G1; a.dst = G2; a.spec = nullptr; a.tw = tw_py;
        a.src_es = hxp; a.src_outer = (long long)hxp * ky;          // per kz plane
        a.dst_es = hxp; a.dst_outer = plane;
        a.lmap = DimMap{ky, py, ky - ky / 2, ky / 2, 1, ky / 2};
        MVSIM_TRY(launch_lines(s, py, FWD, true, a, hxp / tile_y, kz));
        LinesArgs c{};
        c.src = G2; c.dst = G; c.spec = nullptr; c.tw = tw_pz;
        c.src_es = plane; c.src_outer = hxp;                         // per ky row
        c.dst_es = plane; c.dst_outer = hxp;
        c.lmap = DimMap{kz, pz, kz - kz / 2, kz / 2, 1, kz / 2};
        MVSIM_TRY(launch_lines(s, pz, FWD, true, c, hxp / tile_z, py));
    }
    ev_end(ctx, ST_PSF);

    // ---- image: A, B, C (with product), D, E
    ev_begin(ctx, ST_CONVOLVE);
    {
        SrcMap m;
        const int n[3] = {(int)dim[0], (int)dim[1], (int)dim[2]};
        const int Pd[3] = {px, py, pz};
        DimMap* dm[3] = {&m.x, &m.y, &m.z};
        for (int d = 0; d < 3; ++d) {
            const int c = (int)(kdim[d] / 2);
            const int left = (int)(kdim[d] - 1 - kdim[d] / 2);
            *dm[d] = DimMap{n[d], Pd[d], n[d] + c, left, 0, 0};
        }
        MVSIM_TRY(launch_r2c(s, M, img, m, F, tw_m, tw_px, hxp, rows_all));
        LinesArgs b{};
        b.src = F; b.dst = F; b.tw = tw_py; b.src_es = b.dst_es = hxp; b.src_outer = b.dst_outer = plane;
        b.lmap = ident_none;
        MVSIM_TRY(launch_lines(s, py, FWD, false, b, hxp / tile_y, pz));
        LinesArgs c{};
        c.src = F; c.dst = F; c.spec = G; c.tw = tw_pz;
        c.src_es = c.dst_es = c.spec_es = plane; c.src_outer = c.dst_outer = c.spec_outer = hxp;
        c.lmap = ident_none;
        MVSIM_TRY(launch_lines(s, pz, CONV, false, c, hxp / tile_z, py));
        b.tw = tw_py;
        MVSIM_TRY(launch_lines(s, py, INV, false, b, hxp / tile_y, (int)dim[2]));   // planes z >= Nz are never read
        const float scale = (float)(1.0 / ((double)px * (double)py * (double)pz));
        int nblk = 0;
        MVSIM_TRY(launch_c2r(s, M, F, out, tw_m, tw_px, hxp, py, (int)dim[0], (int)dim[1], rows_out, scale,
                             ctx->partials_e.as<double>(), &nblk));
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, ctx->partials_e.as<double>(), (long long)nblk, scal);
        MVSIM_HIP(hipGetLastError());
    }
    ev_end(ctx, ST_CONVOLVE);
    return MVSIM_OK;
}

}  // namespace mvsim
