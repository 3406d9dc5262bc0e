// LDS-tiled direct 3-D convolution (north_star: "an LDS-tiled direct stencil for small PSFs"; BASELINE configs[4] names it
// for the measured 63^3 PSF; SimulateMultiViewDataset.java:579 loads 51^3 stacks) and the FFT-independent cross-check of
// SimulateMultiViewDataset.convolve (:253-264: mirror-single image boundary, kernel centre K/2, no flip).  Bound by fp32
// vector FMA throughput -- 2 Kx Ky Kz flop per voxel against 8 B per voxel -- not by HBM; DESIGN.md section 4.6.
//
// Block = 256 threads (4 waves), output tile 32 x 16 x 8.  A lane owns 8 consecutive x outputs of TWO rows, y and y + 8, as
// the halves of 8 packed accumulators.  The LDS tile stores float2 {row iy, row iy + 8}: one ds_read_b64 delivers the
// operand pair of a v_pk_fma_f32 whose tap -- wave-uniform, read by scalar loads from a copy of the PSF reversed along x
// and zero-padded to the chunk length -- is shared by both halves.  Per PSF row a lane reads the 8 + kxc - 1 pairs its
// outputs touch once and runs kxc x 8 packed FMAs on them (no duplicated LDS reads, no register moves: 512 v_pk_fma_f32
// against 36 ds_read2_b64 for kxc = 64).  The PSF is cut into chunks of kxc x kyc x kzc taps (kxc = 4 NG, one template
// instance per NG) so that the tile with its mirror halo, 8 B x (32 + kxc - 1) x (7 + kyc) x (7 + kzc), fits HALF the CU's
// LDS for ANY PSF up to 64 taps per axis: two blocks per CU, one fills its tile (4 rows in flight per wave) while the other
// computes.  fp32 sums inside a chunk, fp64 across chunks (the oracle's direct sum is fp64; 1e-5 range-normalised is the
// contract, measured 2e-7).
#include "common.h"

namespace mvsim {

namespace {
constexpr int TX = 32, R = 8;
constexpr int MAXK = 64;

__device__ __forceinline__ int mirror_i(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// prepared PSF: [kz][ky][kxp], prep[j] = psf[kx-1-j] for j < kx, 0 beyond (one extra zero row at the end is never read)
__global__ void k_stencil_prep(const float* __restrict__ psf, float* __restrict__ prep, int kx, int ky, int kz, int kxp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = kxp * ky * kz;
    if (i >= total) return;
    const int j = i % kxp, row = i / kxp;
    prep[i] = j < kx ? psf[(kx - 1 - j) + kx * row] : 0.0f;
}

typedef void (*stencil_fn_t)(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int);

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int PTY = 16, PTZ = 8;

__device__ __forceinline__ int reflect_i(int i, int n)      // mirror-single for -n < i < 2n - 1
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

template <int NG, bool NEAR>   // NEAR: every halo index, tile overhang included, is one reflection away (N >= tile + K per axis)
__global__ __launch_bounds__(256, 2) void k_stencil_pair(const float* __restrict__ img, const float* __restrict__ prep,
                                                         float* __restrict__ out, int nx, int ny, int nz, int kx, int ky, int kz,
                                                         int kyc, int kzc, int S2, int Hp)
{
    constexpr int KXP = 4 * NG;                      // taps of one x chunk of the PSF
    constexpr int WIN = R + KXP - 1;                 // float2 values of its LDS row a lane touches
    constexpr int W = TX + KXP - 1;                  // tile width
    constexpr int U = 4;                             // tile rows a wave requests before it stores any of them
    extern __shared__ __align__(16) float tile[];
    v2f* __restrict__ tile2 = reinterpret_cast<v2f*>(tile);
    const int tx = threadIdx.x & 3, ty = (threadIdx.x >> 2) & 7, tz = threadIdx.x >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * PTY, z0 = blockIdx.z * PTZ;
    const int cx = kx / 2, cy = ky / 2, cz = kz / 2;
    const int kxt = (kx + KXP - 1) / KXP * KXP;      // row length of the prepared PSF

    v2f accs[R];
    double accd[2][R];
#pragma unroll
    for (int r = 0; r < R; ++r) { accd[0][r] = 0.0; accd[1][r] = 0.0; }

    for (int a0 = 0; a0 < kxt; a0 += KXP) {
        // reversed tap index j = kx-1-a in [a0, a0+KXP): output xo reads tile column xo + (j - a0)
        const int gx0 = x0 - (kx - 1 - cx) + a0;
        const bool x_inside = gx0 >= 0 && gx0 + W <= nx;
        const bool one = lane < W, two = lane + 64 < W;
        const int l0 = one ? lane : W - 1;
        const int sx0 = x_inside ? gx0 + l0 : (NEAR ? reflect_i(gx0 + l0, nx) : mirror_i(gx0 + l0, nx));
        const int sx1 = !two ? sx0 : (x_inside ? gx0 + lane + 64 : (NEAR ? reflect_i(gx0 + lane + 64, nx) : mirror_i(gx0 + lane + 64, nx)));
        for (int c0 = 0; c0 < kz; c0 += kzc) {
            const int cn = (kz - c0) < kzc ? (kz - c0) : kzc;
            const int D = PTZ + cn - 1;
            const int gz0 = z0 - (c0 + cn - 1 - cz);
            for (int b0 = 0; b0 < ky; b0 += kyc) {
                const int bn = (ky - b0) < kyc ? (ky - b0) : kyc;
                const int Hn = 8 + bn - 1;           // pair rows in use; input rows: Hn + 8
                const int rin = Hn + 8;
                const int gy0 = y0 - (b0 + bn - 1 - cy);
                const int rows = rin * D;
                __syncthreads();
                for (int q0 = wave * U; q0 < rows; q0 += 4 * U) {
                    float v0[U], v1[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int q = (q0 + u < rows) ? q0 + u : rows - 1;
                        const int iy = q % rin, iz = q / rin;
                        const int sy = NEAR ? reflect_i(gy0 + iy, ny) : mirror_i(gy0 + iy, ny);
                        const int sz = NEAR ? reflect_i(gz0 + iz, nz) : mirror_i(gz0 + iz, nz);
                        const float* __restrict__ src = img + (long long)nx * (sy + (long long)ny * sz);
                        v0[u] = src[sx0];
                        v1[u] = src[sx1];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int q = q0 + u;
                        if (q < rows) {
                            const int iy = q % rin, iz = q / rin;
                            if (iy < Hn) {           // .x of pair row iy
                                float* __restrict__ d = tile + 2 * S2 * (iy + Hp * iz);
                                if (one) d[2 * lane] = v0[u];
                                if (two) d[2 * (lane + 64)] = v1[u];
                            }
                            if (iy >= 8) {           // .y of pair row iy - 8
                                float* __restrict__ d = tile + 2 * S2 * ((iy - 8) + Hp * iz) + 1;
                                if (one) d[2 * lane] = v0[u];
                                if (two) d[2 * (lane + 64)] = v1[u];
                            }
                        }
                    }
                }
                __syncthreads();

#pragma unroll
                for (int r = 0; r < R; ++r) accs[r] = v2f{0.f, 0.f};
                for (int cl = 0; cl < cn; ++cl) {
                    const int t = tz + (cn - 1 - cl);
                    v2f acc[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = v2f{0.f, 0.f};
                    for (int bl = 0; bl < bn; ++bl) {
                        const v2f* __restrict__ row = tile2 + S2 * ((ty + (bn - 1 - bl)) + Hp * t) + tx * R;
                        const float* __restrict__ prow = prep + a0 + (long long)kxt * ((b0 + bl) + (long long)ky * (c0 + cl));
                        float w[KXP];
#pragma unroll
                        for (int j = 0; j < KXP; ++j) w[j] = prow[j];
                        v2f win[WIN];
#pragma unroll
                        for (int i = 0; i < WIN; ++i) win[i] = row[i];
#pragma unroll
                        for (int j = 0; j < KXP; ++j) {
#pragma unroll
                            for (int r = 0; r < R; ++r) acc[r] = __builtin_elementwise_fma(v2f{w[j], w[j]}, win[j + r], acc[r]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) accs[r] += acc[r];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) { accd[0][r] += (double)accs[r].x; accd[1][r] += (double)accs[r].y; }
            }
        }
    }

    const int z = z0 + tz;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int y = y0 + ty + 8 * h;
        if (y < ny && z < nz) {
            float* __restrict__ o = out + (long long)nx * (y + (long long)ny * z);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int x = x0 + tx * R + r;
                if (x < nx) o[x] = (float)accd[h][r];
            }
        }
    }
}

template <bool NEAR, int... I>
constexpr std::array<stencil_fn_t, sizeof...(I)> pair_table(std::integer_sequence<int, I...>)
{
    return {{k_stencil_pair<I + 1, NEAR>...}};
}

}  // namespace

// pair form: the PSF is cut into chunks of (4 ng) x kyc x kzc taps; LDS = 8 B * S2 * (7 + kyc) * (7 + kzc) with
// S2 = (32 + 4 ng - 1) | 1.  The chunk that minimises the tile fills (elements moved + barriers) under half the CU's LDS.
static bool pair_geometry(const int64_t kdim[3], int g[5])
{
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    if (kx < 1 || ky < 1 || kz < 1 || kx > MAXK || ky > MAXK || kz > MAXK) return false;
    const size_t budget = 78 * 1024;
    double best = 1e300;
    int bng = 0, bkyc = 0, bkzc = 0;
    for (int ng = 1; ng <= (kx + 3) / 4; ++ng) {
        const int kxp = 4 * ng, W = TX + kxp - 1, S2 = W | 1;
        const int xchunks = (kx + kxp - 1) / kxp;
        for (int kyc = 1; kyc <= ky; ++kyc)
            for (int kzc = 1; kzc <= kz; ++kzc) {
                const size_t bytes = (size_t)8 * S2 * (7 + kyc) * (PTZ - 1 + kzc);
                if (bytes > budget) break;
                const double chunks = (double)xchunks * ((ky + kyc - 1) / kyc) * ((kz + kzc - 1) / kzc);
                // per chunk: a fill of (15 + kyc)(7 + kzc) rows by 4 waves, 4 rows in flight per wave (~2500 cycles per trip of
                // latency, ~40 issue cycles per row) and two barriers; per PSF row ~150 cycles of set-up beside the FMAs;
                // the zero taps of the padded last x chunk are wasted FMAs (4 cycles per 16 of them)
                const double rows = (double)(15 + kyc) * (PTZ - 1 + kzc);
                const double fill = rows / 16.0 * 2500.0 + rows / 4.0 * 40.0 + 3000.0;
                const double cost = chunks * fill + (double)xchunks * ky * kz * (150.0 + kxp * 8 / 2 * 4.0);
                if (cost < best) { best = cost; bng = ng; bkyc = kyc; bkzc = kzc; }
            }
    }
    if (!bng) return false;
    g[0] = bng; g[1] = bkyc; g[2] = bkzc; g[3] = (TX + 4 * bng - 1) | 1; g[4] = 7 + bkyc;
    return true;
}

int launch_stencil(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                   const int64_t kdim[3], float* out)
{
    hipStream_t s = ctx->stream;
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    int g[5];
    if (!pair_geometry(kdim, g)) {
        set_error("direct stencil: PSF %dx%dx%d outside 1..%d taps per axis; use the FFT method", kx, ky, kz, MAXK);
        return MVSIM_EINVAL;
    }
    const int ng = g[0], kyc = g[1], kzc = g[2], S2 = g[3], Hp = g[4];
    const int kxt = (kx + 4 * ng - 1) / (4 * ng) * (4 * ng);
    const size_t lds = (size_t)8 * S2 * Hp * (PTZ - 1 + kzc);
    MVSIM_TRY(ctx->stencil_psf.reserve((size_t)kxt * ky * kz * sizeof(float)));
    float* prep = ctx->stencil_psf.as<float>();
    const int total = kxt * ky * kz;
    hipLaunchKernelGGL(k_stencil_prep, dim3((total + 255) / 256), dim3(256), 0, s, psf, prep, kx, ky, kz, kxt);
    static constexpr auto near_t = pair_table<true>(std::make_integer_sequence<int, MAXK / 4>{});
    static constexpr auto far_t = pair_table<false>(std::make_integer_sequence<int, MAXK / 4>{});
    // one reflection reaches every halo index, tile overhang and padded taps included, when N >= tile + K (+ pad in x)
    const bool near = nx >= TX + kx + 4 * ng && ny >= PTY + ky && nz >= PTZ + kz;
    const stencil_fn_t fn = near ? near_t[ng - 1] : far_t[ng - 1];
    MVSIM_TRY(ensure_lds_attr(ctx, reinterpret_cast<const void*>(fn), lds));
    dim3 grid((nx + TX - 1) / TX, (ny + PTY - 1) / PTY, (nz + PTZ - 1) / PTZ);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, img, prep, out, nx, ny, nz, kx, ky, kz, kyc, kzc, S2, Hp);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// {taps of an x chunk, kyc, kzc, LDS bytes per block, blocks per CU}: what tools/stencil_bench.py and DESIGN.md report
bool stencil_chunk_geometry(const int64_t kdim[3], int64_t out[5])
{
    int g[5];
    if (!pair_geometry(kdim, g)) return false;
    out[0] = 4 * g[0]; out[1] = g[1]; out[2] = g[2];
    out[3] = (int64_t)8 * g[3] * g[4] * (PTZ - 1 + g[2]); out[4] = 2;
    return true;
}

}  // namespace mvsim
