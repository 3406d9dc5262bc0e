// LDS-tiled direct 3-D convolution (north_star: "an LDS-tiled direct stencil for small PSFs"; BASELINE configs[4] names it
// for the measured 63^3 PSF) and the FFT-independent cross-check of SimulateMultiViewDataset.convolve (:253-264: mirror-single
// image boundary, kernel centre K/2, no flip).  Bound by fp32 vector FMA throughput -- 2*Kx*Ky*Kz flop per voxel against
// 8 B per voxel -- not by HBM; see DESIGN.md.
//
// Block = 256 threads (4 waves) -> output tile 32 x 8 x 8; a lane owns R = 8 consecutive x outputs (8 fp32 accumulators).
// The PSF is cut into (y, z) chunks of kyc x kzc rows so that the image tile with its mirror halo,
// (32 + kxp - 1) x (8 + kyc - 1) x (8 + kzc - 1) floats, fits HALF the CU's LDS for ANY PSF up to 64 taps per axis (two blocks
// per CU = two waves per SIMD: one wave alone issues a v_fma_f32 every 4 cycles, half the pipe's rate).  For one PSF row
// (ky, kz) a lane holds the 8 + kxp values of its LDS row that its 8 outputs touch in registers and runs kxp x 8 FMAs on
// them; the taps are wave-uniform, read from a prepared copy of the PSF (reversed along x and zero-padded to a multiple of
// 4 taps, so that a group of taps is one aligned scalar load) straight into SGPRs -- the FMA's scalar operand.  The row
// loop is fully unrolled per template instance (NG = groups of 4 taps).  fp32 partial sums per PSF row chunk, fp64 across
// chunks (the oracle's direct sum is fp64; 1e-5 range-normalised is the contract).
#include "common.h"

namespace mvsim {

namespace {
constexpr int TX = 32, TY = 8, TZ = 8, R = 8;
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

template <int NG>
__global__ __launch_bounds__(256, 2) void k_stencil(const float* __restrict__ img, const float* __restrict__ prep,
                                                    float* __restrict__ out, int nx, int ny, int nz, int kx, int ky, int kz,
                                                    int kyc, int kzc, int S, int H)
{
    constexpr int KXP = 4 * NG;
    constexpr int WIN = R + KXP;                     // values of its row a lane touches (the last one only with a zero tap)
    extern __shared__ __align__(16) float tile[];
    const int tx = threadIdx.x & 3, ty = (threadIdx.x >> 2) & 7, tz = threadIdx.x >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;
    const int cx = kx / 2, cy = ky / 2, cz = kz / 2;
    const int W = TX + KXP - 1;                      // tile width actually filled (>= TX + kx - 1)
    const int gx0 = x0 - (kx - 1 - cx);
    const bool x_inside = gx0 >= 0 && gx0 + W <= nx;

    double accd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) accd[r] = 0.0;

    for (int c0 = 0; c0 < kz; c0 += kzc) {
        const int cn = (kz - c0) < kzc ? (kz - c0) : kzc;   // PSF planes c0 .. c0+cn-1
        const int D = TZ + cn - 1;
        // out[z] needs img[z - (c - cz)] for c in [c0, c0+cn): tile plane t <-> gz = z0 - (c0+cn-1-cz) + t
        const int gz0 = z0 - (c0 + cn - 1 - cz);
        for (int b0 = 0; b0 < ky; b0 += kyc) {
            const int bn = (ky - b0) < kyc ? (ky - b0) : kyc;
            const int Hn = TY + bn - 1;
            const int gy0 = y0 - (b0 + bn - 1 - cy);
            __syncthreads();
            // tile fill: one (y, z) row per wave and trip, its mirrored source row wave-uniform
            for (int q = wave; q < Hn * D; q += 4) {
                const int iy = q % Hn, iz = q / Hn;
                const int sy = mirror_i(gy0 + iy, ny), sz = mirror_i(gz0 + iz, nz);
                const float* __restrict__ src = img + (long long)nx * (sy + (long long)ny * sz);
                float* __restrict__ dst = tile + S * (iy + H * iz);
                if (x_inside) {
                    for (int ix = lane; ix < W; ix += 64) dst[ix] = src[gx0 + ix];
                } else {
                    for (int ix = lane; ix < W; ix += 64) dst[ix] = src[mirror_i(gx0 + ix, nx)];
                }
            }
            __syncthreads();

            for (int cl = 0; cl < cn; ++cl) {
                // img plane for output tz and PSF plane c = c0+cl: gz = z0+tz-(c-cz) -> t = tz + (cn-1-cl)
                const int t = tz + (cn - 1 - cl);
                float acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = 0.0f;
                for (int bl = 0; bl < bn; ++bl) {
                    // row for output ty and PSF row b = b0+bl: gy = y0+ty-(b-cy) -> iy = ty + (bn-1-bl)
                    const float* __restrict__ row = tile + S * ((ty + (bn - 1 - bl)) + H * t) + tx * R;
                    const float* __restrict__ prow = prep + (long long)KXP * ((b0 + bl) + (long long)ky * (c0 + cl));
                    float w[KXP];
#pragma unroll
                    for (int j = 0; j < KXP; ++j) w[j] = prow[j];
                    float win[WIN];
#pragma unroll
                    for (int i = 0; i < WIN - 1; ++i) win[i] = row[i];
#pragma unroll
                    for (int j = 0; j < KXP; ++j) {
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = fmaf(w[j], win[j + r], acc[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) accd[r] += (double)acc[r];
            }
        }
    }

    const int y = y0 + ty, z = z0 + tz;
    if (y < ny && z < nz) {
        float* __restrict__ o = out + (long long)nx * (y + (long long)ny * z);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int x = x0 + tx * R + r;
            if (x < nx) o[x] = (float)accd[r];
        }
    }
}

typedef void (*stencil_fn)(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int);
template <int... I>
constexpr std::array<stencil_fn, sizeof...(I)> stencil_table(std::integer_sequence<int, I...>)
{
    return {{k_stencil<I + 1>...}};
}
}  // namespace

// The (y, z) chunk of the PSF per tile: the pair that minimises tile fills + barriers under the LDS budget.
bool stencil_geometry(const int64_t kdim[3], int g[5])
{
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    if (kx < 1 || ky < 1 || kz < 1 || kx > MAXK || ky > MAXK || kz > MAXK) return false;
    const int ng = (kx + 3) / 4, kxp = 4 * ng;
    const int W = TX + kxp - 1;
    const int S = W | 1;                                  // odd row pitch: the 4 x 8 lanes of a half wave hit 32 different banks
    const size_t budget = 78 * 1024;                      // two blocks per CU (160 KB LDS)
    double best = 1e300;
    int bkyc = 0, bkzc = 0;
    for (int kyc = 1; kyc <= ky; ++kyc)
        for (int kzc = 1; kzc <= kz; ++kzc) {
            const size_t bytes = (size_t)S * (TY + kyc - 1) * (TZ + kzc - 1) * sizeof(float);
            if (bytes > budget) break;
            const double chunks = (double)((ky + kyc - 1) / kyc) * ((kz + kzc - 1) / kzc);
            // per chunk: tile elements per thread (~12 cycles each with the mirror arithmetic) + two barriers
            const double cost = chunks * ((double)S * (TY + kyc - 1) * (TZ + kzc - 1) / 256.0 * 12.0 + 3000.0);
            if (cost < best) { best = cost; bkyc = kyc; bkzc = kzc; }
        }
    if (!bkyc) return false;
    g[0] = ng; g[1] = bkyc; g[2] = bkzc; g[3] = S; g[4] = TY + bkyc - 1;
    return true;
}

int launch_stencil(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                   const int64_t kdim[3], float* out)
{
    hipStream_t s = ctx->stream;
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    int g[5];
    if (!stencil_geometry(kdim, g)) {
        set_error("direct stencil: PSF %dx%dx%d outside 1..%d taps per axis; use the FFT method", kx, ky, kz, MAXK);
        return MVSIM_EINVAL;
    }
    const int ng = g[0], kyc = g[1], kzc = g[2], S = g[3], H = g[4];
    const int kxp = 4 * ng;
    const size_t lds = (size_t)S * H * (TZ + kzc - 1) * sizeof(float);
    MVSIM_TRY(ctx->stencil_psf.reserve((size_t)kxp * ky * kz * sizeof(float)));
    float* prep = ctx->stencil_psf.as<float>();
    const int total = kxp * ky * kz;
    hipLaunchKernelGGL(k_stencil_prep, dim3((total + 255) / 256), dim3(256), 0, s, psf, prep, kx, ky, kz, kxp);
    static constexpr auto table = stencil_table(std::make_integer_sequence<int, MAXK / 4>{});
    const stencil_fn fn = table[ng - 1];
    MVSIM_TRY(ensure_lds_attr(ctx, reinterpret_cast<const void*>(fn), lds));
    dim3 grid((nx + TX - 1) / TX, (ny + TY - 1) / TY, (nz + TZ - 1) / TZ);
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, img, prep, out, nx, ny, nz, kx, ky, kz, kyc, kzc, S, H);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

}  // namespace mvsim
