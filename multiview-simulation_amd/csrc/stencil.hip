// LDS-tiled direct 3-D convolution for small PSFs (and as an FFT-independent cross-check of
// SimulateMultiViewDataset.convolve, :253-264).  Bound by fp32 vector FMA throughput
// (2*Kx*Ky*Kz flop per voxel), not HBM; see DESIGN.md.
//
// Block = 256 threads -> output tile 32 x 8 x 8; each lane owns R = 8 consecutive x outputs.
// The input tile with its mirror halo is staged in LDS, PSF z-planes processed in chunks of kc
// so that the tile fits the LDS budget.  Per (ky,kz) a lane slides an R-wide register window
// along its LDS row: one ds_read per R FMAs; PSF taps are wave-uniform (scalar loads).
#include "common.h"

namespace mvsim {

namespace {
constexpr int TX = 32, TY = 8, TZ = 8, R = 8;

__device__ __forceinline__ int mirror_i(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}
}  // namespace

__global__ __launch_bounds__(256) void k_stencil(const float* __restrict__ img, const float* __restrict__ psf,
                                                 float* __restrict__ out, int nx, int ny, int nz, int kx,
                                                 int ky, int kz, int kc, int S, int H)
{
    extern __shared__ __align__(16) float tile[];
    const int tx = threadIdx.x & 3, ty = (threadIdx.x >> 2) & 7, tz = threadIdx.x >> 5;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;
    const int cx = kx / 2, cy = ky / 2, cz = kz / 2;
    const int W = TX + kx - 1;

    double accd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) accd[r] = 0.0;

    for (int c0 = 0; c0 < kz; c0 += kc) {
        const int cn = (kz - c0) < kc ? (kz - c0) : kc;   // PSF planes c0 .. c0+cn-1
        const int D = TZ + cn - 1;
        // out[z] needs img[z - (c - cz)] for c in [c0, c0+cn): tile plane t <-> gz = z0 - (c0+cn-1-cz) + t
        const int gz0 = z0 - (c0 + cn - 1 - cz);
        const int gy0 = y0 - (ky - 1 - cy);
        const int gx0 = x0 - (kx - 1 - cx);
        __syncthreads();
        const int total = W * H * D;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int ix = i % W;
            const int iy = (i / W) % H;
            const int iz = i / (W * H);
            const int sx = mirror_i(gx0 + ix, nx), sy = mirror_i(gy0 + iy, ny), sz = mirror_i(gz0 + iz, nz);
            tile[ix + S * (iy + H * iz)] = img[sx + (long long)nx * (sy + (long long)ny * sz)];
        }
        __syncthreads();

        for (int cl = 0; cl < cn; ++cl) {
            const int c = c0 + cl;
            // img plane for output tz and PSF plane c: gz = z0+tz-(c-cz) -> t = tz + (c0+cn-1) - c
            const int t = tz + (cn - 1 - cl);
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.0f;
            for (int b = 0; b < ky; ++b) {
                // row for output ty and PSF row b: gy = y0+ty-(b-cy) -> iy = ty + (ky-1-b)
                const float* __restrict__ row = tile + S * ((ty + (ky - 1 - b)) + H * t) + tx * R;
                const float* __restrict__ prow = psf + (long long)kx * (b + (long long)ky * c);
                float win[R];
#pragma unroll
                for (int r = 0; r < R; ++r) win[r] = row[r];
                for (int o = 0; o < kx; o += R) {
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        if (o + j < kx) {
                            const float w = prow[kx - 1 - (o + j)];
#pragma unroll
                            for (int r = 0; r < R; ++r) acc[r] = fmaf(w, win[(j + r) % R], acc[r]);
                            win[j % R] = row[o + j + R];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) accd[r] += (double)acc[r];
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

int launch_stencil(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                   const int64_t kdim[3], float* out)
{
    hipStream_t s = ctx->stream;
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    const int W = TX + kx - 1;
    const int S = (W + 1) | 1;          // odd row stride: conflict-free for the 4x8 lane layout
    const int H = TY + ky - 1;
    const size_t slice = (size_t)S * H * sizeof(float);
    const size_t budget_small = 64 * 1024, budget_big = 150 * 1024;
    int kc = (int)(budget_small / slice) - (TZ - 1);
    size_t budget = budget_small;
    if (kc < 4 && kc < kz) {
        kc = (int)(budget_big / slice) - (TZ - 1);
        budget = budget_big;
    }
    if (kc < 1) {
        set_error("direct stencil: PSF %dx%dx%d too large for the LDS tile; use the FFT method", kx, ky, kz);
        return MVSIM_EINVAL;
    }
    if (kc > kz) kc = kz;
    const size_t lds = slice * (size_t)(TZ + kc - 1) + 64;
    (void)budget;
    MVSIM_TRY(ensure_lds_attr(ctx, reinterpret_cast<const void*>(k_stencil), lds));
    dim3 grid((nx + TX - 1) / TX, (ny + TY - 1) / TY, (nz + TZ - 1) / TZ);
    hipLaunchKernelGGL(k_stencil, grid, dim3(256), lds, s, img, psf, out, nx, ny, nz, kx, ky, kz, kc, S, H);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

}  // namespace mvsim
