// FFT-backed convolution (SimulateMultiViewDataset.java:253-264 / imglib2 FFTConvolution):
//   out[x] = sum_k psf[k] * img_mirror[x - (k - K/2)]
// mirror-single image boundary, zero-extended kernel with its centre K/2 on the origin, plain
// complex product (no conjugate).  rocFFT supplies the 3-D r2c / c2r transforms; padding,
// kernel embedding, the complex product and crop+scale(+sum) are the kernels below.
//
// Circular layout of the padded image (size P >= N + K - 1 per dimension), c = K/2:
//   [0, N)            image
//   [N, N + c)        right mirror halo
//   [P-(K-1-c), P)    left mirror halo (wraps around)
// so the valid output occupies [0, N) and the crop is a plain sub-box copy.
#include "common.h"

#include <mutex>

#include <cstdlib>
#include <cstring>

namespace mvsim {

__device__ __forceinline__ long long mirror_idx(long long i, long long n)
{
    if (n == 1) return 0;
    const long long p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// source coordinate for padded coordinate j, or -1 for the zero gap
__device__ __forceinline__ long long pad_src(long long j, long long n, long long P, long long right, long long left)
{
    if (j < n + right) return mirror_idx(j, n);
    if (j >= P - left) return mirror_idx(j - P, n);
    return -1;
}

__global__ __launch_bounds__(256) void k_pad_mirror(const float* __restrict__ img, float* __restrict__ padded,
                                                    int nx, int ny, int nz, int px, int py, int pz,
                                                    int rx, int lx, int ry, int ly, int rz, int lz)
{
    const int y = blockIdx.y, z = blockIdx.z;
    const long long sy = pad_src(y, ny, py, ry, ly);
    const long long sz = pad_src(z, nz, pz, rz, lz);
    float* __restrict__ dst = padded + (long long)px * (y + (long long)py * z);
    if (sy < 0 || sz < 0) {
        for (int x = threadIdx.x; x < px; x += 256) dst[x] = 0.0f;
        return;
    }
    const float* __restrict__ src = img + (long long)nx * (sy + (long long)ny * sz);
    for (int x = threadIdx.x; x < px; x += 256) {
        const long long sx = pad_src(x, nx, px, rx, lx);
        dst[x] = sx < 0 ? 0.0f : src[sx];
    }
}

int launch_pad_mirror(hipStream_t s, const float* img, const int64_t dim[3], const int64_t kdim[3],
                      float* padded, const int64_t P[3])
{
    int r[3], l[3];
    for (int d = 0; d < 3; ++d) {
        r[d] = (int)(kdim[d] / 2);
        l[d] = (int)(kdim[d] - 1 - kdim[d] / 2);
    }
    dim3 grid(1, (unsigned)P[1], (unsigned)P[2]);
    hipLaunchKernelGGL(k_pad_mirror, grid, dim3(256), 0, s, img, padded, (int)dim[0], (int)dim[1], (int)dim[2],
                       (int)P[0], (int)P[1], (int)P[2], r[0], l[0], r[1], l[1], r[2], l[2]);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

__global__ __launch_bounds__(256) void k_psf_scatter(const float* __restrict__ psf, float* __restrict__ padded,
                                                     int kx, int ky, int kz, int px, int py, int pz)
{
    const long long total = (long long)kx * ky * kz;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int a = (int)(i % kx);
        const int b = (int)((i / kx) % ky);
        const int c = (int)(i / ((long long)kx * ky));
        int x = a - kx / 2; if (x < 0) x += px;
        int y = b - ky / 2; if (y < 0) y += py;
        int z = c - kz / 2; if (z < 0) z += pz;
        padded[x + (long long)px * (y + (long long)py * z)] = psf[i];
    }
}

int launch_psf_embed(hipStream_t s, const float* psf, const int64_t kdim[3], float* padded, const int64_t P[3])
{
    const size_t bytes = (size_t)P[0] * P[1] * P[2] * sizeof(float);
    MVSIM_HIP(hipMemsetAsync(padded, 0, bytes, s));
    const long long total = (long long)kdim[0] * kdim[1] * kdim[2];
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_psf_scatter, dim3(blocks), dim3(256), 0, s, psf, padded, (int)kdim[0], (int)kdim[1],
                       (int)kdim[2], (int)P[0], (int)P[1], (int)P[2]);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

__global__ __launch_bounds__(256) void k_cmul(float2* __restrict__ f, const float2* __restrict__ g, long long n)
{
    const long long nthreads = (long long)gridDim.x * 256;
    const long long n2 = n >> 1;
    float4* f4 = reinterpret_cast<float4*>(f);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += nthreads) {
        const float4 a = f4[i], b = g4[i];
        float4 r;
        r.x = a.x * b.x - a.y * b.y;
        r.y = a.x * b.y + a.y * b.x;
        r.z = a.z * b.z - a.w * b.w;
        r.w = a.z * b.w + a.w * b.z;
        f4[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const float2 a = f[n - 1], b = g[n - 1];
        f[n - 1] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}

int launch_cmul(hipStream_t s, float2* f, const float2* g, int64_t n)
{
    long long want = (n / 2 + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    hipLaunchKernelGGL(k_cmul, dim3(blocks), dim3(256), 0, s, f, g, (long long)n);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// rows = ny*nz output rows, distributed block-cyclically over a fixed grid (deterministic sums)
__global__ __launch_bounds__(256) void k_crop_scale_sum(const float* __restrict__ real, float* __restrict__ out,
                                                        int nx, int ny, int nz, int px, int py, float scale,
                                                        double* __restrict__ partial)
{
    __shared__ double sh[4];
    double acc = 0.0;
    const long long rows = (long long)ny * nz;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int y = (int)(r % ny), z = (int)(r / ny);
        const float* __restrict__ src = real + (long long)px * (y + (long long)py * z);
        float* __restrict__ dst = out + (long long)nx * r;
        for (int x = threadIdx.x; x < nx; x += 256) {
            const float v = src[x] * scale;
            dst[x] = v;
            acc += (double)v;
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void k_sum_final2(const double* __restrict__ partial, int count,
                                                    double* __restrict__ scal)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scal[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int launch_crop_scale_sum(hipStream_t s, const float* real, const int64_t P[3], float* out,
                          const int64_t dim[3], float scale, double* partial, double* scal)
{
    const long long rows = (long long)dim[1] * dim[2];
    int blocks = (int)(rows < SUM_BLOCKS ? rows : SUM_BLOCKS);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_crop_scale_sum, dim3(blocks), dim3(256), 0, s, real, out, (int)dim[0], (int)dim[1],
                       (int)dim[2], (int)P[0], (int)P[1], scale, partial);
    hipLaunchKernelGGL(k_sum_final2, dim3(1), dim3(256), 0, s, partial, blocks, scal);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ---- padded size ------------------------------------------------------------------------------
static bool smooth7(int64_t n)
{
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) n /= p;
    return n == 1;
}

void choose_padded(const int64_t dim[3], const int64_t kdim[3], int64_t P[3], const Options& opt)
{
    // optional override for experiments: option fft_pad = "px,py,pz"
    const int64_t* ov = opt.fft_pad;
    for (int d = 0; d < 3; ++d) {
        const int64_t need = dim[d] + kdim[d] - 1;
        int64_t p = need;
        if (ov[d] >= need) { P[d] = ov[d]; continue; }
        while (!smooth7(p) || (d == 0 && (p & 1))) ++p;
        P[d] = p;
    }
}

// ---- rocFFT plan cache ---------------------------------------------------------------------------
// rocfft_setup() once per process, whatever thread gets here first (contexts live on different host threads)
static std::once_flag g_rocfft_once;
static rocfft_status g_rocfft_status = rocfft_status_success;

static int get_plan(mvsim_ctx* ctx, const int64_t P[3], FftPlan** out)
{
    std::call_once(g_rocfft_once, [] { g_rocfft_status = rocfft_setup(); });
    MVSIM_FFT(g_rocfft_status);
    char key[96];
    snprintf(key, sizeof(key), "%lldx%lldx%lld", (long long)P[0], (long long)P[1], (long long)P[2]);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) { *out = &it->second; return MVSIM_OK; }
    FftPlan pl;
    pl.P[0] = P[0]; pl.P[1] = P[1]; pl.P[2] = P[2];
    const size_t lengths[3] = {(size_t)P[0], (size_t)P[1], (size_t)P[2]};
    MVSIM_FFT(rocfft_plan_create(&pl.fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                 rocfft_precision_single, 3, lengths, 1, nullptr));
    MVSIM_FFT(rocfft_plan_create(&pl.inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                 rocfft_precision_single, 3, lengths, 1, nullptr));
    size_t wf = 0, wi = 0;
    MVSIM_FFT(rocfft_plan_get_work_buffer_size(pl.fwd, &wf));
    MVSIM_FFT(rocfft_plan_get_work_buffer_size(pl.inv, &wi));
    pl.work_bytes = wf > wi ? wf : wi;
    MVSIM_FFT(rocfft_execution_info_create(&pl.info));
    auto res = ctx->plans.emplace(key, pl);
    *out = &res.first->second;
    return MVSIM_OK;
}

void fft_release(mvsim_ctx* ctx)
{
    for (auto& kv : ctx->plans) {
        if (kv.second.fwd) rocfft_plan_destroy(kv.second.fwd);
        if (kv.second.inv) rocfft_plan_destroy(kv.second.inv);
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
    }
    ctx->plans.clear();
    ctx->fft_real.release();
    ctx->fft_spec_img.release();
    ctx->fft_spec_psf.release();
    ctx->fft_work.release();
    custom_fft_release(ctx);
}


int fft_convolve(mvsim_ctx* ctx, const float* img_dev, const int64_t dim[3], const float* psf_dev,
                 const int64_t kdim[3], float* out_dev, ConvTail* tail)
{
    int64_t P[3];
    if (custom_fft_sizes(dim, kdim, P, ctx->opt)) return custom_fft_convolve(ctx, img_dev, dim, psf_dev, kdim, P, out_dev, tail);
    if (tail) { tail->zstride = 1; tail->corr_done = false; }   // the library path produces the whole volume
    choose_padded(dim, kdim, P, ctx->opt);
    FftPlan* pl = nullptr;
    MVSIM_TRY(get_plan(ctx, P, &pl));
    const size_t nreal = (size_t)P[0] * P[1] * P[2];
    const size_t ncplx = (size_t)(P[0] / 2 + 1) * P[1] * P[2];
    MVSIM_TRY(ctx->fft_real.reserve(nreal * sizeof(float)));
    MVSIM_TRY(ctx->fft_spec_img.reserve(ncplx * sizeof(float2)));
    MVSIM_TRY(ctx->fft_spec_psf.reserve(ncplx * sizeof(float2)));
    MVSIM_TRY(ctx->partials.reserve(PARTIALS_BYTES));
    if (pl->work_bytes) {
        MVSIM_TRY(ctx->fft_work.reserve(pl->work_bytes));
        MVSIM_FFT(rocfft_execution_info_set_work_buffer(pl->info, ctx->fft_work.p, pl->work_bytes));
    }
    MVSIM_FFT(rocfft_execution_info_set_stream(pl->info, ctx->stream));

    float* real = ctx->fft_real.as<float>();
    float2* F = ctx->fft_spec_img.as<float2>();
    float2* G = ctx->fft_spec_psf.as<float2>();
    double* partial = ctx->partials.as<double>();
    double* scal = scal_of(ctx);

    // kernel spectrum
    ev_begin(ctx, ST_PSF);
    MVSIM_TRY(launch_psf_embed(ctx->stream, psf_dev, kdim, real, P));
    {
        void* in[1] = {real};
        void* out[1] = {G};
        MVSIM_FFT(rocfft_execute(pl->fwd, in, out, pl->info));
    }
    ev_end(ctx, ST_PSF);

    // image spectrum, product, inverse, crop
    ev_begin(ctx, ST_CONVOLVE);
    MVSIM_TRY(launch_pad_mirror(ctx->stream, img_dev, dim, kdim, real, P));
    {
        void* in[1] = {real};
        void* out[1] = {F};
        MVSIM_FFT(rocfft_execute(pl->fwd, in, out, pl->info));
    }
    MVSIM_TRY(launch_cmul(ctx->stream, F, G, (int64_t)ncplx));
    {
        void* in[1] = {F};
        void* out[1] = {real};
        MVSIM_FFT(rocfft_execute(pl->inv, in, out, pl->info));
    }
    const float scale = (float)(1.0 / ((double)P[0] * (double)P[1] * (double)P[2]));
    MVSIM_TRY(launch_crop_scale_sum(ctx->stream, real, P, out_dev, dim, scale, partial, scal));
    ev_end(ctx, ST_CONVOLVE);
    return MVSIM_OK;
}

}  // namespace mvsim
