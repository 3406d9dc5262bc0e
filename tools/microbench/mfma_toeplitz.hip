// What a Toeplitz formulation of the direct stencil on the matrix cores could reach: one x row of the PSF
// applied to a 32 (x) x 32 (y) output tile is D[32 x 32] += T[32 x (32 + K - 1)] * IN[(32 + K - 1) x 32] with T[i][j] = w[j - i] (zero outside
// 0 <= j - i < K) -- a dense product of which only K / (32 + K - 1) is useful work.  This loop issues exactly that instruction stream per
// (ky, kz) tap row: for every pair of columns of T one v_mfma_f32_32x32x2_f32 whose A operand is gathered from the tap row in LDS (zero
// outside the band) and whose B operand is read from the input tile in LDS.  Reported: the raw MFMA rate and the USEFUL rate
// (2 K flops per output per tap row), to be read beside the packed-FMA stencil's 78.7 Tflop/s (profiles/r03_stencil_bench.txt).
//   hipcc --offload-arch=gfx950 -O3 mfma_toeplitz.hip -o mfma_toeplitz && ./mfma_toeplitz
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(256) void k_toeplitz(float* out, const float* taps, const float* in, int rows)
{
    constexpr int W = 32 + K - 1, WP = (W + 1) / 2 * 2;             // columns of T, padded to pairs
    __shared__ float w[64 + 2 * 64];                                // tap row with zero aprons: w[64 + t], t in [0, K)
    __shared__ float tile[(32 + 2) * (WP + 1)];                     // input tile: 32 y rows (+ slack) x WP x positions
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 + 2 * 64; i += 256) w[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < (32 + 2) * (WP + 1); i += 256) tile[i] = in[i % 4096];
    v16f acc = {0};
    const int i32 = lane & 31, kk = lane >> 5;                      // A: row i32, column 2 step + kk;  B: column i32 (y), row 2 step + kk
    for (int r = 0; r < rows; ++r) {                                // tap rows (ky, kz) of the PSF
        __syncthreads();
        if (threadIdx.x < K) w[64 + threadIdx.x] = taps[(r * K + threadIdx.x) & 4095];
        __syncthreads();
#pragma unroll
        for (int step = 0; step < WP / 2; ++step) {
            const int j = 2 * step + kk;
            const float a = w[64 + j - i32];                        // T[i32][j] = w[j - i32] (zero apron outside the band)
            const float b = tile[i32 * (WP + 1) + j];               // IN[j][y = i32]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K> static void run(float* out, const float* taps, const float* in, int rows, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_toeplitz<K>, dim3(blocks), dim3(256), 0, 0, out, taps, in, rows);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_toeplitz<K>, dim3(blocks), dim3(256), 0, 0, out, taps, in, rows);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    constexpr int W = 32 + K - 1, WP = (W + 1) / 2 * 2;
    const double waves = 4.0 * blocks;
    const double raw = waves * rows * (WP / 2) * (2.0 * 32 * 32 * 2);
    const double useful = waves * rows * (2.0 * K * 32 * 32);
    printf("K = %2d taps: T is 32 x %3d (%2.0f %% of it inside the band), %d blocks x 4 waves, %d tap rows: %7.3f ms  raw %6.1f Tflop/s  useful %6.1f Tflop/s\n",
           K, WP, 100.0 * K / WP, blocks, rows, ms, raw / (ms * 1e-3) / 1e12, useful / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out, *taps, *in;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipMalloc(&taps, 4096 * sizeof(float));
    hipMalloc(&in, 4096 * sizeof(float));
    hipMemset(taps, 0, 4096 * sizeof(float));
    hipMemset(in, 0, 4096 * sizeof(float));
    for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc * 2;
        run<15>(out, taps, in, 15 * 15 * 4, blocks);
        run<31>(out, taps, in, 31 * 31, blocks);
        run<63>(out, taps, in, 63 * 16, blocks);
    }
    return 0;
}
