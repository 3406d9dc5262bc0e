// fp32 FMA issue rate on gfx950: plain v_fma_f32 against v_pk_fma_f32 with a scalar (SGPR) multiplier, 8 or 16 independent
// accumulator chains per lane, 1/2/4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void k_plain(float* out, const float* wsrc, int iters)
{
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9f + i;
    float x = threadIdx.x * 1e-7f;
    for (int it = 0; it < iters; ++it) {
        const float w0 = wsrc[it & 15], w1 = wsrc[(it + 1) & 15];   // wave-uniform -> SGPR
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"((rep & 1) ? w1 : w0), "v"(x));
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_packed(float* out, const float* wsrc, int iters)
{
    v2f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v2f{threadIdx.x * 1e-9f + i, 1.f};
    v2f x = v2f{threadIdx.x * 1e-7f, 2.f};
    for (int it = 0; it < iters; ++it) {
        const v2f w = *reinterpret_cast<const v2f*>(wsrc + 2 * (it & 7));   // SGPR pair
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(w), "v"(x));
        }
    }
    v2f s = v2f{0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <class F>
static void run(const char* name, F launch, double flop_per_thread_iter, int iters, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = flop_per_thread_iter * iters * 256.0 * blocks;
    printf("%-34s blocks/CU %d: %8.3f ms  %7.1f Tflop/s\n", name, blocks / 256, ms, flop / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out, *w;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    hipMalloc(&w, 64 * sizeof(float));
    float hw[64];
    for (int i = 0; i < 64; ++i) hw[i] = 1.0f - 1e-6f * i;
    hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = 256 * bpc;
        run("v_fma_f32 x8 chains", [&] { hipLaunchKernelGGL(k_plain<8>, dim3(blocks), dim3(256), 0, 0, out, w, iters); }, 2.0 * 8 * 8, iters, blocks);
        run("v_fma_f32 x16 chains", [&] { hipLaunchKernelGGL(k_plain<16>, dim3(blocks), dim3(256), 0, 0, out, w, iters); }, 2.0 * 16 * 8, iters, blocks);
        run("v_pk_fma_f32 x8 chains", [&] { hipLaunchKernelGGL(k_packed<8>, dim3(blocks), dim3(256), 0, 0, out, w, iters); }, 4.0 * 8 * 8, iters, blocks);
        run("v_pk_fma_f32 x16 chains", [&] { hipLaunchKernelGGL(k_packed<16>, dim3(blocks), dim3(256), 0, 0, out, w, iters); }, 4.0 * 16 * 8, iters, blocks);
    }
    return 0;
}
