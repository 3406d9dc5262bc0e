// Issue rate of the fp64 instructions the fused rotate + attenuate kernel is made of (v_cvt_f64_f32, v_cvt_f32_f64, v_mul_f64, v_fma_f64,
// v_add_f64, v_max_f64) against v_fma_f32, per SIMD: 8 independent chains per lane, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 f64_rate.hip -o f64_rate && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(STMT) _Pragma("unroll") for (int i = 0; i < 8; ++i) { STMT; }

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-9 + i; f[i] = 1.0f + threadIdx.x * 1e-7f + i; }
    const double c = 1.0000001; const float cf = 1.0000001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (OP == 0) CHAIN8(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(cf)))
            if (OP == 1) CHAIN8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])))
            if (OP == 2) CHAIN8(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i])))
            if (OP == 3) CHAIN8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c)))
            if (OP == 4) CHAIN8(asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(c)))
            if (OP == 5) CHAIN8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c)))
            if (OP == 6) CHAIN8(asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c)))
            if (OP == 7) CHAIN8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(f[i]), "v"(cf) : "vcc"))
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> static void run(const char* name, float* out)
{
    const int iters = 4096, blocks = 256 * 4;             // 4 blocks of 4 waves per CU = 4 waves per SIMD
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = 4.0 * iters * 4 * 8;    // waves per SIMD x instructions per wave
    printf("%-16s %8.3f ms  %5.2f cycles per wave instruction (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("v_fma_f32", out); run<1>("v_cvt_f64_f32", out); run<2>("v_cvt_f32_f64", out); run<3>("v_mul_f64", out);
    run<4>("v_fma_f64", out); run<5>("v_add_f64", out); run<6>("v_max_f64", out); run<7>("v_mad_u64_u32", out);
    return 0;
}
