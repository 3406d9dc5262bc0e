// global_load_lds_dwordx4 on gfx950: where do the 64 x 16 bytes of one wave instruction land in LDS?  (feasibility check for staging tiles
// without registers)      hipcc --offload-arch=gfx950 -O3 lds_direct.hip -o lds_direct && ./lds_direct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float4* __restrict__ dst, const int* __restrict__ perm)
{
    __shared__ float4 tile[256 + 64];
    const int tid = threadIdx.x, wave = tid >> 6;
    // every lane names its own global element (a permutation); the LDS destination is the wave's base: lane l lands at base + l
    const float4* g = src + perm[tid];
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(tile + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dst[tid] = tile[tid];
}

int main()
{
    const int n = 256;
    std::vector<float4> h(n);
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) { h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f); perm[i] = (i * 37 + 11) % n; }
    float4 *src, *dst; int* p;
    hipMalloc(&src, n * sizeof(float4)); hipMalloc(&dst, n * sizeof(float4)); hipMalloc(&p, n * sizeof(int));
    hipMemcpy(src, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(p, perm.data(), n * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, dst, p);
    std::vector<float4> out(n);
    hipMemcpy(out.data(), dst, n * sizeof(float4), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float4 e = h[perm[i]];
        if (out[i].x != e.x || out[i].y != e.y || out[i].z != e.z || out[i].w != e.w) { if (bad < 5) printf("slot %d: got %.2f %.2f expected %.2f %.2f\n", i, out[i].x, out[i].y, e.x, e.y); ++bad; }
    }
    printf("global_load_lds_dwordx4: %d of %d slots differ from 'lane l of wave w lands at base(w) + l'\n", bad, n);
    return 0;
}
