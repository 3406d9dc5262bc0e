// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the convolution's y passes.
// Every kernel below reads (or writes) a byte count that is known exactly; run once plain (HIP-event times, GB/s) and once per
// counter set under `rocprofv3 --pmc ...` (tools/fetch_calib.sh), then compare counter x 1024 with the bytes printed here.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib && ./fetch_calib
//
// Shapes:
//   stream16       a wide coalesced stream, 16 B per lane, grid-stride (the shape MI355X_MICROARCH.md calibrated: counter = 1/2)
//   tile<T>        pass B / D of csrc/fft_kernels.hip (k_fft_lines): grid (tiles, planes), a block of T threads loads the ROWS rows of
//                  one 16-column tile -- 8 lanes x 16 B = one 128-byte segment per row, rows `pitch` bytes apart, all loads requested
//                  before the first wait --, drops them into LDS (dynamic LDS sized like the pass: it sets the blocks per CU) and leaves.
//                  512^3: T = 512, rows 512, pitch 288 x 8 B, plane 560 rows, 18 tiles, 75 KB LDS (two blocks per CU)
//                  1024^3 AS THE PASS RUNS IT (Cfg<1080>: lines above 576 points take 8-COLUMN tiles): T = 512, 4 lanes x 16 B = one
//                  64-byte segment per row, pitch 544 x 8 B, plane 1080 rows, 68 tiles, 78 KB LDS (two blocks per CU) -- in plain grid
//                  order, and with the two tiles of a 128-byte line on one XCD (ids b and b + 8: round 6's order)
//                  1024^3 with 16-column tiles (what the pass would be with 128-byte segments): T = 1024, 34 tiles, 147 KB LDS
//   tile_store<T>  the same shape storing instead of loading (WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ src, long long n4, float* sink)
{
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = src[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) sink[0] = s;
}

struct TileArgs {
    const char* src;
    char*       dst;
    long long   plane_pitch;    // bytes
    int         pitch;          // bytes between rows
    int         rows;           // rows a tile reads
    int         extra;          // rows read a SECOND time (the mirrored halo rows of pass B): rows [1, extra]
    int         pair;           // LPR = 4: the two tiles of a 128-byte line run on one XCD, ids b and b + 8 (k_fft_lines' order)
    float*      sink;
};

template <int T, int NIT, bool STORE, int LPR = 8>
__global__ __launch_bounds__(T) void k_tile(TileArgs p)
{
    extern __shared__ __align__(16) float4 lds[];
    const int tid = threadIdx.x, c = tid % LPR, r0 = tid / LPR;
    constexpr int ROWS = T / LPR;
    unsigned tx = blockIdx.x, ty = blockIdx.y;
    if (LPR == 4 && p.pair) {
        const unsigned nt = gridDim.x, total = nt * gridDim.y, b = blockIdx.y * nt + blockIdx.x;
        if (b < (total & ~15u)) {
            const unsigned xcd = b & 7u, slot = b >> 3, lin = ((((slot >> 1) << 3) + xcd) << 1) | (slot & 1u);
            tx = lin % nt; ty = lin / nt;
        }
    }
    const long long base = (long long)ty * p.plane_pitch + (long long)tx * (LPR * 16) + c * 16;
    const int total = p.rows + p.extra;
    if (STORE) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = r0 + it * ROWS;
            if (n < total) *reinterpret_cast<float4*>(p.dst + base + (long long)n * p.pitch) = make_float4((float)n, 1.f, 2.f, (float)tid);
        }
        return;
    }
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        const int sn = n < p.rows ? n : n - p.rows + 1;             // the halo rows are rows of the tile, read again
        const int cl = n < total ? sn : 0;
        v[it] = *reinterpret_cast<const float4*>(p.src + base + (long long)cl * p.pitch);
    }
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        lds[(tid + it * T) & 2047] = v[it];
        s += v[it].x;
    }
    __syncthreads();
    if (s + lds[(tid * 7) & 2047].y == 123.456f) p.sink[0] = s;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

template <int T, int NIT, bool STORE, int LPR = 8>
static void run_tile(const char* name, TileArgs a, int tiles, int planes, size_t lds, int reps)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<T, NIT, STORE, LPR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_tile<T, NIT, STORE, LPR>), dim3(tiles, planes), dim3(T), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_tile<T, NIT, STORE, LPR>), dim3(tiles, planes), dim3(T), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    const double ms = time_ms(e0, e1) / reps;
    const double uniq = 16.0 * LPR * a.rows * tiles * planes, req = 16.0 * LPR * (a.rows + a.extra) * tiles * planes;
    printf("%-44s T=%4d LDS=%6zu rows=%4d+%2d pitch=%5d tiles=%3d planes=%4d | unique %8.4f GB requested %8.4f GB | %7.3f ms %7.1f GB/s (unique)\n",
           name, T, lds, a.rows, a.extra, a.pitch, tiles, planes, uniq / 1e9, req / 1e9, ms, uniq / ms / 1e6);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const size_t big = (size_t)1024 * 1080 * 544 * 8;     // the 1024^3 view's half spectrum: 4.81 GB
    char *src, *dst; float* sink;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&dst, big)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 0, big)); CK(hipMemset(dst, 0, big));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // the calibrated shape
        const long long n4 = (long long)big / 16;
        hipLaunchKernelGGL(k_stream16, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<const float4*>(src), n4, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream16, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<const float4*>(src), n4, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        const double ms = time_ms(e0, e1) / reps;
        printf("%-44s bytes %8.4f GB | %7.3f ms %7.1f GB/s\n", "k_stream16 (16 B per lane, contiguous)", big / 1e9, ms, big / ms / 1e6);
    }
    TileArgs s512{src, dst, 560LL * 288 * 8, 288 * 8, 512, 0, 0, sink};
    TileArgs s1024{src, dst, 1080LL * 544 * 8, 544 * 8, 1024, 0, 0, sink};
    TileArgs s1024m = s1024; s1024m.extra = 30;
    TileArgs s512m = s512; s512m.extra = 30;
    const size_t l512 = (16 * 561 + 560) * 8 + 256, l1024 = (16 * 1081 + 1080) * 8 + 256;
    const size_t l1024_8 = (8 * 1081 + 1080) * 8 + 256;
    TileArgs s1024p = s1024; s1024p.pair = 1;
    TileArgs s1024pm = s1024m; s1024pm.pair = 1;
    // pass B as it runs: 512^3 (16-column tiles, two blocks per CU), 1024^3 (8-column tiles: 64-byte segments, two blocks per CU)
    run_tile<512, 9, false>("tile 512^3 as pass B runs it", s512, 18, 512, l512, reps);
    run_tile<512, 9, false>("tile 512^3 + 30 mirrored rows", s512m, 18, 512, l512, reps);
    run_tile<512, 9, false, 4>("tile 1024^3 AS PASS B RAN IT (8 col, grid order)", s1024, 68, 1024, l1024_8, reps);
    run_tile<512, 9, false, 4>("tile 1024^3 8 col, pair on one XCD (round 6)", s1024p, 68, 1024, l1024_8, reps);
    run_tile<512, 9, false, 4>("tile 1024^3 8 col, paired + 30 mirrored rows", s1024pm, 68, 1024, l1024_8, reps);
    run_tile<1024, 9, false>("tile 1024^3 with 16-column tiles, 1/CU", s1024, 34, 1024, l1024, reps);
    run_tile<1024, 9, false>("tile 1024^3 16 col + 30 mirrored rows", s1024m, 34, 1024, l1024, reps);
    // which of the differences matters: block size, blocks per CU, pitch
    run_tile<512, 17, false>("tile 1024^3 geometry, 512 threads, 2/CU", s1024, 34, 1024, l512, reps);
    run_tile<1024, 9, false>("tile 1024^3 geometry, 1024 threads, 2/CU", s1024, 34, 1024, l512, reps);
    run_tile<1024, 9, false>("tile 1024^3 geometry, 1024 threads, small LDS", s1024, 34, 1024, 32768, reps);
    run_tile<1024, 5, false>("tile 512^3 geometry, 1024 threads, 1/CU", s512, 18, 512, l1024, reps);
    run_tile<512, 9, false>("tile 512^3 geometry, 512 threads, 1/CU", s512, 18, 512, l1024, reps);
    // the same plane count for both (is it the footprint?): 512^3 geometry over 1024 planes... and 1024 geometry over 256 planes
    run_tile<1024, 9, false>("tile 1024^3 geometry, 128 planes", s1024, 34, 128, l1024, reps);
    // stores in the same shape
    run_tile<512, 9, true>("tile_store 512^3", s512, 18, 512, l512, reps);
    run_tile<512, 9, true, 4>("tile_store 1024^3 8 col, grid order", s1024, 68, 1024, l1024_8, reps);
    run_tile<512, 9, true, 4>("tile_store 1024^3 8 col, paired", s1024p, 68, 1024, l1024_8, reps);
    run_tile<1024, 9, true>("tile_store 1024^3 16 col", s1024, 34, 1024, l1024, reps);
    return 0;
}
