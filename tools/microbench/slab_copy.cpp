// What the Java facade adds on the host around a native call, measured without a JDK: java/src/main/java/net/preibisch/simulation/gpu/
// Buffers.java moves an image between the JVM heap and page-locked staging blocks with two bulk copies --
//   toBlock / toSlabs:  FloatBuffer.put(float[])       heap array -> page-locked direct buffer   (one memcpy per z slab)
//   toImg:              new float[n]; FloatBuffer.get  a fresh, zero-filled heap array <- page-locked buffer (the JVM zeroes the
//                                                      array first: calloc + first touch, then the memcpy)
// -- single-threaded, as the JVM does them.  This program times exactly those memory operations in C on the GPU box's host
// (hipHostMalloc for the staging block, malloc / calloc for the heap side) for a 512^3 float32 volume and prints ms and GB/s.
//   hipcc -O2 tools/microbench/slab_copy.cpp -o tools/microbench/slab_copy && tools/microbench/slab_copy [edge]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 512;
    const size_t count = (size_t)(n * n * n), bytes = count * sizeof(float);
    void* pinned = nullptr;
    if (hipHostMalloc(&pinned, bytes, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "hipHostMalloc failed\n"); return 1; }
    float* heap = (float*)malloc(bytes);
    for (size_t i = 0; i < count; ++i) heap[i] = (float)(i & 1023);              // a live Java array: every page touched
    memset(pinned, 0, bytes);
    double put_ms = 1e30, get_ms = 1e30, get_into_live_ms = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        memcpy(pinned, heap, bytes);                                           // FloatBuffer.put(float[])
        put_ms = std::min(put_ms, ms_since(t0));
        t0 = std::chrono::steady_clock::now();
        float* fresh = (float*)calloc(count, sizeof(float));                   // new float[n]
        memcpy(fresh, pinned, bytes);                                          // FloatBuffer.get(float[])
        get_ms = std::min(get_ms, ms_since(t0));
        volatile float sink = fresh[count - 1];
        (void)sink;
        free(fresh);
        t0 = std::chrono::steady_clock::now();
        memcpy(heap, pinned, bytes);                                           // into an array that already exists (copyBack-like bulk)
        get_into_live_ms = std::min(get_into_live_ms, ms_since(t0));
    }
    printf("%lld^3 float32 = %.3f GB, one host thread (best of 5)\n", n, bytes / 1e9);
    printf("  heap array -> page-locked block   (Buffers.toBlock/toSlabs: FloatBuffer.put)        %8.2f ms  %6.2f GB/s\n", put_ms, bytes / put_ms / 1e6);
    printf("  page-locked block -> NEW heap array (Buffers.toImg: new float[n] + FloatBuffer.get) %8.2f ms  %6.2f GB/s\n", get_ms, bytes / get_ms / 1e6);
    printf("  page-locked block -> existing heap array                                           %8.2f ms  %6.2f GB/s\n", get_into_live_ms, bytes / get_into_live_ms / 1e6);
    free(heap);
    (void)hipHostFree(pinned);
    return 0;
}
