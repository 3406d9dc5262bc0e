// What the Java facade adds on the host around a native call, measured without a JDK: java/src/main/java/net/preibisch/simulation/gpu/
// Buffers.java moves an image between the JVM heap and page-locked staging blocks with two bulk copies --
//   toBlock / toSlabs:  FloatBuffer.put(float[])       heap array -> page-locked direct buffer   (one memcpy per z slab)
//   toImg:              new float[n]; FloatBuffer.get  a fresh, zero-filled heap array <- page-locked buffer (the JVM zeroes the
//                                                      array first: calloc + first touch, then the memcpy)
// -- single-threaded, as the JVM does them.  This program times exactly those memory operations in C on the GPU box's host
// (hipHostMalloc for the staging block, malloc / calloc for the heap side) for a 512^3 float32 volume and prints ms and GB/s.
// Round 6 adds what the facade does since then (MvsimNative.copyFloats -> mvsim_host_copy: the array held with GetPrimitiveArrayCritical, the
// library's host threads copy): the same two directions through mvsim_host_copy, for an array the JVM has already zeroed (new float[n]
// touches every page before the copy starts: timed as a single-threaded memset, then the threaded copy) and for an untouched one (the
// copying threads take the first-touch faults: what a JVM that defers or elides the zeroing would see).
//   hipcc -O2 tools/microbench/slab_copy.cpp -Iinclude -Lmultiview-simulation_amd -lmvsim -Wl,-rpath,$PWD/multiview-simulation_amd \
//         -o tools/microbench/slab_copy && tools/microbench/slab_copy [edge]
#include <hip/hip_runtime.h>

#include "mvsim.h"

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
    // the same through the library's host threads (mvsim_host_copy, ctx = NULL: min(16, hardware threads))
    double tput_ms = 1e30, tget_zeroed_ms = 1e30, tget_copy_only_ms = 1e30, tget_untouched_ms = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        mvsim_host_copy(nullptr, pinned, heap, bytes);
        tput_ms = std::min(tput_ms, ms_since(t0));
        t0 = std::chrono::steady_clock::now();
        float* fresh = (float*)malloc(bytes);
        static void* (*volatile zero)(void*, int, size_t) = memset;           // (opaque: malloc + memset must not become a lazy calloc)
        zero(fresh, 0, bytes);                                                 // new float[n]: the JVM zeroes the array (one thread)
        auto t1 = std::chrono::steady_clock::now();
        mvsim_host_copy(nullptr, fresh, pinned, bytes);
        tget_copy_only_ms = std::min(tget_copy_only_ms, ms_since(t1));
        tget_zeroed_ms = std::min(tget_zeroed_ms, ms_since(t0));
        free(fresh);
        t0 = std::chrono::steady_clock::now();
        fresh = (float*)malloc(bytes);                                         // untouched pages: the copying threads fault them in
        mvsim_host_copy(nullptr, fresh, pinned, bytes);
        tget_untouched_ms = std::min(tget_untouched_ms, ms_since(t0));
        volatile float sink = fresh[count - 1];
        (void)sink;
        free(fresh);
    }
    printf("through mvsim_host_copy (the library's host threads; MvsimNative.copyFloats)\n");
    printf("  heap array -> page-locked block                                                    %8.2f ms  %6.2f GB/s\n", tput_ms, bytes / tput_ms / 1e6);
    printf("  page-locked block -> NEW heap array, zeroed by one thread first (new float[n])     %8.2f ms  %6.2f GB/s   (the copy alone: %.2f ms)\n", tget_zeroed_ms,
           bytes / tget_zeroed_ms / 1e6, tget_copy_only_ms);
    printf("  page-locked block -> NEW heap array, untouched (the copying threads fault it in)   %8.2f ms  %6.2f GB/s\n", tget_untouched_ms, bytes / tget_untouched_ms / 1e6);
    free(heap);
    (void)hipHostFree(pinned);
    return 0;
}
