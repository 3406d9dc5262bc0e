// Ground-truth phantom of SimulateMultiViewDataset (SimulateMultiViewDataset.java:366-522): the step in front of
// the per-view path.
//
//   drawSpheres   :436-522  walk the voxels of one large sphere; per voxel draw radius = rnd.nextInt(10*scale)+1 and
//                           a double; one voxel in (7*scale)^3 (by a rounding test on that double) becomes the centre
//                           of a small sphere whose voxels are max-composited with a second random intensity.
//   downSample2x  :394-424  n-linear samples at 2l + 0.5 (all eight weights 1/8), dims N/2 - 1.
//
// Split between host and GPU.  The random stream is ONE sequential java.util.Random shared by all voxels of the large
// sphere (3 LCG steps per voxel, 5 where a sphere is drawn), so the walk that decides which voxels become centres
// is replayed on the host (~0.1 s for the reference's 580^3 canvas: 31 M voxels) in the HyperSphereCursor's raster
// order; what it produces is a list of (centre, radius, value).  Math.max makes the compositing order-free, so the
// ~11 000 spheres (~10^8 voxel updates) are splatted on the GPU with one atomic max per voxel, and the 2x
// down-sampling is a streaming kernel.
//
// Sphere geometry (ImgLib2 HyperSphereCursor, restated from the published algorithm): z in [cz-R, cz+R],
// r1 = floor(sqrt(R^2 - dz^2)), y in [cy-r1, cy+r1], r0 = floor(sqrt(r1^2 - dy^2)), x in [cx-r0, cx+r0]
// (nested truncated radii, x fastest).  (long)Math.sqrt of an exact integer equals the integer floor square root,
// which is what isqrt() computes without floating-point rounding questions.
#include "common.h"

#include <cmath>
#include <vector>

namespace mvsim {

namespace {

// java.util.Random (JDK specification): 48-bit LCG
struct JRandom {
    uint64_t s;
    int32_t next(int bits)
    {
        s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)((int64_t)s >> (48 - bits));
    }
    int32_t next_int(int32_t bound)
    {
        int32_t r = next(31);
        const int32_t m = bound - 1;
        if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
        for (int32_t u = r; (int32_t)((uint32_t)u - (uint32_t)(r = u % bound) + (uint32_t)m) < 0; u = next(31)) {}
        return r;
    }
    double next_double()
    {
        const int64_t hi = (int64_t)next(26) << 27;
        return (double)(hi + next(27)) * 0x1.0p-53;
    }
};

inline int64_t isqrt_host(int64_t v)
{
    int64_t r = (int64_t)std::sqrt((double)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

}  // namespace

// device-side item = the C ABI's mvsim_sphere (centre, radius, value)
struct SphereItem {
    int cx, cy, cz, r;
    float v;
};
static_assert(sizeof(SphereItem) == sizeof(mvsim_sphere), "SphereItem must mirror mvsim_sphere");

__device__ __forceinline__ int isqrt_dev(int v)
{
    int r = (int)sqrtf((float)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

// max-composite that is valid for any mix of signs (integer order of IEEE floats)
__device__ __forceinline__ void atomic_max_float(float* p, float v)
{
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}

// one block per (sphere, z slice of the sphere); lanes over the (2r+1)^2 candidates of the slice
__global__ __launch_bounds__(256) void k_splat_spheres(float* __restrict__ img, int nx, int ny, int nz,
                                                       const SphereItem* __restrict__ items)
{
    const SphereItem it = items[blockIdx.x];
    const int dz = (int)blockIdx.y - it.r;
    if (dz > it.r) return;
    const int r1 = isqrt_dev(it.r * it.r - dz * dz);
    const int w = 2 * r1 + 1;
    const int z = it.cz + dz;
    if (z < 0 || z >= nz) return;
    for (int e = threadIdx.x; e < w * w; e += 256) {
        const int dy = e / w - r1, dx = e % w - r1;
        const int r0 = isqrt_dev(r1 * r1 - dy * dy);
        if (dx < -r0 || dx > r0) continue;
        const int x = it.cx + dx, y = it.cy + dy;
        if (x < 0 || x >= nx || y < 0 || y >= ny) continue;      // validated on the host; belt and braces
        atomic_max_float(img + (x + (long long)nx * (y + (long long)ny * z)), it.v);
    }
}

// SMVD:394-424.  All eight n-linear weights are 0.5^3: each tap (float)(v * 0.125), float accumulation in the
// interpolator's Gray-code order 000,100,110,010,011,111,101,001 (x is the first digit).
__global__ __launch_bounds__(256) void k_downsample2x(const float* __restrict__ in, float* __restrict__ out, int nx,
                                                      int ny, int ox, int oy, int oz)
{
    const long long total = (long long)ox * oy * oz;
    const long long nthreads = (long long)gridDim.x * 256;
    const long long row = nx, plane = (long long)nx * ny;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += nthreads) {
        const int x = (int)(o % ox);
        const long long t = o / ox;
        const int y = (int)(t % oy), z = (int)(t / oy);
        const float* p = in + 2 * x + row * (2 * y) + plane * (2 * z);
        const double w = 0.5 * 0.5 * 0.5;
        float acc = (float)((double)p[0] * w);
        acc += (float)((double)p[1] * w);
        acc += (float)((double)p[1 + row] * w);
        acc += (float)((double)p[row] * w);
        acc += (float)((double)p[row + plane] * w);
        acc += (float)((double)p[1 + row + plane] * w);
        acc += (float)((double)p[1 + plane] * w);
        acc += (float)((double)p[plane] * w);
        out[o] = acc;
    }
}

int launch_downsample2x(hipStream_t s, const float* in, const int64_t dim[3], float* out)
{
    const int ox = (int)(dim[0] / 2 - 1), oy = (int)(dim[1] / 2 - 1), oz = (int)(dim[2] / 2 - 1);
    const long long total = (long long)ox * oy * oz;
    long long want = (total + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
    hipLaunchKernelGGL(k_downsample2x, dim3(blocks), dim3(256), 0, s, in, out, (int)dim[0], (int)dim[1], ox, oy, oz);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// drawSpheres on a device image.  rnd_state: the 48-bit state of the caller's java.util.Random, advanced exactly
// as the reference advances it.
int draw_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], double min_value, double max_value, int scale,
                     int half_pixel_offset, uint64_t* rnd_state, int64_t* n_spheres)
{
    int64_t c[3], min_size = dim[0];
    for (int d = 0; d < 3; ++d) {
        c[d] = dim[d] / 2;
        if (dim[d] < min_size) min_size = dim[d];
    }
    const int max_radius = 10 * scale;
    const int64_t R = min_size / 2 - 47 * (int64_t)scale - 1;
    if (R < 0) {
        set_error("drawSpheres: image too small for scale %d (large-sphere radius %lld)", scale, (long long)R);
        return MVSIM_EINVAL;
    }
    int64_t modulus = 1;
    for (int d = 0; d < 3; ++d) modulus *= 7 * (int64_t)scale;

    JRandom rnd{*rnd_state & ((1ULL << 48) - 1)};
    std::vector<SphereItem> items;
    const int off_xy = half_pixel_offset ? 1 : 0;
    for (int64_t dz = -R; dz <= R; ++dz) {
        const int64_t r1 = isqrt_host(R * R - dz * dz);
        for (int64_t dy = -r1; dy <= r1; ++dy) {
            const int64_t r0 = isqrt_host(r1 * r1 - dy * dy);
            for (int64_t dx = -r0; dx <= r0; ++dx) {
                const int radius = rnd.next_int(max_radius) + 1;
                const double rv = rnd.next_double();
                const int64_t rounded = (int64_t)std::floor(rv * 10000 + 0.5);     // Math.round
                if (rounded % modulus != 0) continue;
                const double value = rnd.next_double() * (max_value - min_value) + min_value;
                SphereItem it;
                it.cx = (int)(c[0] + dx + off_xy); it.cy = (int)(c[1] + dy + off_xy); it.cz = (int)(c[2] + dz);
                it.r = radius; it.v = (float)value;
                if (it.cx - radius < 0 || it.cy - radius < 0 || it.cz - radius < 0 || it.cx + radius >= dim[0] ||
                    it.cy + radius >= dim[1] || it.cz + radius >= dim[2]) {
                    set_error("drawSpheres: a small sphere leaves the image (the reference throws here)");
                    return MVSIM_EINVAL;
                }
                items.push_back(it);
            }
        }
    }
    *rnd_state = rnd.s;
    if (n_spheres) *n_spheres = (int64_t)items.size();
    return splat_spheres_dev(ctx, img, dim, reinterpret_cast<const mvsim_sphere*>(items.data()), (int64_t)items.size());
}

// Max-compositing of a list of small spheres (the GPU half of drawSpheres; also the entry point for hosts that walk
// the large sphere themselves with their own java.util.Random, mvsim_splat_spheres).
int splat_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], const mvsim_sphere* spheres, int64_t n)
{
    if (n <= 0) return MVSIM_OK;
    int max_radius = 0;
    for (int64_t i = 0; i < n; ++i) {
        const mvsim_sphere& q = spheres[i];
        if (q.radius < 0 || q.radius > 4096) { set_error("invalid argument: sphere %lld has radius %d", (long long)i, q.radius); return MVSIM_EINVAL; }
        if (q.cx - q.radius < 0 || q.cy - q.radius < 0 || q.cz - q.radius < 0 || (int64_t)q.cx + q.radius >= dim[0] ||
            (int64_t)q.cy + q.radius >= dim[1] || (int64_t)q.cz + q.radius >= dim[2]) {
            set_error("drawSpheres: a small sphere leaves the image (the reference throws here)");
            return MVSIM_EINVAL;
        }
        if (q.radius > max_radius) max_radius = q.radius;
    }
    MVSIM_TRY(ctx->sphere_list.reserve((size_t)n * sizeof(SphereItem)));
    // the list lives in pageable host memory: a synchronous copy (the host walk dominates anyway)
    MVSIM_HIP(hipStreamSynchronize(ctx->stream));
    MVSIM_HIP(hipMemcpy(ctx->sphere_list.p, spheres, (size_t)n * sizeof(SphereItem), hipMemcpyHostToDevice));
    // Math.max( value, existing ) stores (float)value when value > existing: with monotonic rounding that is the
    // float max of (float)value and existing
    const int64_t chunk = 32768;        // grid.x limit is far away; chunking only bounds a single launch
    for (int64_t i0 = 0; i0 < n; i0 += chunk) {
        const int64_t m = n - i0 < chunk ? n - i0 : chunk;
        hipLaunchKernelGGL(k_splat_spheres, dim3((unsigned)m, (unsigned)(2 * max_radius + 1)), dim3(256), 0, ctx->stream,
                           img, (int)dim[0], (int)dim[1], (int)dim[2], ctx->sphere_list.as<SphereItem>() + i0);
    }
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

}  // namespace mvsim
