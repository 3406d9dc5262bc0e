// Streaming kernels of the per-view path (gfx950, wave64): rotate, attenuate, sum/adjust,
// slice extraction + Poisson, makeIsotropic, weight image.  All are HBM-bound; design notes and
// algorithmic bytes per voxel are in DESIGN.md.
#include <vector>

#include "common.h"
#include "poisson_dev.h"

namespace mvsim {

// ------------------------------------------------------------------------------------------------
// rotateAroundAxis (SimulateMultiViewDataset.java:104-135)
// out[l] = trilinear(in zero-extended, Minv * l); ImgLib2 NLinearInterpolator arithmetic: weights
// in double, each tap rounded (float)(v * w), float accumulation in Gray-code tap order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tap0(const float* __restrict__ in, int nx, int ny, int nz, long long x,
                                      long long y, long long z)
{
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0.0f;
    return in[x + (long long)nx * (y + (long long)ny * z)];
}

__global__ __launch_bounds__(256) void k_rotate_generic(const float* __restrict__ in, float* __restrict__ out,
                                                        int nx, int ny, int nz, Affine a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int z = blockIdx.z;
    if (x >= nx) return;
    const double l0 = (double)x, l1 = (double)y, l2 = (double)z;
    const double px = l0 * a.m[0] + l1 * a.m[1] + l2 * a.m[2] + a.m[3];
    const double py = l0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
    const double pz = l0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    float acc = 0.0f;
    // whole 2x2x2 support outside the volume -> exact zero (also keeps the long long casts safe)
    if (fx >= -1.0 && fy >= -1.0 && fz >= -1.0 && fx < (double)nx && fy < (double)ny && fz < (double)nz) {
        const long long sx = (long long)fx, sy = (long long)fy, sz = (long long)fz;
        const double w0 = px - fx, w1 = py - fy, w2 = pz - fz;
        const double w0n = 1.0 - w0, w1n = 1.0 - w1, w2n = 1.0 - w2;
        acc = (float)((double)tap0(in, nx, ny, nz, sx, sy, sz) * (w0n * w1n * w2n));
        acc += (float)((double)tap0(in, nx, ny, nz, sx + 1, sy, sz) * (w0 * w1n * w2n));
        acc += (float)((double)tap0(in, nx, ny, nz, sx + 1, sy + 1, sz) * (w0 * w1 * w2n));
        acc += (float)((double)tap0(in, nx, ny, nz, sx, sy + 1, sz) * (w0n * w1 * w2n));
        acc += (float)((double)tap0(in, nx, ny, nz, sx, sy + 1, sz + 1) * (w0n * w1 * w2));
        acc += (float)((double)tap0(in, nx, ny, nz, sx + 1, sy + 1, sz + 1) * (w0 * w1 * w2));
        acc += (float)((double)tap0(in, nx, ny, nz, sx + 1, sy, sz + 1) * (w0 * w1n * w2));
        acc += (float)((double)tap0(in, nx, ny, nz, sx, sy, sz + 1) * (w0n * w1n * w2));
    }
    out[x + (long long)nx * (y + (long long)ny * z)] = acc;
}

// Rotation about x (the only axis SimulateMultiViewDataset.main uses, :557,:570,:591): the inverse
// model has row 0 = (1,0,0,0) exactly, so an output x-row reads 4 source rows at the same x with
// wave-uniform weights.  4 voxels per lane (16-B loads/stores), ROT_ROWS consecutive output rows per
// block with all 4*ROT_ROWS row loads issued before the first blend.
constexpr int ROT_ROWS = 4;

__global__ __launch_bounds__(128) void k_rotate_axis0_v4(const float* __restrict__ in, float* __restrict__ out,
                                                         int nx, int ny, int nz, Affine a)
{
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y0 = blockIdx.y * ROT_ROWS;
    const int z = blockIdx.z;
    if (x4 >= nx) return;
    const long long row = (long long)nx;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v00[ROT_ROWS], v10[ROT_ROWS], v11[ROT_ROWS], v01[ROT_ROWS];
    double w00[ROT_ROWS], w10[ROT_ROWS], w11[ROT_ROWS], w01[ROT_ROWS];
#pragma unroll
    for (int r = 0; r < ROT_ROWS; ++r) {
        const int y = y0 + r;
        const double l1 = (double)y, l2 = (double)z;
        // l0 * m4 with m4 == 0 contributes +0 exactly; keep the reference's evaluation order
        const double py = 0.0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
        const double pz = 0.0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
        const double fy = floor(py), fz = floor(pz);
        v00[r] = v10[r] = v11[r] = v01[r] = zero;
        w00[r] = w10[r] = w11[r] = w01[r] = 0.0;
        if (y < ny && fy >= -1.0 && fz >= -1.0 && fy < (double)ny && fz < (double)nz) {
            const int sy = (int)fy, sz = (int)fz;
            const double w1 = py - fy, w2 = pz - fz;
            const double w1n = 1.0 - w1, w2n = 1.0 - w2;
            w00[r] = 1.0 * w1n * w2n; w10[r] = 1.0 * w1 * w2n; w11[r] = 1.0 * w1 * w2; w01[r] = 1.0 * w1n * w2;
            const bool ya = sy >= 0, yb = sy + 1 < ny, za = sz >= 0, zb = sz + 1 < nz;
            if (ya && za) v00[r] = *reinterpret_cast<const float4*>(in + x4 + row * (sy + (long long)ny * sz));
            if (yb && za) v10[r] = *reinterpret_cast<const float4*>(in + x4 + row * (sy + 1 + (long long)ny * sz));
            if (yb && zb) v11[r] = *reinterpret_cast<const float4*>(in + x4 + row * (sy + 1 + (long long)ny * (sz + 1)));
            if (ya && zb) v01[r] = *reinterpret_cast<const float4*>(in + x4 + row * (sy + (long long)ny * (sz + 1)));
        }
    }
#pragma unroll
    for (int r = 0; r < ROT_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= ny) break;
        float4 o;
#define MVSIM_BLEND(c)                                         \
        o.c = (float)((double)v00[r].c * w00[r]);              \
        o.c += (float)((double)v10[r].c * w10[r]);             \
        o.c += (float)((double)v11[r].c * w11[r]);             \
        o.c += (float)((double)v01[r].c * w01[r]);
        MVSIM_BLEND(x) MVSIM_BLEND(y) MVSIM_BLEND(z) MVSIM_BLEND(w)
#undef MVSIM_BLEND
        *reinterpret_cast<float4*>(out + x4 + (long long)nx * (y + (long long)ny * z)) = o;
    }
}

int launch_rotate(hipStream_t s, const float* in, float* out, const int64_t dim[3], const Affine& inv)
{
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    const bool x_identity = inv.m[0] == 1.0 && inv.m[1] == 0.0 && inv.m[2] == 0.0 && inv.m[3] == 0.0 &&
                            inv.m[4] == 0.0 && inv.m[8] == 0.0;
    const bool aligned = (nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    if (x_identity && aligned) {
        const int threads = 128;
        dim3 grid((nx / 4 + threads - 1) / threads, (ny + ROT_ROWS - 1) / ROT_ROWS, nz);
        hipLaunchKernelGGL(k_rotate_axis0_v4, grid, dim3(threads), 0, s, in, out, nx, ny, nz, inv);
    } else {
        const int threads = 256;
        dim3 grid((nx + threads - 1) / threads, ny, nz);
        hipLaunchKernelGGL(k_rotate_generic, grid, dim3(threads), 0, s, in, out, nx, ny, nz, inv);
    }
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// attenuate3d (SimulateMultiViewDataset.java:318-364).  One (x,z) column per lane, lanes along x
// (coalesced 256-B rows per wave per y step), serial fp64 recurrence from y = Ny-1 downwards:
//   phi = v*delta*n ; n = max(n - phi, 0) ; out = (float)(v*n)
// `steps` = Nx (the reference loops dimension(0) times, Q1); rows below stay zero.
// ------------------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(64) void k_attenuate(const float* __restrict__ in, float* __restrict__ out, int nx,
                                                  int ny, int steps, double delta)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int z = blockIdx.y;
    if (x >= nx) return;
    const long long plane = (long long)nx * ny;
    const float* __restrict__ pin = in + plane * z + x;
    float* __restrict__ pout = out + plane * z + x;
    double n = 1.0;
    int y = ny - 1;
    int left = steps;
    while (left >= U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = pin[(long long)(y - u) * nx];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double dv = (double)v[u];
            const double phi = dv * delta * n;
            n = fmax(n - phi, 0.0);
            pout[(long long)(y - u) * nx] = (float)(dv * n);
        }
        y -= U;
        left -= U;
    }
    for (; left > 0; --left, --y) {
        const double dv = (double)pin[(long long)y * nx];
        const double phi = dv * delta * n;
        n = fmax(n - phi, 0.0);
        pout[(long long)y * nx] = (float)(dv * n);
    }
    for (; y >= 0; --y) pout[(long long)y * nx] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// attenuate3d as a wavefront-level prefix scan (option attenuate=scan; north_star's formulation).  In exact arithmetic
// the recurrence n <- max(n - v delta n, 0) is n_k = PROD_{j<=k} max(1 - v_j delta, 0): lanes run ALONG the illumination
// axis (64 consecutive y per wave), every lane forms its factor, an inclusive product scan across the wave (6 shuffle
// steps in fp64) gives the prefix products, and the product of the chunk carries into the next 64 rows.  Tiles of
// 64 (y) x 64 (x) go through LDS so that HBM sees whole rows although lanes walk columns.  This re-associates the fp64
// roundings of the reference's serial loop: results agree with k_attenuate to ~1e-15 relative in n, i.e. the float outputs
// are identical except where v*n falls within that distance of a rounding boundary (measured: < 1e-6 of the voxels, by one
// ulp).  Parallelism comes from y as well as from (x, z): the form for thin volumes; the serial kernels, which are
// bit-exact against the oracle, stay the default.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attenuate_scan(const float* __restrict__ in, float* __restrict__ out, int nx, int ny,
                                                        int steps, double delta)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, z = blockIdx.y;
    const long long plane = (long long)nx * ny;
    const float* __restrict__ pin = in + plane * z;
    float* __restrict__ pout = out + plane * z;
    const bool xin = x0 + lane < nx;
    double carry[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) carry[i] = 1.0;
    for (int j0 = 0; j0 < steps; j0 += 64) {
        // rows y = ny - 1 - (j0 + r), r = 0..63: wave w brings in rows 16 w .. 16 w + 15, lanes along x
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = wave * 16 + rr;
            float v = 0.f;
            if (j0 + r < steps && xin) v = pin[(long long)(ny - 1 - (j0 + r)) * nx + x0 + lane];
            tile[r][lane] = v;
        }
        __syncthreads();
        const bool live = j0 + lane < steps;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = wave * 16 + i;
            const double dv = (double)tile[lane][c];
            double p = live ? fmax(1.0 - dv * delta, 0.0) : 1.0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double t = __shfl_up(p, d, 64);
                if (lane >= d) p *= t;
            }
            const double n = carry[i] * p;
            tile[lane][c] = (float)(dv * n);
            carry[i] = carry[i] * __shfl(p, 63, 64);
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = wave * 16 + rr;
            if (j0 + r < steps && xin) pout[(long long)(ny - 1 - (j0 + r)) * nx + x0 + lane] = tile[r][lane];
        }
        __syncthreads();
    }
    // rows the reference never visits (Ny > Nx): the attenuated image stays zero there
    for (int yy = ny - 1 - steps - wave; yy >= 0; yy -= 4)
        if (xin) pout[(long long)yy * nx + x0 + lane] = 0.f;
}

int launch_attenuate_scan(hipStream_t s, const float* in, float* out, const int64_t dim[3], double delta)
{
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    dim3 grid((nx + 63) / 64, nz);
    hipLaunchKernelGGL(k_attenuate_scan, grid, dim3(256), 0, s, in, out, nx, ny, nx, delta);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

int launch_attenuate(hipStream_t s, const float* in, float* out, const int64_t dim[3], double delta)
{
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    dim3 grid((nx + 63) / 64, nz);
    hipLaunchKernelGGL(k_attenuate<16>, grid, dim3(64), 0, s, in, out, nx, ny, nx, delta);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused rotate (about x) + attenuate for the per-view pipeline: a lane owns two adjacent (x,z) columns and
// walks y from Ny-1 downwards; every step blends the 4 source rows exactly as k_rotate_axis0_v4 does and feeds
// the value straight into the attenuation recurrence, so `rot` never makes the round trip through HBM
// (saves 8 N bytes per view).  Same arithmetic, same order => bit-identical to the two separate kernels.
// rot_out may be null.
// ------------------------------------------------------------------------------------------------
template <int U, bool WRITE_ROT>
__global__ __launch_bounds__(64) void k_rotate_attenuate_axis0(const float* __restrict__ in, float* __restrict__ rot_out,
                                                               float* __restrict__ att_out, int nx, int ny, int nz,
                                                               int steps, Affine a, double delta, int z_off)
{
    // blockIdx.y counts the planes of the OUTPUT buffers, which start at plane z_off of the view (z-slab tiling; 0
    // for a whole view); the source volume is always the whole ground truth
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int z = blockIdx.y + z_off;
    if (x >= nx) return;
    const long long row = (long long)nx;
    const long long plane = row * ny;
    float* __restrict__ patt = att_out + plane * blockIdx.y + x;
    float* __restrict__ prot = WRITE_ROT ? rot_out + plane * blockIdx.y + x : nullptr;
    const float* __restrict__ pin = in + x;
    const double l2 = (double)z;
    double n = 1.0;
    int y = ny - 1;
    int left = steps;
    while (left > 0) {
        float v00[U], v10[U], v11[U], v01[U];
        double w00[U], w10[U], w11[U], w01[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int yy = y - u;
            v00[u] = v10[u] = v11[u] = v01[u] = 0.f;
            w00[u] = w10[u] = w11[u] = w01[u] = 0.0;
            if (u < left) {
                const double l1 = (double)yy;
                const double py = 0.0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
                const double pz = 0.0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
                const double fy = floor(py), fz = floor(pz);
                if (fy >= -1.0 && fz >= -1.0 && fy < (double)ny && fz < (double)nz) {
                    const int sy = (int)fy, sz = (int)fz;
                    const double w1 = py - fy, w2 = pz - fz;
                    const double w1n = 1.0 - w1, w2n = 1.0 - w2;
                    w00[u] = 1.0 * w1n * w2n; w10[u] = 1.0 * w1 * w2n; w11[u] = 1.0 * w1 * w2; w01[u] = 1.0 * w1n * w2;
                    const bool ya = sy >= 0, yb = sy + 1 < ny, za = sz >= 0, zb = sz + 1 < nz;
                    if (ya && za) v00[u] = pin[row * (sy + (long long)ny * sz)];
                    if (yb && za) v10[u] = pin[row * (sy + 1 + (long long)ny * sz)];
                    if (yb && zb) v11[u] = pin[row * (sy + 1 + (long long)ny * (sz + 1))];
                    if (ya && zb) v01[u] = pin[row * (sy + (long long)ny * (sz + 1))];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < left) {
                const int yy = y - u;
                float r = (float)((double)v00[u] * w00[u]);
                r += (float)((double)v10[u] * w10[u]);
                r += (float)((double)v11[u] * w11[u]);
                r += (float)((double)v01[u] * w01[u]);
                if (WRITE_ROT) prot[(long long)yy * nx] = r;
                const double d = (double)r;
                n = fmax(n - d * delta * n, 0.0);
                patt[(long long)yy * nx] = (float)(d * n);
            }
        }
        y -= U;
        left -= U;
    }
    // rows the reference never visits (Ny > Nx): attenuated image stays zero; rot still needs its values
    for (int yy = ny - 1 - steps; yy >= 0; --yy) {
        patt[(long long)yy * nx] = 0.f;
        if (WRITE_ROT) {
            const double l1 = (double)yy;
            const double py = 0.0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
            const double pz = 0.0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
            const double fy = floor(py), fz = floor(pz);
            float o = 0.f;
            if (fy >= -1.0 && fz >= -1.0 && fy < (double)ny && fz < (double)nz) {
                const int sy = (int)fy, sz = (int)fz;
                const double w1 = py - fy, w2 = pz - fz;
                const double w1n = 1.0 - w1, w2n = 1.0 - w2;
                const double q00 = 1.0 * w1n * w2n, q10 = 1.0 * w1 * w2n, q11 = 1.0 * w1 * w2, q01 = 1.0 * w1n * w2;
                const bool ya = sy >= 0, yb = sy + 1 < ny, za = sz >= 0, zb = sz + 1 < nz;
                const float a00 = (ya && za) ? pin[row * (sy + (long long)ny * sz)] : 0.f;
                const float a10 = (yb && za) ? pin[row * (sy + 1 + (long long)ny * sz)] : 0.f;
                const float a11 = (yb && zb) ? pin[row * (sy + 1 + (long long)ny * (sz + 1))] : 0.f;
                const float a01 = (ya && zb) ? pin[row * (sy + (long long)ny * (sz + 1))] : 0.f;
                o = (float)((double)a00 * q00); o += (float)((double)a10 * q10);
                o += (float)((double)a11 * q11); o += (float)((double)a01 * q01);
            }
            prot[(long long)yy * nx] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Same arithmetic, different decomposition (the production form): the geometry of an output row (y, z) -- the two
// source rows, the four fp64 weights -- is the same for every x, and the scalar unit has no fp64, so the kernel
// above spends half of its vector instructions recomputing wave-uniform values in every lane.  Here a block covers
// the WHOLE x extent of one plane (up to 16 waves), the threads first compute the geometry of the rows of a chunk
// once, one row per thread, into an LDS table, and every lane then fetches a row's entry with same-address (broadcast)
// LDS reads while it walks y.  The element offset of the first source row and the row class come back to the scalar
// unit through v_readfirstlane, so the four row loads and the stores use scalar base addresses and the lane's
// constant x offset.  Row classes: 0 = no tap inside the volume (output is +0, attenuation state unchanged -- what the
// generic arithmetic produces for four zero taps), 1 = all four source rows inside, 2 = some inside (per-tap mask).
// Rounding points unchanged => bit-identical to k_rotate_axis0_v4 + k_attenuate and to the kernel above.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) RowGeo {
    double    w00, w10, w11, w01;
    long long off00;            // byte offset of source row (sy, sz), x = 0; only meaningful for the taps that are inside
    int       kind;             // 0 none, 1 all four taps inside, 2 partial: bits 8..11 = inside(00, 10, 11, 01)
    int       pad;
};
constexpr int GEO_CHUNK = 512;  // rows per LDS table (24 KB)

template <int U, bool WRITE_ROT>
__global__ __launch_bounds__(1024) void k_rotate_attenuate_axis0_lds(const float* __restrict__ in, float* __restrict__ rot_out,
                                                                     float* __restrict__ att_out, int nx, int ny, int nz,
                                                                     int steps, Affine a, double delta, int z_off, int z_cnt,
                                                                     const Affine* __restrict__ atab, long long view_stride)
{
    __shared__ RowGeo geo[GEO_CHUNK];
    __shared__ int bclass[GEO_CHUNK / U];
    if (atab) {
        // stacked views of ONE ground truth (mvsim_simulate_views_dev): blockIdx.z names the view -- its inverse model from the table,
        // its planes `view_stride` voxels further on in the outputs
        a = atab[blockIdx.z];
        att_out += (long long)blockIdx.z * view_stride;
        if (WRITE_ROT) rot_out += (long long)blockIdx.z * view_stride;
    }
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    // Plane order: consecutive block ids go to different XCDs (round-robin dispatch, for speed only), and the output
    // planes z and z + 1 read the same source rows; giving every XCD a contiguous slab of planes keeps that reuse inside
    // one L2 (otherwise each source row is fetched from HBM by two XCDs: measured 1.07 GB instead of 0.54 GB of reads).
    const int slab = (gridDim.y + 7) / 8;
    const int zl = (int)(blockIdx.y % 8u) * slab + (int)(blockIdx.y / 8u);
    if (zl >= z_cnt) return;                          // whole block: uniform
    const int z = zl + z_off;
    const bool active = x < nx;                       // no early exit past this point: every thread takes part in the barriers
    const long long row = (long long)nx;
    const long long plane = row * ny;
    const long long out_plane = plane * zl;           // output buffers start at plane z_off of the view
    const double l2 = (double)z;
    // byte-addressed views for the straight-line path: scalar 64-bit bases plus one 32-bit per-lane offset
    const char* __restrict__ in_b = reinterpret_cast<const char*>(in);
    const long long row_b = row * 4, plane_b = plane * 4;
    const unsigned xoff = (unsigned)(active ? x : nx - 1) * 4u;
    double n = 1.0;
    for (int c0 = 0; c0 < steps; c0 += GEO_CHUNK) {
        const int cnt = min(GEO_CHUNK, steps - c0);
        if (c0 > 0) __syncthreads();                  // the previous chunk's readers are done
        for (int r = threadIdx.x; r < cnt; r += blockDim.x) {
            const int yy = ny - 1 - (c0 + r);
            const double l1 = (double)yy;
            // l0 * m4 with m4 == 0 contributes +0 exactly; keep the reference's evaluation order
            const double py = 0.0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
            const double pz = 0.0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
            const double fy = floor(py), fz = floor(pz);
            RowGeo g;
            g.w00 = g.w10 = g.w11 = g.w01 = 0.0;
            g.off00 = 0;
            g.kind = 0;
            g.pad = 0;
            if (fy >= -1.0 && fz >= -1.0 && fy < (double)ny && fz < (double)nz) {
                const int sy = (int)fy, sz = (int)fz;
                const double w1 = py - fy, w2 = pz - fz;
                const double w1n = 1.0 - w1, w2n = 1.0 - w2;
                g.w00 = 1.0 * w1n * w2n; g.w10 = 1.0 * w1 * w2n; g.w11 = 1.0 * w1 * w2; g.w01 = 1.0 * w1n * w2;
                const bool ya = sy >= 0, yb = sy + 1 < ny, za = sz >= 0, zb = sz + 1 < nz;
                g.off00 = row * (sy + (long long)ny * sz) * 4;          // BYTE offset
                const int m = ((ya && za) ? 1 : 0) | ((yb && za) ? 2 : 0) | ((yb && zb) ? 4 : 0) | ((ya && zb) ? 8 : 0);
                g.kind = m == 15 ? 1 : (m == 0 ? 0 : (2 | (m << 8)));
            }
            geo[r] = g;
        }
        __syncthreads();
        // class of every batch of U rows: 1 = all rows have their four taps inside (straight-line code), 0 = all rows are
        // outside (zeros), 2 = mixed or incomplete batch (generic code)
        for (int bq = threadIdx.x; bq < (cnt + U - 1) / U; bq += blockDim.x) {
            bool all1 = true, all0 = true;
            for (int u = 0; u < U; ++u) {
                const int r = bq * U + u;
                if (r < cnt) {
                    const int k = geo[r].kind;
                    all1 = all1 && k == 1;
                    all0 = all0 && k == 0;
                } else {
                    all1 = false;
                    all0 = false;
                }
            }
            bclass[bq] = all1 ? 1 : (all0 ? 0 : 2);
        }
        __syncthreads();
        for (int r0 = 0; r0 < cnt; r0 += U) {
            const int cls = __builtin_amdgcn_readfirstlane(bclass[r0 / U]);
            const int y0 = ny - 1 - (c0 + r0);                        // rows y0, y0 - 1, ..., y0 - U + 1
            if (cls == 1) {
                // straight-line: U offsets from the table, U x 4 row loads (scalar base + the lane's constant byte offset;
                // lanes beyond nx read the last column, only their stores are masked), then the U dependent steps
                long long offs[U];
#pragma unroll
                for (int u = 0; u < U; ++u) offs[u] = geo[r0 + u].off00;
                float v00[U], v10[U], v11[U], v01[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)offs[u]);
                    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)offs[u] >> 32));
                    const char* __restrict__ p00 = in_b + (long long)(((unsigned long long)hi << 32) | lo);   // scalar base
                    v00[u] = *reinterpret_cast<const float*>(p00 + xoff);
                    v10[u] = *reinterpret_cast<const float*>(p00 + row_b + xoff);
                    v11[u] = *reinterpret_cast<const float*>(p00 + row_b + plane_b + xoff);
                    v01[u] = *reinterpret_cast<const float*>(p00 + plane_b + xoff);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const RowGeo* g = &geo[r0 + u];
                    float r = (float)((double)v00[u] * g->w00);
                    r += (float)((double)v10[u] * g->w10);
                    r += (float)((double)v11[u] * g->w11);
                    r += (float)((double)v01[u] * g->w01);
                    const double d = (double)r;
                    n = fmax(n - d * delta * n, 0.0);
                    if (active) {
                        const long long ob = (out_plane + (long long)(y0 - u) * row) * 4;                    // scalar
                        if (WRITE_ROT) *reinterpret_cast<float*>(reinterpret_cast<char*>(rot_out) + ob + xoff) = r;
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(att_out) + ob + xoff) = (float)(d * n);
                    }
                }
            } else if (cls == 0) {
                // four zero taps per row: r = +0, n = max(n - 0 * delta * n, 0) = n, out = (float)(0 * n) = +0
                if (active) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const long long o = out_plane + (long long)(y0 - u) * row;
                        if (WRITE_ROT) rot_out[o + x] = 0.f;
                        att_out[o + x] = 0.f;
                    }
                }
            } else {
                for (int u = 0; u < U && r0 + u < cnt; ++u) {
                    const RowGeo* g = &geo[r0 + u];
                    const int kind = __builtin_amdgcn_readfirstlane(g->kind);
                    const long long o = out_plane + (long long)(y0 - u) * row;
                    float r = 0.f;
                    if (kind != 0) {
                        const float* __restrict__ p00 = reinterpret_cast<const float*>(in_b + g->off00);
                        const bool t00 = kind == 1 || (kind & (1 << 8)), t10 = kind == 1 || (kind & (2 << 8));
                        const bool t11 = kind == 1 || (kind & (4 << 8)), t01 = kind == 1 || (kind & (8 << 8));
                        const float a00 = (active && t00) ? p00[x] : 0.f;
                        const float a10 = (active && t10) ? p00[row + x] : 0.f;
                        const float a11 = (active && t11) ? p00[row + plane + x] : 0.f;
                        const float a01 = (active && t01) ? p00[plane + x] : 0.f;
                        r = (float)((double)a00 * g->w00);
                        r += (float)((double)a10 * g->w10);
                        r += (float)((double)a11 * g->w11);
                        r += (float)((double)a01 * g->w01);
                        const double d = (double)r;
                        n = fmax(n - d * delta * n, 0.0);
                        if (active) att_out[o + x] = (float)(d * n);
                    } else if (active) {
                        att_out[o + x] = 0.f;
                    }
                    if (WRITE_ROT && active) rot_out[o + x] = r;
                }
            }
        }
    }
    // rows the reference never visits (Ny > Nx): attenuated image stays zero; rot still needs its values
    for (int yy = ny - 1 - steps; yy >= 0; --yy) {
        if (!active) break;
        att_out[out_plane + (long long)yy * row + x] = 0.f;
        if (WRITE_ROT) {
            const double l1 = (double)yy;
            const double py = 0.0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
            const double pz = 0.0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
            const double fy = floor(py), fz = floor(pz);
            float o = 0.f;
            if (fy >= -1.0 && fz >= -1.0 && fy < (double)ny && fz < (double)nz) {
                const int sy = (int)fy, sz = (int)fz;
                const double w1 = py - fy, w2 = pz - fz;
                const double w1n = 1.0 - w1, w2n = 1.0 - w2;
                const double q00 = 1.0 * w1n * w2n, q10 = 1.0 * w1 * w2n, q11 = 1.0 * w1 * w2, q01 = 1.0 * w1n * w2;
                const bool ya = sy >= 0, yb = sy + 1 < ny, za = sz >= 0, zb = sz + 1 < nz;
                const float* __restrict__ pin = in + x;
                const float a00 = (ya && za) ? pin[row * (sy + (long long)ny * sz)] : 0.f;
                const float a10 = (yb && za) ? pin[row * (sy + 1 + (long long)ny * sz)] : 0.f;
                const float a11 = (yb && zb) ? pin[row * (sy + 1 + (long long)ny * (sz + 1))] : 0.f;
                const float a01 = (ya && zb) ? pin[row * (sy + (long long)ny * (sz + 1))] : 0.f;
                o = (float)((double)a00 * q00); o += (float)((double)a10 * q10);
                o += (float)((double)a11 * q11); o += (float)((double)a01 * q01);
            }
            rot_out[out_plane + (long long)yy * row + x] = o;
        }
    }
}

// returns MVSIM_OK and sets *fused = false when the fast-path conditions do not hold (caller runs the two kernels)
int launch_rotate_attenuate(hipStream_t s, const float* in, float* rot_or_null, float* att, const int64_t dim[3],
                            const Affine& inv, double delta, int fused_mode, bool* fused)
{
    return launch_rotate_attenuate_planes(s, in, rot_or_null, att, dim, inv, delta, 0, (int)dim[2], fused_mode, fused);
}

// planes [z_begin, z_begin + z_count) of the view into buffers that start at plane z_begin
int launch_rotate_attenuate_planes(hipStream_t s, const float* in, float* rot_or_null, float* att, const int64_t dim[3],
                                   const Affine& inv, double delta, int z_begin, int z_count, int fused_mode, bool* fused)
{
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    const bool x_identity = inv.m[0] == 1.0 && inv.m[1] == 0.0 && inv.m[2] == 0.0 && inv.m[3] == 0.0 &&
                            inv.m[4] == 0.0 && inv.m[8] == 0.0;
    // fused_mode: 0 separate kernels, 1 geometry table in LDS, 2 every lane recomputes the row geometry, 3 = 1 when the
    // columns fill the chip (>= 2 waves per SIMD), else 0
    // (the separate rotate is only fast in its 16-byte form: Nx % 4 == 0, aligned rows; the generic one is 3x slower than
    // the fused kernel -- the reference's own 289^3 volume: 0.38 ms against 0.10 ms)
    if (fused_mode == 3) {
        const bool vec4 = nx % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(att)) % 16 == 0);
        fused_mode = ((int64_t)nx * z_count >= 131072 || !vec4) ? 1 : 0;
    }
    *fused = x_identity && fused_mode != 0;
    if (!*fused) return MVSIM_OK;
    if (fused_mode == 1) {
        // one block per plane and 1024-column stripe: nx / 64 waves (at most 16) share one geometry table
        const int waves = (nx + 63) / 64 < 16 ? (nx + 63) / 64 : 16;
        dim3 grid_l((nx + waves * 64 - 1) / (waves * 64), (z_count + 7) / 8 * 8), block_l(waves * 64);
        if (rot_or_null)
            hipLaunchKernelGGL((k_rotate_attenuate_axis0_lds<8, true>), grid_l, block_l, 0, s, in, rot_or_null, att, nx, ny, nz, nx, inv, delta, z_begin, z_count,
                               (const Affine*)nullptr, 0ll);
        else
            hipLaunchKernelGGL((k_rotate_attenuate_axis0_lds<8, false>), grid_l, block_l, 0, s, in, rot_or_null, att, nx, ny, nz, nx, inv, delta, z_begin, z_count,
                               (const Affine*)nullptr, 0ll);
        MVSIM_HIP(hipGetLastError());
        return MVSIM_OK;
    }
    dim3 grid((nx + 63) / 64, z_count);
    if (rot_or_null)
        hipLaunchKernelGGL((k_rotate_attenuate_axis0<8, true>), grid, dim3(64), 0, s, in, rot_or_null, att, nx, ny, nz, nx, inv, delta, z_begin);
    else
        hipLaunchKernelGGL((k_rotate_attenuate_axis0<8, false>), grid, dim3(64), 0, s, in, rot_or_null, att, nx, ny, nz, nx, inv, delta, z_begin);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// `nviews` views of one ground truth (rotation about x each, inverse models in the device table `atab`) into the stacked attenuated
// volumes att[v][Nz][Ny][Nx]: ONE launch whose grid carries the view index, so that views too small to fill the chip fill it
// together -- a 128^3 view is 256 waves of serial latency, eight of them are two waves per SIMD.
int launch_rotate_attenuate_views(hipStream_t s, const float* in, float* att, const int64_t dim[3], const Affine* atab, int nviews, double delta)
{
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    const int waves = (nx + 63) / 64 < 16 ? (nx + 63) / 64 : 16;
    dim3 grid_l((nx + waves * 64 - 1) / (waves * 64), (nz + 7) / 8 * 8, nviews), block_l(waves * 64);
    hipLaunchKernelGGL((k_rotate_attenuate_axis0_lds<8, false>), grid_l, block_l, 0, s, in, (float*)nullptr, att, nx, ny, nz, nx, Affine{}, delta, 0, nz,
                       atab, (long long)nx * ny * nz);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Sum (Tools.sumImage, Tools.java:124-132; mpicbg RealSum ~ exact-to-double): per-thread double
// accumulation, wave shuffle + LDS block tree, fixed-order final pass => deterministic.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum_256(double v)
{
    __shared__ double sh[4];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((sh[0] + sh[1]) + (sh[2] + sh[3]));
    __syncthreads();
    return r;  // valid in thread 0
}

__global__ __launch_bounds__(256) void k_sum_partial(const float* __restrict__ in, long long n,
                                                     double* __restrict__ partial)
{
    double acc = 0.0;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * 256;
    const long long n4 = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) ? (n >> 2) : 0;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    for (long long i = tid; i < n4; i += nthreads) {
        const float4 v = in4[i];
        acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
    for (long long i = (n4 << 2) + tid; i < n; i += nthreads) acc += (double)in[i];
    const double b = block_sum_256(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = b;
}

__global__ __launch_bounds__(256) void k_sum_final(const double* __restrict__ partial, int count,
                                                   double* __restrict__ scal)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
    const double b = block_sum_256(acc);
    if (threadIdx.x == 0) scal[0] = b;
}

int launch_sum(hipStream_t s, const float* in, int64_t n, double* partial, double* scal)
{
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > SUM_BLOCKS) blocks = SUM_BLOCKS;
    hipLaunchKernelGGL(k_sum_partial, dim3(blocks), dim3(256), 0, s, in, (long long)n, partial);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, s, partial, blocks, scal);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// adjustImage (Tools.java:143-159): corr = (double)(target - min) / (sum / n)
__global__ void k_adjust_corr(double* scal, long long n, float min_value, float target)
{
    const double avg = scal[0] / (double)n;
    scal[1] = (double)(target - min_value) / avg;
}

int launch_adjust_corr(hipStream_t s, double* scal, int64_t n, float min_value, float target)
{
    hipLaunchKernelGGL(k_adjust_corr, dim3(1), dim3(1), 0, s, scal, (long long)n, min_value, target);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

__global__ __launch_bounds__(256) void k_adjust_apply(float* __restrict__ img, long long n,
                                                      const double* __restrict__ scal, float min_value)
{
    const double corr = scal[1];
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * 256;
    const long long n4 = ((reinterpret_cast<uintptr_t>(img) & 15) == 0) ? (n >> 2) : 0;
    float4* img4 = reinterpret_cast<float4*>(img);
    for (long long i = tid; i < n4; i += nthreads) {
        float4 v = img4[i];
        v.x = adjust_one(v.x, corr, min_value);
        v.y = adjust_one(v.y, corr, min_value);
        v.z = adjust_one(v.z, corr, min_value);
        v.w = adjust_one(v.w, corr, min_value);
        img4[i] = v;
    }
    for (long long i = (n4 << 2) + tid; i < n; i += nthreads) img[i] = adjust_one(img[i], corr, min_value);
}

int launch_adjust_apply(hipStream_t s, float* img, int64_t n, const double* scal, float min_value)
{
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adjust_apply, dim3(blocks), dim3(256), 0, s, img, (long long)n, scal, min_value);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// normImage (Tools.java:112-118): img <- (float)((double)img / sum)
__global__ __launch_bounds__(256) void k_norm_apply(float* __restrict__ img, long long n,
                                                    const double* __restrict__ scal)
{
    const double sum = scal[0];
    const long long nthreads = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += nthreads)
        img[i] = (float)((double)img[i] / sum);
}

int launch_norm_apply(hipStream_t s, float* img, int64_t n, const double* scal)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_norm_apply, dim3(blocks), dim3(256), 0, s, img, (long long)n, scal);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// extractSlices + adjust + Poisson (SimulateMultiViewDataset.java:195-251, Tools.java:73-86):
// out[x,y,k] = f(in[x,y,k*inc]);  ADJUST applies the two adjustImage passes on the fly (fused
// path: the scaled volume is never materialised); NOISE draws Poisson((double)v * mul).
// ------------------------------------------------------------------------------------------------
template <bool ADJUST, bool NOISE>
__global__ __launch_bounds__(256) void k_extract(const float* __restrict__ in, float* __restrict__ out,
                                                 long long plane, long long nzo, int inc, int idx_inc,
                                                 const double* __restrict__ scal, float min_value, double mul,
                                                 uint32_t k0, uint32_t k1, uint32_t stream,
                                                 unsigned long long index_offset, const ExtractView* __restrict__ vt)
{
    if (vt) { const ExtractView e = vt[blockIdx.y]; in = e.in; out = e.out; scal = e.scal; k0 = e.k0; k1 = e.k1; stream = e.stream; }
    double corr = 1.0;
    if (ADJUST) corr = scal[1];
    const long long total = plane * nzo;
    const long long nthreads = (long long)gridDim.x * 256;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += nthreads) {
        const long long k = o / plane;
        const long long i = o - k * plane;
        const long long src = k * inc * plane + i;
        float v = in[src];
        if (ADJUST) v = adjust_one(v, corr, min_value);
        if (NOISE) v = poisson_counter((double)v * mul, k0, k1, stream, index_offset + (unsigned long long)(k * idx_inc * plane + i));
        out[o] = v;
    }
}

// Vector form: 4 consecutive voxels per lane (16-B loads/stores); the lane's 4 voxels are exactly one
// Philox group.  Requires plane % 4 == 0, index_offset % 4 == 0 and 16-B aligned buffers.
template <bool ADJUST, bool NOISE>
__global__ __launch_bounds__(256) void k_extract4(const float* __restrict__ in, float* __restrict__ out,
                                                  long long plane4, long long nzo, int inc,
                                                  const double* __restrict__ scal, float min_value, double mul,
                                                  uint32_t k0, uint32_t k1, uint32_t stream,
                                                  unsigned long long index_offset, const ExtractView* __restrict__ vt)
{
    if (vt) { const ExtractView e = vt[blockIdx.y]; in = e.in; out = e.out; scal = e.scal; k0 = e.k0; k1 = e.k1; stream = e.stream; }
    double corr = 1.0;
    if (ADJUST) corr = scal[1];
    const long long total4 = plane4 * nzo;
    const long long nthreads = (long long)gridDim.x * 256;
    const float4* __restrict__ in4 = reinterpret_cast<const float4*>(in);
    float4* __restrict__ out4 = reinterpret_cast<float4*>(out);
    const bool small32 = total4 < (1ll << 32);
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total4; o += nthreads) {
        long long src4 = o;
        if (inc != 1) {
            const long long k = small32 ? (long long)((unsigned)o / (unsigned)plane4) : o / plane4;
            src4 = k * inc * plane4 + (o - k * plane4);
        }
        float4 v = in4[src4];
        if (ADJUST) {
            v.x = adjust_one(v.x, corr, min_value);
            v.y = adjust_one(v.y, corr, min_value);
            v.z = adjust_one(v.z, corr, min_value);
            v.w = adjust_one(v.w, corr, min_value);
        }
        if (NOISE)
            v = poisson_counter4((double)v.x * mul, (double)v.y * mul, (double)v.z * mul, (double)v.w * mul, k0, k1,
                                 stream, index_offset + 4ull * (unsigned long long)src4);
        out4[o] = v;
    }
}

// Noise form for production sizes, two launches.
//   k_extract4_noise2: every lane busy -- adjust, then phase 1 of the sampler (poisson_dev.h: poisson_phase1): the
//                      "count is 0" shortcut of the low-lambda inversion, and the attempt-0 squeeze of PTRS run densely
//                      over the wave's bright voxels (ballot compaction into a wave-private LDS list).  The ~1/3 of
//                      bright voxels that still need the exact test or a retry, and the few low-lambda voxels whose
//                      count may be >= 1, are appended to a work queue in HBM (per-block segments, LDS append counters:
//                      one global counter would serialise at ~88 atomics/us chip-wide; bright items grow from the front
//                      of the segment, inversion items from its back).
//   k_poisson_resolve: one queue item per lane, looped until resolved; no LDS, no barriers, full occupancy, and
//                      every lane starts with real work -- the divergent fp64 code (logs, divisions) no longer
//                      runs once per voxel slot with 1-in-7 lanes active.
// Same arithmetic per (voxel, attempt) as poisson_counter: bit-identical counts.
template <bool ADJUST, bool CHECKED>
__global__ __launch_bounds__(256) void k_extract4_noise2(const float* __restrict__ in, float* __restrict__ out,
                                                         long long plane4, long long nzo, int inc, int idx_inc,
                                                         const double* __restrict__ scal, float min_value, double mul,
                                                         uint32_t k0, uint32_t k1, uint32_t stream,
                                                         unsigned long long index_offset, PItem* __restrict__ queue,
                                                         unsigned int* __restrict__ qcount, unsigned int segcap,
                                                         const ExtractView* __restrict__ vt)
{
    if (vt) {
        const ExtractView e = vt[blockIdx.y];
        in = e.in; out = e.out; scal = e.scal; k0 = e.k0; k1 = e.k1; stream = e.stream;
        queue = reinterpret_cast<PItem*>(e.queue); qcount = e.qcount;
    }
    __shared__ unsigned long long qctr;
    __shared__ unsigned int qovf[2];
    __shared__ P1Scratch scratch[4];
    if (threadIdx.x == 0) { qctr = 0ull; qovf[0] = 0u; qovf[1] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    P1Args pa;
    pa.mul = mul; pa.mulf = (float)mul; pa.k0 = k0; pa.k1 = k1; pa.stream = stream;
    pa.seg = queue + (unsigned long long)blockIdx.x * segcap; pa.segcap = segcap; pa.ctr = &qctr; pa.ovf = qovf;
    double corr = 1.0;
    if (ADJUST) corr = scal[1];
    const long long total4 = plane4 * nzo;
    const long long nthreads = (long long)gridDim.x * 256;
    const float4* __restrict__ in4 = reinterpret_cast<const float4*>(in);
    float4* __restrict__ out4 = reinterpret_cast<float4*>(out);
    const bool small32 = total4 < (1ll << 32);
    // the trip count is uniform per wave (lanes past the end carry invalid voxels): ballots need every lane
    const long long wave_first = (long long)blockIdx.x * 256 + wave * 64;
    for (long long o0 = wave_first; o0 < total4; o0 += nthreads) {
        const long long o = o0 + lane;
        const bool valid = o < total4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        long long src4 = o, idx4 = o;                 // where the voxels are read / what the RNG counter says they are
        if (valid) {
            if (inc != 1 || idx_inc != 1) {
                const long long k = small32 ? (long long)((unsigned)o / (unsigned)plane4) : o / plane4;
                src4 = k * inc * plane4 + (o - k * plane4);
                idx4 = k * idx_inc * plane4 + (o - k * plane4);
            }
            v = in4[src4];
            if (ADJUST) {
                v.x = adjust_one(v.x, corr, min_value);
                v.y = adjust_one(v.y, corr, min_value);
                v.z = adjust_one(v.z, corr, min_value);
                v.w = adjust_one(v.w, corr, min_value);
            }
        }
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float ov[4];
        poisson_phase1<CHECKED>(vv, valid, index_offset + 4ull * (unsigned long long)idx4, 4ull * (unsigned long long)o, pa, &scratch[wave], lane, ov);
        if (valid) out4[o] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p1_publish(qcount + (size_t)QCOUNT_WORDS * blockIdx.x, qctr, qovf);
        if (blockIdx.x == 0) p1_publish_header(qcount, gridDim.x, segcap);
    }
}

// The same two-launch sampler for planes that are no multiple of four voxels (the reference's own 289^3 run: 83 521 voxels per
// plane) or buffers that are not 16-byte aligned.  Philox groups are four consecutive voxels of the SOURCE index (poisson_dev.h),
// and a plane then starts anywhere inside a group: a lane takes one group of one acquired plane -- up to four voxels, the ones
// that fall inside the plane (scalar loads and stores under a mask; the others enter phase 1 as zeros, which it ignores) --, and the
// 64 lanes of a wave take 64 consecutive groups of the SAME plane, so that a wave's outputs stay consecutive (what phase 1's pair
// compaction assumes).  Same arithmetic per (voxel, attempt) as every other form: bit-identical counts.
template <bool ADJUST, bool CHECKED>
__global__ __launch_bounds__(256) void k_extract_noise2_any(const float* __restrict__ in, float* __restrict__ out,
                                                            long long plane, long long nzo, int inc, int idx_inc,
                                                            const double* __restrict__ scal, float min_value, double mul,
                                                            uint32_t k0, uint32_t k1, uint32_t stream,
                                                            unsigned long long index_offset, PItem* __restrict__ queue,
                                                            unsigned int* __restrict__ qcount, unsigned int segcap,
                                                            long long slots_per_plane, const ExtractView* __restrict__ vt)
{
    if (vt) {
        const ExtractView e = vt[blockIdx.y];
        in = e.in; out = e.out; scal = e.scal; k0 = e.k0; k1 = e.k1; stream = e.stream;
        queue = reinterpret_cast<PItem*>(e.queue); qcount = e.qcount;
    }
    __shared__ unsigned long long qctr;
    __shared__ unsigned int qovf[2];
    __shared__ P1Scratch scratch[4];
    if (threadIdx.x == 0) { qctr = 0ull; qovf[0] = 0u; qovf[1] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    P1Args pa;
    pa.mul = mul; pa.mulf = (float)mul; pa.k0 = k0; pa.k1 = k1; pa.stream = stream;
    pa.seg = queue + (unsigned long long)blockIdx.x * segcap; pa.segcap = segcap; pa.ctr = &qctr; pa.ovf = qovf;
    double corr = 1.0;
    if (ADJUST) corr = scal[1];
    const long long slots = slots_per_plane * nzo;        // wave slots: 64 groups each
    for (long long sl = (long long)blockIdx.x * 4 + wave; sl < slots; sl += (long long)gridDim.x * 4) {
        const long long k = sl / slots_per_plane, jb = sl - k * slots_per_plane;
        const unsigned long long ibase = index_offset + (unsigned long long)(k * idx_inc) * (unsigned long long)plane;   // RNG index of the plane's voxel 0
        const unsigned long long g = (ibase >> 2) + (unsigned long long)(jb * 64 + lane);      // this lane's Philox group
        const long long i0 = (long long)(4ull * g - ibase);                                   // its first voxel inside the plane (may be < 0)
        const float* __restrict__ src = in + k * inc * plane;
        float vv[4];
        bool any = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long i = i0 + c;
            const bool ok = i >= 0 && i < plane;
            float v = 0.f;
            if (ok) {
                v = src[i];
                if (ADJUST) v = adjust_one(v, corr, min_value);
            }
            vv[c] = ok ? v : 0.f;
            any |= ok;
        }
        float ov[4];
        // (output position of component 0; negative for a plane's first group when the plane starts inside it -- the valid components
        // land at non-negative positions all the same, in 64-bit wrap-around arithmetic)
        poisson_phase1<CHECKED>(vv, any, 4ull * g, (unsigned long long)(k * plane + i0), pa, &scratch[wave], lane, ov);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long i = i0 + c;
            if (i >= 0 && i < plane) out[k * plane + i] = ov[c];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p1_publish(qcount + (size_t)QCOUNT_WORDS * blockIdx.x, qctr, qovf);
        if (blockIdx.x == 0) p1_publish_header(qcount, gridDim.x, segcap);
    }
}

// One block per queue segment (same grid as k_extract4_noise2; the grid-stride walk of that kernel spreads the
// bright voxels evenly over the segments).
__global__ __launch_bounds__(256) void k_poisson_resolve(ResolveJob job, const ExtractView* __restrict__ vt)
{
    if (vt) {
        const ExtractView e = vt[blockIdx.y];
        job.out = e.out; job.queue = reinterpret_cast<const PItem*>(e.queue); job.qcount = e.qcount;
        job.k0 = e.k0; job.k1 = e.k1; job.stream = e.stream;
    }
    __shared__ unsigned int ticket;
    __shared__ double tab[RESOLVE_TAB];
    resolve_segment_body(job, (long long)blockIdx.x, (int)threadIdx.x, &ticket, tab);
}

// Queues whose segments hold a SHARE of their blocks' voxels (poisson_queue_share < 16): the voxels a full segment refused, sampled
// where they stand.  A kernel of its own so that this divergent fp64 code costs neither phase 1 nor the resolver a register; its blocks
// return at once unless the resolver has recorded a refusal in the queue's header -- on the bench's volumes, always.
constexpr int REFUSED_BLOCKS = 2048;
constexpr unsigned int REFUSED_LIST = 4096;                // positions the block gathers before it samples them (16 KB of LDS)
__global__ __launch_bounds__(256) void k_poisson_refused(ResolveJob job, int segments, unsigned int full_items, unsigned int* hint,
                                                         const ExtractView* __restrict__ vt)
{
    if (vt) {
        const ExtractView e = vt[blockIdx.y];
        job.out = e.out; job.qcount = e.qcount; job.k0 = e.k0; job.k1 = e.k1; job.stream = e.stream;
    }
    if (job.qcount[QCOUNT_HEADER + 2] == 0u) return;                 // no block of this view was refused anything (the resolver's word)
    __shared__ unsigned int list[REFUSED_LIST];
    __shared__ unsigned int count;
    const int t = (int)threadIdx.x;
    if (t == 0) count = 0u;
    __syncthreads();
    // refused voxels are a few per cent of a block's voxels, scattered: sampled where the walk finds them, one lane in twenty would work.
    // So the walk only gathers positions, and the list is sampled whenever another trip (1024 candidates) might not fit: all lanes busy.
    auto flush = [&]() {
        __syncthreads();                                              // the list is complete
        const unsigned int m = count;
        for (unsigned int i = (unsigned int)t; i < m; i += 256u) {
            const unsigned int o = list[i];
            job.out[o] = resolve_in_place(-job.out[o], job, resolve_index_of(job, o));
        }
        __syncthreads();                                              // every lane has read `count` and its entries
        if (t == 0) count = 0u;
        __syncthreads();
    };
    unsigned int need = 0u;                                           // sixteenths of its voxels the fullest of this block's segments had pending
    for (int seg = (int)blockIdx.x; seg < segments; seg += (int)gridDim.x) {
        const unsigned int* qc = job.qcount + (size_t)QCOUNT_WORDS * seg;
        if (qc[2] == 0u) continue;                                    // block-uniform
        const unsigned int sixteenths = (unsigned int)((16ull * (qc[0] + qc[1] + qc[2]) + full_items - 1u) / full_items);
        need = sixteenths > need ? sixteenths : need;
        for (long long trip = 0;; ++trip) {
            if (!refused_collect(job, seg, trip, t, segments, list, &count)) break;      // block-uniform
            __syncthreads();
            const unsigned int gathered = count;                      // the same for every lane: read between two barriers
            __syncthreads();
            if (gathered + 1024u > REFUSED_LIST) flush();             // the next trip adds up to 1024 positions
        }
    }
    flush();
    // what a context on the automatic share builds its next queue with (api.cpp: queue_mode_next reads the word without synchronising)
    if (t == 0 && hint && need != 0u) __hip_atomic_fetch_max(hint, need > 16u ? 16u : need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int launch_poisson_resolve(hipStream_t s, float* out, void* queue_items, const unsigned int* qcount, int segments, unsigned int segcap,
                           double mul, uint64_t seed, uint32_t stream, long long plane, int idx_inc, uint64_t index_offset)
{
    const ResolveJob job{out, reinterpret_cast<const PItem*>(queue_items), qcount, segcap, mul, (uint32_t)seed, (uint32_t)(seed >> 32), stream,
                         (unsigned int)plane, (unsigned int)idx_inc, (unsigned long long)index_offset, 0, 0, 0};     // full segments: no refusals
    hipLaunchKernelGGL(k_poisson_resolve, dim3(segments), dim3(256), 0, s, job, (const ExtractView*)nullptr);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// Work-queue geometry for n_out output voxels: `blocks` blocks of 256 lanes x 4 voxels walk the volume with a
// grid stride; each owns a segment of `share` sixteenths of its voxels (16: every voxel, 16 B per output voxel of HBM
// workspace and no refusals; less: what does not fit is sampled in place by phase 1, poisson_dev.h).
constexpr size_t QCOUNT_BYTES = (size_t)QCOUNT_HEADER * sizeof(unsigned int) + 256;    // the [counts][header] in front of the segments
static_assert(QCOUNT_BYTES % 256 == 0, "the segments start 16-byte aligned");

// the share a queue of n_out voxels is built with.  Auto: what the context has learned its views need, from QUEUE_SHARE_START sixteenths up
// (at 512^3 and the bench's SNR the sphere phantom's fullest block has 14 % of its voxels pending, that of a volume without an empty voxel
// 68 %: profiles/r05_queue_share.txt), but small queues (<= 64 MiB at full size: up to 160^3 acquired
// voxels) are not worth the extra launch that looks for refused voxels
static int share_for(long long n_out, int share)
{
    if (share >= QUEUE_SHARE_AUTO) {
        const int learned = share - QUEUE_SHARE_AUTO;
        return n_out <= (4ll << 20) ? 16 : (learned >= 16 ? 16 : (learned > QUEUE_SHARE_START ? learned : QUEUE_SHARE_START));
    }
    return share >= 16 ? 16 : (share < 1 ? 1 : share);
}

static unsigned int segment_share(long long worst, int share)
{
    if (share >= 16) return (unsigned int)worst;
    long long c = (worst * (share < 1 ? 1 : share) + 15) / 16;
    c = (c + 63) & ~63ll;                                   // at least one wave of items, whole waves after that
    return (unsigned int)(c < worst ? c : worst);
}

static void poisson_geometry(int64_t n_out, int share, int* blocks, unsigned int* segcap)
{
    long long want = (n_out / 4 + 255) / 256;
    const int b = (int)(want < 1 ? 1 : (want > POISSON_MAX_BLOCKS ? POISSON_MAX_BLOCKS : want));
    const long long iters = (n_out / 4 + (long long)b * 256 - 1) / ((long long)b * 256);
    *blocks = b;
    *segcap = segment_share(iters * 1024, share);   // worst case every voxel of the block: the squeeze accepts only ~35 % at lambda = 10
}

static size_t poisson_queue_bytes(int64_t n_out, int share)
{
    int blocks;
    unsigned int segcap;
    poisson_geometry(n_out, share, &blocks, &segcap);
    return QCOUNT_BYTES + (size_t)blocks * segcap * sizeof(PItem);   // [counts][segments]
}

// The same for k_extract_noise2_any: a wave slot is 64 Philox groups of ONE plane (a plane of `plane` voxels that starts anywhere
// inside a group touches up to plane / 4 + 1 of them, rounded up to whole slots), `blocks` blocks of four waves walk the slots with a
// grid stride, and a block's segment holds `share` sixteenths of the voxels of its trips.
static void poisson_geometry_any(long long plane, long long nzo, int share, int* blocks, unsigned int* segcap, long long* slots_per_plane)
{
    const long long spp = ((plane + 3) / 4 + 1 + 63) / 64;
    const long long slots = spp * nzo;
    long long want = (slots + 3) / 4;
    const int b = (int)(want < 1 ? 1 : (want > POISSON_MAX_BLOCKS ? POISSON_MAX_BLOCKS : want));
    const long long trips = (slots + (long long)b * 4 - 1) / ((long long)b * 4);
    *blocks = b; *segcap = segment_share(trips * 1024, share); *slots_per_plane = spp;
}

// bytes of queue workspace for nzo acquired planes of `plane` voxels, whichever of the two kernels takes them
size_t poisson_queue_bytes_planes(long long plane, long long nzo, int share)
{
    int blocks;
    unsigned int segcap;
    long long spp;
    share = share_for(plane * nzo, share);
    poisson_geometry_any(plane, nzo, share, &blocks, &segcap, &spp);
    const size_t any = QCOUNT_BYTES + (size_t)blocks * segcap * sizeof(PItem);
    const size_t vec = poisson_queue_bytes(plane * nzo, share);
    return any > vec ? any : vec;
}

// What the last two-launch sampler that used this workspace queued (mvsim_get_queue_stats): {items a segment holds, bright items,
// inversion items, voxels refused and sampled in place, pending voxels of the fullest block}.  The caller has synchronised the stream.
int poisson_queue_read_stats(const void* queue_ws, size_t bytes, long long stats[5])
{
    stats[0] = stats[1] = stats[2] = stats[3] = stats[4] = 0;
    if (!queue_ws || bytes < QCOUNT_BYTES) return MVSIM_OK;           // not a queue of the two-launch sampler
    std::vector<unsigned int> h((size_t)QCOUNT_HEADER + 2);
    MVSIM_HIP(hipMemcpy(h.data(), queue_ws, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
    const unsigned int blocks = h[QCOUNT_HEADER] <= (unsigned int)POISSON_MAX_BLOCKS ? h[QCOUNT_HEADER] : 0u;
    stats[0] = h[(size_t)QCOUNT_HEADER + 1];
    stats[1] = stats[2] = stats[3] = stats[4] = 0;
    for (unsigned int b = 0; b < blocks; ++b) {
        const long long f = h[(size_t)QCOUNT_WORDS * b], k = h[(size_t)QCOUNT_WORDS * b + 1], r = h[(size_t)QCOUNT_WORDS * b + 2];
        stats[1] += f;
        stats[2] += k;
        stats[3] += r;
        if (f + k + r > stats[4]) stats[4] = f + k + r;     // what the fullest block had to settle: the segment size that refuses nothing
    }
    return MVSIM_OK;
}

// nviews > 0: the same launch for `nviews` views whose inputs, outputs, [sum, factor] slots, RNG keys and queue workspaces come from
// the device table `vt` (blockIdx.y = view; `vec_all`: every view's buffers allow the 16-byte form; in / out / scal / seed / stream /
// queue_ws arguments unused)
static int launch_extract_impl(hipStream_t s, const float* in, float* out, const int64_t dim[3], int inc, bool adjust,
                               const double* scal, float min_value, bool noise, double mul, uint64_t seed,
                               uint32_t stream, uint64_t index_offset, void* queue_ws, QueueMode queue_mode, int index_inc,
                               int nviews, const ExtractView* vt, bool vec_all)
{
    // index_inc: plane stride of the RNG counter when it differs from the plane stride of the reads (a compact input
    // that holds only the planes k * index_inc of the source volume); 0 = the same as inc
    if (index_inc <= 0) index_inc = inc;
    const unsigned gy = nviews > 0 ? (unsigned)nviews : 1u;
    const long long plane = (long long)dim[0] * dim[1];
    // phase 1 hands a slot's RNG counters across lanes as 32-bit offsets from lane 0's (poisson_phase1): a slot that straddles
    // two acquired planes must not see them 2^32 voxels apart
    // ... and a work item carries its output position in 32 bits
    const bool use_queue = queue_mode.share != 0 && (long long)(index_inc - 1) * plane < (1ll << 31) && plane * ((dim[2] - 1) / inc + 1) < (1ll << 32) &&
                           plane < (1ll << 32);
    const long long nzo = (dim[2] - 1) / inc + 1;
    const long long total = plane * nzo;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int qshare = share_for(total, queue_mode.share);
    const bool vec = (plane % 4 == 0) && (index_offset % 4 == 0) &&
                     (nviews > 0 ? vec_all : ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0));
    if (vec) {
        long long want = (total / 4 + 255) / 256;
        int blocks = (int)(want < 1 ? 1 : (want > 256 * 64 ? 256 * 64 : want));
#define MVSIM_LAUNCH_EX4(A, N)                                                                               \
    hipLaunchKernelGGL((k_extract4<A, N>), dim3(blocks, gy), dim3(256), 0, s, in, out, plane / 4, nzo, inc, scal, \
                       min_value, mul, k0, k1, stream, (unsigned long long)index_offset, vt)
        if (noise && (queue_ws || nviews > 0) && use_queue) {
            int qblocks;
            unsigned int segcap;
            unsigned int full_items;
            poisson_geometry(total, 16, &qblocks, &full_items);
            poisson_geometry(total, qshare, &qblocks, &segcap);
            unsigned int* qcount = reinterpret_cast<unsigned int*>(queue_ws);
            PItem* queue = queue_ws ? reinterpret_cast<PItem*>(reinterpret_cast<char*>(queue_ws) + QCOUNT_BYTES) : nullptr;
#define MVSIM_LAUNCH_N2(A, C)                                                                                                     \
    hipLaunchKernelGGL((k_extract4_noise2<A, C>), dim3(qblocks, gy), dim3(256), 0, s, in, out, plane / 4, nzo, inc, index_inc, scal, \
                       min_value, mul, k0, k1, stream, (unsigned long long)index_offset, queue, qcount, segcap, vt)
            const bool checked = qshare < 16;                 // segments that can fill up: the appends look before they write
            if (adjust && checked) MVSIM_LAUNCH_N2(true, true);
            else if (adjust) MVSIM_LAUNCH_N2(true, false);
            else if (checked) MVSIM_LAUNCH_N2(false, true);
            else MVSIM_LAUNCH_N2(false, false);
#undef MVSIM_LAUNCH_N2
            const ResolveJob rjob{out, queue, qcount, segcap, mul, k0, k1, stream, (unsigned int)plane, (unsigned int)index_inc,
                                  (unsigned long long)index_offset, qshare >= 16 ? 0 : 1, total / 4, 0};
            hipLaunchKernelGGL(k_poisson_resolve, dim3(qblocks, gy), dim3(256), 0, s, rjob, vt);
            if (rjob.walk != 0)
                hipLaunchKernelGGL(k_poisson_refused, dim3(qblocks < REFUSED_BLOCKS ? qblocks : REFUSED_BLOCKS, gy), dim3(256), 0, s, rjob, qblocks,
                                   full_items, queue_mode.hint, vt);
        }
        else if (adjust && noise) MVSIM_LAUNCH_EX4(true, true);
        else if (adjust) MVSIM_LAUNCH_EX4(true, false);
        else if (noise) MVSIM_LAUNCH_EX4(false, true);
        else MVSIM_LAUNCH_EX4(false, false);
#undef MVSIM_LAUNCH_EX4
        MVSIM_HIP(hipGetLastError());
        return MVSIM_OK;
    }
    if (noise && (queue_ws || nviews > 0) && use_queue) {
        // planes that are no multiple of four voxels / unaligned buffers: the same two launches, group by group (k_extract_noise2_any)
        int qblocks;
        unsigned int segcap;
        long long spp;
        unsigned int full_items;
        poisson_geometry_any(plane, nzo, 16, &qblocks, &full_items, &spp);
        poisson_geometry_any(plane, nzo, qshare, &qblocks, &segcap, &spp);
        unsigned int* qcount = reinterpret_cast<unsigned int*>(queue_ws);
        PItem* queue = queue_ws ? reinterpret_cast<PItem*>(reinterpret_cast<char*>(queue_ws) + QCOUNT_BYTES) : nullptr;
#define MVSIM_LAUNCH_ANY(A, C)                                                                                                       \
    hipLaunchKernelGGL((k_extract_noise2_any<A, C>), dim3(qblocks, gy), dim3(256), 0, s, in, out, plane, nzo, inc, index_inc, scal, min_value, \
                       mul, k0, k1, stream, (unsigned long long)index_offset, queue, qcount, segcap, spp, vt)
        const bool checked = qshare < 16;
        if (adjust && checked) MVSIM_LAUNCH_ANY(true, true);
        else if (adjust) MVSIM_LAUNCH_ANY(true, false);
        else if (checked) MVSIM_LAUNCH_ANY(false, true);
        else MVSIM_LAUNCH_ANY(false, false);
#undef MVSIM_LAUNCH_ANY
        const ResolveJob rjob{out, queue, qcount, segcap, mul, k0, k1, stream, (unsigned int)plane, (unsigned int)index_inc,
                              (unsigned long long)index_offset, qshare >= 16 ? 0 : 2, spp * nzo, spp};
        hipLaunchKernelGGL(k_poisson_resolve, dim3(qblocks, gy), dim3(256), 0, s, rjob, vt);
        if (rjob.walk != 0)
            hipLaunchKernelGGL(k_poisson_refused, dim3(qblocks < REFUSED_BLOCKS ? qblocks : REFUSED_BLOCKS, gy), dim3(256), 0, s, rjob, qblocks,
                               full_items, queue_mode.hint, vt);
        MVSIM_HIP(hipGetLastError());
        return MVSIM_OK;
    }
    long long want = (total + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 256 * 32 ? 256 * 32 : want));
#define MVSIM_LAUNCH_EX(A, N)                                                                             \
    hipLaunchKernelGGL((k_extract<A, N>), dim3(blocks, gy), dim3(256), 0, s, in, out, plane, nzo, inc, index_inc, scal, \
                       min_value, mul, k0, k1, stream, (unsigned long long)index_offset, vt)
    if (adjust && noise) MVSIM_LAUNCH_EX(true, true);
    else if (adjust) MVSIM_LAUNCH_EX(true, false);
    else if (noise) MVSIM_LAUNCH_EX(false, true);
    else MVSIM_LAUNCH_EX(false, false);
#undef MVSIM_LAUNCH_EX
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

int launch_extract(hipStream_t s, const float* in, float* out, const int64_t dim[3], int inc, bool adjust,
                   const double* scal, float min_value, bool noise, double mul, uint64_t seed,
                   uint32_t stream, uint64_t index_offset, void* queue_ws, QueueMode queue_mode, int index_inc)
{
    return launch_extract_impl(s, in, out, dim, inc, adjust, scal, min_value, noise, mul, seed, stream, index_offset, queue_ws, queue_mode,
                               index_inc, 0, nullptr, false);
}

// the queue region of one view inside a workspace of poisson_queue_bytes(n_out) bytes: [counts][segments]
void poisson_queue_split(void* queue_ws, void** queue_items, unsigned int** qcount)
{
    *qcount = reinterpret_cast<unsigned int*>(queue_ws);
    *queue_items = reinterpret_cast<char*>(queue_ws) + QCOUNT_BYTES;
}

int launch_extract_views(hipStream_t s, const int64_t dim[3], int inc, bool adjust, float min_value, bool noise, double mul,
                         QueueMode queue_mode, int index_inc, int nviews, const ExtractView* vt_dev, bool vec_all)
{
    return launch_extract_impl(s, nullptr, nullptr, dim, inc, adjust, nullptr, min_value, noise, mul, 0, 0, 0, nullptr, queue_mode, index_inc,
                               nviews, vt_dev, vec_all);
}

// ------------------------------------------------------------------------------------------------
// Acquisitions cross PCIe as 16-bit counts (host-buffer views): Tools.poissonProcess stores raw Poisson COUNTS as floats
// (Tools.java:84), so a view's acquisition is integer-valued and -- at the reference's SNRs -- far below 65 536.  out16[i] = (uint16) in[i];
// *flag is raised when any value does not survive the round trip (a count beyond 65 535, a non-integer because the view was
// simulated without noise, a NaN): the host then fetches the float32 buffer instead.  Exact: counts are integers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_u16(const float* __restrict__ in, unsigned short* __restrict__ out16, long long n,
                                                  unsigned int* __restrict__ flag)
{
    const long long nthreads = (long long)gridDim.x * 256;
    const long long n8 = n >> 3;
    bool bad = false;
    const float4* __restrict__ in4 = reinterpret_cast<const float4*>(in);
    uint4* __restrict__ out8 = reinterpret_cast<uint4*>(out16);
    auto cvt = [&](float v) -> unsigned int {
        const unsigned int u = (unsigned int)fminf(fmaxf(v, 0.f), 65535.f);
        bad |= __float_as_uint((float)u) != __float_as_uint(v);      // bit patterns: -0.0f does not pass for +0.0f
        return u;
    };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += nthreads) {
        const float4 a = in4[2 * i], b = in4[2 * i + 1];
        uint4 o;
        o.x = cvt(a.x) | (cvt(a.y) << 16); o.y = cvt(a.z) | (cvt(a.w) << 16);
        o.z = cvt(b.x) | (cvt(b.y) << 16); o.w = cvt(b.z) | (cvt(b.w) << 16);
        out8[i] = o;
    }
    for (long long i = (n8 << 3) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += nthreads) out16[i] = (unsigned short)cvt(in[i]);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// in: n floats (16-byte aligned), out16: n uint16 (16-byte aligned); *flag must be zero before the launch
int launch_pack_u16(hipStream_t s, const float* in, unsigned short* out16, int64_t n, unsigned int* flag)
{
    long long want = ((n >> 3) + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    hipLaunchKernelGGL(k_pack_u16, dim3(blocks), dim3(256), 0, s, in, out16, (long long)n, flag);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// makeIsotropic (SimulateMultiViewDataset.java:144-171): z = (float)l / (float)inc (Q4), x and y
// integral => the 8-tap interpolator degenerates to two planes (weights of the x+1 / y+1 taps
// are exactly 0); mirror-single extension along z.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long mirror1(long long i, long long n)
{
    if (n == 1) return 0;
    const long long p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// One output plane per blockIdx.y: the plane's z geometry (the float division of Q4, the two source planes, their weights) is the
// same for every voxel of it -- computed once per thread from the block index instead of per voxel with a 64-bit division and two
// 64-bit modulos (0.51 -> 0.2 ms at 512^3); VEC: 16-byte rows (plane % 4 == 0, aligned buffers).  Same arithmetic per voxel.
template <bool VEC>
__global__ __launch_bounds__(256) void k_make_isotropic(const float* __restrict__ in, float* __restrict__ out,
                                                        long long plane, long long nz, long long onz, int inc)
{
    const long long z = blockIdx.y;
    const double pz = (double)((float)z / (float)inc);
    const double fz = floor(pz);
    const double w2 = pz - fz, w2n = 1.0 - w2;
    const long long z0 = mirror1((long long)fz, nz), z1 = mirror1((long long)fz + 1, nz);
    const double wa = 1.0 * 1.0 * w2n, wb = 1.0 * 1.0 * w2;
    // tap order 000 ... 001: only 000 (w = 1*1*w2n) and 001 (w = 1*1*w2) carry weight
    const long long nthreads = (long long)gridDim.x * 256;
    if (VEC) {
        const float4* __restrict__ a4 = reinterpret_cast<const float4*>(in + z0 * plane);
        const float4* __restrict__ b4 = reinterpret_cast<const float4*>(in + z1 * plane);
        float4* __restrict__ o4 = reinterpret_cast<float4*>(out + z * plane);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < plane / 4; i += nthreads) {
            const float4 a = a4[i], b = b4[i];
            float4 r;
            r.x = (float)((double)a.x * wa); r.x += (float)((double)b.x * wb);
            r.y = (float)((double)a.y * wa); r.y += (float)((double)b.y * wb);
            r.z = (float)((double)a.z * wa); r.z += (float)((double)b.z * wb);
            r.w = (float)((double)a.w * wa); r.w += (float)((double)b.w * wb);
            o4[i] = r;
        }
    } else {
        const float* __restrict__ a = in + z0 * plane;
        const float* __restrict__ b = in + z1 * plane;
        float* __restrict__ o = out + z * plane;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < plane; i += nthreads) {
            float acc = (float)((double)a[i] * wa);
            acc += (float)((double)b[i] * wb);
            o[i] = acc;
        }
    }
}

int launch_make_isotropic(hipStream_t s, const float* in, float* out, const int64_t dim[3], int inc)
{
    const long long plane = (long long)dim[0] * dim[1];
    const long long onz = (dim[2] - 1) * inc + 1;
    if (onz > 65535) { set_error("makeIsotropic: %lld output planes exceed one launch", onz); return MVSIM_EINVAL; }
    const bool vec = plane % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const long long per = vec ? plane / 4 : plane;
    long long want = (per + 255) / 256;
    const unsigned bx = (unsigned)(want < 1 ? 1 : (want > 64 ? 64 : want));
    if (vec) hipLaunchKernelGGL(k_make_isotropic<true>, dim3(bx, (unsigned)onz), dim3(256), 0, s, in, out, plane, (long long)dim[2], onz, inc);
    else hipLaunchKernelGGL(k_make_isotropic<false>, dim3(bx, (unsigned)onz), dim3(256), 0, s, in, out, plane, (long long)dim[2], onz, inc);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// computeWeightImage (SimulateMultiViewDataset.java:280-316): cosine ramp along y.
__global__ __launch_bounds__(256) void k_weight_image(float* __restrict__ out, int nx, int ny, long long total)
{
    const long long nthreads = (long long)gridDim.x * 256;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += nthreads) {
        const int y = (int)((o / nx) % ny);
        const int l = ny - y - 1;
        float value;
        if (l < ny / 2) value = 1.0f;
        else if (l > ny / 2 + 40) value = 0.0f;
        else {
            const double pos = ((double)(l - ny / 2) / 40.0) * 3.141592653589793;
            value = (float)((cos(pos) + 1.0) / 2.0);
        }
        out[o] = value;
    }
}

int launch_weight_image(hipStream_t s, float* out, const int64_t dim[3])
{
    const long long total = (long long)dim[0] * dim[1] * dim[2];
    long long want = (total + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    hipLaunchKernelGGL(k_weight_image, dim3(blocks), dim3(256), 0, s, out, (int)dim[0], (int)dim[1], total);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Cross-view weight normalisation (SimulateMultiViewDataset.java:615-640): per voxel, float sum over the
// views in view order; zero sum -> all zero, else w <- min(1, osem * (w / sum)).
// ------------------------------------------------------------------------------------------------
struct ViewPtrs {
    float* p[MVSIM_MAX_VIEWS];
};

template <bool HAVE_SUM, bool WRITE_SUM_ONLY>
__global__ __launch_bounds__(256) void k_weights(ViewPtrs vp, int nv, long long n, const float* __restrict__ sum_in,
                                                 float* __restrict__ sum_out, float osem)
{
    const long long nthreads = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += nthreads) {
        float sum;
        if (HAVE_SUM) {
            sum = sum_in[i];
        } else {
            sum = 0.0f;
            for (int v = 0; v < nv; ++v) sum += vp.p[v][i];
        }
        if (WRITE_SUM_ONLY) { sum_out[i] = sum; continue; }
        for (int v = 0; v < nv; ++v) {
            float w = 0.0f;
            if (sum != 0.0f) w = fminf(1.0f, osem * (vp.p[v][i] / sum));
            vp.p[v][i] = w;
        }
    }
}

int launch_weights(hipStream_t s, float* const* views, int nv, int64_t n, const float* sum_in, float* sum_out,
                   float osem, bool sum_only)
{
    ViewPtrs vp;
    for (int v = 0; v < MVSIM_MAX_VIEWS; ++v) vp.p[v] = v < nv ? views[v] : nullptr;
    long long want = (n + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    if (sum_only)
        hipLaunchKernelGGL((k_weights<false, true>), dim3(blocks), dim3(256), 0, s, vp, nv, (long long)n, sum_in, sum_out, osem);
    else if (sum_in)
        hipLaunchKernelGGL((k_weights<true, false>), dim3(blocks), dim3(256), 0, s, vp, nv, (long long)n, sum_in, sum_out, osem);
    else
        hipLaunchKernelGGL((k_weights<false, false>), dim3(blocks), dim3(256), 0, s, vp, nv, (long long)n, sum_in, sum_out, osem);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

}  // namespace mvsim
