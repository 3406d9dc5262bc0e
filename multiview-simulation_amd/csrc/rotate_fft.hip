// Rotate (about x) + attenuate + the x transform of the convolution as ONE kernel: the attenuated volume never makes its
// round trip through HBM (SimulateMultiViewDataset.java:570-580 -- rotateAroundAxis, attenuate3d, and the first pass of
// convolve's FFT).  The rotate+attenuate kernel (kernels.hip: k_rotate_attenuate_axis0_lds) already holds complete x rows
// of the attenuated plane, one value per lane, while it walks y; pass A of the hand-written convolution
// (fft_kernels.hip: k_fft_x_r2c) transforms exactly those rows.  Here a block -- all the x columns of one plane -- walks
// y in batches of U = 8 rows: every lane blends, attenuates (same arithmetic, same order: `att`, and `rot`, stay
// bit-identical and are still written when the caller asks for them) and drops its value into an LDS row with the
// mirror halo of the padded row; after one barrier each wave owns a row end to end -- packed real FFT of M = Px/2
// points, half-spectrum post-processing, 16-byte stores into the spectrum buffer -- while the row buffers are double
// buffered, so that the faster waves already walk the next batch.  Saves 8 N bytes of HBM traffic per view and a launch;
// the spectrum is bit-identical to pass A's (same plan, same twiddles, same post-processing on the same values).
#include "common.h"
#include "fft_dev.h"

#include <type_traits>

namespace mvsim {
namespace fft {

struct __attribute__((aligned(16))) RowGeoF {
    double    w00, w10, w11, w01;
    long long off00;            // byte offset of source row (sy, sz), x = 0
    int       kind;             // 0 none, 1 all four taps inside, 2 partial: bits 8..11 = inside(00, 10, 11, 01)
    int       pad;
};
// rows per LDS geometry table: 512 (24 KB) for blocks of up to 8 waves, 128 for the 16-wave blocks of rows longer than 512
// voxels, whose double-buffered 16-row transform rounds take 144 KB of the CU's 160
constexpr int geo_chunk_f(int G) { return G == 1 ? 512 : 128; }
constexpr int UF = 8;
// rows of the NEXT batch requested before this batch's transforms (their latency would hide behind the FFTs).  Measured at
// 512^3: 0 -> 0.43 ms, 2 -> 0.46 ms, 4 -> 0.46 ms (128 VGPRs, 4 spilled), also with the LDS-only round barrier that lets loads
// stay in flight across it (0.40 against 0.42 ms): the kernel is not waiting for these loads -- no prefetch in the product build.
#ifndef MVSIM_ROTFFT_PREFETCH
#define MVSIM_ROTFFT_PREFETCH 0
#endif

// G: batches of U rows per transform round.  One wave transforms one row, so a block of up to 8 waves transforms after every
// batch (G = 1); a 16-wave block (rows of 513 .. 1024 voxels) collects two batches first (G = 2), or half its waves would
// sit out every transform.
template <class PLAN, bool WRITE_OUT, int G>
__global__ __launch_bounds__(1024) void k_rotate_attenuate_fftx(RotFftArgs p)
{
    constexpr int M = PLAN::len, LP = M + 1, U = UF, GEO_CHUNK_F = geo_chunk_f(G);
    extern __shared__ __align__(16) float2 lds[];
    float2* rowbuf = lds;                                         // [2][G * U][LP]
    float2* tw = lds + 2 * G * U * LP;                            // [M]
    float2* twx_l = tw + M + (M & 1);                             // [M + 1] (+1): the post-processing twiddles, read by every row of the plane
    RowGeoF* geo = reinterpret_cast<RowGeoF*>(twx_l + (M + 1) + ((M + 1) & 1));  // 16-byte aligned: every part above is a multiple of 2 float2
    int* bclass = reinterpret_cast<int*>(geo + GEO_CHUNK_F);
    // [2][G][16 waves]: bit u set = the wave's 64 columns of row u of that batch hold a non-zero value (see flush_round)
    unsigned int* wmask = reinterpret_cast<unsigned int*>(bclass + GEO_CHUNK_F / U);
    const int nx = p.nx, ny = p.ny, nz = p.nz, steps = p.steps;
    const int x = threadIdx.x;
    const int lane = x & 63, wave = __builtin_amdgcn_readfirstlane(x >> 6), nwaves = (int)blockDim.x >> 6;
    // Plane order.  Consecutive block ids go to different XCDs (round-robin dispatch, for speed only) and planes z, z + 1 read
    // the same source rows, so every XCD gets contiguous runs of planes -- TWO runs half a volume apart: the work of a plane
    // follows its content (empty rows skip the fp64 blends and the transforms), all blocks are resident at once (two per
    // CU), and a specimen in the middle of the volume would otherwise put all the dense planes on two XCDs
    // (measured 0.40 -> 0.36 ms at 512^3 on the sphere phantom).
    const int slab = (gridDim.x + 7) / 8;
    const int hs = slab / 2, jb = (int)(blockIdx.x / 8u), kb = (int)(blockIdx.x % 8u);
    // zl: the plane's index in this launch (rows of `dst`, planes of rot_out / att_out, plane_nz); z: which plane of the rotated
    // volume it is -- a z slab of a tiled view (mvsim_view_slab_*: planes z_first .. z_first + nzl - 1) computes its own planes only
    const int zl = (hs > 0 && slab == 2 * hs) ? ((jb < hs) ? kb * hs + jb : (int)gridDim.x / 2 + kb * hs + (jb - hs))
                                              : kb * slab + jb;
    if (zl >= p.nzl) return;                                      // whole block: uniform
    const int z = p.z_first + zl;
    const bool active = x < nx;
    const long long row = (long long)nx;
    const long long plane = row * ny;
    const long long out_plane = plane * zl;
    const double l2 = (double)z;
    const char* __restrict__ in_b = reinterpret_cast<const char*>(p.in);
    const long long row_b = row * 4, plane_b = plane * 4;
    const unsigned xoff = (unsigned)(active ? x : nx - 1) * 4u;
    const Affine& a = p.a;
    const double delta = p.delta;
    const int Px = 2 * M;
    // where this lane's value goes in the padded row: position x, its mirror image right of the row, its mirror image at the
    // end of the padded row (one reflection each: kx <= nx is guaranteed by the launcher); the zero gap between them
    const int pos_r = (active && x >= nx - 1 - p.halo_r && x <= nx - 2) ? 2 * nx - 2 - x : -1;
    const int pos_l = (active && x >= 1 && x <= p.halo_l) ? Px - x : -1;
    const int gap_lo = nx + p.halo_r, gap_len = Px - p.halo_l - gap_lo;
    for (int i = x; i < M; i += (int)blockDim.x) tw[i] = p.twg[i];
    for (int i = x; i <= M; i += (int)blockDim.x) twx_l[i] = p.twx[i];

    // one row, owned by the calling wave: transform, post-process to the half spectrum, store (k_fft_x_r2c's arithmetic)
    auto transform_store = [&](float2* wbuf, int y) {
        PLAN::template run<1>(wbuf, tw, lane);
        float2* __restrict__ drow = p.dst + ((long long)zl * p.py + y) * p.hxp;
        constexpr int HQ = M / 4;
#pragma unroll 1
        for (int q = lane; q < HQ; q += 64) {
            float2 lo[2], hi[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * q + h;
                const float2 zk = wbuf[k];
                const float2 zm = cconj(wbuf[k == 0 ? 0 : M - k]);
                const float2 sm = cadd(zk, zm), d = csub(zk, zm);
                const float2 wd = cmul(twx_l[k], d);
                const float2 t = make_float2(wd.y, -wd.x);
                lo[h] = cadd(sm, t);
                hi[h] = cconj(csub(sm, t));
            }
            *reinterpret_cast<float4*>(drow + 2 * q) = make_float4(lo[0].x, lo[0].y, lo[1].x, lo[1].y);
            *reinterpret_cast<Pair16*>(drow + M - 2 * q - 1) = Pair16{hi[1].x, hi[1].y, hi[0].x, hi[0].y};
        }
        const int nmid = M - 4 * HQ + 1;
        const int ntail = nmid + (p.hxp - M - 1);
        for (int u = lane; u < ntail; u += 64) {
            if (u < nmid) {
                const int k = 2 * HQ + u;
                const float2 zk = wbuf[k == M ? 0 : k];
                const float2 zm = cconj(wbuf[k == 0 ? 0 : M - k]);
                const float2 sm = cadd(zk, zm), d = csub(zk, zm);
                const float2 wd = cmul(twx_l[k], d);
                drow[k] = cadd(sm, make_float2(wd.y, -wd.x));
            } else {
                drow[M + 1 + (u - nmid)] = make_float2(0.f, 0.f);
            }
        }
    };
    auto zero_row = [&](int y) {                                  // the spectrum of a zero row
        float4* __restrict__ drow = reinterpret_cast<float4*>(p.dst + ((long long)zl * p.py + y) * p.hxp);
        for (int i = lane; i < p.hxp / 2; i += 64) drow[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    };

    double n = 1.0;
    unsigned int plane_any = 0u;                                  // this wave's columns: any non-zero attenuated value in the plane so far
    int buf = 0;
    int fill = 0;                                                 // batches waiting in the current round buffer (block-uniform)
    int ysub[G], nsub[G];                                         // first row (y) and row count of each of them
    // one barrier per round: the rows are complete; every wave transforms its share; the other buffer takes the next round
    // (a wave reaches the NEXT round's barrier only after its transforms of this one, so two buffers suffice)
    auto flush_round = [&]() {
        // LDS-only barrier: __syncthreads() is a workgroup-scope fence as well and drains vmcnt, i.e. it would wait for every
        // global store of the previous round's spectra (and for the next batch's rows, were they requested ahead) before the
        // waves may meet.  What the round hands over travels through LDS alone: the wave's own ds_writes are complete
        // (lgkmcnt(0)), then the barrier; global loads and stores stay in flight across it.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int j = wave; j < fill * U; j += nwaves) {
            const int sub = j / U, u = j - sub * U;
            if (u < nsub[sub]) {
                // a row without a single non-zero voxel (everything outside the specimen: well over half the rows of a sphere
                // phantom) has the zero spectrum: no transform, sixteen-byte zero stores.  Exact, not an approximation.
                unsigned int any = 0u;
                for (int w = 0; w < nwaves; ++w) any |= wmask[(buf * G + sub) * 16 + w];
#ifdef MVSIM_EXP_ROTFFT_NOFFT
                any = 0u;
#endif
                if ((__builtin_amdgcn_readfirstlane(any) >> u) & 1u) transform_store(rowbuf + (((size_t)buf * G + sub) * U + u) * LP, ysub[sub] - u);
                else zero_row(ysub[sub] - u);
            }
        }
        buf ^= 1;
        fill = 0;
    };
    float pv00[U], pv10[U], pv11[U], pv01[U];                     // source rows of a class-1 batch (see issue_loads)
    bool have_pref = false;                                       // block-uniform
    // straight-line loads of a batch whose rows all have their four taps inside the volume: scalar row bases from the
    // table (v_readfirstlane) plus the lane's constant byte offset; lanes beyond nx read the last column
    constexpr int UPRE = MVSIM_ROTFFT_PREFETCH;                  // rows of the next batch requested ahead (registers are the limit)
    auto issue_loads = [&](int r0n, auto lo_c, auto hi_c) {
#pragma unroll
        for (int u = decltype(lo_c)::value; u < decltype(hi_c)::value; ++u) {
            const long long off = geo[r0n + u].off00;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)off);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)off >> 32));
            const char* __restrict__ p00 = in_b + (long long)(((unsigned long long)hi << 32) | lo);
            pv00[u] = *reinterpret_cast<const float*>(p00 + xoff);
            pv10[u] = *reinterpret_cast<const float*>(p00 + row_b + xoff);
            pv11[u] = *reinterpret_cast<const float*>(p00 + row_b + plane_b + xoff);
            pv01[u] = *reinterpret_cast<const float*>(p00 + plane_b + xoff);
        }
    };
    for (int c0 = 0; c0 < steps; c0 += GEO_CHUNK_F) {
        const int cnt = min(GEO_CHUNK_F, steps - c0);
        __syncthreads();                                          // the previous chunk's readers are done (and tw is staged)
        for (int r = x; r < cnt; r += (int)blockDim.x) {
            const int yy = ny - 1 - (c0 + r);
            const double l1 = (double)yy;
            const double py = 0.0 * a.m[4] + l1 * a.m[5] + l2 * a.m[6] + a.m[7];
            const double pz = 0.0 * a.m[8] + l1 * a.m[9] + l2 * a.m[10] + a.m[11];
            const double fy = floor(py), fz = floor(pz);
            RowGeoF g;
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
                g.off00 = row * (sy + (long long)ny * sz) * 4;
                const int m = ((ya && za) ? 1 : 0) | ((yb && za) ? 2 : 0) | ((yb && zb) ? 4 : 0) | ((ya && zb) ? 8 : 0);
                g.kind = m == 15 ? 1 : (m == 0 ? 0 : (2 | (m << 8)));
            }
            geo[r] = g;
        }
        __syncthreads();
        for (int bq = x; bq < (cnt + U - 1) / U; bq += (int)blockDim.x) {
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
        have_pref = false;
        for (int r0 = 0; r0 < cnt; r0 += U) {
            const int cls = __builtin_amdgcn_readfirstlane(bclass[r0 / U]);
            const int y0 = ny - 1 - (c0 + r0);                    // rows y0, y0 - 1, ..., y0 - U + 1
            const int nrows = min(U, cnt - r0);
            if (cls == 0) {
                // four zero taps per row: rot = +0, the attenuation state is unchanged, att = +0: zero rows, zero spectra
                if (WRITE_OUT && active) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const long long o = out_plane + (long long)(y0 - u) * row;
                        if (p.rot_out) p.rot_out[o + x] = 0.f;
                        if (p.att_out) p.att_out[o + x] = 0.f;
                    }
                }
                for (int j = wave; j < U; j += nwaves) zero_row(y0 - j);
                continue;
            }
            float val[U];
            if (cls == 1) {
                // the batch's 4 U source rows: already in flight when the previous batch asked for them before its transforms
                if (!have_pref) issue_loads(r0, std::integral_constant<int, 0>{}, std::integral_constant<int, UPRE>{});
                issue_loads(r0, std::integral_constant<int, UPRE>{}, std::integral_constant<int, U>{});
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // four zero taps in EVERY lane of the wave (empty space): rot = +0, the attenuation state does not move
                    // (n - 0 delta n = n), att = +0 -- what the arithmetic below gives, without the fp64 blends.  Wave-uniform.
#ifdef MVSIM_EXP_ROTFFT_NOBLEND
                    const unsigned bits = 0u; val[u] = pv00[u] + pv10[u] + pv11[u] + pv01[u];
#else
                    const unsigned bits = __float_as_uint(pv00[u]) | __float_as_uint(pv10[u]) | __float_as_uint(pv11[u]) | __float_as_uint(pv01[u]);
#endif
                    if (__builtin_amdgcn_ballot_w64((bits << 1) != 0u) == 0ull) {
                        val[u] = 0.f;
                        if (WRITE_OUT && active) {
                            const long long ob = (out_plane + (long long)(y0 - u) * row) * 4;
                            if (p.rot_out) *reinterpret_cast<float*>(reinterpret_cast<char*>(p.rot_out) + ob + xoff) = 0.f;
                            if (p.att_out) *reinterpret_cast<float*>(reinterpret_cast<char*>(p.att_out) + ob + xoff) = 0.f;
                        }
                        continue;
                    }
                    const RowGeoF* g = &geo[r0 + u];
                    float r = (float)((double)pv00[u] * g->w00);
                    r += (float)((double)pv10[u] * g->w10);
                    r += (float)((double)pv11[u] * g->w11);
                    r += (float)((double)pv01[u] * g->w01);
                    const double d = (double)r;
                    n = fmax(n - d * delta * n, 0.0);
                    val[u] = (float)(d * n);
                    if (WRITE_OUT && active) {
                        const long long ob = (out_plane + (long long)(y0 - u) * row) * 4;
                        if (p.rot_out) *reinterpret_cast<float*>(reinterpret_cast<char*>(p.rot_out) + ob + xoff) = r;
                        if (p.att_out) *reinterpret_cast<float*>(reinterpret_cast<char*>(p.att_out) + ob + xoff) = val[u];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    val[u] = 0.f;
                    if (u < nrows) {
                        const RowGeoF* g = &geo[r0 + u];
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
                            val[u] = (float)(d * n);
                        }
                        if (WRITE_OUT && active) {
                            if (p.rot_out) p.rot_out[o + x] = r;
                            if (p.att_out) p.att_out[o + x] = val[u];
                        }
                    }
                }
            }
            // the batch's rows into LDS as the padded real rows pass A would read: position x, the two mirror images, the gap
            {
                unsigned int rowmask = 0u;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (__builtin_amdgcn_ballot_w64(active && (__float_as_uint(val[u]) << 1) != 0u) != 0ull) rowmask |= 1u << u;
                if (lane == 0) wmask[(buf * G + fill) * 16 + wave] = rowmask;
                plane_any |= rowmask;
            }
            float* __restrict__ rb = reinterpret_cast<float*>(rowbuf + ((size_t)buf * G + fill) * U * LP);
            ysub[fill] = y0;
            nsub[fill] = nrows;
            fill += 1;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float* __restrict__ rr = rb + (size_t)u * 2 * LP;
                if (active) rr[x] = val[u];
                if (pos_r >= 0) rr[pos_r] = val[u];
                if (pos_l >= 0) rr[pos_l] = val[u];
                for (int gpos = x; gpos < gap_len; gpos += (int)blockDim.x) rr[gap_lo + gpos] = 0.f;
            }
            // the next batch's rows are requested BEFORE this batch's transforms: their latency hides behind the FFTs
            have_pref = r0 + U < cnt && __builtin_amdgcn_readfirstlane(bclass[r0 / U + 1]) == 1;
            if (have_pref) issue_loads(r0 + U, std::integral_constant<int, 0>{}, std::integral_constant<int, UPRE>{});
            if (fill == G) flush_round();
        }
        if (fill > 0) flush_round();                              // before the geometry table (and with it nothing the round needs) moves on
    }
    // Planes that stay empty (a specimen in empty space: a third of the planes of the sphere phantom) need no convolution passes at
    // all: their spectrum is exactly zero.  The passes skip what these flags call empty (custom_fft_convolve_slab, ConvTail::plane_nz).
    // (every block writes its plane's flag, 0 or 1: nothing has to be cleared beforehand)
    if (p.plane_nz) {
        unsigned int* blk_any = wmask + 2 * G * 16;
        __syncthreads();                                          // (also: every round's readers of wmask are done)
        if (x == 0) *blk_any = 0u;
        __syncthreads();
        if (lane == 0 && plane_any != 0u) atomicOr(blk_any, 1u);
        __syncthreads();
        if (x == 0) p.plane_nz[zl] = (int)*blk_any;
    }
    // rows the reference never visits (Ny > Nx): the attenuated image stays zero there; rot still has its values
    for (int yy = ny - 1 - steps; yy >= 0; --yy) {
        if (WRITE_OUT && active) {
            if (p.att_out) p.att_out[out_plane + (long long)yy * row + x] = 0.f;
            if (p.rot_out) {
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
                    const float* __restrict__ pin = p.in + x;
                    const float a00 = (ya && za) ? pin[row * (sy + (long long)ny * sz)] : 0.f;
                    const float a10 = (yb && za) ? pin[row * (sy + 1 + (long long)ny * sz)] : 0.f;
                    const float a11 = (yb && zb) ? pin[row * (sy + 1 + (long long)ny * (sz + 1))] : 0.f;
                    const float a01 = (ya && zb) ? pin[row * (sy + (long long)ny * (sz + 1))] : 0.f;
                    o = (float)((double)a00 * q00); o += (float)((double)a10 * q10);
                    o += (float)((double)a11 * q11); o += (float)((double)a01 * q01);
                }
                p.rot_out[out_plane + (long long)yy * row + x] = o;
            }
        }
        if ((ny - 1 - steps - yy) % nwaves == wave) zero_row(yy);
    }
}

template <class PLAN>
static int launch_rot_fftx_t(mvsim_ctx* ctx, const RotFftArgs& a, bool write_out)
{
    constexpr int M = PLAN::len;
    const int waves = (a.nx + 63) / 64;
    const int G = waves > 8 ? 2 : 1;
    const size_t lds = (size_t)(2 * G * UF * (M + 1) + M + (M & 1) + (M + 1) + ((M + 1) & 1)) * sizeof(float2) + (size_t)geo_chunk_f(G) * sizeof(RowGeoF) +
                       (size_t)(geo_chunk_f(G) / UF) * sizeof(int) + (size_t)(2 * G * 16 + 4) * sizeof(unsigned int);
    if (lds > 160 * 1024) { set_error("fused rotate + x transform: %zu bytes of LDS", lds); return MVSIM_EINVAL; }
    dim3 grid((unsigned)((a.nzl + 7) / 8 * 8)), block((unsigned)(waves * 64));
#define MVSIM_RF(W_, G_)                                                                                            \
    do {                                                                                                            \
        MVSIM_TRY(ensure_lds_attr(ctx, reinterpret_cast<const void*>(k_rotate_attenuate_fftx<PLAN, W_, G_>), lds)); \
        hipLaunchKernelGGL((k_rotate_attenuate_fftx<PLAN, W_, G_>), grid, block, lds, ctx->stream, a);              \
    } while (0)
    if constexpr (M >= 280) {                                     // rows of more than 512 voxels: padded half length >= 280
        if (G == 2) {
            if (write_out) MVSIM_RF(true, 2); else MVSIM_RF(false, 2);
            MVSIM_HIP(hipGetLastError());
            return MVSIM_OK;
        }
    }
    if (G == 2) { set_error("fused rotate + x transform: no 16-wave instance for half length %d", M); return MVSIM_EINVAL; }
    if (write_out) MVSIM_RF(true, 1); else MVSIM_RF(false, 1);
#undef MVSIM_RF
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// half lengths the fused kernel is instantiated for (rows of up to 1024 voxels with PSFs of up to 64 taps); the plans --
// radices and with them the layout of the twiddle table -- are the size table's own
constexpr bool rot_fftx_len_ok(int len) { return len >= 72 && len <= 576; }

bool rot_fftx_has_plan(int M)
{
    switch (M) {
#define X(LL, ...) case LL: return rot_fftx_len_ok(LL);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    return false;
}

template <int LL, int... Rs>
static int launch_rot_fftx_pick(mvsim_ctx* ctx, const RotFftArgs& a, bool write_out)
{
    if constexpr (rot_fftx_len_ok(LL)) return launch_rot_fftx_t<Plan<LL, Rs...>>(ctx, a, write_out);
    set_error("fused rotate + x transform: half length %d is not instantiated", LL);
    return MVSIM_EINVAL;
}

int launch_rot_fftx(mvsim_ctx* ctx, int M, const RotFftArgs& a, bool write_out)
{
    switch (M) {
#define X(LL, ...) case LL: return launch_rot_fftx_pick<LL, __VA_ARGS__>(ctx, a, write_out);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("fused rotate + x transform: no plan for half length %d", M);
    return MVSIM_EINVAL;
}

}  // namespace fft
}  // namespace mvsim
