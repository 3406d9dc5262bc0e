// Hand-written LDS-tiled FFT convolution for gfx950 (the production path of
// SimulateMultiViewDataset.convolve, :253-264, for padded sizes of the form 2^a 3^b).
//
// A 3-D real convolution is five streaming passes over one half-spectrum buffer F
// (complex float, [Pz][Py][Hxp], Hxp = Px/2+1 rounded up to the tile width):
//   A  x: mirror-pad on the fly + real->complex FFT along x          read 4N      write C
//   B  y: complex FFT along y, in place                               read C       write C
//   C  z: FFT along z, multiply by the PSF spectrum, inverse FFT z    read C (+G)  write C
//   D  y: inverse FFT along y, in place                               read C       write C
//   E  x: complex->real FFT along x + crop + 1/P^3 + block sums       read C       write 4N
// Each pass stages a tile of NL lines in LDS (line = the FFT axis, contiguous in LDS; NL adjacent
// kx columns = 128 B contiguous in HBM).  Every wave then OWNS whole lines of the tile and runs the
// mixed-radix Stockham passes on them with register butterflies; because LDS operations of one wave
// execute in issue order, the passes need no workgroup barrier at all -- the waves of a block run
// decoupled and hide each other's LDS latency.  Block barriers remain only around the transposed
// staging (and around the spectrum product of pass C).  Twiddles come from a host-computed (double
// precision, rounded once) table staged in LDS.  Inverse transforms use conj(FFT(conj(.))).
#include "common.h"
#include "fft_dev.h"
#include "poisson_dev.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mvsim {
static int ensure_twiddles(mvsim_ctx* ctx, int L, int kind, const float2** out);   // (defined behind the kernels; launch_lines needs the split form's tables)
namespace fft {

// ---------------------------------------------------------------------------------- y / z pass kernel
// CONVZ: the z pass as FFT -> product -> inverse FFT on a spectrum WITHOUT z padding (the geometry of the direct z pass): the
// mirrored halo planes are read from their mirror images through the index map, and the PSF's z spectrum is never stored --
// every block transforms the Kz taps of its own 16 lines first and keeps the result in registers for the product.
enum Mode { FWD = 0, INV = 1, CONV = 2, CONVZ = 3 };

struct LinesArgs {
    const float2* src;      // element (line c, position n) of tile (bx, by): src[by*src_outer + n'*src_es + bx*NL + c]
    float2*       dst;
    const float2* spec;     // CONV: PSF spectrum, same tiling
    const float2* tw;
    long long     src_es, src_outer, dst_es, dst_outer, spec_es, spec_outer;
    DimMap        lmap;     // SPARSE: position n reads source position map_src(lmap, n) (or zero)
    int           gap_lo, gap_hi;   // positions n in [gap_lo, gap_hi) are structurally zero: not loaded (gap_hi <= gap_lo: none)
    int           outer_skip_lo, outer_skip_len;   // outer index by >= outer_skip_lo is shifted by outer_skip_len (skipped planes)
    int           store_limit;      // > 0: positions n >= store_limit are never read downstream and are not stored
    int           dst_tile_major;   // FWD: store tile (bx,by) as NL contiguous lines of L: dst[((by*gridDim.x+bx)*NL + c)*L + n]
    // z-blocked layout of the spectrum between passes B, C' and D (0: off): position n of a line sits at
    // (n >> ZBS) * blk + (n & (ZB - 1)) * es -- ZB rows of every plane stay together, so that the z pass, which walks the
    // planes of ONE row, finds them ZB * Hxp elements apart instead of a whole plane apart (see line_off)
    long long     src_blk, dst_blk;
    int           src_mirror;       // != 0 (non-SPARSE): position n reads source position map_src(lmap, n) -- the mirrored halo rows
                                    // of the padded image are the rows themselves, read twice instead of transformed twice
    // planes known to be empty (null: none): a block whose outer index names an empty plane has nothing to read and -- because every
    // reader of its output asks the same flags -- nothing to write.  Outer index k = plane k * nz_stride of the flag array
    const int*    nzflags;
    int           nz_stride;
    // blocked OUTER addressing (0: linear): outer index by sits at (by >> ZBS) * oblk + (by & (ZB - 1)) * outer -- the z-blocked
    // spectrum layout seen from a pass whose lines run along z (CONVZ)
    long long     src_oblk, dst_oblk;
    // CONVZ: the PSF's (x, y) spectrum, Kz planes in the z-blocked layout of the direct z pass (ZConvArgs::taps); pmap embeds tap
    // t at position (t - Kz / 2) mod L; adjustImage's sum from this pass's output as in k_zconv (null: not wanted)
    const float2* taps;
    long long     taps_es, taps_oblk, taps_outer;
    DimMap        pmap;
    const double2* wx;
    const double2* wy;
    double*       sum_partial;
    int           pair_tiles;       // 8-column tiles: the two tiles of a 128-byte line on one XCD (see k_fft_lines; exp bit 4 clears it)
};

#ifndef MVSIM_ZBS
#define MVSIM_ZBS 4
#endif
#ifndef MVSIM_ZINLINE_MIN_KZ
#define MVSIM_ZINLINE_MIN_KZ 48           // measured (profiles/r04_zpass_sweep.txt): see custom_fft_convolve_slab
#endif
constexpr int ZBS = MVSIM_ZBS, ZB = 1 << ZBS;        // rows per block of the z-blocked layout
constexpr int NZ_EXT = 96;                          // mirrored planes the flag bit string holds in front of plane 0 (>= Kz - 1 + tap padding)

#if defined(MVSIM_DEV_ATTRIBUTION) && defined(MVSIM_EXP_LINES_STAMPS)
// Attribution build only (tools/lines_timeline.py): shader-clock stamps of the phases of every block of the image's y passes --
// [mode FWD / INV][block][8]: start, loads requested, loads arrived, tile staged (barrier), this wave's transform done, barrier, stores
// requested.  The last launch of each mode stays in the buffer.
constexpr int STAMP_BLOCKS = 1 << 20;
__device__ unsigned long long g_line_stamps[2][STAMP_BLOCKS][8];
#define MVSIM_STAMP(k)                                                                                                       \
    do {                                                                                                                     \
        if (!SPARSE && (MODE == FWD || MODE == INV) && threadIdx.x == 0) {                                                   \
            const unsigned long long sb_ = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;                          \
            if (sb_ < (unsigned long long)STAMP_BLOCKS) g_line_stamps[MODE == INV][sb_][k] = __builtin_amdgcn_s_memtime();   \
        }                                                                                                                    \
    } while (0)
#else
#define MVSIM_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int mirror_index(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

__device__ __forceinline__ long long line_off(int n, long long es, long long blk)
{
    return blk ? (long long)(n >> ZBS) * blk + (long long)(n & (ZB - 1)) * es : (long long)n * es;
}

// second launch bound: as many blocks as the LDS lets a CU hold (two, or one for the long lines) must stay resident
// (w = blocks*T/256 waves per SIMD, rounded up) -- the register budget follows from that
template <int L> constexpr int lines_blocks_per_cu() { return 2 * Cfg<L>::LDS <= 160 * 1024 ? 2 : 1; }
template <class PLAN, int MODE, bool SPARSE>
__global__ __launch_bounds__(Cfg<PLAN::len>::T, (lines_blocks_per_cu<PLAN::len>() * Cfg<PLAN::len>::T + 255) / 256)
void k_fft_lines(LinesArgs p)
{
    constexpr int L = PLAN::len;
    using C = Cfg<L>;
    constexpr int NL = C::NL, LP = C::LP, T = C::T, LW = C::LW, NW = C::NW;
    constexpr int LPR = NL / 2;             // lanes per position: one float4 = two adjacent columns
    constexpr int ROWS = T / LPR;
    constexpr int NIT = (L + ROWS - 1) / ROWS;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw = lds + NL * LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c2 = (tid % LPR) * 2;
    const int r0 = tid / LPR;
    float2* wbuf = buf + wave * LW * LP;    // the lines this wave transforms
    // all global loads of the tile are issued before anything waits
    // Which tile this block takes.  Tiles of 8 columns (the long lines: L > 576) move 64-BYTE row segments, i.e. two neighbouring tiles
    // share every 128-byte line they read and write.  In plain grid order those two blocks have consecutive ids, which the dispatcher
    // hands to DIFFERENT XCDs: both L2s fetch the whole line for their half (measured at 1024^3, profiles/r05_pmc_hbm_traffic_1024.txt
    // and tools/microbench/fetch_calib.hip: passes B and D read 2.0-2.1 x the bytes they need, TCC_EA0_RDREQ_128B = 2 per line, every
    // half-line write misses).  So the pair goes to ONE XCD, one directly behind the other in that XCD's dispatch order (ids b and
    // b + 8): the second half hits the line the first one brought in, and the two half-line writes meet in one L2 before they leave.
    int tile_x = (int)blockIdx.x, tile_o = (int)blockIdx.y;
    if constexpr (NL * sizeof(float2) < 128) {
        const unsigned nt = gridDim.x, total = nt * gridDim.y, b = blockIdx.y * nt + blockIdx.x;
        if ((nt & 1u) == 0 && b < (total & ~15u) && p.pair_tiles) {
            const unsigned xcd = b & 7u, slot = b >> 3;
            const unsigned lin = ((((slot >> 1) << 3) + xcd) << 1) | (slot & 1u);
            tile_x = (int)(lin % nt); tile_o = (int)(lin / nt);
        }
    }
    MVSIM_STAMP(0);
    const int by = tile_o >= p.outer_skip_lo ? tile_o + p.outer_skip_len : tile_o;
    if (p.nzflags) {
        // block-uniform (one scalar load): nothing of an empty plane is read, transformed or stored -- its readers skip it on the
        // same flags.  Pass B: flags of the input planes; pass D: the dilated flags (a plane of the z pass's output is empty iff
        // every plane its Kz taps reach is), outer index k = plane k * nz_stride
        if (p.nzflags[by * p.nz_stride] == 0) return;
    }
    auto outer_off = [&](long long outer, long long oblk) {
        return oblk ? (long long)(by >> ZBS) * oblk + (long long)(by & (ZB - 1)) * outer : (long long)by * outer;
    };
    const float2* sbase = p.src + outer_off(p.src_outer, p.src_oblk) + (long long)tile_x * NL + c2;
    float4 vp[MODE == CONVZ ? NIT : 1];
    if (MODE == CONVZ) {
        const float2* tbase = p.taps + outer_off(p.taps_outer, p.taps_oblk) + (long long)tile_x * NL + c2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = r0 + it * ROWS;
            vp[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((L % ROWS == 0) || n < L) {
                const int t = map_src(p.pmap, n);
                if (t >= 0) vp[it] = *reinterpret_cast<const float4*>(tbase + (long long)t * p.taps_es);
            }
        }
    }
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((L % ROWS == 0) || n < L) {
            if (SPARSE) {
                const int sn = map_src(p.lmap, n);
                if (sn >= 0) v[it] = *reinterpret_cast<const float4*>(sbase + sn * p.src_es);
            } else if (n < p.gap_lo || n >= p.gap_hi) {
                // (the compiler waits for the tile's FIRST row before it requests the second -- the merge of loaded and zero registers behind
                // this branch costs it a register copy --; without the branch, all rows requested at once, passes B and D measured the same)
                const int sn = p.src_mirror ? map_src(p.lmap, n) : n;
                v[it] = *reinterpret_cast<const float4*>(sbase + line_off(sn, p.src_es, p.src_blk));
            }
        }
    }
    MVSIM_STAMP(1);
    for (int i = tid; i < L; i += T) tw[i] = p.tw[i];
    MVSIM_STAMP(2);                                             // (behind the wait for the twiddles, i.e. for every load before them)
    constexpr int R1 = PLAN::R1, IT1 = (LW * (L / R1) + 63) / 64;
    float2 ps[MODE == CONVZ ? IT1 : 1][MODE == CONVZ ? R1 : 1];
    if constexpr (MODE == CONVZ) {
        // the PSF's z lines of this tile: taps into the (zeroed) lines, transform, keep the spectrum as the product's operands
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = r0 + it * ROWS;
            if ((L % ROWS == 0) || n < L) {
                buf[c2 * LP + n] = make_float2(vp[it].x, vp[it].y);
                buf[(c2 + 1) * LP + n] = make_float2(vp[it].z, vp[it].w);
            }
        }
        __syncthreads();
        PLAN::template run<LW>(wbuf, tw, lane);
        PLAN::template capture_first_pass<LW, IT1>(wbuf, lane, ps);
        __syncthreads();                                  // every wave has read its PSF lines: the image tile may overwrite them
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        if ((L % ROWS == 0) || n < L) {
            float2 a = make_float2(v[it].x, v[it].y), b = make_float2(v[it].z, v[it].w);
            if (MODE == INV) { a = cconj(a); b = cconj(b); }
            buf[c2 * LP + n] = a;
            buf[(c2 + 1) * LP + n] = b;
        }
    }
    __syncthreads();
    MVSIM_STAMP(3);
    PLAN::template run<LW>(wbuf, tw, lane);
    MVSIM_STAMP(4);
    if (MODE == CONV) {
        // x PSF spectrum, conjugate, transform again (inverse = conj FFT conj).  The spectrum is stored tile-major
        // (each line contiguous), so the wave that owns a line streams its spectrum line straight into the first
        // radix pass of the second transform: no barrier, no extra trip through LDS.
        const float2* gl = p.spec + (((long long)tile_o * gridDim.x + tile_x) * NL + (long long)wave * LW) * L;
        PLAN::template run_op<LW>(wbuf, tw, lane, LoadMulConj{gl, L});
    }
    if constexpr (MODE == CONVZ) PLAN::template run_op<LW>(wbuf, tw, lane, LoadMulConjReg<IT1, R1>{ps});
    if (MODE == FWD && p.dst_tile_major) {
        // wave-private store: every line of the tile contiguous (consumed by LoadMulConj above)
        float2* gl = p.dst + (((long long)tile_o * gridDim.x + tile_x) * NL + (long long)wave * LW) * L;
#pragma unroll
        for (int j = 0; j < LW; ++j)
            for (int n2 = lane * 2; n2 < L; n2 += 128) {
                const float2 a = wbuf[j * LP + n2], b = wbuf[j * LP + n2 + 1];
                *reinterpret_cast<float4*>(gl + j * L + n2) = make_float4(a.x, a.y, b.x, b.y);
            }
        return;
    }
    __syncthreads();
    MVSIM_STAMP(5);
    float2* dbase = p.dst + outer_off(p.dst_outer, p.dst_oblk) + (long long)tile_x * NL + c2;
    const int nstore = p.store_limit > 0 ? p.store_limit : L;
    float2 sa = make_float2(0.f, 0.f), sb = make_float2(0.f, 0.f);       // CONVZ: this thread's share of its two lines' sums over z
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        if (((L % ROWS == 0) || n < L) && n < nstore) {
            float2 a = buf[c2 * LP + n], b = buf[(c2 + 1) * LP + n];
            if (MODE != FWD) { a = cconj(a); b = cconj(b); }
            *reinterpret_cast<float4*>(dbase + line_off(n, p.dst_es, p.dst_blk)) = make_float4(a.x, a.y, b.x, b.y);
            if (MODE == CONVZ) { sa = cadd(sa, a); sb = cadd(sb, b); }
        }
    }
    MVSIM_STAMP(6);
    if constexpr (MODE == CONVZ) {
        if (p.sum_partial) {
            // adjustImage's sum from the spectrum side, as k_zconv's epilogue: SUM_z of every line (fp32 over a thread's <= NIT rows,
            // double from there on), weighted with wx[kx] * wy[ky], one double per block
            __syncthreads();                              // everybody has read its outputs: the tile serves as scratch
            float2* red = buf;                            // [ROWS][NL]
            red[r0 * NL + c2] = sa;
            red[r0 * NL + c2 + 1] = sb;
            __syncthreads();
            double t = 0.0;
            if (tid < NL) {
                double sre = 0.0, sim = 0.0;
                for (int r = 0; r < ROWS; ++r) { const float2 q = red[r * NL + tid]; sre += (double)q.x; sim += (double)q.y; }
                const double2 a = p.wx[tile_x * NL + tid], b = p.wy[by];
                const double wr = a.x * b.x - a.y * b.y, wi = a.x * b.y + a.y * b.x;
                t = sre * wr - sim * wi;                  // Re( s * W )
            }
            if (wave == 0) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
                if (lane == 0) p.sum_partial[(long long)tile_o * gridDim.x + tile_x] = t;
            }
        }
    }
}

// The image's y passes on lines that leave a CU room for ONE tile (L >= 1280: 8 lines x (L + 1) x 8 B > 80 KB; configs[4]'s 2160-point lines),
// round 6 -- instantiated for L = 2048, 2160 and 2240 (split_half_ok).  With one block per CU nothing overlaps: the block clock stamps (profiles/r06_lines_timeline.txt) show a 2160-point block alive for
// 60 k ticks of which 32 k are its loads, 20 k its transform and the barrier behind it, 5 k its stores -- HBM idles while it transforms, the
// SIMDs idle while it loads (2.6 TB/s per pass against 4.4 at 1080 points, where two blocks interleave).  Here the FIRST radix-2 stage of a
// decimation-in-frequency transform runs in REGISTERS as the rows arrive -- a lane loads rows n and n + L/2 of its two columns,
//     e[n] = x[n] + x[n + L/2],   o[n] = (x[n] - x[n + L/2]) w_L^n      (X[2k] = FFT_{L/2}(e)[k],  X[2k + 1] = FFT_{L/2}(o)[k])
// -- and the two half-length transforms pass through an 8-line LDS tile of L/2 points one after the other: e is staged, transformed with the
// half length's own plan and stored to the even rows; then o (kept in registers meanwhile: 36 VGPRs at 2160 points) to the odd rows.  Half
// the LDS: TWO blocks per CU, as on the 1080-point lines.  Same result as the L-point plan up to rounding (another factorisation: 2 x the
// half plan instead of the table's radices for L; measured 2-4e-7 of the range apart, tests/test_gpu_parity.py); option exp bit 8 keeps the
// one-block form for A/B runs.  tw2 = exp(-2 pi i n / L), n < L/2 (ensure_twiddles kind 1); p.tw = the HALF plan's pass table.
template <class PLANH, int MODE>
__global__ __launch_bounds__(Cfg<PLANH::len>::T, (2 * Cfg<PLANH::len>::T + 255) / 256)
void k_fft_lines_split(LinesArgs p, const float2* __restrict__ tw2)
{
    static_assert(MODE == FWD || MODE == INV, "the image's y passes");
    constexpr int LH = PLANH::len;
    using C = Cfg<LH>;
    constexpr int NL = C::NL, LP = C::LP, T = C::T, LW = C::LW;
    static_assert(2 * C::LDS <= 160 * 1024, "tiles of the half length, two per CU");
    constexpr int LPR = NL / 2;             // lanes per position: one float4 = two adjacent columns
    constexpr int ROWS = T / LPR;
    constexpr int NIT = (LH + ROWS - 1) / ROWS;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw = lds + NL * LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c2 = (tid % LPR) * 2;
    const int r0 = tid / LPR;
    float2* wbuf = buf + wave * LW * LP;
    // 8-column tiles: the two tiles of a 128-byte line on one XCD, one behind the other in its dispatch order (see k_fft_lines)
    int tile_x = (int)blockIdx.x, tile_o = (int)blockIdx.y;
    if constexpr (NL * sizeof(float2) < 128) {
        const unsigned nt = gridDim.x, total = nt * gridDim.y, b = blockIdx.y * nt + blockIdx.x;
        if ((nt & 1u) == 0 && b < (total & ~15u) && p.pair_tiles) {
            const unsigned xcd = b & 7u, slot = b >> 3;
            const unsigned lin = ((((slot >> 1) << 3) + xcd) << 1) | (slot & 1u);
            tile_x = (int)(lin % nt); tile_o = (int)(lin / nt);
        }
    }
    const int by = tile_o >= p.outer_skip_lo ? tile_o + p.outer_skip_len : tile_o;
    if (p.nzflags) {
        if (p.nzflags[by * p.nz_stride] == 0) return;              // an empty plane (block-uniform): see k_fft_lines
    }
    auto outer_off = [&](long long outer, long long oblk) {
        return oblk ? (long long)(by >> ZBS) * oblk + (long long)(by & (ZB - 1)) * outer : (long long)by * outer;
    };
    const float2* sbase = p.src + outer_off(p.src_outer, p.src_oblk) + (long long)tile_x * NL + c2;
    // position `pos` of the padded line, this lane's two columns.  The mirrored halo rows (pass B behind the fused rotate kernel) by ONE
    // reflection -- the launcher takes this kernel only where the halo is shorter than the image --: map_src's general path (a modulo)
    // costs the 18 address computations of a lane registers they do not have
    const int mir_n = p.src_mirror ? p.lmap.n : (1 << 30), mir_a = p.src_mirror ? p.lmap.a : (1 << 30), mir_P = p.lmap.P, mir_b = p.lmap.b;
    auto row = [&](int pos) -> float4 {
        if (pos >= p.gap_lo && pos < p.gap_hi) return make_float4(0.f, 0.f, 0.f, 0.f);
        int sn = pos;
        if (pos >= mir_a) {                                        // (never without src_mirror)
            if (pos < mir_P - mir_b) return make_float4(0.f, 0.f, 0.f, 0.f);
            sn = pos - mir_P;
        }
        sn = sn < 0 ? -sn : sn;
        sn = sn >= mir_n ? 2 * mir_n - 2 - sn : sn;
        return *reinterpret_cast<const float4*>(sbase + line_off(sn, p.src_es, p.src_blk));
    };
    float4 lo[NIT], hi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        lo[it] = hi[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((LH % ROWS == 0) || n < LH) {
            lo[it] = row(n);
            hi[it] = row(n + LH);
        }
    }
    // the first stage's twiddles w_L^n pass through the LDS region of the half plan's table, which takes their place before the first
    // transform (registers: the 72 of the tile's rows leave no room for 18 more across the loads)
    for (int i = tid; i < LH; i += T) tw[i] = tw2[i];
    __syncthreads();
    // first stage in registers: lo <- e, hi <- o (inverse transform: conj FFT conj, as everywhere)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int n = r0 + it * ROWS;
        const float2 w = tw[((LH % ROWS == 0) || n < LH) ? n : 0];
        float2 a0 = make_float2(lo[it].x, lo[it].y), a1 = make_float2(lo[it].z, lo[it].w);
        float2 b0 = make_float2(hi[it].x, hi[it].y), b1 = make_float2(hi[it].z, hi[it].w);
        if (MODE == INV) { a0 = cconj(a0); a1 = cconj(a1); b0 = cconj(b0); b1 = cconj(b1); }
        const float2 e0 = cadd(a0, b0), e1 = cadd(a1, b1);
        const float2 o0 = cmul(csub(a0, b0), w), o1 = cmul(csub(a1, b1), w);
        lo[it] = make_float4(e0.x, e0.y, e1.x, e1.y);
        hi[it] = make_float4(o0.x, o0.y, o1.x, o1.y);
    }
    __syncthreads();                                                // every lane has read its twiddles: the plan's table takes the region
    for (int i = tid; i < LH; i += T) tw[i] = p.tw[i];
    float2* dbase = p.dst + outer_off(p.dst_outer, p.dst_oblk) + (long long)tile_x * NL + c2;
    const int nstore = p.store_limit > 0 ? p.store_limit : 2 * LH;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int n = r0 + it * ROWS;
            if ((LH % ROWS == 0) || n < LH) {
                const float4 v = h == 0 ? lo[it] : hi[it];
                buf[c2 * LP + n] = make_float2(v.x, v.y);
                buf[(c2 + 1) * LP + n] = make_float2(v.z, v.w);
            }
        }
        __syncthreads();
        // a wave's lines one at a time: with the odd half pinned in registers a two-line transform (its butterflies' operands side by
        // side) does not fit the 128 VGPRs of two blocks per CU; one line after the other costs the same pass iterations (60 .. 180
        // butterflies of a 540-point line fill the 64 lanes as well as 120 .. 360 of two)
#pragma unroll 1
        for (int j = 0; j < LW; ++j) PLANH::template run<1>(wbuf + j * LP, tw, lane);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = r0 + it * ROWS, n = 2 * k + h;            // output k of this half is row 2 k + h of the line
            if (((LH % ROWS == 0) || k < LH) && n < nstore) {
                float2 a = buf[c2 * LP + k], b = buf[(c2 + 1) * LP + k];
                if (MODE != FWD) { a = cconj(a); b = cconj(b); }
                *reinterpret_cast<float4*>(dbase + line_off(n, p.dst_es, p.dst_blk)) = make_float4(a.x, a.y, b.x, b.y);
            }
        }
        if (h == 0) __syncthreads();                                // the odd half overwrites the tile
    }
}

// ---------------------------------------------------------------------------------- x passes
// A: rows of the (virtually) padded real volume -> half spectrum along x.  M = Px/2.
// Each wave owns LW rows end to end (load, transform, post-process, store): no block barrier after the
// twiddle table is staged.  Lanes run along x with 16-B loads / stores; dst row pitch hxp (complex).
template <class PLAN>
__global__ __launch_bounds__(CfgX<PLAN::len>::T) void k_fft_x_r2c(const float* __restrict__ src, SrcMap map,
                                                                 float2* __restrict__ dst,
                                                                 const float2* __restrict__ twg,
                                                                 const float2* __restrict__ twx, int hxp,
                                                                 long long rows)
{
    constexpr int M = PLAN::len;
    using C = CfgX<M>;
    constexpr int NR = C::NL, LP = C::LP, T = C::T, LW = C::LW;
    constexpr int PAIRS = (M + 1) / 2;               // lanes needed for one row (2 complex = 4 floats per lane)
    extern __shared__ __align__(16) float2 lds[];
    float2* tw = lds + NR * LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* wbuf = lds + wave * LW * LP;
    const long long row0 = (long long)blockIdx.x * NR + wave * LW;

    const int nxs = map.x.n;
    const bool fast_x = (nxs & 3) == 0 && map.x.mode == 0;     // 16-B aligned interior loads
    // row offsets of the wave's LW rows (wave-uniform)
    long long offs[LW], drows[LW];                    // source offset (-1: zero row) and destination row of the wave's rows
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const long long row = row0 + j;
        offs[j] = -1;
        drows[j] = row;
        if (row < rows) {
            const unsigned py = (unsigned)(map.enum_y ? map.enum_y : map.y.P);
            const unsigned urow = (unsigned)row;      // rows = Py*Pz < 2^31
            const int z = (int)(urow / py), y = (int)(urow - (unsigned)z * py);
            const int sy = map_src(map.y, y), sz = map_src(map.z, z);
            if (sy >= 0 && sz >= 0) offs[j] = (long long)nxs * (sy + (long long)map.y.n * sz);
            drows[j] = (long long)z * map.y.P + y;
        }
    }
    // rows in the zero gap of the padded volume are never read by pass B/C (they skip the gap): if none of the
    // wave's rows carries data the wave only has to keep the block barrier company
    bool any_row = false;
#pragma unroll
    for (int j = 0; j < LW; ++j) any_row |= offs[j] >= 0;
    if (!any_row && map.x.mode == 0) {
        for (int i = tid; i < M; i += T) tw[i] = twg[i];
        __syncthreads();
        return;
    }
    // (row, lane-pair) items flattened over the wave's rows: every load iteration has all 64 lanes busy
    constexpr int LITEMS = LW * PAIRS;
    constexpr int LIT = (LITEMS + 63) / 64;
    float4 v[LIT];
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
        const int e = lane + it * 64;
        const int j = e / PAIRS, q = e - j * PAIRS;   // floats 4q .. 4q+3 of padded row j
        long long off = -1;
#pragma unroll
        for (int jj = 0; jj < LW; ++jj) off = (j == jj) ? offs[jj] : off;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (off >= 0 && e < LITEMS) {
            const float* __restrict__ srow = src + off;
            if (fast_x && 4 * q + 3 < nxs) {
                t = *reinterpret_cast<const float4*>(srow + 4 * q);
            } else {
                const int s0 = map_src(map.x, 4 * q), s1 = map_src(map.x, 4 * q + 1);
                const int s2 = map_src(map.x, 4 * q + 2), s3 = map_src(map.x, 4 * q + 3);
                t.x = s0 >= 0 ? srow[s0] : 0.f;
                t.y = s1 >= 0 ? srow[s1] : 0.f;
                t.z = s2 >= 0 ? srow[s2] : 0.f;
                t.w = s3 >= 0 ? srow[s3] : 0.f;
            }
        }
        v[it] = t;
    }
    for (int i = tid; i < M; i += T) tw[i] = twg[i];
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
        const int e = lane + it * 64;
        if (e < LITEMS) {
            const int j = e / PAIRS, q = e - j * PAIRS;
            wbuf[j * LP + 2 * q] = make_float2(v[it].x, v[it].y);
            if (2 * q + 1 < M) wbuf[j * LP + 2 * q + 1] = make_float2(v[it].z, v[it].w);
        }
    }
    __syncthreads();                                  // twiddle table complete (rows are wave-private)
    PLAN::template run<LW>(wbuf, tw, lane);
    // Post-process to the half spectrum, in symmetric pairs.  With s = Z[k] + conj Z[M-k], d = Z[k] - conj Z[M-k],
    // t = -i w_P^k d:   2 X[k] = s + t,   2 X[M-k] = conj(s - t).   The factor 2 is NOT divided out here: image and
    // PSF spectra both carry it and pass E folds the exact power of two into its final scale.
    // Item (row j, q) owns k = 2q, 2q+1 (one 16-B store) and the mirrored M-2q, M-2q-1 (two 8-B stores); items are
    // flattened over the wave's rows so that all lanes stay busy.
    constexpr int HQ = M / 4;
    constexpr int PITEMS = LW * HQ;
#pragma unroll 1
    for (int e = lane; e < PITEMS; e += 64) {
        const int j = e / HQ, q = e - j * HQ;
        const long long row = row0 + j;
        if (row >= rows) continue;
        long long drw = drows[0];
#pragma unroll
        for (int jj = 1; jj < LW; ++jj) drw = (j == jj) ? drows[jj] : drw;
        float2* __restrict__ drow = dst + drw * hxp;
        const float2* __restrict__ zrow = wbuf + j * LP;
        float2 lo[2], hi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * q + h;
            const float2 zk = zrow[k];
            const float2 zm = cconj(zrow[k == 0 ? 0 : M - k]);
            const float2 sm = cadd(zk, zm), d = csub(zk, zm);
            const float2 wd = cmul(twx[k], d);
            const float2 t = make_float2(wd.y, -wd.x);                 // -i * w^k * d
            lo[h] = cadd(sm, t);
            hi[h] = cconj(csub(sm, t));                                 // element M-k (k = 0: element M)
        }
        *reinterpret_cast<float4*>(drow + 2 * q) = make_float4(lo[0].x, lo[0].y, lo[1].x, lo[1].y);
        // the mirrored pair is adjacent too (elements M-2q-1, M-2q): one 16-byte store at 8-byte alignment
        *reinterpret_cast<Pair16*>(drow + M - 2 * q - 1) = Pair16{hi[1].x, hi[1].y, hi[0].x, hi[0].y};
    }
    // per row: the middle elements not covered by the pairs (k = M/2 when M % 4 == 0) and the zero padding up
    // to hxp; flattened over rows as well
    const int nmid = M - 4 * HQ + 1;
    const int ntail = nmid + (hxp - M - 1);
    for (int e = lane; e < LW * ntail; e += 64) {
        const int j = e / ntail, u = e - j * ntail;
        const long long row = row0 + j;
        if (row >= rows) continue;
        long long drw = drows[0];
#pragma unroll
        for (int jj = 1; jj < LW; ++jj) drw = (j == jj) ? drows[jj] : drw;
        float2* __restrict__ drow = dst + drw * hxp;
        const float2* __restrict__ zrow = wbuf + j * LP;
        if (u < nmid) {
            const int k = 2 * HQ + u;
            const float2 zk = zrow[k == M ? 0 : k];
            const float2 zm = cconj(zrow[k == 0 ? 0 : M - k]);
            const float2 sm = cadd(zk, zm), d = csub(zk, zm);
            const float2 wd = cmul(twx[k], d);
            drow[k] = cadd(sm, make_float2(wd.y, -wd.x));
        } else {
            drow[M + 1 + (u - nmid)] = make_float2(0.f, 0.f);
        }
    }
}

// E: half spectrum rows -> real rows, cropped to nx, scaled; one partial sum (double) per block.
// Wave-private rows as in pass A; rows are processed in batches of RB to bound register use.
// FUSE: the rows do not go to HBM as the convolved volume; the epilogue applies adjustImage (the mean is known from the
// z pass), stores the adjusted row only if the caller wants the volume, and turns the rows of acquired planes into the
// acquisition: a copy (no noise) or phase 1 of the Poisson sampler (poisson_phase1; the rest is queued for
// k_poisson_resolve).  Saves the 8 N bytes of the convolved volume's round trip and a launch.
struct C2RFuse {
    const double* scal;           // scal[1] = adjustImage's factor
    float         min_value;
    float*        con;            // adjusted volume, rows as this pass produces them (null: not wanted)
    float*        acq;            // acquisition [planes][ny][nx]
    int           acq_every;      // plane k of this pass is acquired iff k % acq_every == 0, as acquisition plane k / acq_every
    int           idx_zstride;    // plane k of this pass is plane k * idx_zstride of the source volume (RNG counter)
    int           noise;
    double        mul;
    uint32_t      k0, k1, stream;
    PItem*        queue;          // per-block segments of `segcap` items
    unsigned int* qcount;
    unsigned int  segcap;
};

// planes of pass E's input known to be empty (pass D has not written them): plane k of this pass is plane k * stride of the
// dilated flag array (null: none)
struct C2REmpty {
    const int* flags;
    int        stride;
};

template <class PLAN, bool FUSE>
__global__ __launch_bounds__(CfgX<PLAN::len>::T) void k_fft_x_c2r(const float2* __restrict__ srcc,
                                                                 float* __restrict__ out,
                                                                 const float2* __restrict__ twg,
                                                                 const float2* __restrict__ twx, int hxp, int py,
                                                                 int nx, int ny, long long rows, float scale,
                                                                 double* __restrict__ partial, C2RFuse f, C2REmpty em)
{
    constexpr int M = PLAN::len;
    using C = CfgX<M>;
    constexpr int NR = C::NL, LP = C::LP, T = C::T, LW = C::LW, NW = C::NW;
    extern __shared__ __align__(16) float2 lds[];
    float2* tw = lds + NR * LP;
    double* red = reinterpret_cast<double*>(lds + NR * LP + M);              // NW doubles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* wbuf = lds + wave * LW * LP;
    // FUSE: behind the 32 doubles of `red`: the append counter (64 bits) and the two refusal counters, then one P1Scratch per wave
    unsigned long long* qctr = reinterpret_cast<unsigned long long*>(red + 32);
    unsigned int* qovf = reinterpret_cast<unsigned int*>(red + 33);
    constexpr size_t SCR_OFF = ((size_t)(NR * LP + M) * sizeof(float2) + 34 * sizeof(double) + 15) & ~(size_t)15;
    P1Scratch* scratch = reinterpret_cast<P1Scratch*>(reinterpret_cast<char*>(lds) + SCR_OFF);
    if (FUSE && tid == 0) { *qctr = 0ull; qovf[0] = 0u; qovf[1] = 0u; }
    // Every global load of the wave -- its share of the transform's twiddle table, the plane flags of its rows, the rows themselves and
    // the post-processing twiddles of its items -- is requested before anything waits: requested where they are used, they made a chain
    // of five dependent round trips to the caches in a block that lives for a few microseconds.
    constexpr int TWN = (M + T - 1) / T;
    float2 twreg[TWN];
#pragma unroll
    for (int n = 0; n < TWN; ++n) { const int i = tid + n * T; twreg[n] = i < M ? twg[i] : make_float2(0.f, 0.f); }
    const long long row0 = (long long)blockIdx.x * NR + wave * LW;           // output rows (y < ny, z < nz)

    // Pre-process in symmetric pairs.  With A = X[k], B = conj X[M-k], s = A + B, d = A - B, t = i w_P^{-k} d:
    //   Z[k] = s + t,  Z[M-k] = conj(s - t);  the inverse FFT runs as conj(FFT(conj Z)), so conj(Z) is staged:
    //   buf[k] = conj(s + t),  buf[M-k] = s - t.
    // Item (row j, q) owns k = 2q, 2q+1 (one 16-B load) and the mirrored M-2q, M-2q-1 (two 8-B loads), q < M/4;
    // items are flattened over the wave's rows so that every iteration has all 64 lanes busy.
    constexpr int HQ = M / 4;
    constexpr int ITEMS = LW * HQ;
    constexpr int HIT = (ITEMS + 63) / 64;
    long long offs[LW];
    int rowflag[LW];                                  // the plane flag of each of the wave's rows, all requested before the first is looked at
    int zrow_[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const long long row = row0 + j;
        const bool live = row < rows;
        const unsigned urow = live ? (unsigned)row : 0u;             // rows = Ny*Nz < 2^31
        const int z = (int)(urow / (unsigned)ny), y = (int)(urow - (unsigned)z * (unsigned)ny);
        offs[j] = live ? ((long long)z * py + y) * hxp : -1;
        zrow_[j] = __builtin_amdgcn_readfirstlane(z);                // (a wave's rows are wave-uniform)
        rowflag[j] = 1;
    }
    if (em.flags) {
        // scalar loads without a branch between them: they leave together
#pragma unroll
        for (int j = 0; j < LW; ++j) rowflag[j] = em.flags[zrow_[j] * em.stride];
    }
    // a row of an empty plane is a row of zeros -- not read (nobody wrote it), transformed to zeros
#pragma unroll
    for (int j = 0; j < LW; ++j)
        if (rowflag[j] == 0) offs[j] = -1;
    // every row of this wave lies in an empty plane (wave-uniform): nothing to read or transform, its outputs are zeros
    bool wave_empty = !FUSE && em.flags != nullptr;
#pragma unroll
    for (int j = 0; j < LW; ++j) wave_empty = wave_empty && (offs[j] < 0);
    constexpr int NMID = M - 4 * HQ + 1;             // middle elements not covered by the pairs (k = M/2 when M % 4 == 0)
    static_assert(LW * NMID <= 64, "one lane per middle element");
    if (!wave_empty) {
    float4 xa[HIT], xw[HIT];                          // X[2q], X[2q+1] and the twiddles w^{2q}, w^{2q+1} of the item
    Pair16 xb[HIT];                                   // X[M-2q-1], X[M-2q]: adjacent too, one 16-byte load at 8-byte alignment
    bool live_item[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = lane + it * 64;
        const int j = e / HQ, q = e - j * HQ;
        long long off = -1;
#pragma unroll
        for (int jj = 0; jj < LW; ++jj) off = (j == jj) ? offs[jj] : off;
        // no branch around the loads (a merge of loaded and zero registers behind one made the compiler wait for the data at once): a
        // dead item -- a row of an empty plane, a lane beyond the items -- reads the first elements of the buffer and is zeroed below
        const bool item = off >= 0 && e < ITEMS;
        live_item[it] = item;
        const int ql = item ? q : 0;
        const float2* __restrict__ sp = srcc + (item ? off : 0);
        xw[it] = *reinterpret_cast<const float4*>(twx + 2 * (e < ITEMS ? q : 0));
        xa[it] = *reinterpret_cast<const float4*>(sp + 2 * ql);      // X[2q], X[2q+1]
        xb[it] = *reinterpret_cast<const Pair16*>(sp + M - 2 * ql - 1);
    }
    // ... and the middle elements, one per lane
    const int mj = lane / NMID, mk = 2 * HQ + (lane - mj * NMID);
    const bool mid = lane < LW * NMID && mk < M;
    long long moff = -1;
#pragma unroll
    for (int jj = 0; jj < LW; ++jj) moff = (mj == jj) ? offs[jj] : moff;
    const bool mid_row = mid && moff >= 0;
    const int mkl = mid_row ? mk : 0;                                 // (no branch around these loads either)
    const float2* __restrict__ msp = srcc + (mid_row ? moff : 0);
    const float2 ma = msp[mkl], mb = msp[M - mkl], mw = twx[mkl];
    // the transform's twiddles go to LDS now that everything is under way (the barrier below covers them)
#pragma unroll
    for (int n = 0; n < TWN; ++n) { const int i = tid + n * T; if (i < M) tw[i] = twreg[n]; }
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = lane + it * 64;
        if (e < ITEMS) {
            const int j = e / HQ, q = e - j * HQ;
            float2* zrow = wbuf + j * LP;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * q + h;
                float2 a = h == 0 ? make_float2(xa[it].x, xa[it].y) : make_float2(xa[it].z, xa[it].w);
                float2 b = cconj(h == 0 ? make_float2(xb[it].c, xb[it].d) : make_float2(xb[it].a, xb[it].b));
                if (!live_item[it]) { a = make_float2(0.f, 0.f); b = make_float2(0.f, 0.f); }
                const float2 sm = cadd(a, b), d = csub(a, b);
                const float2 wk = h == 0 ? make_float2(xw[it].x, xw[it].y) : make_float2(xw[it].z, xw[it].w);
                const float2 wd = cmul(cconj(wk), d);                       // w^{-k} d
                const float2 t = make_float2(-wd.y, wd.x);                  // i w^{-k} d
                zrow[k] = cconj(cadd(sm, t));                               // dead rows were loaded as zeros
                if (k > 0) zrow[M - k] = csub(sm, t);
            }
        }
    }
    if (mid) {
        float2 zk = make_float2(0.f, 0.f);
        if (mid_row) {
            const float2 a = ma, b = cconj(mb);
            const float2 sm = cadd(a, b), d = csub(a, b);
            const float2 wd = cmul(cconj(mw), d);
            zk = cconj(cadd(sm, make_float2(-wd.y, wd.x)));
        }
        wbuf[mj * LP + mk] = zk;
    }
    } else {
#pragma unroll
        for (int n = 0; n < TWN; ++n) { const int i = tid + n * T; if (i < M) tw[i] = twreg[n]; }
    }
    __syncthreads();                                  // twiddle table complete (rows are wave-private)
    if (!wave_empty) PLAN::template run<LW>(wbuf, tw, lane);
    // z[n] = conj(buf[n]) = x[2n] + i x[2n+1]; lane q writes x[4q..4q+3]
    if (FUSE) {
        // nx % 4 == 0 and 16-byte aligned outputs are guaranteed by the launcher
        const double corr = f.scal[1];
        P1Args pa;
        pa.mul = f.mul; pa.mulf = (float)f.mul; pa.k0 = f.k0; pa.k1 = f.k1; pa.stream = f.stream;
        pa.seg = f.queue + (unsigned long long)blockIdx.x * f.segcap; pa.segcap = f.segcap; pa.ctr = qctr; pa.ovf = qovf;
        const int nq4 = nx >> 2;
        for (int j = 0; j < LW; ++j) {
            const long long row = row0 + j;
            if (row >= rows) break;                                          // wave-uniform
            const unsigned urow = (unsigned)row;
            const int k = (int)(urow / (unsigned)ny), y = (int)(urow - (unsigned)k * (unsigned)ny);
            const bool acquired = (k % f.acq_every) == 0;
            const float2* __restrict__ zrow = wbuf + j * LP;
            const unsigned long long idx_row = (unsigned long long)nx * ((unsigned long long)y + (unsigned long long)ny * ((unsigned long long)k * (unsigned long long)f.idx_zstride));
            const unsigned long long acq_row = (unsigned long long)nx * ((unsigned long long)y + (unsigned long long)ny * (unsigned long long)(k / f.acq_every));
            for (int q0 = 0; q0 < nq4; q0 += 64) {                           // uniform trip count: phase 1 needs every lane
                const int q = q0 + lane;
                const bool valid = q < nq4;
                float vv[4] = {0.f, 0.f, 0.f, 0.f};
                if (valid) {
                    const float2 z0 = zrow[2 * q];
                    const float2 z1 = (2 * q + 1 < M) ? zrow[2 * q + 1] : make_float2(0.f, 0.f);
                    vv[0] = adjust_one_f(z0.x * scale, corr, f.min_value);
                    vv[1] = adjust_one_f(-z0.y * scale, corr, f.min_value);
                    vv[2] = adjust_one_f(z1.x * scale, corr, f.min_value);
                    vv[3] = adjust_one_f(-z1.y * scale, corr, f.min_value);
                    if (f.con) *reinterpret_cast<float4*>(f.con + row * nx + 4 * q) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
                if (acquired) {
                    float ov[4] = {vv[0], vv[1], vv[2], vv[3]};
                    if (f.noise)
                        poisson_phase1<false>(vv, valid, idx_row + 4ull * (unsigned long long)q, acq_row + 4ull * (unsigned long long)q, pa,
                                       &scratch[wave], lane, ov);
                    if (valid) *reinterpret_cast<float4*>(f.acq + acq_row + 4 * q) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                }
            }
        }
        __syncthreads();
        if (tid == 0 && f.noise) p1_publish(f.qcount + (size_t)QCOUNT_WORDS * blockIdx.x, *qctr, qovf);   // no queue without noise (qcount is null then)
        return;
    }
    double acc = 0.0;
    const bool vec_out = (nx & 3) == 0;
    const bool want_sum = partial != nullptr;             // null: adjustImage's sum came from the z pass (early sum)
    for (int j = 0; j < LW; ++j) {
        const long long row = row0 + j;
        if (row >= rows) break;
        float* __restrict__ o = out + row * nx;
        if (wave_empty) {
            if (vec_out) for (int q = lane; 4 * q < nx; q += 64) *reinterpret_cast<float4*>(o + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            else for (int q = lane; q < nx; q += 64) o[q] = 0.f;
            continue;
        }
        const float2* __restrict__ zrow = wbuf + j * LP;
        for (int q = lane; 4 * q < nx; q += 64) {
            const float2 z0 = zrow[2 * q];
            const float2 z1 = (2 * q + 1 < M) ? zrow[2 * q + 1] : make_float2(0.f, 0.f);
            const float x0 = z0.x * scale, x1 = -z0.y * scale, x2 = z1.x * scale, x3 = -z1.y * scale;
            if (vec_out) {
                *reinterpret_cast<float4*>(o + 4 * q) = make_float4(x0, x1, x2, x3);
                if (want_sum) acc += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
            } else {
                o[4 * q] = x0; acc += (double)x0;
                if (4 * q + 1 < nx) { o[4 * q + 1] = x1; acc += (double)x1; }
                if (4 * q + 2 < nx) { o[4 * q + 2] = x2; acc += (double)x2; }
                if (4 * q + 3 < nx) { o[4 * q + 3] = x3; acc += (double)x3; }
            }
        }
    }
    if (!want_sum) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (int w = 0; w < NW; ++w) sum += red[w];
        partial[blockIdx.x] = sum;
    }
}

// Empty-plane bookkeeping for the convolution passes (one small block per view).  flags[z] = 1 where the fused rotate kernel found a
// non-zero voxel in plane z.  dil[z] = 1 iff any plane the z pass's Kz taps reach from output plane z -- z + c - t, t = 0 .. kz - 1,
// mirror-single at the faces -- is non-empty (passes D and E).  bits: bit i = flags[mirror(i - NZ_EXT)] (pass C').
__global__ __launch_bounds__(1024) void k_plane_flags_finish(const int* __restrict__ flags, int n, int kz, int c, unsigned int* __restrict__ bits,
                                                             int nwords, int* __restrict__ dil, int* __restrict__ empty_hint)
{
    // the flags once through LDS (volumes of up to 16384 planes; beyond that straight from memory): every thread then reads up to 64 of them
    __shared__ unsigned char sf[16384];
    __shared__ int cnt, cnt_dil;
    const int t = threadIdx.x;
    const bool in_lds = n <= 16384;
    if (t == 0) { cnt = 0; cnt_dil = 0; }
    __syncthreads();
    int mine = 0;
    for (int z = t; z < n; z += 1024) {
        const int f = flags[z] != 0;
        if (in_lds) sf[z] = (unsigned char)f;
        mine += !f;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (t == 0 && empty_hint) *empty_hint = cnt;          // how many planes are empty: a hint for the host (page-locked), see rotate_attenuate_fftx
    auto flag = [&](int z) { return in_lds ? (int)sf[z] : (flags[z] != 0 ? 1 : 0); };
    // (one reflection covers every index asked for here unless the PSF is deeper than the volume: the general form only then)
    auto mir = [&](int i) {
        int j = i < 0 ? -i : i;
        if (j >= n) j = 2 * n - 2 - j;
        return (j >= 0 && j < n) ? j : mirror_index(i, n);
    };
    // one bit per lane, 64 bits per ballot: two words of the bit string per wave and trip
    for (int i0 = 0; i0 < nwords * 32; i0 += 1024) {
        const int i = i0 + t;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(i < nwords * 32 && flag(mir(i - NZ_EXT)) != 0);
        if ((t & 63) == 0) {
            if (i / 32 < nwords) bits[i / 32] = (unsigned int)m;
            if (i / 32 + 1 < nwords) bits[i / 32 + 1] = (unsigned int)(m >> 32);
        }
    }
    int mine_dil = 0;
    for (int z = t; z < n; z += 1024) {
        int any = 0;
        for (int tt = 0; tt < kz; ++tt) any |= flag(mir(z + c - tt));
        dil[z] = any;
        mine_dil += !any;
    }
    if (empty_hint) {
        // (statistics only: how many planes of the z pass's OUTPUT are empty -- what passes D and E skip; mvsim_get_plane_stats)
        if (mine_dil) atomicAdd(&cnt_dil, mine_dil);
        __syncthreads();
        if (t == 0) { empty_hint[1] = cnt_dil; empty_hint[2] = n; }
    }
}

// scal[0] = factor * sum of the partials; with corr_n > 0 also adjustImage's factor scal[1] = (target - min) / (scal[0] / n)
// (Tools.java:146-147: the subtraction in float, the rest in double) -- one launch less per view than a separate kernel.
__global__ __launch_bounds__(1024) void k_reduce_partials(const double* __restrict__ partial, long long count,
                                                          double* __restrict__ scal, double factor, long long corr_n,
                                                          float min_value, float target)
{
    __shared__ double sh[16];
    partial += (long long)blockIdx.x * count;             // stacked views: block v reduces view v's partials into its own pair
    scal += 2 * blockIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    long long i = threadIdx.x;
    for (; i + 3 * 1024 < count; i += 4 * 1024) {
        a0 += partial[i];
        a1 += partial[i + 1024];
        a2 += partial[i + 2 * 1024];
        a3 += partial[i + 3 * 1024];
    }
    for (; i < count; i += 1024) a0 += partial[i];
    double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];
        const double sum = t * factor;
        scal[0] = sum;
        if (corr_n > 0) scal[1] = (double)(target - min_value) / (sum / (double)corr_n);
    }
}

// ---------------------------------------------------------------------------------- z pass without transforms
// Pass C as a DIRECT convolution along z in the mixed domain.  After the x and y transforms every z line (kx, ky) of
// the image spectrum has to be convolved with the z line of the PSF's (x,y) spectrum, which has only Kz taps:
//     out(z) = sum_j g[j] * f(mirror(z + c - j)),   g[j] = G2[j][ky][kx],  c = Kz/2.
// For the PSF depths of this workload (Kz <= 64) that is fewer, and far better shaped, flops than a forward
// transform, a product and an inverse transform of the padded length (4 FMAs per tap against add-heavy butterflies),
// and it removes whole pieces of the FFT formulation: no z padding (the mirrored halo planes are the SAME planes,
// re-read through an index map, so passes A and B run on Nz planes instead of Nz+Kz-1), no expansion of the PSF
// spectrum to Pz planes (one launch and 2 C bytes of HBM traffic), no zero-gap planes.
// Tile: NLZ = 16 adjacent kx columns (128-B rows) x a z chunk with its halo, staged in LDS row by row exactly as it
// lies in HBM (pitch 17 to spread the banks); taps of the 16 lines in LDS.  A lane owns one line and ZU consecutive
// outputs; taps are consumed in chunks of ZJ with a sliding register window over the inputs, so each LDS word is
// read once per lane and chunk.  Out of place (chunks read each other's halo).
// (MVSIM_EXP_NLZ = 32 -- 256-byte rows -- measured 0.35 ms with 160-output tiles, 0.42 ms with 112-output tiles, against 0.32 ms)
#ifndef MVSIM_EXP_NLZ
#define MVSIM_EXP_NLZ 16
#endif
constexpr int NLZ = MVSIM_EXP_NLZ, ZU = 16, ZJ = 8, ZT = 256, ZPITCH = NLZ + 1;
constexpr int ZLPR = NLZ / 2;                    // lanes per staged row (16 B each)
constexpr int ZRPI = ZT / ZLPR;                  // rows per staging iteration
#ifndef MVSIM_ZNIT
#define MVSIM_ZNIT 9
#endif
#ifndef MVSIM_ZBLK
#define MVSIM_ZBLK 3
#endif
#ifndef MVSIM_ZLDS
#define MVSIM_ZLDS (44 * 1024)
#endif
constexpr int ZNIT = MVSIM_ZNIT;                 // staged rows <= ZNIT * ZRPI (zconv_chunk keeps the tile below that)
constexpr int ZBLOCKS_PER_CU = MVSIM_ZBLK;
constexpr size_t ZLDS_TARGET = MVSIM_ZLDS;
// float2 elements of the staged input region (rounded to 16 bytes: the tap region behind it is accessed as float4)
__host__ __device__ constexpr size_t zconv_frows(int zc, int kz, int kzp)
{
    return (((size_t)(kzp - kz + zc + kz - 1) * ZPITCH) + 1) & ~(size_t)1;
}

struct ZConvArgs {
    const float2* src;      // z-blocked (see below)
    float2*       dst;      // z-blocked
    const float2* taps;     // z-blocked as well, Kz planes: (ky >> ZBS) * taps_blk + t * zs + (ky & (ZB-1)) * Hxp + kx
    // src and dst in the z-blocked layout (LinesArgs::src_blk): element (z, ky, kx) at (ky >> ZBS) * blk + z * zs + (ky & (ZB-1)) * Hxp + kx
    long long     src_blk, dst_blk, taps_blk, zs;
    int           hxp, nz, kz, c, zc;   // nz: output planes; zc: outputs per tile along z (multiple of ZU)
    // z-slab tiling: the mirror boundary acts on the GLOBAL plane index; src holds the global planes from z_in0 on,
    // dst plane 0 is global plane z_out0 (whole volume: nz_global = nz, both offsets 0)
    int           nz_global, z_in0, z_out0;
    // adjustImage's sum taken from the spectrum side (null: not wanted).  The sum of the cropped convolved volume is a
    // linear functional of this pass's output: sum = scale * SUM_{z, ky, kx <= M} Re( F[z][ky][kx] * wx[kx] * wy[ky] ) with
    // wx[kx] = c(kx) * SUM_{x < Nx} e^{+2 pi i kx x / Px} (c = 1 for kx = 0 and M, else 2: Hermitian half), wy[ky] alike
    // without c.  One partial (double) per block.
    const double2* wx;
    const double2* wy;
    double*       sum_partial;
    // planes of the input known to be empty (pass B has not even written them; null: none): their rows are not loaded, and a tile
    // whose rows are all empty computes and stores nothing (pass D skips its planes on the same flags)
    // bit i = plane mirror(i - NZ_EXT) is non-empty: the flags as a bit string that already holds the mirrored halo on both sides, so
    // that the planes a tile reaches are ONE run of bits (k_plane_flags_finish); null: every plane is read
    const unsigned int* nzbits;
    // k_zconv_strided only: outputs are the planes k * stride (the compact planes extractSlices reads); LDS rows reserved for the tile
    int           stride, frows;
    // tiles: nch chunks along z x nkxb groups of NLZ columns x py rows, on a 1-D grid (zconv_tile)
    int           nch, nkxb, py, plain_order;
    // stacked views (mvsim_simulate_views_dev; blockIdx.y = view): view v finds its planes, its outputs and the taps of ITS PSF that
    // many elements further on, and its partial sums behind those of the views before it
    long long     src_view, dst_view, taps_view;
};

__device__ __forceinline__ void zconv_view(ZConvArgs& p)
{
    const long long v = blockIdx.y;                       // 0 for a single view (strides unused)
    p.src += v * p.src_view;
    p.dst += v * p.dst_view;
    p.taps += v * p.taps_view;
    if (p.sum_partial) p.sum_partial += v * ((long long)p.nch * p.nkxb * p.py);
}

// Which tile a block of the z pass works on.  Workgroups go to the eight XCDs round robin by their linear id, and the chunks of one
// column group share their halo rows and all of their taps: with the chunk index fastest in the GRID, neighbours landed on different
// XCDs and both fetched those rows into their own L2 (k_zconv read 0.60 GB where pass B had written 0.48: profiles/r04_g_pmc_hbm_traffic.txt).
// So every XCD walks a contiguous range of the tile order (chunk fastest, then column group, then ky) instead.
__device__ __forceinline__ bool zconv_tile(const ZConvArgs& p, int& chunk, int& kxb, int& ky, long long& tile)
{
    const long long total = (long long)p.nch * p.nkxb * p.py, per = (total + 7) / 8;
    const long long l = blockIdx.x;
    tile = p.plain_order ? l : (l & 7) * per + (l >> 3);
    if (tile >= total) return false;
    chunk = (int)(tile % p.nch);
    const long long col = tile / p.nch;
    kxb = (int)(col % p.nkxb);
    ky = (int)(col / p.nkxb);
    return true;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// The z pass loads its tile without a branch around any load: behind `if (row wanted) v = load` the compiler merges the loaded registers
// with the zeros of the other path, pays for that with a register copy -- and a wait for the tile's FIRST row before the second is
// requested (and again before the sum's weights): two extra trips to memory in a block that lives for 12 us.  A row that is not wanted
// (padding, past the end, an empty plane -- nobody wrote it) reads the tile's column in the first plane instead (a cache hit) and is cleared
// with this mask; an AND, not a select -- a select whose operand is a load is turned back into the branch.
__device__ __forceinline__ float4 zconv_keep4(float4 t, bool keep)
{
    const unsigned int m = keep ? 0xFFFFFFFFu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(t.x) & m), __uint_as_float(__float_as_uint(t.y) & m),
                       __uint_as_float(__float_as_uint(t.z) & m), __uint_as_float(__float_as_uint(t.w) & m));
}

constexpr int TNIT = 64 / ZRPI;                       // kzp <= 64 rows of taps

// One block per tile = (z chunk, 16-column group, row ky): grid (chunks, Hxp/16, Py) -- the chunks of a column are
// dispatched together, so a chunk's halo rows are usually still in L2 / Infinity Cache from its neighbour.
// (Measured and dropped on the plane-major layout: a persistent variant that prefetches the next tile into registers, and a
// wave-private sliding-ring formulation -- one wave per column group, taps in registers, every input plane read once, no
// barriers: 16 % fewer vector instructions and no halo re-reads, yet 0.38 ms against 0.34 ms.  What bound all of them was
// the distance between the rows of a tile -- a whole plane -- and not the vector pipe: with the z-blocked layout
// (LinesArgs::src_blk) the rows of a tile are ZB * Hxp elements apart and this kernel runs in 0.32 ms; with a quarter of its
// FMAs it would take 0.30 ms, with two instead of three blocks per CU the same 0.32 ms.)
__global__ __launch_bounds__(ZT, ZBLOCKS_PER_CU) void k_zconv(ZConvArgs p)
{
    extern __shared__ __align__(16) float2 lds[];
    zconv_view(p);
    const int kzp = (p.kz + ZJ - 1) / ZJ * ZJ;            // taps padded with zeros to whole chunks
    const int padf = kzp - p.kz;                          // leading zero rows the padded taps may touch
    int chunk, kxb, ky;
    long long tile;
    if (!zconv_tile(p, chunk, kxb, ky, tile)) return;
    const int zc0 = chunk * p.zc;
    const int zn = min(p.zc, p.nz - zc0);                 // outputs of this block
    const int rows = padf + zn + p.kz - 1;                // staged input rows
    float2* f = lds;                                      // [rows][ZPITCH]
    float2* g = lds + zconv_frows(p.zc, p.kz, kzp);      // [kzp][NLZ]
    const int tid = threadIdx.x;
    const long long rowb = (long long)(ky & (ZB - 1)) * p.hxp + (long long)kxb * NLZ;
    const long long scol = (long long)(ky >> ZBS) * p.src_blk + rowb, dcol = (long long)(ky >> ZBS) * p.dst_blk + rowb;
    const long long tcol = (long long)(ky >> ZBS) * p.taps_blk + rowb;
    const int c2 = (tid % ZLPR) * 2;                      // staging: ZLPR lanes x 16 B per row
    const int hl = p.kz - 1 - p.c;                        // halo below z = 0
    // all global loads of the tile are issued before anything waits
    float4 v[ZNIT], tv[TNIT];
    // which of the staged rows belong to non-empty planes: ZRPI = 32 rows per staging iteration = one 32-bit word per iteration, cut
    // out of the flag bit string with scalar loads and shifts (block-uniform; all ones without flags).  Rows of empty planes are not
    // read -- pass B has not written them --, and a tile without a single non-empty row has nothing to compute or store.
    static_assert(ZRPI == 32, "one word of plane bits per staging iteration");
    unsigned int rowbits[ZNIT];
    unsigned int zbmask = 0xFFFFFFFFu;                               // z blocks of this tile that have anything to compute (up to ZNIT * 2 = 18)
#pragma unroll
    for (int it = 0; it < ZNIT; ++it) rowbits[it] = 0xFFFFFFFFu;
    if (p.nzbits) {
        const int b0 = p.z_out0 + zc0 - hl + NZ_EXT - padf;           // bit of staged row 0 (>= 0: NZ_EXT covers halo and padding)
        const unsigned int* __restrict__ wp = p.nzbits + (b0 >> 5);
        const int sh = b0 & 31;
        unsigned int anyrow = 0u, missing = 0u;                        // any non-empty row / any empty row among the tile's rows
#pragma unroll
        for (int it = 0; it < ZNIT; ++it) {
            const unsigned long long two = ((unsigned long long)wp[it + 1] << 32) | wp[it];
            unsigned int m = (unsigned int)(two >> sh);
            // rows below padf are padding, rows from `rows` on do not exist
            const int lo = padf - it * 32, hi = rows - it * 32;
            unsigned int valid = ~0u;
            if (lo > 0) valid &= lo >= 32 ? 0u : ~0u << lo;
            if (hi < 32) valid &= hi <= 0 ? 0u : ~0u >> (32 - hi);
            m &= valid;
            rowbits[it] = m;
            anyrow |= m;
            missing |= m ^ valid;
        }
        if (anyrow == 0u) {
            if (p.sum_partial && tid == 0) p.sum_partial[tile] = 0.0;
            return;
        }
        // the same per z block of ZU = 16 outputs: block zb reads the staged rows [padf + 16 zb, padf + 16 zb + 16 + kz - 1) -- at most
        // 79 rows from bit padf + 16 (zb & 1) of word zb / 2 on (static word indices: padf < 16) --; a block that reaches nothing but
        // empty planes computes and stores nothing (pass D asks the dilated flags, which say the same)
        static_assert(ZU == 16 && ZNIT * 2 <= 32, "z blocks of 16 outputs, two per word of plane bits");
        const int len = ZU + p.kz - 1;
        if (missing != 0u) zbmask = 0u;                               // (a tile without an empty row -- every tile of a dense volume -- skips this)
#pragma unroll
        for (int zb = 0; zb < ZNIT * 2 && missing != 0u; ++zb) {
            const int w0 = zb / 2, off = padf + 16 * (zb & 1);
            auto word = [&](int i) { return (w0 + i) < ZNIT ? rowbits[(w0 + i) < ZNIT ? (w0 + i) : 0] : 0u; };
            const unsigned long long lo = ((unsigned long long)word(1) << 32) | word(0), hi = ((unsigned long long)word(3) << 32) | word(2);
            // bits [off, off + len) of the 128-bit string hi:lo, len <= 79, off < 32
            const unsigned long long a = (lo >> off) | (off ? hi << (64 - off) : 0ull);
            const unsigned long long b = off ? hi >> off : hi;
            const unsigned long long ma = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
            const unsigned long long mb = len > 64 ? ((1ull << (len - 64)) - 1ull) : 0ull;
            if (((a & ma) | (b & mb)) != 0ull) zbmask |= 1u << zb;
        }
    }
#pragma unroll
    for (int it = 0; it < ZNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        const bool want = r >= padf && r < rows && ((rowbits[it] >> (tid / ZLPR)) & 1u);
        int z = p.z_out0 + zc0 + (r - padf) - hl;
        if ((unsigned)z >= (unsigned)p.nz_global) z = mirror_index(z, p.nz_global);
        const long long zo = want ? (long long)(z - p.z_in0) * p.zs : 0ll;
        v[it] = zconv_keep4(*reinterpret_cast<const float4*>(p.src + zo + scol + c2), want);
    }
#pragma unroll
    for (int it = 0; it < TNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        const bool want = r < p.kz;
        tv[it] = zconv_keep4(*reinterpret_cast<const float4*>(p.taps + (long long)(want ? r : 0) * p.zs + tcol + c2), want);
    }
    // the sum's weights travel with the tile (requested in the epilogue they cost a block 1-4 k cycles of exposed latency: clock stamps)
    double2 wxa = make_double2(0.0, 0.0), wyb = make_double2(0.0, 0.0);
    if (p.sum_partial) { wxa = p.wx[kxb * NLZ + (tid & (NLZ - 1))]; wyb = p.wy[ky]; }
#pragma unroll
    for (int it = 0; it < ZNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        if (r < rows) {
            f[r * ZPITCH + c2] = make_float2(v[it].x, v[it].y);
            f[r * ZPITCH + c2 + 1] = make_float2(v[it].z, v[it].w);
        }
    }
#pragma unroll
    for (int it = 0; it < TNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        if (r < kzp) *reinterpret_cast<float4*>(g + r * NLZ + c2) = tv[it];
    }
    __syncthreads();
    const int line = tid & (NLZ - 1);
    const int nzb = (zn + ZU - 1) / ZU;
    double sre = 0.0, sim = 0.0;                          // this lane's sum over z of its line's outputs
    for (int zb = tid / NLZ; zb < nzb; zb += ZT / NLZ) {
        if (!((zbmask >> zb) & 1u)) continue;
        const int z0 = zb * ZU;
        v2f acc[ZU];
#pragma unroll
        for (int u = 0; u < ZU; ++u) acc[u] = v2f{0.f, 0.f};
        // window w[i] = f[base + i]; output u, tap j0 + t reads w[u - t + ZJ - 1]
        float2 w[ZU + ZJ - 1];
        const int base = padf + z0 + p.kz - 1 - (ZJ - 1);   // j0 = 0
        // No clamps on the row index: rows above the staged range (the last z block of a chunk may be partial) lie inside the LDS
        // region all the same (it is sized for zc outputs) and whatever they hold only feeds outputs that are neither stored nor
        // summed; base never goes below 0 (it ends at z0: padf + kz = kzp).  One address register plus immediate offsets instead
        // of a 64-bit multiply-add per window element (24 quarter-rate v_mad_u64_u32 per z block).
        const float2* __restrict__ fw = f + base * ZPITCH + line;
#pragma unroll
        for (int i = 0; i < ZU + ZJ - 1; ++i) w[i] = fw[i * ZPITCH];
#pragma unroll 1
        for (int j0 = 0; j0 < kzp; j0 += ZJ) {
            float2 gg[ZJ];
#pragma unroll
            for (int t = 0; t < ZJ; ++t) gg[t] = g[(j0 + t) * NLZ + line];
#pragma unroll
            for (int t = 0; t < ZJ; ++t) {
#pragma unroll
                for (int u = 0; u < ZU; ++u) {
                    // complex multiply-accumulate as two packed FMAs: (gr, gr) * (xr, xi), then (-gi, gi) * (xi, xr)
                    const float2 x = w[u - t + ZJ - 1];
                    acc[u] = __builtin_elementwise_fma(v2f{gg[t].x, gg[t].x}, v2f{x.x, x.y}, acc[u]);
                    acc[u] = __builtin_elementwise_fma(v2f{-gg[t].y, gg[t].y}, v2f{x.y, x.x}, acc[u]);
                }
            }
            if (j0 + ZJ < kzp) {
                // next chunk of taps looks ZJ rows further down
#pragma unroll
                for (int i = ZU + ZJ - 2; i >= ZJ; --i) w[i] = w[i - ZJ];
                fw -= ZJ * ZPITCH;
#pragma unroll
                for (int i = 0; i < ZJ; ++i) w[i] = fw[i * ZPITCH];
            }
        }
        float2* d = p.dst + (long long)(zc0 + z0) * p.zs + dcol + line;
#pragma unroll
        for (int u = 0; u < ZU; ++u) {
            if (z0 + u < zn) *d = make_float2(acc[u].x, acc[u].y);
            d += p.zs;
        }
        if (p.sum_partial) {
            // pairwise in single precision over the (at most) 16 outputs, then double: error ~2 roundings of fp32
            v2f t8[ZU / 2];
#pragma unroll
            for (int u = 0; u < ZU / 2; ++u) {
                const v2f a = (z0 + 2 * u < zn) ? acc[2 * u] : v2f{0.f, 0.f};
                const v2f b = (z0 + 2 * u + 1 < zn) ? acc[2 * u + 1] : v2f{0.f, 0.f};
                t8[u] = a + b;
            }
#pragma unroll
            for (int w = ZU / 4; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) t8[u] = t8[u] + t8[u + w];
            sre += (double)t8[0].x;
            sim += (double)t8[0].y;
        }
    }
    if (p.sum_partial) {
        __shared__ double red[ZT / 64];
        const double wr = wxa.x * wyb.x - wxa.y * wyb.y, wi = wxa.x * wyb.y + wxa.y * wyb.x;
        double t = sre * wr - sim * wi;                   // Re( s * W )
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = t;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < ZT / 64; ++w) sum += red[w];
            p.sum_partial[tile] = sum;
        }
    }
}

// The z pass of a view that only returns the acquisition (compact planes, inc = S > 1): passes D and E read the planes k * S alone, so
// only those are convolved -- 1 / S of the taps' work of k_zconv, which at Kz = 63 is what bounds it (configs[3]: 1024^3, inc 4) -- and
// adjustImage's sum, which is over EVERY plane of the convolved volume, is taken from the tile's input rows instead of its outputs:
//   SUM_{z in tile} out[z] = SUM_r f[r] * W[r],  W[r] = SUM of the taps t with a_t <= r < a_t + zn  (a_t = padf + Kz - 1 - t),
// a difference of two prefix sums of the line's taps (interior rows: the sum of all taps); one complex product per staged row, in fp64.
// Same tile, same staging (empty planes included) as k_zconv.  The outputs in polyphase form: plane z0 + S k needs the rows
// padf + Kz - 1 - rho + S (k - q) for the taps t = S q + rho -- per phase rho a stride-1 convolution of every S-th row with every S-th tap.
// Work unit = (phase, chunk of ZJ taps of that phase); the four 16-lane groups of a wave take the units round robin on the SAME 16
// outputs per lane (register window of ZU + ZJ - 1 rows S apart: 8 packed FMAs per LDS word, as in k_zconv) and add up their partial
// sums over the lane crossbar (reduce-scatter: xor 32, xor 16), each group storing a quarter of the wave's outputs.  The waves share a
// trip's (up to) 64 outputs evenly.  The summation order differs from k_zconv's tap-by-tap order: ~1e-7 of the range
// (test_compact_planes_view_equals_full_view; option zconv_strided=0 keeps the planes of a compact view bit-identical to a full one).
// LDS rows are NLZ float2 apart here (no padding): the groups of a half wave read rows an odd distance apart -> other 32 banks.
// a + b after v_permlane32_swap (W = 32: the upper 32 lanes of a against the lower 32 of b) or v_permlane16_swap (W = 16: the odd rows
// of 16 lanes of a against the even rows of b).  (Scalars first: __builtin_bit_cast on an element of an ext_vector reads element 0.)
template <int W> __device__ __forceinline__ v2f swap_add(v2f a, v2f b)
{
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    const auto x = W == 32 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(ax), __float_as_uint(bx), false, false)
                           : __builtin_amdgcn_permlane16_swap(__float_as_uint(ax), __float_as_uint(bx), false, false);
    const auto y = W == 32 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(ay), __float_as_uint(by), false, false)
                           : __builtin_amdgcn_permlane16_swap(__float_as_uint(ay), __float_as_uint(by), false, false);
    const unsigned int x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
    return v2f{__uint_as_float(x0), __uint_as_float(y0)} + v2f{__uint_as_float(x1), __uint_as_float(y1)};
}

template <int S>
__global__ __launch_bounds__(ZT, ZBLOCKS_PER_CU) void k_zconv_strided(ZConvArgs p)
{
    extern __shared__ __align__(16) float2 lds[];
    zconv_view(p);
    constexpr int PITCH = NLZ;
    const int nq = (p.kz + S - 1) / S;                    // taps of phase 0 (the longest)
    const int qp = (nq + ZJ - 1) / ZJ * ZJ;               // ... padded to whole chunks
    const int gr = S * qp;                                // tap rows (zero from Kz on)
    const int padf = gr - p.kz;                           // leading zero rows the padded taps may touch
    int chunk, kxb, ky;
    long long tile;
    if (!zconv_tile(p, chunk, kxb, ky, tile)) return;
    const int zc0 = chunk * p.zc;
    const int zn = min(p.zc, p.nz - zc0);                 // planes of this tile (outputs: every S-th, from the tile's first)
    const int rows = padf + zn + p.kz - 1;                // staged input rows
    float2* f = lds;                                      // [frows][PITCH]
    float2* g = lds + (size_t)p.frows * PITCH;            // [gr][NLZ]
    float2* gc = g + (size_t)gr * NLZ;                    // [Kz + 1][NLZ] exclusive prefix sums of the taps; [16][NLZ] segment totals behind
    const int tid = threadIdx.x;
    const long long rowb = (long long)(ky & (ZB - 1)) * p.hxp + (long long)kxb * NLZ;
    const long long scol = (long long)(ky >> ZBS) * p.src_blk + rowb, dcol = (long long)(ky >> ZBS) * p.dst_blk + rowb;
    const long long tcol = (long long)(ky >> ZBS) * p.taps_blk + rowb;
    const int c2 = (tid % ZLPR) * 2;
    const int hl = p.kz - 1 - p.c;
    float4 v[ZNIT], tv[TNIT];
    static_assert(ZRPI == 32, "one word of plane bits per staging iteration");
    unsigned int rowbits[ZNIT];
#pragma unroll
    for (int it = 0; it < ZNIT; ++it) rowbits[it] = 0xFFFFFFFFu;
    if (p.nzbits) {
        const int b0 = p.z_out0 + zc0 - hl + NZ_EXT - padf;           // bit of staged row 0 (>= 0: the host checks hl + padf <= NZ_EXT)
        const unsigned int* __restrict__ wp = p.nzbits + (b0 >> 5);
        const int sh = b0 & 31;
        unsigned int anyrow = 0u;
#pragma unroll
        for (int it = 0; it < ZNIT; ++it) {
            const unsigned long long two = ((unsigned long long)wp[it + 1] << 32) | wp[it];
            unsigned int m = (unsigned int)(two >> sh);
            const int lo = padf - it * 32, hi = rows - it * 32;
            unsigned int valid = ~0u;
            if (lo > 0) valid &= lo >= 32 ? 0u : ~0u << lo;
            if (hi < 32) valid &= hi <= 0 ? 0u : ~0u >> (32 - hi);
            m &= valid;
            rowbits[it] = m;
            anyrow |= m;
        }
        if (anyrow == 0u) {
            if (p.sum_partial && tid == 0) p.sum_partial[tile] = 0.0;
            return;
        }
    }
#pragma unroll
    for (int it = 0; it < ZNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        const bool want = r >= padf && r < rows && ((rowbits[it] >> (tid / ZLPR)) & 1u);
        int z = p.z_out0 + zc0 + (r - padf) - hl;
        if ((unsigned)z >= (unsigned)p.nz_global) z = mirror_index(z, p.nz_global);
        const long long zo = want ? (long long)(z - p.z_in0) * p.zs : 0ll;
        v[it] = zconv_keep4(*reinterpret_cast<const float4*>(p.src + zo + scol + c2), want);
    }
#pragma unroll
    for (int it = 0; it < TNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        const bool want = r < p.kz;
        tv[it] = zconv_keep4(*reinterpret_cast<const float4*>(p.taps + (long long)(want ? r : 0) * p.zs + tcol + c2), want);
    }
    // the sum's weights travel with the tile (requested in the epilogue they cost a block 1-4 k cycles of exposed latency: clock stamps)
    double2 wxa = make_double2(0.0, 0.0), wyb = make_double2(0.0, 0.0);
    if (p.sum_partial) { wxa = p.wx[kxb * NLZ + (tid & (NLZ - 1))]; wyb = p.wy[ky]; }
#pragma unroll
    for (int it = 0; it < ZNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        if (r < rows) *reinterpret_cast<float4*>(f + r * PITCH + c2) = v[it];
    }
#pragma unroll
    for (int it = 0; it < TNIT; ++it) {
        const int r = (tid / ZLPR) + it * ZRPI;
        if (r < gr) *reinterpret_cast<float4*>(g + r * NLZ + c2) = tv[it];
    }
    for (int r = TNIT * ZRPI + tid / ZLPR; r < gr; r += ZRPI) *reinterpret_cast<float4*>(g + r * NLZ + c2) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int line = lane & (NLZ - 1), grp = lane >> 4;
    if (p.sum_partial) {
        // exclusive prefix sums of every line's taps, gc[t] = g[0] + .. + g[t - 1] (t <= Kz): a thread per (line, four taps), the
        // sixteen segments' totals through LDS
        static_assert(ZT / NLZ == 16 && TNIT * ZRPI <= 64, "sixteen segments of four taps cover Kz <= 64");
        const int ln = tid & (NLZ - 1), seg = tid >> 4;
        float2* tot = gc + (size_t)(p.kz + 1) * NLZ;
        v2f loc[4], run = v2f{0.f, 0.f}, last = v2f{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 4 * seg + i;
            loc[i] = run;
            if (t < p.kz) { const float2 x = g[t * NLZ + ln]; last = v2f{x.x, x.y}; run += last; }
        }
        tot[seg * NLZ + ln] = make_float2(run.x, run.y);
        __syncthreads();
        v2f off = v2f{0.f, 0.f};
#pragma unroll
        for (int q = 0; q < ZT / NLZ - 1; ++q) {
            const float2 x = tot[q * NLZ + ln];
            if (q < seg) off += v2f{x.x, x.y};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 4 * seg + i;
            const v2f e = loc[i] + off;
            if (t < p.kz) gc[t * NLZ + ln] = make_float2(e.x, e.y);
            if (t == p.kz - 1) { const v2f all = e + last; gc[p.kz * NLZ + ln] = make_float2(all.x, all.y); }
        }
    }
    const int nout = (zn + S - 1) / S;                    // outputs of the tile: planes zc0 + S k
    const int nunits = S * (qp / ZJ);                     // unit i: phase i % S, taps S (q0 + t) + phase with q0 = (i / S) * ZJ
    for (int kb = 0; kb < nout; kb += 64) {
        const int nt = min(64, nout - kb), zu = (nt + 3) >> 2;   // this trip's outputs, dealt evenly over the four waves
        const int k0 = kb + wave * zu, kend = min(k0 + zu, kb + nt);
        if (k0 >= kend) continue;
        v2f acc[ZU];
#pragma unroll
        for (int u = 0; u < ZU; ++u) acc[u] = v2f{0.f, 0.f};
#pragma unroll 1
        for (int i = grp; i < nunits; i += 4) {
            const int rho = i % S, q0 = (i / S) * ZJ;
            // output u, tap q0 + t reads row base + S (u - t); base - S (ZJ - 1) >= 0 (padf), rows beyond the staged ones (outputs past
            // kend) lie inside the reserved region (p.frows) and feed nothing that is stored
            const int base = padf + p.kz - 1 - rho + S * (k0 - q0);
            const float2* __restrict__ fw = f + (base - S * (ZJ - 1)) * PITCH + line;
            const float2* __restrict__ gw = g + (S * q0 + rho) * NLZ + line;
            float2 w[ZU + ZJ - 1], gg[ZJ];
#pragma unroll
            for (int j = 0; j < ZU + ZJ - 1; ++j) w[j] = fw[j * S * PITCH];
#pragma unroll
            for (int t = 0; t < ZJ; ++t) gg[t] = gw[t * S * NLZ];
#pragma unroll
            for (int t = 0; t < ZJ; ++t) {
#pragma unroll
                for (int u = 0; u < ZU; ++u) {
                    const float2 x = w[u - t + ZJ - 1];
                    acc[u] = __builtin_elementwise_fma(v2f{gg[t].x, gg[t].x}, v2f{x.x, x.y}, acc[u]);
                    acc[u] = __builtin_elementwise_fma(v2f{-gg[t].y, gg[t].y}, v2f{x.y, x.x}, acc[u]);
                }
            }
        }
        // reduce-scatter over the four groups: group g ends up with the outputs 4 g .. 4 g + 3 of the wave.  v_permlane32_swap
        // exchanges the upper half of one register with the lower half of another, so a = acc[u], b = acc[8 + u] become (own a | b of
        // lane - 32) and (a of lane + 32 | own b): their sum is what the lower lanes keep of u and the upper lanes of 8 + u.
        // v_permlane16_swap does the same with the odd / even rows of 16 lanes.
        static_assert(ZU == 16, "four groups x four outputs");
        v2f h8[8], h4[4];
#pragma unroll
        for (int u = 0; u < 8; ++u) h8[u] = swap_add<32>(acc[u], acc[8 + u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) h4[u] = swap_add<16>(h8[u], h8[4 + u]);
        float2* d = p.dst + (long long)(zc0 + S * (k0 + 4 * grp)) * p.zs + dcol + line;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + 4 * grp + u < kend) *d = make_float2(h4[u].x, h4[u].y);
            d += p.zs * S;
        }
    }
    if (p.sum_partial) {
        __shared__ double red[ZT / 64];
        __syncthreads();                                  // the taps' prefix sums (wave 0)
            // a thread's <= 18 rows in single precision (as k_zconv adds its 16 outputs), then double; every row the same three LDS reads
        // (rows that every tap reaches weigh gc[Kz] - gc[0]), unrolled so that they are all in flight together
        const int ln = tid & (NLZ - 1);
        v2f s2 = v2f{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < ZNIT * ZRPI / (ZT / NLZ); ++i) {
            const int r = padf + (tid >> 4) + i * (ZT / NLZ);
            const bool in = r < rows;
            const int tlo = in ? max(0, padf + p.kz - 1 - r) : 0, thi = in ? min(p.kz, padf + p.kz - 1 + zn - r) : 0;   // taps [tlo, thi)
            float2 x = make_float2(0.f, 0.f);
            if (in) x = f[r * PITCH + ln];
            const float2 a = gc[thi * NLZ + ln], b = gc[tlo * NLZ + ln];
            const float wr = a.x - b.x, wi = a.y - b.y;
            s2 = __builtin_elementwise_fma(v2f{wr, wr}, v2f{x.x, x.y}, s2);
            s2 = __builtin_elementwise_fma(v2f{-wi, wi}, v2f{x.y, x.x}, s2);
        }
        const double sre = (double)s2.x, sim = (double)s2.y;
        const double wr = wxa.x * wyb.x - wxa.y * wyb.y, wi = wxa.x * wyb.y + wxa.y * wyb.x;
        double t = sre * wr - sim * wi;                   // Re( s * W )
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = t;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < ZT / 64; ++w) sum += red[w];
            p.sum_partial[tile] = sum;
        }
    }
}

struct PlanDesc {
    int len;
    int nrad;
    int rad[5];
};
static const PlanDesc kPlans[] = {
#define X(L, ...) {L, (int)(sizeof((int[]){__VA_ARGS__}) / sizeof(int)), {__VA_ARGS__}},
    MVSIM_FFT_SIZES(X)
#undef X
};

static const int kSizes[] = {
#define X(L, ...) L,
    MVSIM_FFT_SIZES(X)
#undef X
};

template <class K> static int set_lds(mvsim_ctx* ctx, K kernel, size_t bytes)
{
    return ensure_lds_attr(ctx, reinterpret_cast<const void*>(kernel), bytes);
}

static size_t zconv_lds(int zc, int kz)
{
    const int kzp = (kz + ZJ - 1) / ZJ * ZJ;
    return (zconv_frows(zc, kz, kzp) + (size_t)kzp * NLZ) * sizeof(float2);
}

// outputs per block along z: as many as fit a quarter of a CU's LDS (four resident blocks), at most Nz rounded up to ZU
static int zconv_chunk(int nz, int kz)
{
    int zc = (nz + ZU - 1) / ZU * ZU;
    const int kzp = (kz + ZJ - 1) / ZJ * ZJ;
    while (zc > ZU && (zconv_lds(zc, kz) > ZLDS_TARGET || (kzp - kz) + zc + kz - 1 > ZNIT * ZRPI)) zc -= ZU;
    // balance the chunks
    const int nchunks = (nz + zc - 1) / zc;
    const int even = ((nz + nchunks - 1) / nchunks + ZU - 1) / ZU * ZU;
    return even < zc ? even : zc;
}

// blocks (= partial sums) of the z pass for this geometry
static long long zconv_blocks(const ZConvArgs& a, int py, int cols = 0)
{
    return (long long)((a.nz + a.zc - 1) / a.zc) * ((cols > 0 ? cols : a.hxp) / NLZ) * py;
}

// cols > 0: only that many kx columns from the column the pointers of `a` start at (a kx panel)
static dim3 zconv_grid(ZConvArgs& a, int py, int cols, int env_exp)
{
    a.nch = (a.nz + a.zc - 1) / a.zc; a.nkxb = (cols > 0 ? cols : a.hxp) / NLZ; a.py = py;
    a.plain_order = (env_exp & 1) ? 1 : 0;
    const long long total = (long long)a.nch * a.nkxb * a.py;
    return dim3((unsigned)((total + 7) / 8 * 8));
}

static int launch_zconv(mvsim_ctx* ctx, const ZConvArgs& args, int py, int views = 1)
{
    hipStream_t s = ctx->stream;
    ZConvArgs a = args;
    const size_t lds = zconv_lds(a.zc, a.kz);
    dim3 grid = zconv_grid(a, py, 0, ctx->opt.exp);
    grid.y = (unsigned)views;
    MVSIM_TRY(set_lds(ctx, k_zconv, lds));
    hipLaunchKernelGGL(k_zconv, grid, dim3(ZT), lds, s, a);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// k_zconv_strided: planes per tile (a multiple of the stride; 0: this geometry keeps k_zconv), the LDS rows to reserve and the bytes
#ifndef MVSIM_ZLDS_S
#define MVSIM_ZLDS_S (52 * 1024)
#endif
static int zconv_strided_chunk(int nz, int kz, int s, bool force, int* frows_out, size_t* lds_out)
{
    if (s < 2 || s > 4 || kz > 64) return 0;
    const int nq = (kz + s - 1) / s, qp = (nq + ZJ - 1) / ZJ * ZJ, gr = s * qp, padf = gr - kz;
    if (kz - 1 - kz / 2 + padf > NZ_EXT) return 0;       // the flag bit string holds that many planes in front of plane 0
    auto geometry = [&](int zc, int* frows, size_t* lds) {
        // the last output a lane computes (not stores): trips of 64 outputs, four waves with zu each, 16 per lane
        const int nout = (zc + s - 1) / s;
        int kmax = 0;
        for (int kb = 0; kb < nout; kb += 64) {
            const int nt = std::min(64, nout - kb), zu = (nt + 3) / 4;
            kmax = std::max(kmax, kb + 3 * zu + ZU - 1);
        }
        const int staged = padf + zc + kz - 1;
        *frows = std::max(staged, padf + kz - 1 + s * kmax + 1);
        *lds = ((size_t)*frows * NLZ + (size_t)gr * NLZ + (size_t)(kz + 1 + ZT / NLZ) * NLZ) * sizeof(float2);
        return staged;
    };
    int zc = (nz + s - 1) / s * s, frows = 0;
    size_t lds = 0;
    while (zc > s && (geometry(zc, &frows, &lds) > ZNIT * ZRPI || lds > (size_t)MVSIM_ZLDS_S)) zc -= s;
    if (geometry(zc, &frows, &lds) > ZNIT * ZRPI || lds > (size_t)MVSIM_ZLDS_S) return 0;
    const int nchunks = (nz + zc - 1) / zc;
    // Both kernels take about the same time per TILE whatever the tile holds (memory latency and the fixed phases: ~15.5 ns here, 13 ns
    // + 0.17 ns per tap in k_zconv, over the sizes of profiles/r04_zstrided.txt); the padded tap rows of this kernel cost LDS rows, so a
    // volume can need more chunks here (512 planes, 31 taps, stride 3: three instead of two) -- then the plain kernel is the faster one.
    if (!force) {
        const int zc1 = zconv_chunk(nz, kz), nch1 = (nz + zc1 - 1) / zc1;
        if (nchunks * 15.5 >= nch1 * (13.0 + 0.17 * kz)) return 0;
    }
    const int even = ((nz + nchunks - 1) / nchunks + s - 1) / s * s;
    if (even < zc) { zc = even; geometry(zc, &frows, &lds); }
    *frows_out = frows; *lds_out = lds;
    return zc;
}

static int launch_zconv_strided(mvsim_ctx* ctx, const ZConvArgs& args, int py, size_t lds, int views = 1)
{
    hipStream_t s = ctx->stream;
    ZConvArgs a = args;
    dim3 grid = zconv_grid(a, py, 0, ctx->opt.exp);
    grid.y = (unsigned)views;
    switch (a.stride) {
    case 2: MVSIM_TRY(set_lds(ctx, k_zconv_strided<2>, lds)); hipLaunchKernelGGL(k_zconv_strided<2>, grid, dim3(ZT), lds, s, a); break;
    case 3: MVSIM_TRY(set_lds(ctx, k_zconv_strided<3>, lds)); hipLaunchKernelGGL(k_zconv_strided<3>, grid, dim3(ZT), lds, s, a); break;
    case 4: MVSIM_TRY(set_lds(ctx, k_zconv_strided<4>, lds)); hipLaunchKernelGGL(k_zconv_strided<4>, grid, dim3(ZT), lds, s, a); break;
    default: set_error("k_zconv_strided: stride %d", a.stride); return MVSIM_EINVAL;
    }
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

template <class PLAN>
static int launch_lines_t(mvsim_ctx* ctx, int mode, bool sparse, const LinesArgs& a, int tiles, int nouter)
{
    hipStream_t s = ctx->stream;
    using C = Cfg<PLAN::len>;
    dim3 grid(tiles, nouter), block(C::T);
#define MVSIM_LL(MODE_, SP_)                                                                 \
    do {                                                                                     \
        MVSIM_TRY(set_lds(ctx, k_fft_lines<PLAN, MODE_, SP_>, C::LDS));                           \
        hipLaunchKernelGGL((k_fft_lines<PLAN, MODE_, SP_>), grid, block, C::LDS, s, a);      \
    } while (0)
    if (mode == FWD && sparse) MVSIM_LL(FWD, true);
    else if (mode == FWD) MVSIM_LL(FWD, false);
    else if (mode == INV) MVSIM_LL(INV, false);
    else if (mode == CONVZ) MVSIM_LL(CONVZ, false);
    else MVSIM_LL(CONV, false);
#undef MVSIM_LL
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

template <class PLAN>
static int launch_r2c_t(mvsim_ctx* ctx, const float* src, const SrcMap& map, float2* dst, const float2* tw,
                        const float2* twx, int hxp, long long rows)
{
    using C = CfgX<PLAN::len>;
    hipStream_t s = ctx->stream;
    const long long blocks = (rows + C::NL - 1) / C::NL;
    MVSIM_TRY(set_lds(ctx, k_fft_x_r2c<PLAN>, C::LDS));
    hipLaunchKernelGGL((k_fft_x_r2c<PLAN>), dim3((unsigned)blocks), dim3(C::T), C::LDS, s, src, map, dst, tw, twx, hxp, rows);
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

template <class PLAN>
static int launch_c2r_t(mvsim_ctx* ctx, const float2* srcc, float* out, const float2* tw, const float2* twx, int hxp,
                        int py, int nx, int ny, long long rows, float scale, double* partial, int* nblocks, const C2RFuse* fuse,
                        const C2REmpty& em = C2REmpty{})
{
    using C = CfgX<PLAN::len>;
    const long long groups = (rows + C::NL - 1) / C::NL;
    const int blocks = (int)groups;
    *nblocks = blocks;
    hipStream_t s = ctx->stream;
    if (fuse) {
        const size_t lds = C::LDS + 32 + (size_t)C::NW * sizeof(P1Scratch);
        MVSIM_TRY(set_lds(ctx, k_fft_x_c2r<PLAN, true>, lds));
        hipLaunchKernelGGL((k_fft_x_c2r<PLAN, true>), dim3((unsigned)blocks), dim3(C::T), lds, s, srcc, out, tw, twx, hxp, py, nx, ny,
                           rows, scale, partial, *fuse, em);
    } else {
        MVSIM_TRY(set_lds(ctx, k_fft_x_c2r<PLAN, false>, C::LDS));
        hipLaunchKernelGGL((k_fft_x_c2r<PLAN, false>), dim3((unsigned)blocks), dim3(C::T), C::LDS, s, srcc, out, tw, twx, hxp, py, nx, ny,
                           rows, scale, partial, C2RFuse{}, em);
    }
    MVSIM_HIP(hipGetLastError());
    return MVSIM_OK;
}

// rows per block of pass E for this half length (the queue segments of the fused tail are per block)
template <class PLAN> static int c2r_rows_per_block_t() { return CfgX<PLAN::len>::NL; }
static int c2r_rows_per_block(int M)
{
    switch (M) {
#define X(LL, ...) case LL: return c2r_rows_per_block_t<Plan<LL, __VA_ARGS__>>();
        MVSIM_FFT_SIZES(X)
#undef X
    }
    return 0;
}

// the half-length plans k_fft_lines_split is instantiated for: 2 LH is a table length whose own tile leaves a CU room for one block only,
// and the 8-line tile of LH points fits twice
// (... and whose blocks are eight waves: measured, profiles/r06_split_ab.txt -- 2160 = 2 x 1080 and 2048 = 2 x 1024 gain
// 15-25 % per pass; 1792 = 2 x 896, four waves of two lines with 112 registers of rows per lane, loses in pass B what it gains in pass D).
// The same form would serve the lines of 1024 .. 1152 points for another reason -- their halves (512 .. 576 points) take SIXTEEN-column
// tiles: 128-byte rows in every load and store, two blocks per CU either way -- and was built and run for them (the kernel below takes
// 16-column half tiles as it stands: WIDE in the launcher; same tests, green).  It is NOT instantiated there: with two lines
// per wave the compiler does not fit the 72 registers of a lane's rows, their addresses and the pinned odd half into the 128 of two blocks
// per CU (104-228 bytes of scratch in every variant tried: all rows at once, groups of three with the even half written to LDS at once,
// one-reflection addressing), spills rows as they arrive -- i.e. waits for them one by one -- and pass B at 1024^3 takes 2.66 instead
// of 1.66 ms (profiles/r06_split_ab.txt).
template <int LH> constexpr bool split_half_ok()
{
    return Cfg<LH>::NL == 8 && Cfg<LH>::T == 512 && 2 * Cfg<LH>::LDS <= 160 * 1024 && 2 * Cfg<2 * LH>::LDS > 160 * 1024;
}

template <int LH, int... Rs>
static int launch_lines_split_t(mvsim_ctx* ctx, int mode, const LinesArgs& a, const float2* tw2, int tiles, int nouter)
{
    if constexpr (LH >= 1024 && LH <= 1120 && split_half_ok<LH>()) {
        using PLANH = Plan<LH, Rs...>;
        using C = Cfg<LH>;
        // `tiles` counts tiles of the whole length's width (8 columns); the half length's tiles may be 16 wide
        constexpr int WIDE = C::NL / Cfg<2 * LH>::NL;
        if (tiles % WIDE) { set_error("custom FFT: %d tiles of 8 columns do not pair up", tiles); return MVSIM_EINVAL; }
        dim3 grid(tiles / WIDE, nouter), block(C::T);
        if (mode == FWD) {
            MVSIM_TRY(set_lds(ctx, k_fft_lines_split<PLANH, FWD>, C::LDS));
            hipLaunchKernelGGL((k_fft_lines_split<PLANH, FWD>), grid, block, C::LDS, ctx->stream, a, tw2);
        } else {
            MVSIM_TRY(set_lds(ctx, k_fft_lines_split<PLANH, INV>, C::LDS));
            hipLaunchKernelGGL((k_fft_lines_split<PLANH, INV>), grid, block, C::LDS, ctx->stream, a, tw2);
        }
        MVSIM_HIP(hipGetLastError());
        return MVSIM_OK;
    }
    set_error("custom FFT: no split form for half length %d", LH);
    return MVSIM_EINVAL;
}

static bool lines_split_available(int L)
{
    if (L % 2) return false;
    switch (L / 2) {
#define X(LL, ...) case LL: if constexpr (LL >= 1024 && LL <= 1120) return split_half_ok<LL>(); else return false;
        MVSIM_FFT_SIZES(X)
#undef X
    }
    return false;
}

static bool lines_has_plan(int L)
{
    switch (L) {
#define X(LL, ...) case LL: return true;
        MVSIM_FFT_SIZES(X)
#undef X
    }
    return false;
}

static int launch_lines(mvsim_ctx* s, int L, int mode, bool sparse, const LinesArgs& a0, int tiles, int nouter)
{
    LinesArgs a = a0;
    a.pair_tiles = (s->opt.exp & 4) ? 0 : 1;              // exp bit 4: 8-column tiles in plain grid order (A/B, tools/ab_env.sh)
    // lines of one block per CU: the image's y passes as two half-length transforms (k_fft_lines_split) unless exp bit 8 asks for the
    // one-block form; the PSF's sparse passes and the z-pass forms keep k_fft_lines
    if ((mode == FWD || mode == INV) && !sparse && !a.dst_tile_major && !(s->opt.exp & 8) && lines_has_plan(L) && lines_split_available(L) &&
        tiles % 2 == 0 && (!a.src_mirror || (a.lmap.mode == 0 && a.lmap.b < a.lmap.n && a.lmap.a - a.lmap.n < a.lmap.n))) {
        const float2 *twh = nullptr, *tw2 = nullptr;
        MVSIM_TRY(ensure_twiddles(s, L / 2, 0, &twh));
        MVSIM_TRY(ensure_twiddles(s, L, 1, &tw2));
        a.tw = twh;
        switch (L / 2) {
#define X(LL, ...) case LL: return launch_lines_split_t<LL, __VA_ARGS__>(s, mode, a, tw2, tiles, nouter);
            MVSIM_FFT_SIZES(X)
#undef X
        }
    }
    switch (L) {
#define X(LL, ...) \
    case LL: return launch_lines_t<Plan<LL, __VA_ARGS__>>(s, mode, sparse, a, tiles, nouter);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("custom FFT: unsupported length %d", L);
    return MVSIM_EINVAL;
}

static int launch_r2c(mvsim_ctx* s, int M, const float* src, const SrcMap& map, float2* dst, const float2* tw,
                      const float2* twx, int hxp, long long rows)
{
    switch (M) {
#define X(LL, ...) \
    case LL: return launch_r2c_t<Plan<LL, __VA_ARGS__>>(s, src, map, dst, tw, twx, hxp, rows);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("custom FFT: unsupported half length %d", M);
    return MVSIM_EINVAL;
}

static int launch_c2r(mvsim_ctx* s, int M, const float2* srcc, float* out, const float2* tw, const float2* twx, int hxp,
                      int py, int nx, int ny, long long rows, float scale, double* partial, int* nblocks, const C2RFuse* fuse,
                        const C2REmpty& em = C2REmpty{})
{
    switch (M) {
#define X(LL, ...) \
    case LL: return launch_c2r_t<Plan<LL, __VA_ARGS__>>(s, srcc, out, tw, twx, hxp, py, nx, ny, rows, scale, partial, nblocks, fuse, em);
        MVSIM_FFT_SIZES(X)
#undef X
    }
    set_error("custom FFT: unsupported half length %d", M);
    return MVSIM_EINVAL;
}

template <int L> static constexpr int nl_of() { return Cfg<L>::NL; }
static int lines_per_tile(int L)
{
    switch (L) {
#define X(LL, ...) case LL: return nl_of<LL>();
        MVSIM_FFT_SIZES(X)
#undef X
    }
    return 16;
}

static int pick_size(int64_t need)
{
    for (int v : kSizes)
        if (v >= need) return v;
    return 0;
}

}  // namespace fft

// Padded sizes for the custom path, or false if some dimension has no supported size.
bool custom_fft_sizes(const int64_t dim[3], const int64_t kdim[3], int64_t P[3], const Options& opt)
{
    if (opt.rocfft) return false;
    for (int d = 0; d < 3; ++d) {
        const int64_t need = dim[d] + kdim[d] - 1;
        if (d == 0) {
            const int m = fft::pick_size((need + 1) / 2);
            if (!m) return false;
            P[0] = 2 * (int64_t)m;
        } else {
            const int v = fft::pick_size(need);
            if (!v) return false;
            P[d] = v;
        }
    }
    return true;
}

// kind 0: per-pass transform table of the plan for length L (see wpasses); kind 1: plain exp(-2 pi i k / L),
// k = 0..L (the real<->complex post/pre-processing of the x passes).
static int ensure_twiddles(mvsim_ctx* ctx, int L, int kind, const float2** out)
{
    const int key = L * 2 + kind;
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *out = reinterpret_cast<const float2*>(it->second); return MVSIM_OK; }
    std::vector<float2> h((size_t)L + 1, make_float2(0.f, 0.f));
    if (kind == 1) {
        for (int k = 0; k <= L; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)L;
            h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    } else {
        const fft::PlanDesc* pd = nullptr;
        for (const auto& d : fft::kPlans) if (d.len == L) pd = &d;
        if (!pd) { set_error("custom FFT: no plan for length %d", L); return MVSIM_EINVAL; }
        int P = 1, off = 0;
        for (int i = 0; i < pd->nrad; ++i) {
            const int R = pd->rad[i];
            if (P > 1) {
                for (int r = 1; r < R; ++r)
                    for (int k = 0; k < P; ++k) {
                        const double a = -2.0 * M_PI * (double)r * (double)k / ((double)P * (double)R);
                        h[(size_t)off + (size_t)(r - 1) * P + k] = make_float2((float)std::cos(a), (float)std::sin(a));
                    }
                off += (R - 1) * P;
            }
            P *= R;
        }
    }
    void* d = nullptr;
    MVSIM_HIP(hipMalloc(&d, h.size() * sizeof(float2)));
    // synchronous copy from a temporary: happens once per length per context
    hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); set_error("twiddle upload failed: %s", hipGetErrorString(e)); return MVSIM_EHIP; }
    ctx->twiddles[key] = d;
    *out = reinterpret_cast<const float2*>(d);
    return MVSIM_OK;
}

// Weights of the early sum (see ZConvArgs): w[k] = c(k) * SUM_{n < N} e^{+2 pi i k n / P}, k < len; c(k) = 1 unless `half`,
// where c = 1 for k = 0 and k = P/2, 2 for 0 < k < P/2 and 0 beyond (the half spectrum's padding columns).  Built on the
// host in double from an exact P-entry table of e^{2 pi i j / P} (the angle index k n mod P is integer arithmetic).
static int ensure_box_weights(mvsim_ctx* ctx, int N, int P, int len, bool half, const double2** out)
{
    char key[64];
    snprintf(key, sizeof(key), "%d/%d/%d/%d", N, P, len, half ? 1 : 0);
    auto it = ctx->box_weights.find(key);
    if (it != ctx->box_weights.end()) { *out = reinterpret_cast<const double2*>(it->second); return MVSIM_OK; }
    std::vector<double> ct((size_t)P), st((size_t)P);
    for (int j = 0; j < P; ++j) {
        const double a = 2.0 * M_PI * (double)j / (double)P;
        ct[j] = std::cos(a); st[j] = std::sin(a);
    }
    std::vector<double2> h((size_t)len, make_double2(0.0, 0.0));
    for (int k = 0; k < len; ++k) {
        if (half && k > P / 2) break;
        if (!half && k >= P) break;
        long double re = 0.0L, im = 0.0L;
        for (int n = 0; n < N; ++n) {
            const int j = (int)(((long long)k * n) % P);
            re += ct[j]; im += st[j];
        }
        const double c = half ? ((k == 0 || 2 * k == P) ? 1.0 : 2.0) : 1.0;
        h[k] = make_double2(c * (double)re, c * (double)im);
    }
    void* d = nullptr;
    MVSIM_HIP(hipMalloc(&d, h.size() * sizeof(double2)));
    hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); set_error("weight upload failed: %s", hipGetErrorString(e)); return MVSIM_EHIP; }
    ctx->box_weights[key] = d;
    *out = reinterpret_cast<const double2*>(d);
    return MVSIM_OK;
}

// Geometry of the fused tail: pass E blocks own queue segments that can hold every voxel of their rows.
static bool fused_tail_geometry(const int64_t dim[3], const int64_t kdim[3], int inc, bool con_wanted, const Options& opt,
                                long long* blocks, unsigned int* segcap)
{
    int64_t P[3];
    if (!custom_fft_sizes(dim, kdim, P, opt)) return false;
    const bool zdirect = opt.zpass == 2 ? false : kdim[2] <= 64;
    if (!zdirect || !opt.early_sum || (dim[0] & 3) != 0) return false;
    const int nr = fft::c2r_rows_per_block((int)(P[0] / 2));
    if (nr <= 0) return false;
    const long long nk = con_wanted ? dim[2] : (dim[2] - 1) / inc + 1;
    const long long rows = dim[1] * nk;
    if (rows * dim[0] >= (1ll << 32)) return false;               // work items carry the output position in 32 bits
    *blocks = (rows + nr - 1) / nr;
    *segcap = (unsigned int)(nr * dim[0]);
    return true;
}

size_t fused_tail_queue_bytes(const int64_t dim[3], const int64_t kdim[3], int inc, bool con_wanted, const Options& opt)
{
    long long blocks = 0;
    unsigned int segcap = 0;
    if (!fused_tail_geometry(dim, kdim, inc, con_wanted, opt, &blocks, &segcap)) return 0;
    const size_t counts = ((size_t)QCOUNT_WORDS * blocks * sizeof(unsigned int) + 255) & ~(size_t)255;
    return counts + (size_t)blocks * segcap * sizeof(PItem);
}

void custom_fft_release(mvsim_ctx* ctx)
{
    for (auto& kv : ctx->twiddles) (void)hipFree(kv.second);
    ctx->twiddles.clear();
    for (auto& kv : ctx->box_weights) (void)hipFree(kv.second);
    ctx->box_weights.clear();
    ctx->partials_z.release();
    ctx->cfft_f.release();
    ctx->cfft_g.release();
    ctx->cfft_g1.release();
    ctx->cfft_g2.release();
}

// {Px, Py, Pz, Hxp, direct z pass?} of the hand-written path for this volume / PSF (what the passes move through HBM)
bool custom_fft_geometry(const int64_t dim[3], const int64_t kdim[3], int64_t g[5], const Options& opt)
{
    using namespace fft;
    int64_t P[3];
    if (!custom_fft_sizes(dim, kdim, P, opt)) return false;
    const bool zdirect = opt.zpass == 2 ? false : kdim[2] <= 64;
    const int tile_y = lines_per_tile((int)P[1]), tile_z = lines_per_tile((int)P[2]);
    int tw_max = tile_y > tile_z ? tile_y : tile_z;
    if (zdirect) tw_max = tile_y > NLZ ? tile_y : NLZ;
    const int M = (int)(P[0] / 2);
    g[0] = P[0]; g[1] = P[1]; g[2] = zdirect ? dim[2] : P[2];
    g[3] = ((M + 1 + tw_max - 1) / tw_max) * tw_max;
    g[4] = zdirect ? 1 : 0;
    return true;
}

bool custom_fft_batchable(const mvsim_ctx* ctx, const int64_t dim[3], const int64_t kdim[3])
{
    using namespace fft;
    const Options& o = ctx->opt;
    int64_t P[3];
    if (!custom_fft_sizes(dim, kdim, P, o)) return false;
    const int ny = (int)dim[1], ky = (int)kdim[1], kz = (int)kdim[2];
    const bool zdirect = o.zpass == 2 ? false : kz <= 64;
    if (!zdirect || !o.early_sum || o.fuse_tail || o.zpass == 3) return false;
    if (o.zpass == 0 && kz >= MVSIM_ZINLINE_MIN_KZ && P[2] >= 512 && lines_per_tile((int)P[2]) == 16) return false;   // the inline FFT z pass
    return ny > 1 && ky / 2 < ny && ky - 1 - ky / 2 < ny;         // one reflection reaches every halo row (ymirror)
}

// Rotate (about x) + attenuate + pass A as one kernel (rotate_fft.hip) when the view's geometry allows it: the attenuated
// volume then never crosses HBM unless the caller asked for it.  *done = false: nothing was enqueued, the caller runs the
// separate kernels.  On success the context's spectrum buffer F holds what pass A would have written for `att`, and the
// convolution is told so through ConvTail::x_done.
// z_first / nzl: the planes of the rotated volume to compute (a z slab of a tiled view: its own planes and the halo the PSF reaches;
// nzl < 0: all of them).  F then holds those planes from index 0, as pass A leaves them for custom_fft_convolve_slab.
int rotate_attenuate_fftx(mvsim_ctx* ctx, const float* gt, float* rot_or_null, float* att_or_null, const int64_t dim[3],
                          const int64_t kdim[3], const Affine& inv, double delta, bool* done, const int** plane_nz, int z_first, int nzl)
{
    using namespace fft;
    *done = false;
    if (plane_nz) *plane_nz = nullptr;
    if (ctx->opt.fused_fftx == 0) return MVSIM_OK;
    const bool x_identity = inv.m[0] == 1.0 && inv.m[1] == 0.0 && inv.m[2] == 0.0 && inv.m[3] == 0.0 &&
                            inv.m[4] == 0.0 && inv.m[8] == 0.0;
    if (!x_identity) return MVSIM_OK;
    int64_t P[3];
    if (!custom_fft_sizes(dim, kdim, P, ctx->opt)) return MVSIM_OK;
    const int nx = (int)dim[0], ny = (int)dim[1], nz = (int)dim[2];
    if (nzl < 0) { z_first = 0; nzl = nz; }
    if (z_first < 0 || nzl < 1 || z_first + nzl > nz) return MVSIM_OK;
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    const bool zdirect = ctx->opt.zpass == 2 ? false : kz <= 64;
    // pass B must be reading the mirrored halo rows from their mirror images (pass A then transforms the Ny rows of a plane
    // only -- what this kernel produces), the padded x row must be one reflection deep, one block must span the row
    const bool ymirror = zdirect && ny > 1 && ky / 2 < ny && ky - 1 - ky / 2 < ny;
    if (!ymirror || nx > 1024 || nx < 64 || kx > nx || nx > ny) return MVSIM_OK;
    const int px = (int)P[0], py = (int)P[1], M = px / 2;
    if (!rot_fftx_has_plan(M)) return MVSIM_OK;
    // auto: from two waves per SIMD of columns up (below that the kernel is a handful of serial waves, like its parent)
    if (ctx->opt.fused_fftx == 2 && (int64_t)nx * nzl < 131072) return MVSIM_OK;
    const int tile_y = lines_per_tile(py);
    const int tw_max = tile_y > NLZ ? tile_y : NLZ;
    const int hxp = ((M + 1 + tw_max - 1) / tw_max) * tw_max;
    const int pyb = (py + ZB - 1) / ZB * ZB;
    const size_t cbytes = (size_t)hxp * pyb * nzl * sizeof(float2);    // the size custom_fft_convolve_slab reserves
    MVSIM_TRY(ctx->cfft_f.reserve(cbytes));
    MVSIM_TRY(ctx->cfft_g.reserve(cbytes));
    const float2 *tw_m, *tw_px;
    MVSIM_TRY(ensure_twiddles(ctx, M, 0, &tw_m));
    MVSIM_TRY(ensure_twiddles(ctx, px, 1, &tw_px));
    RotFftArgs a{};
    a.in = gt; a.rot_out = rot_or_null; a.att_out = att_or_null; a.dst = ctx->cfft_f.as<float2>();
    a.twg = tw_m; a.twx = tw_px;
    a.nx = nx; a.ny = ny; a.nz = nz; a.steps = nx;                     // Q1: attenuate3d walks dimension(0) steps along y
    a.z_first = z_first; a.nzl = nzl;
    a.hxp = hxp; a.py = py;
    a.halo_r = kx / 2; a.halo_l = kx - 1 - kx / 2;
    a.a = inv; a.delta = delta;
    // The flags cost a volume WITHOUT empty planes about 1.5 % of a view (a scalar load in front of every block's tile loads, the bit
    // strings of the z pass): the last flagged view says how many planes were empty (a page-locked word the device writes, read here
    // without waiting -- a hint, nothing depends on its being current), and while that was none only every sixteenth view carries
    // flags, to notice when the data change.  Results are the same with and without flags.
    bool want_flags = plane_nz && ctx->opt.skip_empty && nzl == nz;     // (a slab keeps the unflagged passes)
    if (want_flags) {
        if (!ctx->empty_hint) {
            MVSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->empty_hint), 4 * sizeof(int), hipHostMallocDefault));
            ctx->empty_hint[0] = -1; ctx->empty_hint[1] = -1; ctx->empty_hint[2] = 0; ctx->empty_hint[3] = 0;
        }
        const int hint = *reinterpret_cast<volatile int*>(ctx->empty_hint);
        if (hint == 0 && ctx->views_since_flags < 15) { want_flags = false; ctx->views_since_flags += 1; }
        else ctx->views_since_flags = 0;
    }
    if (want_flags) {
        MVSIM_TRY(ctx->plane_flags.reserve((size_t)(3 * nz + 1024) * sizeof(int)));   // flags, the dilated flags, the flag bit string
        a.plane_nz = ctx->plane_flags.as<int>();
        *plane_nz = a.plane_nz;
    }
    MVSIM_TRY(launch_rot_fftx(ctx, M, a, rot_or_null != nullptr || att_or_null != nullptr));
    *done = true;
    return MVSIM_OK;
}

int custom_fft_convolve(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                        const int64_t kdim[3], const int64_t P[3], float* out, ConvTail* tail)
{
    const SlabRange whole{0, (int)dim[2], 0, (int)dim[2]};
    return custom_fft_convolve_slab(ctx, img, dim, psf, kdim, P, whole, out, tail);
}

int custom_fft_convolve_slab(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                             const int64_t kdim[3], const int64_t P[3], const SlabRange& slab, float* out, ConvTail* tail)
{
    using namespace fft;
    const bool is_slab = slab.nz_in != (int)dim[2] || slab.nz_out != (int)dim[2];
    // stacked views (ConvTail::views = V > 1; the caller has asked custom_fft_batchable): `img` holds V attenuated volumes back to back,
    // `psf` V PSFs, `out` receives V convolved volumes (compact planes: V x nk), the context's scalar pairs 0 .. V-1 the sums and factors.
    // Passes A, B, D, E and the PSF's passes work on planes (rows) and simply see V x as many; the z pass carries the view in its grid.
    const int V = (tail && tail->views > 1) ? tail->views : 1;
    const int px = (int)P[0], py = (int)P[1], pz = (int)P[2];
    const int kx = (int)kdim[0], ky = (int)kdim[1], kz = (int)kdim[2];
    const int M = px / 2;
    const int tile_y = lines_per_tile(py), tile_z = lines_per_tile(pz);
    // z pass: direct convolution with the Kz taps (k_zconv) unless the PSF is deep or the FFT formulation is asked for
    const bool zdirect = ctx->opt.zpass == 2 ? false : kz <= 64;
    const bool early = zdirect && ctx->opt.early_sum;             // adjustImage's sum from pass C' instead of pass E
    // with the sum known before passes D and E, they only have to produce the planes extractSlices reads
    // (a z slab may ask for it too: its first output plane is then a multiple of the stride -- the caller's promise -- so that the planes
    // k * zstride of the SLAB are planes k' * zstride of the view)
    int zstride = (tail && early && tail->zstride > 1 && (!is_slab || slab.z_out0 % tail->zstride == 0)) ? tail->zstride : 1;
    // fused tail: pass E adjusts, extracts and samples (needs the sum first, whole rows of float4 groups, aligned outputs)
    long long fblocks = 0;
    unsigned int fsegcap = 0;
    bool fuse = tail && tail->want_fuse && early && !is_slab && tail->acq &&
                fused_tail_geometry(dim, kdim, tail->inc, tail->con_adj != nullptr, ctx->opt, &fblocks, &fsegcap) &&
                ((reinterpret_cast<uintptr_t>(tail->acq) | reinterpret_cast<uintptr_t>(tail->con_adj)) & 15) == 0;
    if (fuse) zstride = tail->con_adj ? 1 : tail->inc;
    if (tail) { tail->zstride = zstride; tail->fused = fuse; }
    // adjustImage's factor rides in the reduction of the sum when the caller described it (a view; not the stage operator)
    const long long corr_n = (tail && tail->corr_n > 0 && !is_slab) ? tail->corr_n : 0;
    const float corr_min = tail ? tail->min_value : 0.f, corr_target = tail ? tail->target_average : 1.f;
    if (tail) tail->corr_done = corr_n > 0;
    if (is_slab && !zdirect) {
        set_error("z-slab tiling needs the direct z pass (PSF depth %d > 64 or option fft_zpass=fft)", kz);
        return MVSIM_EINVAL;
    }
    if (V > 1 && (is_slab || !zdirect || !early || fuse || (tail && (tail->x_done || tail->plane_nz)))) {
        set_error("stacked views need the direct z pass with the early sum on whole views");
        return MVSIM_EINVAL;
    }
    const int nzs = slab.nz_in;                                   // planes the image spectrum holds when zdirect
    const int nzo = slab.nz_out;                                  // planes that leave the z pass
    int tw_max = tile_y > tile_z ? tile_y : tile_z;
    if (zdirect) tw_max = tile_y > NLZ ? tile_y : NLZ;
    const int hxp = ((M + 1 + tw_max - 1) / tw_max) * tw_max;
    const int pyb = (py + ZB - 1) / ZB * ZB;                        // rows of a plane in the z-blocked layout
    const int nkv = (nzo - 1) / zstride + 1;                        // planes per view that leave passes D and E
    const int nzp = V > 1 ? nkv * zstride : nzo;                    // plane pitch of a view in the z pass's output: the planes k * zstride of
                                                                    // every view at multiples of zstride of the stacked index (pass D's stride)
    const size_t cbytes = (size_t)hxp * (zdirect ? pyb : py) * (zdirect ? (size_t)V * std::max(nzs, nzp) : (size_t)pz) * sizeof(float2);
    const long long rows_out_early = (long long)dim[1] * nzo * V;
    MVSIM_TRY(ctx->cfft_f.reserve(cbytes));
    MVSIM_TRY(ctx->cfft_g.reserve(cbytes));
    // compact PSF intermediates: G1 [kz][ky][hxp] (x transformed), G2 [kz][py][hxp] (x,y transformed)
    MVSIM_TRY(ctx->cfft_g1.reserve((size_t)V * hxp * ky * kz * sizeof(float2)));
    MVSIM_TRY(ctx->cfft_g2.reserve((size_t)V * hxp * pyb * kz * sizeof(float2)));
    MVSIM_TRY(ctx->partials.reserve(PARTIALS_BYTES));
    MVSIM_TRY(ctx->partials_e.reserve((size_t)((rows_out_early + 3) / 4 + 16) * sizeof(double)));
    double* scal = scal_of(ctx);

    // the z pass on the unpadded spectrum as FFT -> product -> inverse FFT with the PSF's z spectrum computed per tile (CONVZ): from
    // MVSIM_ZINLINE_MIN_KZ taps up the Kz-tap direct convolution is bound by its FMAs and this form by HBM (profiles/r04_zpass_sweep.txt)
    // Measured, whole views with 31 x 31 x Kz PSFs (profiles/r04_zpass_sweep.txt): 2048 x 2048 x 512 (padded z length 576: tiles of 16
    // lines, 128-byte rows) -- pass C 8.1 ms whatever Kz against 8.1 / 9.0 / 10.2 ms for the direct form at Kz = 41 / 51 / 63; 1024^3
    // (length 1120: tiles of 8 lines, 64-byte rows, one line per wave) -- 5.6 ms against 3.9 / 4.4 / 4.9 ms.  Hence auto = deep PSFs on
    // z lengths of 512 .. 576 (the sizes measured to win); everything else keeps the direct form unless asked.
    // compact planes: the direct form computes 1 / zstride of the planes (k_zconv_strided) and is then ahead of the inline FFT at every depth
    int zs_frows = 0;
    size_t zs_lds = 0;
    const int zs_chunk = (zdirect && zstride > 1 && ctx->opt.zconv_strided && ctx->opt.zpass != 3)
                             ? zconv_strided_chunk(slab.nz_out, kz, zstride, (ctx->opt.exp & 2) != 0, &zs_frows, &zs_lds) : 0;
    const bool zinline = zdirect && !is_slab && zs_chunk == 0 && V == 1 &&
                         (ctx->opt.zpass == 3 || (ctx->opt.zpass == 0 && kz >= MVSIM_ZINLINE_MIN_KZ && pz >= 512 && lines_per_tile(pz) == 16));
    const float2 *tw_m, *tw_px, *tw_py, *tw_pz;
    MVSIM_TRY(ensure_twiddles(ctx, M, 0, &tw_m));
    MVSIM_TRY(ensure_twiddles(ctx, px, 1, &tw_px));
    MVSIM_TRY(ensure_twiddles(ctx, py, 0, &tw_py));
    tw_pz = nullptr;
    if (!zdirect || zinline) MVSIM_TRY(ensure_twiddles(ctx, pz, 0, &tw_pz));

    float2* F = ctx->cfft_f.as<float2>();
    float2* G = ctx->cfft_g.as<float2>();
    float2* G1 = ctx->cfft_g1.as<float2>();
    float2* G2 = ctx->cfft_g2.as<float2>();
    hipStream_t s = ctx->stream;
    const long long rows_all = (long long)py * pz;
    const long long plane = (long long)hxp * py;
    const DimMap ident_none = DimMap{0, 0, 0, 0, 1, 0};

    // ---- PSF spectrum G.  The embedded kernel is zero outside Kx*Ky*Kz taps, so the x pass runs on the
    //      Ky*Kz non-zero rows only, the y pass on the Kz non-zero planes only (sparse loads), and the z
    //      pass expands Kz planes to the full spectrum.
    // It depends on nothing the image passes A and B produce and is a handful of small launches, so it runs on the
    // context's side stream beside them (fork here, join before the z pass reads G2 / G).
    // Only where it pays: the fork and the join cost ~10 us of cross-stream dependency, more than the spectrum of a small view.
    const bool side = ctx->opt.psf_overlap && (int64_t)dim[0] * dim[1] * dim[2] >= (int64_t)1 << 24;
    ctx->psf_on_side = side;
    if (side) {
        if (!ctx->side_stream) {
            MVSIM_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            MVSIM_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        MVSIM_HIP(hipEventRecord(ctx->ev_fork, s));
        MVSIM_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        ctx->stream = ctx->side_stream;                 // the launch helpers enqueue on ctx->stream
    }
    int psf_rc = MVSIM_OK;
    ev_begin(ctx, ST_PSF);
    do {
        SrcMap m{};
        m.x = DimMap{kx, px, kx - kx / 2, kx / 2, 1, kx / 2};   // embed along x with wrap-around
        m.y = DimMap{ky, ky, ky, 0, 1, 0};                       // compact: identity
        m.z = DimMap{kz * V, kz * V, kz * V, 0, 1, 0};             // (stacked PSFs: V x Kz planes of taps)
        if ((psf_rc = launch_r2c(ctx, M, psf, m, G1, tw_m, tw_px, hxp, (long long)ky * kz * V)) != MVSIM_OK) break;
        LinesArgs a{};
        a.src = G1; a.dst = G2; a.spec = nullptr; a.tw = tw_py;
        a.src_es = hxp; a.src_outer = (long long)hxp * ky;          // per kz plane
        a.dst_es = hxp; a.dst_outer = plane;
        if (zdirect) { a.dst_outer = (long long)ZB * hxp; a.dst_blk = (long long)kz * V * ZB * hxp; }   // the z pass reads its taps z-blocked
        a.lmap = DimMap{ky, py, ky - ky / 2, ky / 2, 1, ky / 2};
        if ((psf_rc = launch_lines(ctx, py, FWD, true, a, hxp / tile_y, kz * V)) != MVSIM_OK) break;
        if (!zdirect) {
            LinesArgs c{};
            c.src = G2; c.dst = G; c.spec = nullptr; c.tw = tw_pz;
            c.src_es = plane; c.src_outer = hxp;                         // per ky row
            c.dst_es = plane; c.dst_outer = hxp;
            c.lmap = DimMap{kz, pz, kz - kz / 2, kz / 2, 1, kz / 2};
            c.dst_tile_major = 1;                                         // consumed line by line in pass C
            psf_rc = launch_lines(ctx, pz, FWD, true, c, hxp / tile_z, py);
        }
    } while (false);
    ev_end(ctx, ST_PSF);
    if (side) {
        // always joined, also after a failed launch: a stream capture must not end with the side stream dangling
        ctx->stream = s;
        MVSIM_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
    }
    MVSIM_TRY(psf_rc);

    // ---- image: A, B, C (with product), D, E
    ev_begin(ctx, ST_CONVOLVE);
    {
        SrcMap m{};
        const int n[3] = {(int)dim[0], (int)dim[1], (int)dim[2]};
        const int Pd[3] = {px, py, pz};
        DimMap* dm[3] = {&m.x, &m.y, &m.z};
        for (int d = 0; d < 3; ++d) {
            const int c = (int)(kdim[d] / 2);
            const int left = (int)(kdim[d] - 1 - kdim[d] / 2);
            *dm[d] = DimMap{n[d], Pd[d], n[d] + c, left, 0, 0};
        }
        if (zdirect) m.z = DimMap{nzs * V, nzs * V, nzs * V, 0, 0, 0};   // no z padding: k_zconv mirrors through an index map
        // direct z pass: the mirrored halo rows along y are copies of rows pass A transforms anyway, so it transforms the Ny
        // rows of a plane only and pass B reads the halo positions from their mirror images (same bytes, from L2)
        SrcMap ma = m;
        const bool ymirror = zdirect && m.y.n > 1 && m.y.a - m.y.n < m.y.n && m.y.b < m.y.n;   // one reflection reaches every halo row
        if (ymirror) { ma.y = DimMap{m.y.n, m.y.P, m.y.n, 0, 0, 0}; ma.enum_y = m.y.n; }   // and visits no other row
        if (tail && tail->x_done) {
            // the fused rotate + attenuate + x transform has left the spectrum of the attenuated rows in F already
            // (a slab's fused kernel has computed its nz_in input planes -- the slab and its halo -- and left them in F from plane 0)
            if (!(zdirect && ymirror) || V != 1) { set_error("x_done without the geometry of the fused x transform"); return MVSIM_EINVAL; }
        } else {
            ev_begin(ctx, ST_PASS_A);
            MVSIM_TRY(launch_r2c(ctx, M, img, ma, F, tw_m, tw_px, hxp, zdirect ? (long long)(ymirror ? m.y.n : py) * nzs * V : rows_all));
            ev_end(ctx, ST_PASS_A);
        }
        // zero gap of the padded volume: y in [Ny + cy, Py - lefty), z in [Nz + cz, Pz - leftz).  Pass A does not
        // transform (or write) rows there, pass B skips the gap planes and does not load gap rows, pass C does
        // not load gap planes.
        const int ygap_lo = m.y.a, ygap_hi = py - m.y.b, zgap_lo = m.z.a, zgap_hi = pz - m.z.b;
        LinesArgs b{};
        b.src = F; b.dst = F; b.tw = tw_py; b.src_es = b.dst_es = hxp; b.src_outer = b.dst_outer = plane;
        b.lmap = ident_none;
        if (ymirror) { b.lmap = m.y; b.src_mirror = 1; }
        b.gap_lo = ygap_lo; b.gap_hi = ygap_hi;
        b.outer_skip_lo = zgap_lo; b.outer_skip_len = zgap_hi > zgap_lo ? zgap_hi - zgap_lo : 0;
        if (zdirect) {
            // out of place into the z-blocked layout the z pass reads and writes: G[(ky >> ZBS)][z][ky & (ZB-1)][kx]
            b.outer_skip_lo = 1 << 30; b.outer_skip_len = 0;
            b.dst = G; b.dst_outer = (long long)ZB * hxp; b.dst_blk = (long long)nzs * V * ZB * hxp;
        }
        float2* Fz = F;                                               // where passes D and E find the z-convolved spectrum
        const int* em_flags = nullptr;                                // planes passes D and E skip (see pnz below)
        const int nzd = zdirect ? nzo : (int)dim[2];                  // planes z >= Nz are never read
        const int nk = ((nzd - 1) / zstride + 1) * V;                 // planes 0, zstride, 2 zstride, ... (of every stacked view)
        {
        // planes the fused rotate kernel found empty: B skips them, C' does not load them and skips tiles made of nothing else, D and E
        // skip the planes whose taps reach nothing but empty planes (exact: their spectra are zero) -- a specimen in empty space
        const int* pnz = (tail && tail->x_done && tail->plane_nz && zdirect && !zinline && !is_slab && !fuse)
                             ? tail->plane_nz : nullptr;
        const int* pnz_dil = nullptr;
        const unsigned int* pnz_bits = nullptr;
        if (pnz) {
            // ctx->plane_flags = [flags (nz)][dilated (nz)][bit string (nwords)] (rotate_attenuate_fftx reserves all three)
            int* base = ctx->plane_flags.as<int>();
            const int nwords = ((int)dim[2] + 2 * NZ_EXT + 64 + 31) / 32 + ZNIT + 2;
            // one block, ~9 us: beside pass B on the side stream when there is one (pass B reads the raw flags; the bit string and the
            // dilated flags are for passes C', D and E, behind the join)
            hipLaunchKernelGGL(k_plane_flags_finish, dim3(1), dim3(1024), 0, side ? ctx->side_stream : s, pnz, (int)dim[2], kz, kz / 2,
                               reinterpret_cast<unsigned int*>(base + 2 * dim[2]), nwords, base + dim[2], ctx->empty_hint);
            MVSIM_HIP(hipGetLastError());
            if (side) MVSIM_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
            pnz_dil = base + dim[2];
            pnz_bits = reinterpret_cast<const unsigned int*>(base + 2 * dim[2]);
            b.nzflags = pnz; b.nz_stride = 1;
        }
        ev_begin(ctx, ST_PASS_B);
        MVSIM_TRY(launch_lines(ctx, py, FWD, false, b, hxp / tile_y, zdirect ? nzs * V : pz - b.outer_skip_len));
        ev_end(ctx, ST_PASS_B);
        b.lmap = ident_none; b.src_mirror = 0;
        if (side) MVSIM_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));    // the z pass reads the PSF spectrum
        ev_begin(ctx, ST_PASS_C);
        b.gap_lo = b.gap_hi = 0; b.outer_skip_lo = 1 << 30; b.outer_skip_len = 0;
        if (zinline) {
            LinesArgs c{};
            c.src = G; c.dst = F; c.tw = tw_pz;
            c.src_es = c.dst_es = (long long)ZB * hxp; c.src_outer = c.dst_outer = hxp;
            c.src_oblk = (long long)nzs * ZB * hxp; c.dst_oblk = (long long)nzo * ZB * hxp;
            c.lmap = DimMap{(int)dim[2], pz, (int)dim[2] + kz / 2, kz - 1 - kz / 2, 0, 0};
            c.src_mirror = 1;
            c.gap_lo = (int)dim[2] + kz / 2; c.gap_hi = pz - (kz - 1 - kz / 2);
            c.outer_skip_lo = 1 << 30;
            c.store_limit = (int)dim[2];
            c.taps = G2; c.taps_es = (long long)ZB * hxp; c.taps_outer = hxp; c.taps_oblk = (long long)kz * ZB * hxp;
            c.pmap = DimMap{kz, pz, kz - kz / 2, kz / 2, 1, kz / 2};
            const float scale_f = (float)(0.25 / ((double)px * (double)py * (double)pz));
            const long long zblocks = (long long)(hxp / tile_z) * py;
            if (early) {
                MVSIM_TRY(ensure_box_weights(ctx, (int)dim[0], px, hxp, true, &c.wx));
                MVSIM_TRY(ensure_box_weights(ctx, (int)dim[1], py, py, false, &c.wy));
                MVSIM_TRY(ctx->partials_z.reserve((size_t)zblocks * sizeof(double)));
                c.sum_partial = ctx->partials_z.as<double>();
            }
            MVSIM_TRY(launch_lines(ctx, pz, CONVZ, false, c, hxp / tile_z, py));
            if (early) {
                hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, s, c.sum_partial, zblocks, scal, (double)scale_f,
                                   corr_n, corr_min, corr_target);
                MVSIM_HIP(hipGetLastError());
            }
            Fz = G;
        } else if (zdirect) {
            ZConvArgs z{};
            z.src = G; z.dst = F; z.taps = G2; z.hxp = hxp; z.nz = nzo; z.kz = kz; z.c = kz / 2;
            z.zs = (long long)ZB * hxp; z.src_blk = (long long)nzs * V * ZB * hxp; z.dst_blk = (long long)nzp * V * ZB * hxp;
            z.taps_blk = (long long)kz * V * ZB * hxp;
            z.src_view = (long long)nzs * z.zs; z.dst_view = (long long)nzp * z.zs; z.taps_view = (long long)kz * z.zs;
            z.nz_global = (int)dim[2]; z.z_in0 = slab.z_in0; z.z_out0 = slab.z_out0;
            z.zc = zs_chunk > 0 ? zs_chunk : zconv_chunk(nzo, kz);
            z.stride = zs_chunk > 0 ? zstride : 1; z.frows = zs_frows;
            z.nzbits = pnz_bits;
            const float scale_f = (float)(0.25 / ((double)px * (double)py));
            long long zblocks = 0;
            if (early) {
                MVSIM_TRY(ensure_box_weights(ctx, (int)dim[0], px, hxp, true, &z.wx));
                MVSIM_TRY(ensure_box_weights(ctx, (int)dim[1], py, py, false, &z.wy));
                zblocks = zconv_blocks(z, py);
                MVSIM_TRY(ctx->partials_z.reserve((size_t)zblocks * V * sizeof(double)));
                z.sum_partial = ctx->partials_z.as<double>();
            }
            if (zs_chunk > 0) MVSIM_TRY(launch_zconv_strided(ctx, z, py, zs_lds, V));
            else MVSIM_TRY(launch_zconv(ctx, z, py, V));
            if (early) {
                // same factor pass E applies to every voxel (a float), so that the two sums estimate the same quantity
                hipLaunchKernelGGL(k_reduce_partials, dim3(V), dim3(1024), 0, s, z.sum_partial, zblocks, scal, (double)scale_f,
                                   corr_n, corr_min, corr_target);
                MVSIM_HIP(hipGetLastError());
            }
            Fz = G;                                                   // pass D: F (z-blocked) -> G (plane-major), out of place
        } else {
        LinesArgs c{};
        c.src = F; c.dst = F; c.spec = G; c.tw = tw_pz;
        c.src_es = c.dst_es = c.spec_es = plane; c.src_outer = c.dst_outer = c.spec_outer = hxp;
        c.lmap = ident_none;
        c.gap_lo = zgap_lo; c.gap_hi = zgap_hi;
        c.outer_skip_lo = 1 << 30;
        c.store_limit = (int)dim[2];                                  // pass D only reads planes z < Nz
        MVSIM_TRY(launch_lines(ctx, pz, CONV, false, c, hxp / tile_z, py));
        }
        ev_end(ctx, ST_PASS_C);
        b.src = b.dst = Fz;
        b.tw = tw_py;
        b.store_limit = (int)dim[1];                                  // pass E only reads rows y < Ny
        b.src_outer = b.dst_outer = plane * zstride;
        b.dst_blk = 0;
        if (zdirect) { b.src = F; b.src_outer = (long long)ZB * hxp * zstride; b.src_blk = (long long)nzp * V * ZB * hxp; }
        b.nzflags = nullptr;
        if (pnz) { b.nzflags = pnz_dil; b.nz_stride = zstride; em_flags = pnz_dil; }
        ev_begin(ctx, ST_PASS_D);
        MVSIM_TRY(launch_lines(ctx, py, INV, false, b, hxp / tile_y, nk));
        ev_end(ctx, ST_PASS_D);
        b.nzflags = nullptr;
        }
        C2RFuse fz{};
        if (fuse) {
            // adjustImage's factor must exist before pass E runs: (target - min) / (sum / n) from the early sum
            if (!tail->corr_done) {
                ev_begin(ctx, ST_ADJUST);
                MVSIM_TRY(launch_adjust_corr(s, scal, (int64_t)dim[0] * dim[1] * dim[2], tail->min_value, tail->target_average));
                ev_end(ctx, ST_ADJUST);
            }
            fz.scal = scal; fz.min_value = tail->min_value; fz.con = tail->con_adj; fz.acq = tail->acq;
            fz.acq_every = tail->con_adj ? tail->inc : 1; fz.idx_zstride = zstride; fz.noise = tail->noise ? 1 : 0;
            fz.mul = tail->mul; fz.k0 = (uint32_t)tail->seed; fz.k1 = (uint32_t)(tail->seed >> 32); fz.stream = tail->stream;
            if (tail->noise) {
                const size_t counts = ((size_t)QCOUNT_WORDS * fblocks * sizeof(unsigned int) + 255) & ~(size_t)255;
                if (ctx->pqueue.bytes < counts + (size_t)fblocks * fsegcap * sizeof(PItem)) {
                    set_error("fused tail: queue workspace not reserved");
                    return MVSIM_EINVAL;
                }
                fz.qcount = ctx->pqueue.as<unsigned int>();
                fz.queue = reinterpret_cast<PItem*>(ctx->pqueue.as<char>() + counts);
                fz.segcap = fsegcap;
            }
        }
        ev_begin(ctx, ST_PASS_E);
        // both half spectra carry the factor 2 left in by pass A (see k_fft_x_r2c): 2 * 2 = 4
        const float scale = (float)(0.25 / ((double)px * (double)py * ((zdirect && !zinline) ? 1.0 : (double)pz)));
        int nblk = 0;
        MVSIM_TRY(launch_c2r(ctx, M, Fz, out, tw_m, tw_px, hxp, py * zstride, (int)dim[0], (int)dim[1], (long long)dim[1] * nk, scale,
                             early ? nullptr : ctx->partials_e.as<double>(), &nblk, fuse ? &fz : nullptr,
                             C2REmpty{em_flags, zstride}));
        if (fuse && nblk != (int)fblocks) { set_error("fused tail: block count mismatch"); return MVSIM_EINVAL; }
        if (!early) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, s, ctx->partials_e.as<double>(), (long long)nblk, scal, 1.0,
                                       corr_n, corr_min, corr_target);
        MVSIM_HIP(hipGetLastError());
        ev_end(ctx, ST_PASS_E);
        if (fuse && tail->noise) {
            ev_end(ctx, ST_CONVOLVE);
            ev_begin(ctx, ST_EXTRACT);
            MVSIM_TRY(launch_poisson_resolve(s, tail->acq, fz.queue, fz.qcount, (int)fblocks, fsegcap, tail->mul, tail->seed, tail->stream,
                                             (long long)dim[0] * dim[1], fz.acq_every * fz.idx_zstride, 0));
            ev_end(ctx, ST_EXTRACT);
            return MVSIM_OK;
        }
    }
    ev_end(ctx, ST_CONVOLVE);
    return MVSIM_OK;
}

}  // namespace mvsim

#if defined(MVSIM_DEV_ATTRIBUTION) && defined(MVSIM_EXP_LINES_STAMPS)
// attribution build only: the stamps of blocks [0, nblocks) of mode `inv` (see g_line_stamps), 8 words per block
extern "C" int mvsim_dev_read_line_stamps(int inv, unsigned long long* out, size_t nblocks)
{
    if (nblocks > (size_t)mvsim::fft::STAMP_BLOCKS) nblocks = (size_t)mvsim::fft::STAMP_BLOCKS;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mvsim::fft::g_line_stamps), nblocks * 8 * sizeof(unsigned long long),
                                    (size_t)(inv ? 1 : 0) * mvsim::fft::STAMP_BLOCKS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif
