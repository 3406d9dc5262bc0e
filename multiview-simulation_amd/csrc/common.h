// Internal declarations shared by the libmvsim translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <roctracer/roctx.h>

#include <array>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/mvsim.h"

// Attribution switches (tools/attribute_flags.sh, attribute_valu.sh, attribution_run.sh: builds that compile a part of a kernel's work
// OUT -- and with it the correctness of the results -- to see what that part costs; the figures are in profiles/r04_attribution.txt).
// They exist only in a build that defines MVSIM_DEV_ATTRIBUTION, which multiview-simulation_amd/build.py never does: in the product
// build none of the names below can be defined, whatever MVSIM_EXTRA_CFLAGS carries.
#ifndef MVSIM_DEV_ATTRIBUTION
#undef MVSIM_EXP_ROTFFT_NOFFT
#undef MVSIM_EXP_ROTFFT_NOBLEND
#undef MVSIM_EXP_NOEXACT
#undef MVSIM_EXP_NOPHILOX
#undef MVSIM_EXP_NOSMALLPUSH
#undef MVSIM_EXP_NOBRIGHT
#undef MVSIM_P1_F64
#undef MVSIM_EXP_NLZ
#undef MVSIM_EXP_LINES_STAMPS
#endif

namespace mvsim {

void set_error(const char* fmt, ...);

#define MVSIM_HIP(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            mvsim::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                             __LINE__);                                                         \
            return (e_ == hipErrorOutOfMemory) ? MVSIM_ENOMEM : MVSIM_EHIP;                     \
        }                                                                                       \
    } while (0)

#define MVSIM_FFT(expr)                                                                         \
    do {                                                                                        \
        rocfft_status s_ = (expr);                                                              \
        if (s_ != rocfft_status_success) {                                                      \
            mvsim::set_error("%s failed: rocfft status %d (%s:%d)", #expr, (int)s_, __FILE__,   \
                             __LINE__);                                                         \
            return MVSIM_EFFT;                                                                  \
        }                                                                                       \
    } while (0)

#define MVSIM_TRY(expr)                                                                         \
    do {                                                                                        \
        int rc_ = (expr);                                                                       \
        if (rc_ != MVSIM_OK) return rc_;                                                        \
    } while (0)

#define MVSIM_CHECK_ARG(cond, msg)                                                              \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            mvsim::set_error("invalid argument: %s", msg);                                      \
            return MVSIM_EINVAL;                                                                \
        }                                                                                       \
    } while (0)

// Grow-only device buffer (steady state: no hipMalloc on the hot path).
struct DevBuf {
    void*  p     = nullptr;
    size_t bytes = 0;
    // allocation epoch of the owning context (null for buffers that belong to no context): bumped whenever this buffer
    // moves or goes away, so that the context's captured view graphs -- which hold raw workspace addresses -- are dropped.
    // Per context, not per process: another thread's context growing a workspace must not cost this one its graphs.
    unsigned long long* epoch = nullptr;
    int    reserve(size_t need);
    void   release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Affine {
    double m[12];
};

struct FftPlan {
    int64_t      P[3]      = {0, 0, 0};
    rocfft_plan  fwd       = nullptr;   // real P^3 -> complex (Px/2+1) Py Pz
    rocfft_plan  inv       = nullptr;   // complex -> real
    rocfft_execution_info info = nullptr;
    size_t       work_bytes = 0;
};

// Small ring of pinned host buffers for asynchronous H2D copies of PSFs (the caller's buffer may be
// pageable and may be reused as soon as the call returns).
struct PinnedRing {
    static constexpr int SLOTS = 4;
    void*      p[SLOTS]     = {};
    size_t     bytes[SLOTS] = {};
    hipEvent_t ev[SLOTS]    = {};
    bool       busy[SLOTS]  = {};
    int        next         = 0;
    int  acquire(size_t need, int* slot);   // waits for the slot's previous copy if still in flight
    void release_all();
};

// Run-time switches of a context.  Defaults come from the environment, read ONCE per process (env_options); the tests
// reach every code path through mvsim_set_option.  Nothing on a launch path calls getenv.
struct Options {
    int     zpass = 0;                 // z pass of the hand-written convolution: 0 auto, 1 direct Kz-tap convolution (Kz <= 64), 2 FFT on a
                                       // z-padded spectrum, 3 FFT inline: no z padding, mirrored planes through the index map, the PSF's z
                                       // spectrum computed per tile (k_fft_lines<CONVZ>); auto: inline from MVSIM_ZINLINE_MIN_KZ taps up, else direct
    bool    rocfft = false;            // library fallback instead of the hand-written passes
    int     fused_rotate = 3;          // rotate+attenuate as one kernel when the rotation is about x: 0 off, 1 row geometry
                                       // shared through LDS, 2 recomputed per lane (kept for A/B runs), 3 auto (production):
                                       // 1 when the view has >= 2 waves per SIMD of columns to walk or the separate rotate
                                       // cannot use 16-byte rows (Nx % 4 != 0), else 0 -- one lane walks
                                       // one (x, z) column, so a 128^3 view is 256 waves of serial latency (measured 30 us
                                       // against 20 us for the two kernels; break-even at 256^3)
    int     fused_fftx = 2;            // rotate + attenuate + the convolution's x transform as ONE kernel in the per-view pipeline
                                       // (rotate_fft.hip; `att` crosses HBM only when requested): 0 off, 1 whenever the geometry
                                       // allows, 2 auto (production): from 131072 columns up, like fused_rotate
    bool    attenuate_scan = false;    // attenuate3d (stage operator) as a wavefront prefix scan along y: re-associates the
                                       // fp64 products (float outputs differ from the serial walk by one ulp on < 1e-6 of the voxels)
    int     poisson_queue = 1;         // 1: two-launch Poisson (streaming kernel with wave-level compaction + work-queue
                                       // resolver, production), 0: one kernel
    int     poisson_queue_share = 16;  // sixteenths of a block's voxels its queue segment holds.  16 (default): every voxel, 16 B per acquired
                                       // voxel (2.25 GiB at 512^3, 34 GB at 2048^2 x 512), nothing is ever refused and phase 1 appends without
                                       // looking.  1..15: smaller segments; the appends check for room (+0.03 ms per 512^3 view) and what a full
                                       // segment refuses is sampled where it stands by a third kernel, same counts (k_poisson_refused).  0 =
                                       // "auto": 16 for queues of up to 64 MiB, else 5 (0.70 GiB at 512^3) and growing to what the context's
                                       // views turn out to need (the phantom: 3, a volume without an empty voxel: 12)
    bool    early_sum = true;          // adjustImage's sum from the spectrum side (pass C epilogue) so that pass E can adjust
    int     exp = 0;                   // experiment bits for A/B runs on one box (tools/): 1 = z pass tiles in plain grid order, 2 = k_zconv_strided wherever its
                                       // geometry allows (without the cost rule of zconv_strided_chunk), 4 = the 8-column tiles of the long y / z lines
                                       // in plain grid order (the two tiles of a 128-byte line on different XCDs, as before round 6),
                                       // 8 = the image's y passes on lines of one block per CU (L >= 1280) as ONE transform (k_fft_lines) instead of two half-length phases (k_fft_lines_split)
    bool    zconv_strided = true;      // views that only return the acquisition (compact planes, inc > 1): the direct z pass computes the
                                       // planes k * inc alone (k_zconv_strided: 1 / inc of the taps' work) and takes adjustImage's sum from
                                       // its INPUT rows (the sum over all planes is a linear functional of them); 0: every plane, as round 3
    bool    psf_overlap = true;        // PSF spectrum on the context's side stream, concurrent with passes A and B (views of
                                       // >= 2^24 voxels; +1 % views/s at 512^3).  On by default since round 3 (bit-identical, tested);
                                       // overlapped kernels share the chip, so their own durations in a profile no longer add up to
                                       // the stage time: profiles and bench.py's roofline leg switch both overlaps off
    int     tail_overlap = 1;          // extract + Poisson of a view concurrent with the next view's rotate+attenuate (device
                                       // views of >= 2^24 voxels; measured +0..6 % views/s at 512^3 depending on the box -- the two
                                       // stages then share the chip, so their own HIP-event times grow while the convolution
                                       // between them is undisturbed): 0 off, 1 on the context's own stream only (nothing outside
                                       // the library can observe the difference), 2 also on a caller's stream (the caller
                                       // calls mvsim_join / any entry point before its stream touches the outputs)
    bool    fuse_tail = false;         // adjust + extract + Poisson phase 1 in the epilogue of the convolution's last pass: saves
                                       // the 8 N bytes of the convolved volume's round trip, but the merged kernel is bound by
                                       // vector issue (0.61 ms against 0.25 + 0.36 ms for pass E and the streaming Poisson kernel
                                       // at 512^3): measured neutral to slightly slower, so off by default
    int     graph = 0;                 // replay simulate_view_dev from a captured hipGraph (small, launch-bound volumes)
    int     acq_u16 = 1;               // host-buffer views (mvsim_simulate_view_async): the acquisition crosses PCIe as uint16 counts when the
                                       // view is sampled (snr >= 0), widened on the host; automatic float32 fallback per view.  0: always float32
    int     host_threads = 0;          // threads of the host-side widening (0 = auto: min(16, hardware threads))
    int     view_lanes = 0;            // mvsim_simulate_views_dev: views in flight side by side (0 = auto: from the size of a view)
    int     view_batch = 2;            // mvsim_simulate_views_dev: the views STACKED -- one launch per stage for all of them (api.cpp:
                                       // views_enqueue_batched): 0 never, 1 whenever the views allow it, 2 auto (views of <= 2^26 voxels)
    bool    bcast_ring = false;        // ground-truth broadcast as one ncclBroadcast instead of scatter + all-gather
    bool    bcast_peer_copy = false;   // ... or as copy-engine transfers between IPC-mapped buffers (comm.cpp: bcast_peer_copy)
    bool    bcast_pipelined = false;   // ... or the pipelined form: the root scatters the WHOLE volume as nranks - 1 chunks while the peers all-gather
                                       // the pieces that have arrived among themselves (mvsim_comm_broadcast_plan; comm.cpp: bcast_pipelined)
    int64_t fft_pad[3] = {0, 0, 0};    // explicit padded sizes on the rocFFT path (0: choose)
    bool    skip_empty = true;         // convolution passes skip planes the fused rotate kernel found empty (exact; option for A/B runs)
};
// How launch_extract samples.  share 0: one launch, no work queue.  1..16: work queue whose per-block segments hold that many sixteenths
// of the block's voxels.  QUEUE_SHARE_AUTO + L (L = 0..16): the library's choice -- every voxel for queues of up to 64 MiB, else
// max(QUEUE_SHARE_START, L) sixteenths, L being what this context's views have needed so far (api.cpp: queue_mode_next).
constexpr int QUEUE_SHARE_AUTO = 32;
constexpr int QUEUE_SHARE_START = 5;
struct QueueMode {
    int           share = 0;
    unsigned int* hint  = nullptr;  // page-locked word k_poisson_refused raises to the sixteenths the fullest refused block would have needed
};
const Options& env_options();
int parse_option(Options& o, const char* name, const char* value);   // MVSIM_OK / MVSIM_EINVAL

// ST_PASS_*: the five passes of the hand-written convolution, nested inside ST_CONVOLVE (not part of the stage sum)
enum Stage { ST_ROTATE = 0, ST_ATTENUATE, ST_PSF, ST_CONVOLVE, ST_ADJUST, ST_EXTRACT, ST_PASS_A, ST_PASS_B, ST_PASS_C, ST_PASS_D,
             ST_PASS_E, ST_COUNT };

}  // namespace mvsim

struct mvsim_ctx {
    mvsim_ctx();
    mvsim_ctx(const mvsim_ctx&) = delete;
    mvsim_ctx& operator=(const mvsim_ctx&) = delete;
    unsigned long long alloc_epoch = 1;     // see DevBuf::epoch
    int         roctx_depth = 0;            // stage ranges open on this context (rebalanced by every entry point after an error return)
    int         device     = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream     = nullptr;
    int         num_cu     = 256;

    // workspaces
    mvsim::DevBuf vol_a, vol_b, vol_c;      // N-sized float staging for host entry points / fused path
    mvsim::DevBuf out_buf;                  // extract output staging
    mvsim::DevBuf psf_dev;                  // K^3 floats
    mvsim::DevBuf view_tab;                 // stacked views: [inverse models][extract tables][PSFs] of the current batch
    mvsim::DevBuf stencil_psf;              // direct stencil: PSF reversed along x, rows zero-padded to 4 taps
    mvsim::PinnedRing pinned;
    mvsim::DevBuf fft_real;                 // P^3 floats
    mvsim::DevBuf fft_spec_img, fft_spec_psf;
    mvsim::DevBuf fft_work;
    mvsim::DevBuf pqueue;                   // Poisson work queue: [count][items]
    mvsim::DevBuf sphere_list;              // phantom generator: (centre, radius, value) items
    mvsim::DevBuf plane_flags;              // per-plane non-zero flags of the current view (rotate_fft.hip -> the convolution passes)
    int*          empty_hint = nullptr;     // page-locked word the device writes: empty planes of the last view that carried flags (-1: none yet)
    unsigned int* queue_hint = nullptr;     // page-locked word the device raises: sixteenths of a block's voxels the fullest refused queue segment needed
    int           queue_share_learned = 0;  // ... as the host has seen it so far (auto share: queue_mode_next)
    int           views_since_flags = 0;    // views run WITHOUT flags since (a volume without empty planes pays nothing for the bookkeeping)
    mvsim::DevBuf weight_img;               // computeWeightImage of the last volume size (mvsim_simulate_iteration_dev)
    int64_t       weight_dim[3] = {0, 0, 0};
    mvsim::DevBuf host_gt, host_rot, host_att, host_con;   // device twins of the host-buffer simulate_view (grow-only)
    mvsim::DevBuf partials;                 // doubles: block partial sums + [sum, corr]
    mvsim::DevBuf partials_e;               // per-block sums of the c2r/crop pass
    mvsim::DevBuf cfft_f, cfft_g;           // custom FFT: image / PSF half spectra [Pz][Py][Hxp]
    mvsim::DevBuf cfft_g1, cfft_g2;         // compact PSF intermediates [Kz][Ky][Hxp], [Kz][Py][Hxp]
    mvsim::Options opt;
    std::unordered_set<const void*> lds_attr_set;   // kernels whose dynamic-LDS limit has been raised on this device
    std::map<std::string, void*> box_weights;   // early-sum weights (fft_kernels.hip: ensure_box_weights)
    mvsim::DevBuf partials_z;               // per-block sums of the z pass
    std::map<int, void*> twiddles;          // 2*length + kind -> device twiddle table (fft_kernels.hip)
    std::map<std::string, mvsim::FftPlan> plans;
    bool   fft_ready = false;

    // timing: a ring of event sets, one set per stage-operator / simulate_view call, so that a host can time
    // many asynchronous calls and read the averages once (mvsim_get_timings)
    static constexpr int TIMING_SLOTS = 64;
    bool       timing = false;
    hipEvent_t evr[TIMING_SLOTS][mvsim::ST_COUNT][2] = {};
    bool       ev_used[TIMING_SLOTS][mvsim::ST_COUNT] = {};
    int        ev_cur = 0;          // slot being recorded
    long long  ev_calls = 0;        // slots started since timing was enabled / last read
    bool       ev_created = false;
    mvsim_timings last = {};

    // pipelined host-buffer views (mvsim_simulate_view_async): upload(v+1) || compute(v) || download(v-1)
    static constexpr int ASYNC_SLOTS = 2, ASYNC_HISTORY = 8;
    double     async_corr_done[ASYNC_HISTORY] = {};   // adjustImage factors of the views that have landed
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    // the PSF's (x,y) spectrum is independent of the image passes A and B: it runs on a side stream beside them
    hipStream_t side_stream = nullptr;
    hipEvent_t  ev_fork = nullptr, ev_join = nullptr;
    bool        psf_on_side = false;          // last convolution: PSF spectrum overlapped (its time is inside passes A/B)
    // extract + Poisson of view v ("the tail") on a stream of its own, so that the next view's rotate+attenuate -- a
    // latency-bound kernel that leaves most of the chip idle -- runs beside it.  While tail_pending, the tail's outputs are
    // NOT yet ordered on ctx->stream: join_tail() restores that and every entry point does it first (set_device).
    hipStream_t tail_stream = nullptr;
    hipEvent_t  ev_tail_fork = nullptr, ev_tail = nullptr;
    bool        tail_pending = false;
    const char *tail_lo[2] = {nullptr, nullptr}, *tail_hi[2] = {nullptr, nullptr};   // byte ranges the pending tail writes / reads
    mvsim::DevBuf async_gt[ASYNC_SLOTS], async_acq[ASYNC_SLOTS];
    // acquisitions cross PCIe as 16-bit counts (option acq_transfer): packed on the device, widened on the host by mvsim_wait
    mvsim::DevBuf async_u16[ASYNC_SLOTS];         // [n_out uint16, padded to 256 bytes][flag word]
    void*      async_u16_host[ASYNC_SLOTS] = {};  // page-locked twin
    size_t     async_u16_host_bytes[ASYNC_SLOTS] = {};
    bool       async_as_u16[ASYNC_SLOTS] = {};
    float*     async_out_acq[ASYNC_SLOTS] = {};   // the caller's acquisition buffer of the slot's view
    long long  async_out_n[ASYNC_SLOTS] = {};
    long long  u16_views = 0, u16_fallbacks = 0;  // statistics (mvsim_get_transfer_stats)
    mvsim::DevBuf sync_u16;                       // the same transfer for the synchronous host-buffer entry points (down_counts)
    void*      sync_u16_host = nullptr;
    size_t     sync_u16_host_bytes = 0;
    hipEvent_t ev_h2d[ASYNC_SLOTS] = {}, ev_compute[ASYNC_SLOTS] = {}, ev_d2h[ASYNC_SLOTS] = {};
    bool       async_inflight[ASYNC_SLOTS] = {};
    long long  async_ticket[ASYNC_SLOTS] = {};
    double*    async_corr = nullptr;          // pinned: one double per slot
    const void* async_gt_src[ASYNC_SLOTS] = {};   // host ground truth a slot holds (identity + generation: re-upload skipped)
    unsigned long long async_gt_gen[ASYNC_SLOTS] = {};
    size_t     async_gt_bytes[ASYNC_SLOTS] = {};  // bytes of that upload (a view of another size never reuses it)
    long long  async_next = 0;
    bool       async_ready = false;
    bool       async_twins_busy = false;      // the single-buffered rot/att/con twins are being downloaded

    // hipGraph replay of views (option "graph"): cache keyed by every pointer and parameter of the view
    struct ViewGraph {
        std::string    key;
        hipGraph_t     graph;
        hipGraphExec_t exec;
        unsigned long long last_use;
    };
    std::vector<ViewGraph> graphs;
    std::unordered_set<std::string> graph_seen;
    unsigned long long graph_tick = 0, graph_epoch = 0;

    // mvsim_simulate_views_dev: views that cannot fill the chip one at a time run side by side on LANES -- child contexts of this
    // one (own stream, own workspaces, same device and options), forked from and joined to this context's stream by events
    std::vector<mvsim_ctx*> lanes;
    std::vector<hipEvent_t> lane_done;
    hipEvent_t  ev_lane_fork = nullptr;
    bool        is_lane = false;

    // z-slab tiling: what mvsim_view_slab_convolve_dev left in vol_a for mvsim_view_slab_finish_dev -- the slab's planes [z0, z1), or only
    // the planes k * slab_zstride of it (compact, slab_zstride > 1: the slab starts at a multiple of the view's spacing)
    int64_t slab_z0 = -1, slab_z1 = -1;
    int     slab_zstride = 1;

    // RCCL
    void* comm = nullptr;
    void* peer_copy = nullptr;             // broadcast=peer_copy: IPC maps, copy streams (comm.cpp)
    int   nranks = 1, rank = 0;
};

inline mvsim_ctx::mvsim_ctx()
{
    for (mvsim::DevBuf* b : {&vol_a, &vol_b, &vol_c, &out_buf, &psf_dev, &stencil_psf, &fft_real, &fft_spec_img, &fft_spec_psf, &fft_work,
                             &pqueue, &view_tab, &sphere_list, &weight_img, &plane_flags, &host_gt, &host_rot, &host_att, &host_con, &partials, &partials_e, &cfft_f, &cfft_g,
                             &cfft_g1, &cfft_g2, &partials_z, &async_gt[0], &async_gt[1], &async_acq[0], &async_acq[1], &async_u16[0], &async_u16[1], &sync_u16})
        b->epoch = &alloc_epoch;
}

namespace mvsim {

// ---- kernel launchers (each enqueues on `s`, returns MVSIM_* status) ----------------------------
int launch_rotate(hipStream_t s, const float* in, float* out, const int64_t dim[3], const Affine& inv);
int launch_attenuate(hipStream_t s, const float* in, float* out, const int64_t dim[3], double delta);
int launch_attenuate_scan(hipStream_t s, const float* in, float* out, const int64_t dim[3], double delta);
int launch_rotate_attenuate(hipStream_t s, const float* in, float* rot_or_null, float* att, const int64_t dim[3],
                            const Affine& inv, double delta, int fused_mode, bool* fused);
int launch_rotate_attenuate_planes(hipStream_t s, const float* in, float* rot_or_null, float* att, const int64_t dim[3],
                                   const Affine& inv, double delta, int z_begin, int z_count, int fused_mode, bool* fused);
// sum -> scal[0]; partial workspace must hold >= SUM_BLOCKS doubles
constexpr int SUM_BLOCKS = 2048;
int launch_sum(hipStream_t s, const float* in, int64_t n, double* partial, double* scal);
// scal[1] = (double)(target - min) / (scal[0] / n)
int launch_adjust_corr(hipStream_t s, double* scal, int64_t n, float min_value, float target);
int launch_adjust_apply(hipStream_t s, float* img, int64_t n, const double* scal, float min_value);
int launch_norm_apply(hipStream_t s, float* img, int64_t n, const double* scal);
// extract (+ optional adjust using scal[1]) (+ optional Poisson).  in: Nx*Ny*Nz, out: Nx*Ny*nzo
int launch_extract(hipStream_t s, const float* in, float* out, const int64_t dim[3], int inc, bool adjust,
                   const double* scal, float min_value, bool noise, double mul, uint64_t seed,
                   uint32_t stream, uint64_t index_offset, void* queue_ws, QueueMode queue_mode, int index_inc = 0);
// bytes of the Poisson work queue (HBM) for nzo planes of `plane` voxels whichever sampler kernel takes them (planes that are no
// multiple of four voxels or unaligned buffers go group by group through k_extract_noise2_any, whose wave slots are padded per plane)
size_t poisson_queue_bytes_planes(long long plane, long long nzo, int share);
int poisson_queue_read_stats(const void* queue_ws, size_t bytes, long long stats[5]);
// ---- stacked views (mvsim_simulate_views_dev): one launch per stage for V views of one ground truth; blockIdx.y / .z = view
struct ExtractView {              // per-view operands of the extract + Poisson kernels (device table)
    const float*  in;             // the view's convolved planes
    float*        out;            // its acquisition
    const double* scal;           // its [sum, adjustImage factor]
    void*         queue;          // its Poisson work-queue segments (PItem) ...
    unsigned int* qcount;         // ... and their counters
    uint32_t      k0, k1, stream, pad;
};
int launch_rotate_attenuate_views(hipStream_t s, const float* in, float* att, const int64_t dim[3], const Affine* atab_dev, int nviews, double delta);
void poisson_queue_split(void* queue_ws, void** queue_items, unsigned int** qcount);
int launch_extract_views(hipStream_t s, const int64_t dim[3], int inc, bool adjust, float min_value, bool noise, double mul,
                         QueueMode queue_mode, int index_inc, int nviews, const ExtractView* vt_dev, bool vec_all);
int launch_pack_u16(hipStream_t s, const float* in, unsigned short* out16, int64_t n, unsigned int* flag);
int launch_make_isotropic(hipStream_t s, const float* in, float* out, const int64_t dim[3], int inc);
// phantom generator (phantom.hip)
int launch_downsample2x(hipStream_t s, const float* in, const int64_t dim[3], float* out);
int draw_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], double min_value, double max_value, int scale,
                     int half_pixel_offset, uint64_t* rnd_state, int64_t* n_spheres);
int splat_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], const mvsim_sphere* spheres, int64_t n);
int launch_weight_image(hipStream_t s, float* out, const int64_t dim[3]);
int launch_weights(hipStream_t s, float* const* views, int nv, int64_t n, const float* sum_in, float* sum_out,
                   float osem, bool sum_only);

// FFT convolution pieces
int launch_pad_mirror(hipStream_t s, const float* img, const int64_t dim[3], const int64_t kdim[3],
                      float* padded, const int64_t P[3]);
int launch_psf_embed(hipStream_t s, const float* psf, const int64_t kdim[3], float* padded, const int64_t P[3]);
int launch_cmul(hipStream_t s, float2* f, const float2* g, int64_t n);
// out = crop(real) * scale; also accumulates block partial sums -> partial (SUM_BLOCKS doubles) and scal[0]
int launch_crop_scale_sum(hipStream_t s, const float* real, const int64_t P[3], float* out,
                          const int64_t dim[3], float scale, double* partial, double* scal);
int launch_stencil(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                   const int64_t kdim[3], float* out);
bool stencil_chunk_geometry(const int64_t kdim[3], int64_t out[5]);
// raise a kernel's dynamic-LDS limit once per context (hipFuncSetAttribute is not free on a launch path)
int ensure_lds_attr(mvsim_ctx* ctx, const void* kernel, size_t bytes);

// What the caller would like the tail of the convolution to do beyond "write the convolved volume"; every request is
// optional for the implementation, which reports back what it did.
struct ConvTail {
    // in: `img` / `psf` / `out` hold this many views stacked back to back (mvsim_simulate_views_dev; the caller has checked
    // custom_fft_batchable): one launch per pass for all of them, sums and factors in the context's scalar pairs 0 .. views-1
    int views = 1;
    // in: produce only the planes k * zstride (the planes extractSlices will read) into a COMPACT out[nk][Ny][Nx].
    // Possible when adjustImage's sum comes from the spectrum side (early sum); out: the stride actually applied (1 = full).
    int zstride = 1;
    // in: the caller would like the last pass to adjust (Tools.adjustImage), extract (every inc-th plane) and run phase 1 of
    // the Poisson sampler itself instead of writing the convolved volume: everything below describes that request.
    bool     want_fuse = false;
    float    min_value = 0.f, target_average = 1.f;
    float*   con_adj = nullptr;        // adjusted volume (all planes) if the caller wants it, else null
    float*   acq = nullptr;            // acquisition
    int      inc = 1;
    bool     noise = false;
    double   mul = 0.0;
    uint64_t seed = 0;
    uint32_t stream = 0;
    // out: true when the convolution did all of that (sum, factor in the context's scalar slots; acquisition complete)
    bool     fused = false;
    // in: pass A has run already -- the context's spectrum buffer holds the x transform of the image rows (fused rotate +
    // attenuate + x transform, rotate_fft.hip); `img` is not read
    bool     x_done = false;
    // in: voxels of the view (> 0: compute adjustImage's factor from min_value / target_average together with the sum);
    // out: corr_done = the factor is in the context's scalar slot, no separate k_adjust_corr needed
    long long corr_n = 0;
    bool      corr_done = false;
    // in (with x_done): per-plane flags the fused rotate kernel has set (1 = the attenuated plane holds a non-zero voxel; device
    // array of dim[2] ints): passes B .. E skip the planes whose inputs are all empty -- exact, their spectra are zero
    const int* plane_nz = nullptr;
};
// bytes of queue workspace the fused tail needs for this geometry (0: the geometry has no fused tail)
size_t fused_tail_queue_bytes(const int64_t dim[3], const int64_t kdim[3], int inc, bool con_wanted, const Options& opt);
// plane / idx_inc / index_offset: how the RNG counter of an output element follows from its position (ResolveJob)
int launch_poisson_resolve(hipStream_t s, float* out, void* queue_items, const unsigned int* qcount, int segments, unsigned int segcap,
                           double mul, uint64_t seed, uint32_t stream, long long plane, int idx_inc, uint64_t index_offset);
int comm_allreduce_f64_on_stream(mvsim_ctx* cc, double* value_dev, hipStream_t s);      // comm.cpp
int rotate_attenuate_fftx(mvsim_ctx* ctx, const float* gt, float* rot_or_null, float* att_or_null, const int64_t dim[3],
                          const int64_t kdim[3], const Affine& inv, double delta, bool* done, const int** plane_nz = nullptr,
                          int z_first = 0, int nzl = -1);
int fft_convolve(mvsim_ctx* ctx, const float* img_dev, const int64_t dim[3], const float* psf_dev,
                 const int64_t kdim[3], float* out_dev, ConvTail* tail);
void fft_release(mvsim_ctx* ctx);
void choose_padded(const int64_t dim[3], const int64_t kdim[3], int64_t P[3], const Options& opt);
// hand-written LDS FFT path (fft_kernels.hip)
bool custom_fft_sizes(const int64_t dim[3], const int64_t kdim[3], int64_t P[3], const Options& opt);
// orders a pending tail (extract + Poisson of the last view, on the tail stream) in front of whatever is enqueued on
// ctx->stream next
int  join_tail(mvsim_ctx* ctx);
int  custom_fft_convolve(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                         const int64_t kdim[3], const int64_t P[3], float* out, ConvTail* tail);
// z-slab form (direct z pass only): `img` holds the planes [z_in0, z_in0 + nz_in) of a volume with dim[2] planes,
// `out` receives the planes [z_out0, z_out0 + nz_out); every plane the Kz taps reach (mirrored at the global faces)
// must lie inside the input range.  The sum left in the context's scalar slot is the sum of the output planes.
struct SlabRange {
    int z_in0, nz_in, z_out0, nz_out;
};
int  custom_fft_convolve_slab(mvsim_ctx* ctx, const float* img, const int64_t dim[3], const float* psf,
                              const int64_t kdim[3], const int64_t P[3], const SlabRange& slab, float* out, ConvTail* tail);
void custom_fft_release(mvsim_ctx* ctx);
bool custom_fft_geometry(const int64_t dim[3], const int64_t kdim[3], int64_t g[5], const Options& opt);
// true when views of this geometry can run stacked through the hand-written passes (ConvTail::views > 1): direct z pass, early sum,
// pass B reading the mirrored y halo from its mirror images, no inline-FFT z pass, no fused tail
bool custom_fft_batchable(const mvsim_ctx* ctx, const int64_t dim[3], const int64_t kdim[3]);

// stage markers: a roctx range per stage (what rocprofv3 --marker-trace / a timeline shows around the launches; the reference
// prints a time stamp per stage, SimulateMultiViewDataset.java:553-590) and, when timing is on, HIP events recorded on the
// stream the kernels run on.  begin/end pairs nest strictly.
inline const char* stage_name(int st)
{
    static const char* const names[ST_COUNT] = {"mvsim:rotate", "mvsim:attenuate", "mvsim:psf_spectrum", "mvsim:convolve", "mvsim:adjust",
                                               "mvsim:extract_poisson", "mvsim:pass_A_x_r2c", "mvsim:pass_B_y_fwd", "mvsim:pass_C_z",
                                               "mvsim:pass_D_y_inv", "mvsim:pass_E_x_c2r"};
    return names[st];
}
inline void ev_begin(mvsim_ctx* ctx, int st)
{
    (void)roctxRangePushA(stage_name(st));
    ctx->roctx_depth += 1;
    if (ctx->timing) (void)hipEventRecord(ctx->evr[ctx->ev_cur][st][0], ctx->stream);
}
inline void ev_end(mvsim_ctx* ctx, int st)
{
    if (ctx->roctx_depth > 0) { (void)roctxRangePop(); ctx->roctx_depth -= 1; }
    if (ctx->timing) {
        (void)hipEventRecord(ctx->evr[ctx->ev_cur][st][1], ctx->stream);
        ctx->ev_used[ctx->ev_cur][st] = true;
    }
}
// An error return between ev_begin and ev_end (MVSIM_TRY / MVSIM_HIP) leaves ranges open: every entry point closes what
// the previous call on this context left behind (set_device), and so does mvsim_destroy.
inline void ev_rebalance(mvsim_ctx* ctx)
{
    while (ctx->roctx_depth > 0) { (void)roctxRangePop(); ctx->roctx_depth -= 1; }
}
// start a new slot (called once per timed call)
inline void ev_next(mvsim_ctx* ctx)
{
    if (!ctx->timing) return;
    if (ctx->ev_calls > 0) ctx->ev_cur = (ctx->ev_cur + 1) % mvsim_ctx::TIMING_SLOTS;
    ctx->ev_calls += 1;
    for (int s = 0; s < ST_COUNT; ++s) ctx->ev_used[ctx->ev_cur][s] = false;
}

// [sum, adjustImage factor] of the current view, behind the block partials; stacked views: one pair per view.  The doubles behind
// the pairs are scratch of the one-double all-reduce (comm.cpp)
constexpr int SCAL_DOUBLES = 2 * MVSIM_MAX_VIEWS;
constexpr size_t PARTIALS_BYTES = (size_t)(SUM_BLOCKS + SCAL_DOUBLES + 8) * sizeof(double);
inline double* scal_of(mvsim_ctx* ctx, int view = 0) { return ctx->partials.as<double>() + SUM_BLOCKS + 2 * view; }

// comm.cpp: a device range is about to go away -- drop the peer mappings registered for it (broadcast=peer_copy)
void comm_forget_range(mvsim_ctx* ctx, const void* p, size_t bytes);

void axis_rotation_host(const int64_t dim[3], int axis, int degrees, double m[12]);
void affine_invert_host(const double m[12], double inv[12]);

}  // namespace mvsim
