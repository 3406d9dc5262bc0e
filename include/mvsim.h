/*
 * mvsim.h -- C ABI of libmvsim.so: MI355X (gfx950) implementation of the per-view
 * acquisition pipeline of net.preibisch.simulation.SimulateMultiViewDataset
 *
 *     rotate -> attenuate -> 3-D PSF convolve -> adjust -> axial slice extraction -> Poisson
 *
 * The reference is pure Java with NO foreign-function interface for this path; each entry
 * point below names the Java static method (file:line in the reference tree) whose body it
 * replaces.  The Java-side binding a maintainer would add (JNI shim + facade with the same
 * signatures) is in java/ and described in INTEGRATION.md.
 *
 *   SMVD  = src/main/java/net/preibisch/simulation/SimulateMultiViewDataset.java
 *   Tools = src/main/java/net/preibisch/simulation/Tools.java
 *
 * Conventions
 *   - all images are IEEE float32, x fastest: index = x + Nx*(y + Ny*z)  (ArrayImg order)
 *   - dim[3] = {Nx, Ny, Nz};  kdim[3] likewise for the PSF
 *   - every function returns 0 (MVSIM_OK) or a negative status; mvsim_last_error() gives the
 *     thread-local message of the last failure.  No C++ exception crosses this boundary.
 *   - "host" entry points: caller owns every buffer, pre-sized; the call is synchronous and
 *     keeps no reference to the pointers afterwards.
 *   - "_dev" entry points: pointers are device (HBM) addresses valid on the context's GPU;
 *     work is enqueued on the context's stream (mvsim_set_stream) and the call returns
 *     without synchronising unless stated.
 *   - a context is bound to one GPU and is NOT thread-safe; use one per host thread.
 *   - there is no CPU fallback: without a usable gfx950 device mvsim_create fails.
 */
#ifndef MVSIM_H
#define MVSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSIM_OK       0
#define MVSIM_EINVAL  (-1)   /* bad dims / axis / inc / null pointer              */
#define MVSIM_ENOMEM  (-2)   /* device or host allocation failed                  */
#define MVSIM_EHIP    (-3)   /* HIP runtime error                                 */
#define MVSIM_EFFT    (-4)   /* rocFFT error                                      */
#define MVSIM_ERCCL   (-5)   /* RCCL error                                        */
#define MVSIM_ENODEV  (-6)   /* no usable GPU                                     */

typedef struct mvsim_ctx mvsim_ctx;

/* ---- library / context ------------------------------------------------------------ */
const char* mvsim_version(void);
const char* mvsim_last_error(void);
int  mvsim_device_count(int* count);
int  mvsim_create(int device, mvsim_ctx** ctx);
int  mvsim_destroy(mvsim_ctx* ctx);
/* Use a caller-owned hipStream_t (e.g. the host framework's current stream); NULL restores
 * the context's own stream. */
int  mvsim_set_stream(mvsim_ctx* ctx, void* hip_stream);
int  mvsim_synchronize(mvsim_ctx* ctx);
/* Orders everything the context still has in flight on its internal streams (the extract + Poisson tail of the last
 * device view, option "tail_overlap") in front of whatever is enqueued on its stream next; no host synchronisation.  Every
 * entry point does this first, so only a caller that set its OWN stream (mvsim_set_stream), opted in with
 * tail_overlap = any, and enqueues work of its own on that stream needs to call it. */
int  mvsim_join(mvsim_ctx* ctx);
/* Run-time switches of a context (tests and experiments; production needs none).  Defaults come from the environment
 * variables of the same meaning, read once per process: "fft_zpass" = auto|direct|fft (MVSIM_FFT_ZPASS),
 * "fft_backend" = custom|rocfft (MVSIM_FFT_BACKEND), "fft_pad" = "px,py,pz"|auto (MVSIM_FFT_PAD), "fused_rotate" = auto|1|0|2
 * (MVSIM_NO_FUSED_ROTATE; auto = fused from 131072 columns up, separate kernels for small views; 2 = the variant that
 * recomputes the row geometry in every lane), "poisson_queue" = 1|0 (MVSIM_POISSON_NOQUEUE), "early_sum" = 1|0 (MVSIM_NO_EARLY_SUM),
 * "graph" = 0|1 (MVSIM_GRAPH), "broadcast" = scatter_allgather|ring|peer_copy|pipelined (MVSIM_BROADCAST), "psf_overlap" = 1|0 (the PSF's spectrum on a side
 * stream of the context, concurrent with the image passes A and B), "tail_overlap" = 1|0|any (extract + Poisson of a device
 * view concurrent with the next view's rotate+attenuate; 1 = only on the context's own stream, see mvsim_join; both
 * overlaps are on by default -- results are bit-identical to the serial order -- and are switched off for profiles whose
 * per-kernel durations must add up to the stage times), "fuse_tail" = 0|1 (adjust +
 * extract + Poisson phase 1 in the epilogue of the convolution's last pass), "fused_fftx" = auto|1|0 (per-view pipeline: rotate + attenuate + the x transform of
 * the FFT convolution as one kernel, so that the attenuated volume crosses HBM only when requested; auto = from 131072
 * columns up), "attenuate" = serial|scan (mvsim_attenuate3d
 * as a wavefront-level prefix scan along the illumination axis: parallel in y, not bit-identical to the serial walk).
 * MVSIM_OPTIONS="name=value;name=value" sets any of them process-wide.  Unknown names or values: MVSIM_EINVAL. */
int  mvsim_set_option(mvsim_ctx* ctx, const char* name, const char* value);
/* Release cached FFT plans / workspaces / PSF spectra held by the context. */
int  mvsim_release_caches(mvsim_ctx* ctx);

/* ---- device memory (for hosts that keep volumes resident between calls) ------------- */
int  mvsim_dev_alloc(mvsim_ctx* ctx, size_t bytes, void** dptr);
int  mvsim_dev_free(mvsim_ctx* ctx, void* dptr);
int  mvsim_upload(mvsim_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  mvsim_download(mvsim_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Page-locked host memory for the host-buffer entry points: transfers from/to such blocks run at PCIe speed (the
 * JNI shim hands them to Java as direct ByteBuffers).  Ordinary pageable buffers work everywhere too, just slower. */
int  mvsim_host_alloc(mvsim_ctx* ctx, size_t bytes, void** hptr);
int  mvsim_host_free(mvsim_ctx* ctx_or_null, void* hptr);
/* Host-to-host copy on the library's host threads (option "host_threads"; ctx may be NULL): what the JNI shim uses between a Java heap
 * array held with GetPrimitiveArrayCritical and a page-locked staging block -- the copies of the facade's Buffers.toBlock / toImg
 * (SimulateMultiViewDataset.java:109,198: every operator returns a NEW ArrayImg).  Synchronous; overlapping ranges are moved in order. */
int  mvsim_host_copy(mvsim_ctx* ctx_or_null, void* dst, const void* src, size_t bytes);
/* byte-wise fill, asynchronous on the context stream (e.g. the zero canvas of the phantom generator) */
int  mvsim_dev_memset(mvsim_ctx* ctx, void* dptr, int value, size_t bytes);

/* ---- pure host helpers --------------------------------------------------------------- */
/* SMVD:80-102 axisRotation(Interval,int axis,int degrees): forward model T(+c) R T(-c) as a
 * row-major 3x4 matrix (mpicbg AffineModel3D layout m00..m23).  Zero-min interval assumed. */
int     mvsim_axis_rotation(const int64_t dim[3], int axis, int degrees, double m[12]);
/* (Nz-1)/inc+1, SMVD:197 */
int64_t mvsim_extract_nz(int64_t nz, int inc);
/* (Nz_acq-1)*inc+1, SMVD:146 */
int64_t mvsim_isotropic_nz(int64_t nz_acq, int inc);
/* (SNR/sqrt(5))^2, Tools:76 */
double  mvsim_poisson_mul(double snr);

/* ---- stage operators, host buffers ----------------------------------------------------- */
/* SMVD:104-135 rotateAroundAxis: out[l] = trilinear(in zero-extended, M^-1 l). */
int mvsim_rotate_around_axis(mvsim_ctx* ctx, const float* in, const int64_t dim[3],
                             int axis, int degrees, float* out);
/* SMVD:318-364 attenuate3d: light enters at y = Ny-1; Nx > Ny is rejected (the reference
 * walks dimension(0) steps along y and leaves the interval). */
int mvsim_attenuate3d(mvsim_ctx* ctx, const float* in, const int64_t dim[3], double delta, float* out);
/* Tools:112-118 normImage: img <- (float)(img / sum), in place. */
int mvsim_norm_image(mvsim_ctx* ctx, float* img, int64_t n);
/* SMVD:253-264 convolve(img, psf, service): normalises psf IN PLACE (sum -> 1), then exact
 * linear convolution with mirror-single image boundary, kernel centre kdim/2, no flip.
 * method: 0 = auto (the stencil up to 4 x 4 x 4 taps, where it is measured faster; the FFT passes beyond), 1 = FFT,
 * 2 = direct LDS-tiled stencil (any PSF up to 64 taps per axis; 2 Kx Ky Kz flop per voxel). */
int mvsim_convolve(mvsim_ctx* ctx, const float* img, const int64_t dim[3],
                   float* psf, const int64_t kdim[3], int method, float* out);
/* Tools:143-159 adjustImage: in place; *correction = (target - min) / mean. */
int mvsim_adjust_image(mvsim_ctx* ctx, float* img, int64_t n, float min_value, float target_average,
                       double* correction);
/* SMVD:181-231 extractSlices(img, inc, poissonSNR, rnd): out has mvsim_extract_nz(Nz,inc)
 * planes.  snr < 0 => bit-exact strided copy.  Otherwise per-voxel Poisson(lambda = v * mul)
 * from the counter-based generator keyed by (seed, stream) with counter = source voxel index;
 * the Java facade takes seed = rnd.nextLong(). */
int mvsim_extract_slices(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float snr,
                         uint64_t seed, uint32_t stream, float* out);
/* Tools:73-86 poissonProcess(img, SNR, rnd): in place over n values; counter = index_offset+i. */
int mvsim_poisson_process(mvsim_ctx* ctx, float* img, int64_t n, double snr,
                          uint64_t seed, uint32_t stream, uint64_t index_offset);
/* SMVD:144-171 makeIsotropic: out has mvsim_isotropic_nz(Nz,inc) planes. */
int mvsim_make_isotropic(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float* out);
/* SMVD:280-316 computeWeightImage (the reference ignores its delta argument). */
int mvsim_compute_weight_image(mvsim_ctx* ctx, const int64_t dim[3], float* out);
/* SMVD:436-522 drawSpheres(img, minValue, maxValue, scale, halfPixelOffset, rnd), in place.  rnd_state is the
 * 48-bit state of the caller's java.util.Random ((seed ^ 0x5DEECE66D) & (2^48-1) right after `new Random(seed)`);
 * it is advanced exactly as the reference advances it.  The walk over the large sphere (one shared sequential
 * random stream) runs on the host, the max-compositing of the small spheres on the GPU.  n_spheres may be NULL. */
int mvsim_draw_spheres(mvsim_ctx* ctx, float* img, const int64_t dim[3], double min_value, double max_value,
                       int scale, int half_pixel_offset, uint64_t* rnd_state, int64_t* n_spheres);
/* The GPU half of drawSpheres for hosts that walk the large sphere THEMSELVES with the caller's java.util.Random (any
 * subclass, SimulateTileStitching.java:93,108 passes `new Random(seed)`): max-composite n small spheres -- ImgLib2
 * HyperSphere geometry, nested truncated radii -- into img, in place.  A sphere that leaves the image is rejected
 * (the reference throws there). */
typedef struct mvsim_sphere {
    int32_t cx, cy, cz;       /* centre                                   */
    int32_t radius;           /* rnd.nextInt(10*scale) + 1 (SMVD:468)     */
    float   value;            /* (float) intensity, Math.max-composited   */
} mvsim_sphere;
int mvsim_splat_spheres(mvsim_ctx* ctx, float* img, const int64_t dim[3], const mvsim_sphere* spheres, int64_t n);
int mvsim_splat_spheres_dev(mvsim_ctx* ctx, float* img_dev, const int64_t dim[3], const mvsim_sphere* spheres_host, int64_t n);
/* SMVD:394-424 downSample2x: out has dim[d]/2 - 1 samples per dimension. */
int mvsim_downsample2x(mvsim_ctx* ctx, const float* in, const int64_t dim[3], float* out);

/* ---- stage operators, device-resident buffers (asynchronous on the context stream) ------ */
/* (mvsim_draw_spheres_dev returns after the host walk; the compositing kernels are asynchronous.) */
int mvsim_draw_spheres_dev(mvsim_ctx* ctx, float* img, const int64_t dim[3], double min_value, double max_value,
                           int scale, int half_pixel_offset, uint64_t* rnd_state, int64_t* n_spheres);
int mvsim_downsample2x_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], float* out);
int mvsim_rotate_around_axis_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3],
                                 int axis, int degrees, float* out);
int mvsim_attenuate3d_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], double delta, float* out);
/* psf_host is normalised in place on the host (it is <= ~1 MB), then uploaded. */
int mvsim_convolve_dev(mvsim_ctx* ctx, const float* img, const int64_t dim[3],
                       float* psf_host, const int64_t kdim[3], int method, float* out);
/* Synchronises (the correction is returned to the host). */
int mvsim_adjust_image_dev(mvsim_ctx* ctx, float* img, int64_t n, float min_value, float target_average,
                           double* correction);
int mvsim_extract_slices_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float snr,
                             uint64_t seed, uint32_t stream, float* out);
int mvsim_make_isotropic_dev(mvsim_ctx* ctx, const float* in, const int64_t dim[3], int inc, float* out);
int mvsim_compute_weight_image_dev(mvsim_ctx* ctx, const int64_t dim[3], float* out);

/* ---- cross-view weight normalisation (after the view loop, SMVD:615-640 and :648-661) ------------------ */
#define MVSIM_MAX_VIEWS 32
/* out[i] = sum over views (float accumulation in view order, starting from 0), SMVD:625-628 / :648-661.
 * vols: HOST array of n_views DEVICE pointers. */
int mvsim_sum_views_dev(mvsim_ctx* ctx, const float* const* vols, int n_views, int64_t n, float* out);
/* In place on every view: sum == 0 -> 0, else min(1, osem * (w / sum)) in float (SMVD:630-639).
 * sum_or_null: per-voxel sum over ALL views (e.g. after mvsim_comm_allreduce_sum when views are sharded over
 * ranks); NULL => summed here over the given views. */
int mvsim_normalize_weights_dev(mvsim_ctx* ctx, float* const* weights, int n_views, int64_t n,
                                const float* sum_or_null, float osem);
/* Host buffers: weights[v] are host pointers of n floats each. */
int mvsim_normalize_weights(mvsim_ctx* ctx, float* const* weights, int n_views, int64_t n, float osem);

/* ---- fused per-view pipeline: loop body SMVD:570-585 -------------------------------------- */
typedef struct mvsim_view_params {
    int32_t  axis;            /* 0 (SMVD:570)                                  */
    int32_t  degrees;         /* angle + angleOffset                           */
    double   delta;           /* attenuation, 0.01 (SMVD:533)                  */
    float    min_value;       /* 1e-4f (SMVD:77)                               */
    float    target_average;  /* 1     (SMVD:78)                               */
    int32_t  inc;             /* lightsheetSpacing (SMVD:532)                  */
    float    snr;             /* poissonSNR, 25 (SMVD:531); < 0 => no noise    */
    uint64_t seed;            /* counter-RNG key; 464232194 (SMVD:76)          */
    uint32_t stream;          /* view index                                    */
    int32_t  conv_method;     /* 0 auto, 1 FFT, 2 direct stencil               */
} mvsim_view_params;

/* Optional intermediate outputs (NULL = not wanted).  acq is required. */
typedef struct mvsim_view_outputs {
    float* rot;   /* Nx*Ny*Nz               (SMVD:570) */
    float* att;   /* Nx*Ny*Nz               (SMVD:573) */
    float* con;   /* Nx*Ny*Nz, adjusted     (SMVD:580-582) */
    float* acq;   /* Nx*Ny*extract_nz       (SMVD:585) */
} mvsim_view_outputs;

void mvsim_view_params_default(mvsim_view_params* p);

/* Device-resident: gt and every non-NULL output are device pointers; psf_host is a host
 * buffer, normalised in place.  Intermediates stay in HBM; nothing is copied to the host
 * except the scalar correction when correction != NULL (which forces a synchronise). */
int mvsim_simulate_view_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3],
                            float* psf_host, const int64_t kdim[3],
                            const mvsim_view_params* params, const mvsim_view_outputs* out,
                            double* correction);
/* The view loop of `main` (SimulateMultiViewDataset.java:567-585) for views that cannot fill the chip one at a time -- the
 * reference's own run is 7 views of a 289^3 volume with 51^3 PSFs (:376-380, :399, :531-548): n_views independent views of ONE
 * ground truth in one call.  psf_host[v] (normalised in place like everywhere else), params[v] and outs[v] describe view v; all
 * views share dim and kdim.  The library runs as many of them side by side as pays for their size (option "view_lanes" =
 * auto|1..32: child contexts of `ctx` on the same device, forked from and joined to ctx's stream, so the call is asynchronous on
 * that stream like mvsim_simulate_view_dev).  Every output is bit-identical to what n_views sequential mvsim_simulate_view_dev
 * calls write -- also when several views name the SAME (or overlapping) PSF memory, e.g. one buffer for all views: such PSFs are
 * normalised one after the other in view order, exactly as sequential calls would (a buffer named n times is normalised n times;
 * distinct buffers are normalised by one host thread each).  Output buffers must not overlap each other or the ground truth
 * (MVSIM_EINVAL). */
int mvsim_simulate_views_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* const* psf_host,
                             const int64_t kdim[3], const mvsim_view_params* params, const mvsim_view_outputs* outs,
                             int n_views);
/* The same with HOST buffers: the ground truth goes up once, the views run in one call (stacked / side by side as above), and the
 * acquisitions -- acq_host[v] of dim[0] * dim[1] * mvsim_extract_nz(dim[2], params[v].inc) floats -- come back together, as uint16 counts
 * over PCIe where the views are sampled (see mvsim_get_transfer_stats).  Synchronous; what a JVM calls for the reference's own run
 * (SimulateMultiViewDataset.java:567-585: seven views of one 289^3 volume). */
int mvsim_simulate_views(mvsim_ctx* ctx, const float* gt_host, const int64_t dim[3], float* const* psf_host,
                         const int64_t kdim[3], const mvsim_view_params* params, float* const* acq_host, int n_views);
/* One whole iteration of `main`'s view loop (SimulateMultiViewDataset.java:567-613), device-resident: the view above
 * (rotate, attenuate, convolve, adjust, extractSlices + Poisson) followed by what the loop does with its results --
 *   iso          = makeIsotropic(acq, inc)                                    (:588)   Nx*Ny*isotropic_nz
 *   view         = rotateAroundAxis(iso, axis, back_degrees)                  (:591)   same size (the reference passes -angle)
 *   view_weights = rotateAroundAxis(computeWeightImage(rot, delta), axis, back_degrees)   (:576,592)   Nx*Ny*Nz
 *   view_psf     = rotateAroundAxis(psf, axis, back_degrees)                  (:593)   Kx*Ky*Kz, the PSF as convolve() left it (normalised, Q5)
 * Every non-NULL member of `more` is a device buffer of that size; `iso` may be NULL with `view` set (library scratch).
 * computeWeightImage depends on the dimensions alone: it is rendered once per context and size and reused.
 * `out->acq` is required; the call is asynchronous on the context's stream. */
typedef struct mvsim_iteration_outputs {
    float* iso;
    float* view;
    float* view_weights;
    float* view_psf;
} mvsim_iteration_outputs;
int mvsim_simulate_iteration_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3],
                                 float* psf_host, const int64_t kdim[3],
                                 const mvsim_view_params* params, int back_degrees,
                                 const mvsim_view_outputs* out, const mvsim_iteration_outputs* more);
/* Host buffers in / out (what the JNI shim calls). */
int mvsim_simulate_view(mvsim_ctx* ctx, const float* gt, const int64_t dim[3],
                        float* psf_host, const int64_t kdim[3],
                        const mvsim_view_params* params, const mvsim_view_outputs* out,
                        double* correction);

/* Pipelined host-buffer views for a caller that loops over views (SMVD:567-613, SimulateTileStitching.java:93-111):
 * the call enqueues  upload(gt) -> view -> download(outputs)  on three HIP streams over two staging sets and returns a
 * ticket at once, so that upload(v+1), compute(v) and download(v-1) overlap; with page-locked buffers (mvsim_host_alloc)
 * a 512^3 view costs one PCIe transfer time instead of the sum of three phases.  gt_host, psf_host and every output
 * buffer must stay valid and untouched until mvsim_wait(ticket) returns; at most two tickets may be outstanding (a
 * third call waits for the oldest).  The ground truth is uploaded again only when its pointer or `gt_generation`
 * differs from what the staging set already holds (pass a new generation after changing the buffer's contents; the
 * view loop of `main` passes the same `rendered` image for every angle).  psf_host is normalised in place before the
 * call returns.  Intermediates (rot/att/con) may be requested; they are single-buffered on the device, so views that ask
 * for them do not overlap with their neighbours' downloads. */
int mvsim_simulate_view_async(mvsim_ctx* ctx, const float* gt_host, uint64_t gt_generation, const int64_t dim[3],
                              float* psf_host, const int64_t kdim[3], const mvsim_view_params* params,
                              const mvsim_view_outputs* out_host, int64_t* ticket);
/* Blocks until the view behind `ticket` has landed in its host buffers; *correction (may be NULL) receives the
 * adjustImage factor.  Tickets may be waited for in any order, each once. */
int mvsim_wait(mvsim_ctx* ctx, int64_t ticket, double* correction);
/* How the acquisitions of mvsim_simulate_view_async crossed PCIe.  Tools.poissonProcess stores Poisson COUNTS as floats
 * (Tools.java:84): a sampled view (snr >= 0) is packed to uint16 on the device, downloaded as half the bytes and widened into the
 * caller's float buffer by mvsim_wait (a few host threads, option "host_threads" = auto|1..256); a view holding a value that does not
 * survive the round trip (a count beyond 65 535) is fetched as float32 after all, automatically.  Exact either way.  Option
 * "acq_transfer" = auto|f32 (f32: never pack).  *views_as_u16: views that took the 16-bit path so far, *fallbacks: how many of them
 * had to be fetched as float32.  Either pointer may be NULL. */
int mvsim_get_transfer_stats(mvsim_ctx* ctx, int64_t* views_as_u16, int64_t* fallbacks);
/* Host buffers given as z slabs: volumes beyond 2^31-1 voxels do not fit one Java array / direct buffer
 * (SimulateMultiViewDataset.java:109 uses ArrayImg and cannot hold them at all), so the ground truth arrives as
 * n_gt_slabs pointers of gt_slab_nz[i] planes each (sum = dim[2]) and the acquisition leaves as n_acq_slabs pointers of
 * acq_slab_nz[j] planes each (sum = mvsim_extract_nz(dim[2], inc)).  Same arithmetic as mvsim_simulate_view. */
int mvsim_simulate_view_zslabs(mvsim_ctx* ctx, const float* const* gt_slabs, const int64_t* gt_slab_nz, int n_gt_slabs,
                               const int64_t dim[3], float* psf_host, const int64_t kdim[3],
                               const mvsim_view_params* params, float* const* acq_slabs, const int64_t* acq_slab_nz,
                               int n_acq_slabs, double* correction);

/* The per-stage operators with HOST buffers given as z slabs: an ArrayImg may hold 2^31-1 voxels (SimulateMultiViewDataset.java:109,
 * :198, :235, :321 create their results with ArrayImgFactory) but one direct ByteBuffer only 2^31-1 BYTES, so the Java facade hands
 * volumes beyond 2^29 voxels over as several page-locked blocks.  in_slabs[i] holds in_slab_nz[i] planes (sum = dim[2]); out_slabs[j]
 * receives out_slab_nz[j] planes (sum = dim[2], or mvsim_extract_nz(dim[2], inc) for extractSlices).  Same arithmetic as the
 * single-buffer entry points (SMVD:104-135, :318-364, :253-264, :181-231). */
int mvsim_rotate_around_axis_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                                    int axis, int degrees, float* const* out_slabs, const int64_t* out_slab_nz, int n_out);
int mvsim_attenuate3d_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                             double delta, float* const* out_slabs, const int64_t* out_slab_nz, int n_out);
int mvsim_convolve_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                          float* psf, const int64_t kdim[3], int method, float* const* out_slabs, const int64_t* out_slab_nz, int n_out);
int mvsim_extract_slices_zslabs(mvsim_ctx* ctx, const float* const* in_slabs, const int64_t* in_slab_nz, int n_in, const int64_t dim[3],
                                int inc, float snr, uint64_t seed, uint32_t stream, float* const* out_slabs, const int64_t* out_slab_nz,
                                int n_out);

/* ---- per-stage device timings of the last simulate_view / stage call (milliseconds) ------ */
typedef struct mvsim_timings {
    /* psf_ms is 0 when the PSF spectrum ran on the context's side stream (option psf_overlap, views of >= 2^24 voxels):
     * it then overlaps passes A and B and its time is part of convolve_ms */
    float rotate_ms, attenuate_ms, psf_ms, convolve_ms, adjust_ms, extract_ms, total_ms;
    /* the passes of the hand-written convolution, nested inside convolve_ms (0 on other paths): x real->complex,
     * y forward, z (direct convolution or FFT + product + inverse FFT), y inverse, x complex->real + crop + sum */
    float pass_a_ms, pass_b_ms, pass_c_ms, pass_d_ms, pass_e_ms;
} mvsim_timings;
/* Geometry of the hand-written convolution for this volume / PSF: {Px, Py, planes of the spectrum, Hxp (complex row
 * pitch), 1 if the z pass is the direct convolution}.  A pass moves 8 * Hxp * Py * planes bytes each way (the x passes
 * 4 N on their real side).  MVSIM_EINVAL when the sizes fall outside the pass table (rocFFT path). */
int mvsim_fft_geometry(const int64_t dim[3], const int64_t kdim[3], int64_t geometry[5]);
/* Geometry of the direct LDS-tiled stencil (method 2) for this PSF: {taps of one x chunk, y taps, z taps of the PSF chunk a
 * tile serves, LDS bytes per block, blocks per CU}.  Any PSF of 1..64 taps per axis is accepted (SimulateMultiViewDataset.java:579
 * loads 51^3 stacks); beyond that MVSIM_EINVAL (use the FFT method). */
int mvsim_stencil_geometry(const int64_t kdim[3], int64_t geometry[5]);
/* Planes the convolution passes of the last FLAGGED view skipped (option "skip_empty"; exact: the spectrum of an empty attenuated plane
 * is zero): stats = {planes of the view, planes whose attenuated image is empty (pass B does not transform them, the z pass does not
 * read them), planes of the z pass's output that are empty (passes D and E skip them)}.  Read from a page-locked word the device
 * writes, without synchronising: call after the view has completed.  All zero when no view has carried flags yet. */
int mvsim_get_plane_stats(mvsim_ctx* ctx, int64_t stats[3]);
/* The work queue of the Poisson sampler (Tools.java:73-86 is one sample per voxel; here the voxels whose sample needs the divergent fp64
 * code wait in per-block segments for a second kernel): stats = {bytes of queue workspace the context holds, items one segment holds,
 * bright items (lambda >= 10) and inversion items (lambda < 10) the context's last sampled view queued -- the first view of a stacked
 * call --, voxels its full segments refused (a third kernel samples them where they stand: same counts, slower), pending voxels (queued
 * or refused) of its fullest block: the segment size that refuses nothing}.
 * Option "poisson_queue_share" = 1..16 fixes the sixteenths of a block's voxels its segment holds (16, the default: 16 bytes per
 * acquired voxel, nothing is ever refused, the fastest).  "auto": 16 for queues of up to 64 MiB; otherwise 5 at first, and after a view
 * whose segments refused voxels the context's later views get what that view would have needed plus one sixteenth (memory for ~2 % of
 * a 512^3 view's time).  Counts are the same for every share.
 * Synchronises the context.  (Zeros after a view that did not go through the two-launch sampler: options "poisson_queue=0", "fuse_tail=1".) */
int mvsim_get_queue_stats(mvsim_ctx* ctx, int64_t stats[6]);
int mvsim_enable_timing(mvsim_ctx* ctx, int enable);
int mvsim_get_timings(mvsim_ctx* ctx, mvsim_timings* t);

/* ---- multi-GPU: one process per GPU, views shard across ranks (RCCL over xGMI) ------------ */
#define MVSIM_UNIQUE_ID_BYTES 128
/* rank 0 creates the id, the host passes the 128 bytes to every rank by any means. */
int mvsim_comm_unique_id(unsigned char id[MVSIM_UNIQUE_ID_BYTES]);
int mvsim_comm_init(mvsim_ctx* ctx, int nranks, int rank, const unsigned char id[MVSIM_UNIQUE_ID_BYTES]);
/* The RCCL this library is bound to in the running process: file path of the shared object that provides ncclGetVersion
 * (a process that loaded PyTorch first holds torch/lib/librccl.so under the same soname) and its version code
 * (major * 10000 + minor * 100 + patch).  Either output may be NULL. */
int mvsim_comm_library_info(char* path, size_t path_capacity, int* version);
/* Broadcast the ground-truth volume (device pointer, count floats) from root; enqueued on
 * the context stream.  The only collective on the path (views are independent, SMVD:567).
 * Default form: scatter (root sends chunk r to rank r, nranks-1 concurrent ncclSend: one per xGMI link) followed by
 * an in-place ncclAllGather, so that all links carry traffic; option "broadcast" = ring selects one ncclBroadcast. */
int mvsim_comm_broadcast_volume(mvsim_ctx* ctx, float* vol_dev, int64_t count, int root);
/* Option "broadcast" = peer_copy moves the chunks of the same scatter + all-gather with the copy engines between the ranks'
 * buffers (IPC-mapped into each other's processes; no RCCL kernels beside the views).  The buffer a rank broadcasts into must have
 * been REGISTERED: a collective -- every rank calls it, in the same order as the other collectives of the communicator, each for
 * its own buffer of >= count floats; it synchronises the context's stream.  A registration is never reused for another buffer:
 * registering again replaces it, mvsim_comm_unregister_volume (local), mvsim_dev_free of the buffer and mvsim_comm_destroy drop
 * it -- LOCALLY: the other ranks keep their mappings of this rank's buffer until THEY unregister, register again or destroy their
 * communicator.  So a registered volume is unregistered (or freed) on EVERY rank before any rank runs another peer_copy broadcast;
 * a broadcast into a registration one rank has already dropped writes through a mapping of freed memory.
 * One process per rank (ranks that share a process use the RCCL forms). */
int mvsim_comm_register_volume(mvsim_ctx* ctx, float* vol_dev, int64_t count);
/* Option "broadcast" = pipelined: scatter + all-gather loads the root's outbound links twice (its scatter chunks, then its own chunk of
 * the all-gather: 2 S / (N b) for a volume of S bytes over links of b bytes/s).  The pipelined form scatters the WHOLE volume as N - 1
 * chunks to the N - 1 peers, piece by piece, and lets the peers all-gather the pieces that have arrived among themselves over the
 * peer<->peer links WHILE the next piece leaves the root: one group of ncclSend / ncclRecv per stage, ~ S / ((N - 1) b) in all (1.3-1.4 ms
 * instead of 2.2 ms for a 512^3 volume on 8 GPUs).  The schedule is a pure function of (nranks, rank, root, count, pieces): this entry point
 * returns it -- ops[i] = {stage, kind (0 send, 1 receive), peer rank, first float, floats} in issue order, *n_ops of them (MVSIM_EINVAL when
 * capacity is too small; ops may be NULL to ask for the count) -- so that a host can check it without a GPU (every send has its receive in
 * the same stage, every rank ends with every float).  The last stage holds the unaligned tail as sends from the root. */
typedef struct mvsim_bcast_op { int32_t stage, kind, peer, pad; int64_t first, count; } mvsim_bcast_op;
int mvsim_comm_broadcast_plan(int nranks, int rank, int root, int64_t count, int pieces, mvsim_bcast_op* ops, int capacity, int* n_ops);
int mvsim_comm_unregister_volume(mvsim_ctx* ctx, const float* vol_dev);
/* In-place sum over ranks (RCCL all-reduce) of a device float buffer: the per-voxel weight sums of SMVD:625-628
 * when the views live on different GPUs.  Summation order differs from the sequential reference (<= 1 ulp). */
int mvsim_comm_allreduce_sum(mvsim_ctx* ctx, float* buf_dev, int64_t count);
/* Sum over ranks of ONE double held on the host (in place; synchronises): the adjustImage sum of a view whose
 * z slabs live on different GPUs. */
int mvsim_comm_allreduce_sum_f64(mvsim_ctx* ctx, double* value_host);
/* The same sum for ONE double that lives on the DEVICE (in place, asynchronous, no host trip): enqueued on `hip_stream` (null: the
 * context's stream), so a caller can order it behind the kernel that produces the value.  What mvsim_view_slab_dev runs between a
 * slab's convolution and its adjust (SimulateMultiViewDataset.java:582: Tools.adjustImage needs the whole view's sum). */
int mvsim_comm_allreduce_sum_f64_dev(mvsim_ctx* ctx, double* value_dev, void* hip_stream);
int mvsim_comm_destroy(mvsim_ctx* ctx);
/* view v of n_views belongs to rank v % nranks; returns how many views `rank` owns and writes
 * their indices (capacity max_out). */
int mvsim_shard_views(int n_views, int nranks, int rank, int* view_idx, int max_out);

/* ---- one PROCESS driving several GPUs (what a JVM host is): a group owns one context per device and an RCCL
 * communicator over them (ncclCommInitAll).  The ground truth goes host -> device 0 -> all devices (scatter +
 * all-gather over xGMI), then view v of SimulateMultiViewDataset.main's loop (:567-613) runs on device v % ndev and
 * its acquisition is copied to acq_host[v]; the call returns when every view has landed.  psf_host[v] is normalised in
 * place like everywhere else.  devices == NULL means 0 .. ndev-1. */
typedef struct mvsim_group mvsim_group;
int        mvsim_group_create(int ndev, const int* devices, mvsim_group** group);
int        mvsim_group_destroy(mvsim_group* group);
int        mvsim_group_size(const mvsim_group* group);
mvsim_ctx* mvsim_group_ctx(mvsim_group* group, int index);
/* gt_host is free for reuse when the call returns (the upload has completed; the collectives behind it may still run). */
int        mvsim_group_broadcast_volume(mvsim_group* group, const float* gt_host, const int64_t dim[3]);
int        mvsim_group_simulate_views(mvsim_group* group, float* const* psf_host, const int64_t kdim[3],
                                      const mvsim_view_params* params, int n_views, float* const* acq_host);

/* ---- z-slab tiling of ONE view across GPUs (BASELINE configs[3]/[4]) ----------------------------
 * For volumes whose views should be split over several GPUs: rank r owns the planes [z0, z1) of the view
 * (mvsim_slab_range gives a balanced partition).  Every rank holds the whole ground truth (broadcast); rotation and
 * attenuation of the slab and of the Kz/2 halo planes the PSF reaches are recomputed locally (attenuation runs along
 * y inside a plane, SMVD:335-359), the mirror boundary acts at the global faces only, and the only exchange is the
 * sum of the convolved voxels that adjustImage needs.  Per view and rank:
 *     mvsim_view_slab_convolve_dev(...,&slab_sum)      rotate, attenuate, convolve the slab (kept in the context)
 *     total = sum of slab_sum over ranks               e.g. mvsim_comm_allreduce_sum_f64
 *     mvsim_view_slab_finish_dev(..., total, acq)      adjust with the global mean, extract, Poisson
 * acq receives the acquired planes k with z0 <= k*inc < z1, in order (*n_planes of them); Poisson counters use the
 * global voxel index, so the tiling is invisible in the counts.  Needs rotation about x (axis 0), the hand-written
 * convolution path and a PSF depth <= 64. */
int mvsim_slab_range(int64_t nz, int nranks, int rank, int64_t* z0, int64_t* z1);
int mvsim_view_slab_convolve_dev(mvsim_ctx* ctx, const float* gt, const int64_t dim[3], float* psf_host,
                                 const int64_t kdim[3], const mvsim_view_params* params, int64_t z0, int64_t z1,
                                 double* slab_sum);
int mvsim_view_slab_finish_dev(mvsim_ctx* ctx, const int64_t dim[3], const mvsim_view_params* params, int64_t z0,
                               int64_t z1, double total_sum, float* acq, int64_t* n_planes);
/* The three steps above as ONE asynchronous call that never leaves the device: the slab's share of the sum stays in HBM, is reduced
 * over the ranks in place by the communicator of `comm_ctx` (null: ctx's own; no communicator or one rank: nothing to reduce) ON
 * ctx's stream, and adjust, extract and Poisson follow behind it.  The loop body tiled is SimulateMultiViewDataset.java:570-585.
 * Same results as the three-step form (the reduction adds the same doubles; RCCL's order over ranks may differ from a host
 * loop's in the last bit of the sum). */
int mvsim_view_slab_dev(mvsim_ctx* ctx, mvsim_ctx* comm_ctx, const float* gt, const int64_t dim[3], float* psf_host,
                        const int64_t kdim[3], const mvsim_view_params* params, int64_t z0, int64_t z1, float* acq,
                        int64_t* n_planes);

#ifdef __cplusplus
}
#endif
#endif /* MVSIM_H */
