/*
 * mvsim_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the per-view hot path of
 * net.preibisch.simulation.SimulateMultiViewDataset (reference, Java).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (libmvsim.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path and no JVM exists in the build container, so the reference
 * itself could not be executed.  Part of the arithmetic lives in third-party
 * jars that are absent from /root/reference (ImgLib2 core [pom-scijava 44.0.0
 * BOM], imglib2-algorithm 0.18.3 FFTConvolution, mpicbg 1.6.6 AffineModel3D /
 * RealSum, JDK java.util.Random); their published algorithms are restated
 * here.  What IS pinned: the JDK-specified java.util.Random stream (known
 * answers in tests/golden/), the Random123 Philox4x32-10 known answers, and
 * the analytic known-answer tests in tests/test_oracle_kat.py.
 *
 * All volumes are IEEE float32, x-fastest: index = x + Nx*(y + Ny*z)
 * (ImgLib2 ArrayImg cursor order, SimulateMultiViewDataset.java:115-132).
 *
 * Citations "SMVD:n" = src/main/java/net/preibisch/simulation/SimulateMultiViewDataset.java:n
 *           "Tools:n" = src/main/java/net/preibisch/simulation/Tools.java:n
 */
#ifndef MVSIM_ORACLE_H
#define MVSIM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- java.util.Random (JDK specification; used at SMVD:76, Tools:73) ---- */
typedef struct { uint64_t s; } orc_jrandom;
void     orc_jrandom_seed(orc_jrandom* r, int64_t seed);
int32_t  orc_jrandom_next(orc_jrandom* r, int bits);
int32_t  orc_jrandom_next_int(orc_jrandom* r);
int32_t  orc_jrandom_next_int_bound(orc_jrandom* r, int32_t bound);
int64_t  orc_jrandom_next_long(orc_jrandom* r);
double   orc_jrandom_next_double(orc_jrandom* r);

/* ---- uncommons/PoissonGenerator.java:95-109 (inter-arrival counting) ---- */
int32_t  orc_poisson_interarrival(orc_jrandom* r, double mean);

/* ---- Philox4x32-10 (Random123) + the O(1) counter-based Poisson sampler
 *      that the HIP path uses (independent second implementation). ---- */
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double   orc_det_log(double x);
double   orc_det_exp(double x);
double   orc_det_exp_neg(double lambda);        /* exp(-lambda), 0 < lambda < 10 */
double   orc_det_lgamma_int(int64_t k);            /* log(k!) */
int64_t  orc_poisson_counter(double lambda, uint64_t seed, uint32_t stream, uint64_t index);

/* ---- SMVD:80-102 axisRotation, mpicbg AffineModel3D semantics ---- */
/* m = row-major 3x4 forward model T(+c) R T(-c); minv = createInverse() */
void orc_axis_rotation(const int64_t dim[3], int axis, int degrees, double m[12]);
void orc_affine_invert(const double m[12], double minv[12]);

/* ---- SMVD:104-135 rotateAroundAxis (zero-extended trilinear) ---- */
int orc_rotate_around_axis(const float* in, const int64_t dim[3], int axis, int degrees, float* out);
/* the same loop over the output planes z0 .. z0 + nzp - 1 only (out: nzp planes) */
int orc_rotate_around_axis_planes(const float* in, const int64_t dim[3], int axis, int degrees, int64_t z0, int64_t nzp, float* out);

/* ---- SMVD:318-364 attenuate3d ---- */
int orc_attenuate3d(const float* in, const int64_t dim[3], double delta, float* out);

/* ---- Tools:112-132 normImage / sumImage (mpicbg RealSum) ---- */
double orc_sum_image(const float* img, int64_t n);
void   orc_norm_image(float* img, int64_t n);

/* bench.py's cpu_baseline mode "all_cores": OpenMP over planes / columns in rotate, attenuate, adjustImage (default 0: serial) */
void orc_set_parallel(int on);
int  orc_max_threads(void);

/* ---- Tools:143-159 adjustImage; returns the correction factor ---- */
double orc_adjust_image(float* img, int64_t n, float min_value, float target_average);

/* ---- SMVD:253-264 convolve: exact linear convolution, mirror-single image
 *      boundary, kernel centre K/2, no flip (FFTConvolution semantics).
 *      Direct double-precision summation; normalises psf IN PLACE (SMVD:255). ---- */
int orc_convolve_direct(const float* img, const int64_t dim[3], float* psf, const int64_t kdim[3], float* out);
/* the same sum at n listed voxels (idx = x + Nx*(y + Ny*z)); psf as given (not normalised here); fp64 results */
int orc_convolve_direct_at(const float* img, const int64_t dim[3], const float* psf, const int64_t kdim[3],
                           const int64_t* idx, int64_t n, double* out);

/* ---- SMVD:181-251 extractSlices / poissonProcess ----
 * mode 0: SNR<0 -> pure strided copy (noise flag ignored)
 * rng_mode 0: reference-exact (java.util.Random stream shared across slices, Q10)
 * rng_mode 1: counter-based (Philox key=seed, stream=view, counter=source voxel index) */
int64_t orc_extract_nz(int64_t nz, int inc);
int orc_extract_slices_ref(const float* in, const int64_t dim[3], int inc, float snr,
                           orc_jrandom* rnd, float* out);
int orc_extract_slices_counter(const float* in, const int64_t dim[3], int inc, float snr,
                               uint64_t seed, uint32_t stream, float* out);
/* counter mode on a window of planes z0 .. z0 + dim[2] - 1 of the source volume (z0 % inc == 0): counters = indices in the full volume */
int orc_extract_slices_counter_window(const float* in, const int64_t dim[3], int inc, float snr,
                                      uint64_t seed, uint32_t stream, int64_t z0, float* out);
double orc_poisson_mul(double snr);                /* Tools:76 */

/* ---- "next" items: SMVD:144-171 makeIsotropic, SMVD:280-316 computeWeightImage ---- */
int64_t orc_isotropic_nz(int64_t nz_acq, int inc);
int orc_make_isotropic(const float* in, const int64_t dim[3], int inc, float* out);
int orc_compute_weight_image(const int64_t dim[3], float* out);
/* ---- "next" rank 3: ground-truth phantom, SMVD:366-522.
 *      drawSpheres walks an ImgLib2 HyperSphereCursor (imglib2-algorithm, absent from /root/reference): the
 *      iteration order and the integer radii are restated from the published algorithm as recalled --
 *      raster order, last dimension outermost, nested radii r[d-1] = (long)sqrt(r[d]^2 - pos_d^2) -- and are
 *      specification-to-test like the rest of this oracle (PARITY UNPINNED).
 *      n_spheres (may be NULL) receives the number of small spheres drawn. ---- */
int orc_draw_spheres(float* img, const int64_t dim[3], double min_value, double max_value, int scale,
                     int half_pixel_offset, orc_jrandom* rnd, int64_t* n_spheres);
/* out has dim[d]/2 - 1 samples per dimension */
int orc_downsample2x(const float* in, const int64_t dim[3], float* out);
int64_t orc_hypersphere_size(int64_t radius);       /* voxels a 3-D HyperSphere of that radius iterates */

/* ---- SMVD:615-640 cross-view weight normalisation (in place) ---- */
int orc_normalize_weights(float* const* weights, int n_views, int64_t n, float osem);

#ifdef __cplusplus
}
#endif
#endif
