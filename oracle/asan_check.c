/* CPU ORACLE (test infrastructure): sanitizer run of mvsim_oracle.c.
 *
 * Built by `make -C oracle asan` with -fsanitize=address,undefined and run by the `not gpu` test suite: every
 * oracle function of the per-view path runs once on small, exactly sized heap buffers, so any out-of-bounds
 * access, signed overflow or misaligned access in the restatement aborts the program.  (The GPU side has no
 * sanitizer on this pool; the host-only leg of the C ABI gets the same treatment in tests/c_abi/host_only.c.)
 * Exit code 0 = clean. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvsim_oracle.h"

static float* fbuf(int64_t n) { float* p = (float*)malloc((size_t)n * sizeof(float)); if (!p) exit(2); return p; }

int main(void)
{
    const int64_t dim[3] = {11, 13, 9}, kdim[3] = {5, 3, 7};
    const int64_t n = dim[0] * dim[1] * dim[2], k = kdim[0] * kdim[1] * kdim[2];
    orc_jrandom r;
    orc_jrandom_seed(&r, 464232194);
    float* gt = fbuf(n);
    for (int64_t i = 0; i < n; ++i) gt[i] = (float)orc_jrandom_next_double(&r);
    float *rot = fbuf(n), *att = fbuf(n), *con = fbuf(n), *psf = fbuf(k);
    for (int64_t i = 0; i < k; ++i) psf[i] = 0.1f + (float)orc_jrandom_next_double(&r);
    for (int axis = 0; axis < 3; ++axis)
        if (orc_rotate_around_axis(gt, dim, axis, 37 * (axis + 1), rot)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    if (orc_attenuate3d(rot, dim, 0.01, att)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    if (orc_convolve_direct(att, dim, psf, kdim, con)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    if (fabs(orc_sum_image(psf, k) - 1.0) > 1e-5) { fprintf(stderr, "psf not normalised\n"); { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; } }
    const double corr = orc_adjust_image(con, n, 1e-4f, 1.0f);
    if (!(corr > 0.0)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    for (int inc = 1; inc <= 4; ++inc) {
        const int64_t nzo = orc_extract_nz(dim[2], inc), no = dim[0] * dim[1] * nzo;
        float *a = fbuf(no), *b = fbuf(no), *c = fbuf(no);
        if (orc_extract_slices_counter(con, dim, inc, -1.0f, 1, 0, a)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
        if (orc_extract_slices_counter(con, dim, inc, 25.0f, 464232194ull, 3, b)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
        orc_jrandom q;
        orc_jrandom_seed(&q, 5);
        if (orc_extract_slices_ref(con, dim, inc, 3.0f, &q, c)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
        const int64_t adim[3] = {dim[0], dim[1], nzo};
        const int64_t niso = dim[0] * dim[1] * orc_isotropic_nz(nzo, inc);
        float* iso = fbuf(niso);
        if (orc_make_isotropic(b, adim, inc, iso)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
        free(a); free(b); free(c); free(iso);
    }
    float* w[3];
    for (int v = 0; v < 3; ++v) { w[v] = fbuf(n); if (orc_compute_weight_image(dim, w[v])) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; } }
    if (orc_normalize_weights(w, 3, n, 3.0f)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    /* phantom: the large sphere has radius min/2 - 47*scale - 1; scale 1 on a ~120^3 canvas, scale 2 (+ half-pixel
     * offset) on a ~200^3 one, then 2x down-sampling */
    const int64_t cdim[3] = {120, 112, 130}, cn = 120 * 112 * 130;
    float* canvas = (float*)calloc((size_t)cn, sizeof(float));
    int64_t ns = 0, ns2 = 0;
    if (orc_draw_spheres(canvas, cdim, 0.0, 1.0, 1, 0, &r, &ns)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    const int64_t ddim[3] = {200, 196, 204}, dn = 200 * 196 * 204;
    float* canvas2 = (float*)calloc((size_t)dn, sizeof(float));
    if (orc_draw_spheres(canvas2, ddim, 0.0, 1.0, 2, 1, &r, &ns2)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    ns += ns2;
    float* small = fbuf((ddim[0] / 2 - 1) * (ddim[1] / 2 - 1) * (ddim[2] / 2 - 1));
    if (orc_downsample2x(canvas2, ddim, small)) { fprintf(stderr, "fail at line %d\n", __LINE__); return 1; }
    /* samplers across the lambda regimes */
    int64_t acc = 0;
    const double lams[] = {0.0, 0.0125, 0.99, 9.99, 10.0, 125.0, 4000.0, 22000.0};
    for (unsigned i = 0; i < sizeof(lams) / sizeof(lams[0]); ++i)
        for (uint64_t idx = 0; idx < 64; ++idx) acc += orc_poisson_counter(lams[i], 464232194ull, 1, idx * 7919ull);
    for (int i = 0; i < 32; ++i) acc += orc_poisson_interarrival(&r, 2.5);
    double m[12], mi[12];
    orc_axis_rotation(dim, 1, -52, m);
    orc_affine_invert(m, mi);
    printf("oracle sanitizer run ok (%lld spheres, checksum %lld, hypersphere(3) = %lld)\n", (long long)ns, (long long)acc,
           (long long)orc_hypersphere_size(3));
    free(gt); free(rot); free(att); free(con); free(psf); free(canvas); free(canvas2); free(small);
    for (int v = 0; v < 3; ++v) free(w[v]);
    return 0;
}
