/* Plain-C consumer of the C ABI (include/mvsim.h): no Python, no torch, no C++.  Built by the tests with
 *     gcc -std=c99 -Iinclude tests/c_abi/smoke.c -Lmultiview-simulation_amd -lmvsim -Wl,-rpath,... -lm
 * Exit codes: 0 ok, 3 no usable GPU (MVSIM_ENODEV: what a CPU-only host must see), 1 anything else. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvsim.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != MVSIM_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mvsim_last_error());           \
            return rc_ == MVSIM_ENODEV ? 3 : 1;                                          \
        }                                                                                \
    } while (0)

int main(void)
{
    const int64_t dim[3] = {32, 32, 24}, kdim[3] = {5, 5, 7};
    const int64_t n = dim[0] * dim[1] * dim[2];
    mvsim_ctx* ctx = NULL;
    CHECK(mvsim_create(0, &ctx));

    float* gt = (float*)calloc((size_t)n, sizeof(float));
    float* out = (float*)malloc((size_t)n * sizeof(float));
    for (int64_t z = 8; z < 16; ++z)
        for (int64_t y = 10; y < 22; ++y)
            for (int64_t x = 12; x < 20; ++x) gt[x + dim[0] * (y + dim[1] * z)] = 1.0f + 0.01f * (float)(x + y + z);

    /* rotation by 0 degrees is the identity, bit for bit */
    CHECK(mvsim_rotate_around_axis(ctx, gt, dim, 0, 0, out));
    if (memcmp(gt, out, (size_t)n * sizeof(float)) != 0) { fprintf(stderr, "0-degree rotation is not the identity\n"); return 1; }

    /* one whole view: rotate, attenuate, convolve, adjust, extract every 3rd plane, Poisson */
    float psf[5 * 5 * 7];
    double psum = 0.0;
    for (int i = 0; i < 5 * 5 * 7; ++i) psf[i] = 1.0f + (float)(i % 7);
    mvsim_view_params p;
    mvsim_view_params_default(&p);
    p.degrees = 30; p.inc = 3; p.snr = 25.0f; p.seed = 464232194ull; p.stream = 0;
    const int64_t nzo = mvsim_extract_nz(dim[2], p.inc);
    float* acq = (float*)malloc((size_t)(dim[0] * dim[1] * nzo) * sizeof(float));
    mvsim_view_outputs o = {NULL, NULL, NULL, acq};
    double corr = 0.0;
    CHECK(mvsim_simulate_view(ctx, gt, dim, psf, kdim, &p, &o, &corr));
    for (int i = 0; i < 5 * 5 * 7; ++i) psum += (double)psf[i];
    if (fabs(psum - 1.0) > 1e-5) { fprintf(stderr, "PSF was not normalised in place (sum %.9g)\n", psum); return 1; }
    double mean = 0.0;
    for (int64_t i = 0; i < dim[0] * dim[1] * nzo; ++i) {
        if (acq[i] < 0.0f || acq[i] != floorf(acq[i])) { fprintf(stderr, "count %g is not a non-negative integer\n", acq[i]); return 1; }
        mean += acq[i];
    }
    mean /= (double)(dim[0] * dim[1] * nzo);
    if (!(corr > 0.0) || !(mean > 1.0)) { fprintf(stderr, "implausible view (corr %g, mean count %g)\n", corr, mean); return 1; }

    /* argument errors come back as status codes with a message, never as crashes */
    if (mvsim_rotate_around_axis(ctx, gt, dim, 7, 10, out) != MVSIM_EINVAL || strlen(mvsim_last_error()) == 0) {
        fprintf(stderr, "axis 7 was not rejected\n");
        return 1;
    }
    printf("c-abi smoke ok: %d planes acquired, mean count %.3f, corr %.6g, library %s\n", (int)nzo, mean, corr, mvsim_version());
    free(gt); free(out); free(acq);
    mvsim_destroy(ctx);
    return 0;
}
