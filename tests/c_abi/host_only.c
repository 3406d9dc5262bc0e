/* Host-only leg of the C ABI (include/mvsim.h) for a sanitizer build: every entry point that needs no GPU, called from
 * a consumer compiled with -fsanitize=address,undefined (the library's own host code runs inside the same process, so
 * its reads and writes of the caller's buffers are checked against exactly sized heap blocks).
 * Exit code 0 = clean.  A GPU may or may not be present: mvsim_create must either work or fail with MVSIM_ENODEV. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvsim.h"

#define REQUIRE(cond)                                                               \
    do {                                                                            \
        if (!(cond)) { fprintf(stderr, "line %d: %s\n", __LINE__, #cond); return 1; } \
    } while (0)

int main(void)
{
    REQUIRE(strstr(mvsim_version(), "gfx950") != NULL);
    /* SimulateMultiViewDataset.java:80-102: T(+c) R T(-c), integer-division centre, float-radian angle */
    int64_t* dim = (int64_t*)malloc(3 * sizeof(int64_t));
    double* m = (double*)malloc(12 * sizeof(double));
    dim[0] = 512; dim[1] = 512; dim[2] = 512;
    REQUIRE(mvsim_axis_rotation(dim, 0, 90, m) == MVSIM_OK);
    REQUIRE(m[0] == 1.0 && m[3] == 0.0 && fabs(m[5] + 4.371139e-08) < 1e-13 && fabs(m[6] + 1.0) < 1e-12);
    REQUIRE(fabs(m[7] - (255.0 + 255.0)) < 1e-4);                    /* centre 255, not 255.5 */
    REQUIRE(mvsim_axis_rotation(dim, 3, 90, m) == MVSIM_EINVAL && strlen(mvsim_last_error()) > 0);
    REQUIRE(mvsim_extract_nz(512, 3) == 171 && mvsim_extract_nz(512, 4) == 128 && mvsim_extract_nz(5, 0) == -1);
    REQUIRE(mvsim_isotropic_nz(171, 3) == 511);
    REQUIRE(mvsim_poisson_mul(25.0) == 124.99999999999997);           /* Tools.java:76 */
    mvsim_view_params* p = (mvsim_view_params*)malloc(sizeof(mvsim_view_params));
    mvsim_view_params_default(p);
    REQUIRE(p->delta == (double)0.01f && p->inc == 3 && p->snr == 25.0f && p->seed == 464232194ull && p->min_value == 1e-4f);
    /* views shard v % nranks */
    int* idx = (int*)malloc(3 * sizeof(int));
    REQUIRE(mvsim_shard_views(8, 3, 1, idx, 3) == 3 && idx[0] == 1 && idx[1] == 4 && idx[2] == 7);
    REQUIRE(mvsim_shard_views(8, 3, 2, idx, 1) == 2 && idx[0] == 2);   /* capacity 1: count still reported */
    REQUIRE(mvsim_shard_views(8, 0, 0, idx, 3) < 0);
    int64_t z0 = -1, z1 = -1;
    REQUIRE(mvsim_slab_range(1024, 8, 7, &z0, &z1) == MVSIM_OK && z0 == 896 && z1 == 1024);
    int64_t* kdim = (int64_t*)malloc(3 * sizeof(int64_t));
    int64_t* geo = (int64_t*)malloc(5 * sizeof(int64_t));
    kdim[0] = kdim[1] = kdim[2] = 31;
    REQUIRE(mvsim_fft_geometry(dim, kdim, geo) == MVSIM_OK && geo[0] == 560 && geo[1] == 560 && geo[2] == 512 && geo[4] == 1);
    int count = -1;
    const int rc = mvsim_device_count(&count);
    REQUIRE((rc == MVSIM_OK && count >= 0) || rc == MVSIM_ENODEV);
    mvsim_ctx* ctx = NULL;
    const int rc2 = mvsim_create(0, &ctx);
    REQUIRE((rc2 == MVSIM_OK && ctx != NULL) || (rc2 == MVSIM_ENODEV && ctx == NULL));   /* no CPU fallback */
    if (ctx) {
        REQUIRE(mvsim_set_option(ctx, "fft_zpass", "fft") == MVSIM_OK);
        REQUIRE(mvsim_set_option(ctx, "fft_zpass", "sideways") == MVSIM_EINVAL);
        REQUIRE(mvsim_destroy(ctx) == MVSIM_OK);
    }
    free(dim); free(m); free(p); free(idx); free(kdim); free(geo);
    printf("c-abi host-only sanitizer run ok (%s)\n", rc2 == MVSIM_OK ? "GPU present" : "no GPU: MVSIM_ENODEV");
    return 0;
}
