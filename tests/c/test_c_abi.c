/* The C ABI from plain C99 (gcc -std=c99 -pedantic): the header must be valid C, the library must link without any
 * C++ or HIP headers on the caller's side, and without a GPU every compute entry point must fail loudly
 * (VP_ERR_NO_DEVICE) -- there is no CPU fallback.  With a GPU the same program runs the reference's smallest
 * end-to-end case: one double-exponential fit.  usage: test_c_abi [expect_gpu] */
#include "varpro_hip.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

int main(int argc, char **argv) {
    const int expect_gpu = argc > 1 && strcmp(argv[1], "expect_gpu") == 0;
    int failures = 0;
    vp_lm_opts o;
    vp_model_desc d;
    double t[64], y[64], alpha[2] = {1.5, 4.0}, c[3];
    vp_report rep;
    vp_batch *h = 0;
    int i, rc;
    vp_lm_opts_default(&o, VP_F64);
    if (!(o.patience == 100 && o.stepbound == 100.0)) ++failures;
    memset(&d, 0, sizeof d);
    d.n_basis = 3;
    d.n_params = 2;
    d.kind[0] = VP_BASIS_EXP_DECAY; d.param[0][0] = 0; d.param[0][1] = -1;
    d.kind[1] = VP_BASIS_EXP_DECAY; d.param[1][0] = 1; d.param[1][1] = -1;
    d.kind[2] = VP_BASIS_CONST;     d.param[2][0] = -1; d.param[2][1] = -1;
    for (i = 0; i < 64; ++i) {
        t[i] = 12.5 * i / 63.0;
        y[i] = 4.0 * exp(-t[i] / 1.0) + 2.5 * exp(-t[i] / 3.0) + 1.0;
    }
    rc = vp_batch_create(&h, &d, VP_F64, 64, 1, 1, t, y, 0, -1.0, VP_FLAG_OWN_STREAM, 0, 0);
    if (vp_device_count() <= 0) {
        if (expect_gpu) ++failures;
        if (rc != -4 || h != 0) ++failures; /* VP_ERR_NO_DEVICE, never a fallback */
        printf("no device: create rc=%d (%s)\n", rc, vp_last_error());
    } else {
        if (rc != 0) {
            printf("create failed: %s\n", vp_last_error());
            return 1;
        }
        rc = vp_fit(h, &o, alpha, c, &rep);
        if (rc != 0 || rep.termination <= 0) ++failures;
        if (fabs(alpha[0] - 1.0) > 1e-8 || fabs(alpha[1] - 3.0) > 1e-8) ++failures;
        if (fabs(c[0] - 4.0) > 1e-7 || fabs(c[1] - 2.5) > 1e-7 || fabs(c[2] - 1.0) > 1e-7) ++failures;
        printf("fit: tau = (%.12f, %.12f), c = (%.9f, %.9f, %.9f), %d evaluations\n", alpha[0], alpha[1], c[0], c[1], c[2],
               rep.n_evals);
        vp_batch_destroy(h);
    }
    printf("%d failure(s)\n", failures);
    return failures != 0;
}
