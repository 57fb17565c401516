/* Throughput mode through the C ABI, without Python: a stream of same-shaped batches fitted with TWO batches in flight --
 * two device-pointer handles, each bound to its own HIP stream (one handle <-> one stream, include/varpro_hip.h), batch k
 * enqueued on handle k mod 2: vp_set_observations + vp_fit + vp_summary_device, nothing waits on the host.  The straggler
 * tail of one launch (its longest fits running alone) then overlaps the bulk of the next batch; this is the schedule
 * bench.py quotes as `value` and varpro_amd.FitPipeline packages.
 * Checked: every batch's parameters and reports are bit-identical (its 4-double summary equal: counts exactly, the cost sum
 * to rounding) to the same batch fitted alone on a third handle with a host synchronisation after every step.  The two wall
 * times are printed (6 x 8 192 fits: 8.5 ms one at a time with a host wait per step, 3.3 ms with two in flight).
 * Reference: independent problems, /root/reference/src/solvers/levmar/mod.rs:238-254 (one fit each).
 * usage: test_two_in_flight      (prints "no device" and exits 0 without a GPU) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "varpro_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_VP(x) do { int r_ = (x); if (r_ != 0) { printf("%s: %s\n", #x, vp_last_error()); return 1; } } while (0)

static double lcg(unsigned long long *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) / 9007199254740992.0;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

enum { B = 8192, M = 1024, NBATCH = 6 };

int main(void) {
    int failures = 0, i, b, k;
    unsigned long long seed = 4242;
    vp_model_desc mdl;
    hipStream_t st[3];
    vp_batch *h[3] = {0, 0, 0};
    double *t, *Y, *a0;
    double *d_t, *d_Y[NBATCH], *d_a0[NBATCH], *d_a[2][NBATCH], *d_sum[2][NBATCH];
    vp_report *d_rep[2][NBATCH];
    double t_seq, t_two;
    if (vp_device_count() <= 0) {
        printf("no device: the test needs a GPU (the C ABI has no CPU path)\n");
        return 0;
    }
    CHECK_HIP(hipSetDevice(0));
    for (i = 0; i < 3; ++i) CHECK_HIP(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    memset(&mdl, 0, sizeof mdl);
    mdl.n_basis = 3;
    mdl.n_params = 2;
    mdl.kind[0] = VP_BASIS_EXP_DECAY; mdl.param[0][0] = 0; mdl.param[0][1] = -1;
    mdl.kind[1] = VP_BASIS_EXP_DECAY; mdl.param[1][0] = 1; mdl.param[1][1] = -1;
    mdl.kind[2] = VP_BASIS_CONST; mdl.param[2][0] = -1; mdl.param[2][1] = -1;

    t = (double *)malloc(sizeof(double) * M);
    Y = (double *)malloc(sizeof(double) * (size_t)B * M);
    a0 = (double *)malloc(sizeof(double) * B * 2);
    for (i = 0; i < M; ++i) t[i] = 12.5 * i / (M - 1);
    CHECK_HIP(hipMalloc((void **)&d_t, sizeof(double) * M));
    CHECK_HIP(hipMemcpy(d_t, t, sizeof(double) * M, hipMemcpyHostToDevice));
    for (k = 0; k < NBATCH; ++k) { /* NBATCH different batches, resident on the device */
        for (b = 0; b < B; ++b) {
            const double t1 = 0.5 + 1.5 * lcg(&seed), t2 = 2.5 + 5.5 * lcg(&seed), c1 = 100 * lcg(&seed), c2 = 100 * lcg(&seed), c3 = 100 * lcg(&seed);
            for (i = 0; i < M; ++i) Y[(size_t)b * M + i] = c1 * exp(-t[i] / t1) + c2 * exp(-t[i] / t2) + c3 + 0.1 * (lcg(&seed) - 0.5);
            a0[b * 2] = t1 * (0.75 + 0.5 * lcg(&seed));
            a0[b * 2 + 1] = t2 * (0.75 + 0.5 * lcg(&seed));
        }
        CHECK_HIP(hipMalloc((void **)&d_Y[k], sizeof(double) * (size_t)B * M));
        CHECK_HIP(hipMalloc((void **)&d_a0[k], sizeof(double) * B * 2));
        CHECK_HIP(hipMemcpy(d_Y[k], Y, sizeof(double) * (size_t)B * M, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_a0[k], a0, sizeof(double) * B * 2, hipMemcpyHostToDevice));
        for (i = 0; i < 2; ++i) {
            CHECK_HIP(hipMalloc((void **)&d_a[i][k], sizeof(double) * B * 2));
            CHECK_HIP(hipMalloc((void **)&d_sum[i][k], sizeof(double) * 4));
            CHECK_HIP(hipMalloc((void **)&d_rep[i][k], sizeof(vp_report) * B));
        }
    }
    for (i = 0; i < 3; ++i)
        CHECK_VP(vp_batch_create(&h[i], &mdl, VP_F64, M, 1, B, d_t, d_Y[0], NULL, -1.0, VP_FLAG_DEVICE_PTRS, 0, st[i]));

    /* ---- reference run: one batch at a time on handle 2, host synchronisation after every step ---- */
    t_seq = now_ms();
    for (k = 0; k < NBATCH; ++k) {
        CHECK_HIP(hipMemcpyAsync(d_a[0][k], d_a0[k], sizeof(double) * B * 2, hipMemcpyDeviceToDevice, st[2]));
        CHECK_VP(vp_set_observations(h[2], d_Y[k]));
        CHECK_VP(vp_fit(h[2], NULL, d_a[0][k], NULL, d_rep[0][k]));
        CHECK_VP(vp_summary_device(h[2], d_sum[0][k]));
        CHECK_HIP(hipStreamSynchronize(st[2]));
    }
    t_seq = now_ms() - t_seq;

    /* ---- two batches in flight: batch k on handle / stream k mod 2, no host wait until the end ---- */
    t_two = now_ms();
    for (k = 0; k < NBATCH; ++k) {
        const int j = k & 1;
        CHECK_HIP(hipMemcpyAsync(d_a[1][k], d_a0[k], sizeof(double) * B * 2, hipMemcpyDeviceToDevice, st[j]));
        CHECK_VP(vp_set_observations(h[j], d_Y[k]));
        CHECK_VP(vp_fit(h[j], NULL, d_a[1][k], NULL, d_rep[1][k]));
        CHECK_VP(vp_summary_device(h[j], d_sum[1][k]));
    }
    CHECK_HIP(hipStreamSynchronize(st[0]));
    CHECK_HIP(hipStreamSynchronize(st[1]));
    t_two = now_ms() - t_two;

    {
        double *a_s = (double *)malloc(sizeof(double) * B * 2), *a_p = (double *)malloc(sizeof(double) * B * 2);
        vp_report *r_s = (vp_report *)malloc(sizeof(vp_report) * B), *r_p = (vp_report *)malloc(sizeof(vp_report) * B);
        double s_s[4], s_p[4], fits_ok = 0;
        for (k = 0; k < NBATCH; ++k) {
            CHECK_HIP(hipMemcpy(a_s, d_a[0][k], sizeof(double) * B * 2, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(a_p, d_a[1][k], sizeof(double) * B * 2, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(r_s, d_rep[0][k], sizeof(vp_report) * B, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(r_p, d_rep[1][k], sizeof(vp_report) * B, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(s_s, d_sum[0][k], sizeof s_s, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(s_p, d_sum[1][k], sizeof s_p, hipMemcpyDeviceToHost));
            if (memcmp(a_s, a_p, sizeof(double) * B * 2) != 0) { printf("  batch %d: parameters differ\n", k); ++failures; }
            if (memcmp(r_s, r_p, sizeof(vp_report) * B) != 0) { printf("  batch %d: reports differ\n", k); ++failures; }
            /* counts exactly; the cost sum is accumulated with atomics (summation order not fixed): to rounding */
            if (s_s[1] != s_p[1] || s_s[2] != s_p[2] || s_s[3] != s_p[3] || fabs(s_s[0] - s_p[0]) > 1e-12 * fabs(s_s[0])) {
                printf("  batch %d: summaries differ (%.17g %.0f %.0f %.0f vs %.17g %.0f %.0f %.0f)\n", k, s_s[0], s_s[1], s_s[2], s_s[3],
                       s_p[0], s_p[1], s_p[2], s_p[3]);
                ++failures;
            }
            if (!(s_p[1] + s_p[2] == B && s_p[1] >= 0.95 * B && s_p[3] >= B)) { printf("  batch %d: implausible aggregates\n", k); ++failures; }
            fits_ok += s_p[1];
        }
        printf("%d batches of %d fits: %.0f converged; one at a time (host sync per step) %.2f ms, two in flight %.2f ms\n",
               NBATCH, B, fits_ok, t_seq, t_two);
        free(a_s); free(a_p); free(r_s); free(r_p);
    }
    for (i = 0; i < 3; ++i) { vp_batch_destroy(h[i]); (void)hipStreamDestroy(st[i]); }
    printf("%d failure(s)\n", failures);
    free(t); free(Y); free(a0);
    return failures ? 1 : 0;
}
