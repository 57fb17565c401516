/* The collective path WITHOUT Python ("next" item 8): a plain C host that links
 * librccl itself, one rank (ncclCommInitAll on device 0 -- the GPU box has one device; the call sequence is the one every
 * rank of an N-GPU job makes).
 *   (1) independent fits: vp_fit -> vp_summary_device (4 DEVICE doubles on the handle's stream) -> ncclAllReduce of those 32
 *       bytes on the same stream -> equal to vp_summary's host numbers; and the packaged form vp_reduce_cost(h, comm, out4).
 *   (2) one global fit whose right-hand sides are sharded (vp_set_rhs_allreduce): the callback the library calls once per LM
 *       evaluation issues ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream) -- compared with the same fit
 *       on an unsharded handle (identical alpha / objective to the tolerances of tests/test_gpu_sharded_global_fit.py).
 * Reference: the scalar LM cost reduction and the per-evaluation exchange of BASELINE.json:north_star; the fit itself is
 * /root/reference/src/solvers/levmar/mod.rs:238-254.
 * usage: test_rccl_summary      (prints "no device" and exits 0 without a GPU) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "varpro_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { printf("%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define CHECK_VP(x) do { int r_ = (x); if (r_ != 0) { printf("%s: %s\n", #x, vp_last_error()); return 1; } } while (0)

static double lcg(unsigned long long *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) / 9007199254740992.0;
}

typedef struct {
    ncclComm_t comm;
    int calls;
} ar_ctx;

/* vp_allreduce_fn: sum `count` DEVICE doubles in place over all ranks, ordered on `hip_stream` */
static int allreduce_cb(void *dev_doubles, int64_t count, void *hip_stream, void *user) {
    ar_ctx *c = (ar_ctx *)user;
    c->calls++;
    return ncclAllReduce(dev_doubles, dev_doubles, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)hip_stream) == ncclSuccess ? 0 : -1;
}

static void double_exp_model(vp_model_desc *d, int offset) {
    memset(d, 0, sizeof *d);
    d->n_basis = offset ? 3 : 2;
    d->n_params = 2;
    d->kind[0] = VP_BASIS_EXP_DECAY; d->param[0][0] = 0; d->param[0][1] = -1;
    d->kind[1] = VP_BASIS_EXP_DECAY; d->param[1][0] = 1; d->param[1][1] = -1;
    if (offset) { d->kind[2] = VP_BASIS_CONST; d->param[2][0] = -1; d->param[2][1] = -1; }
}

int main(void) {
    enum { B = 512, M = 256, S = 40, M2 = 512 };
    int failures = 0, i, b, dev0 = 0;
    unsigned long long seed = 99;
    ncclComm_t comm;
    hipStream_t stream;
    vp_model_desc m3, m2;
    if (vp_device_count() <= 0) {
        printf("no device: the RCCL test needs a GPU (the C ABI has no CPU path)\n");
        return 0;
    }
    CHECK_HIP(hipSetDevice(0));
    CHECK_NCCL(ncclCommInitAll(&comm, 1, &dev0));
    CHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    double_exp_model(&m3, 1);
    double_exp_model(&m2, 0);

    /* ---- (1) independent fits on device pointers, the 4-double reduction through RCCL ---- */
    {
        static double t[M], Y[B * M], a0[B * 2];
        double *d_t, *d_Y, *d_a, *d_sum4, loc[4], red[4], pk[4];
        vp_batch *h = 0;
        for (i = 0; i < M; ++i) t[i] = 12.5 * i / (M - 1);
        for (b = 0; b < B; ++b) {
            const double t1 = 0.5 + 1.5 * lcg(&seed), t2 = 2.5 + 5.5 * lcg(&seed), c1 = 100 * lcg(&seed), c2 = 100 * lcg(&seed), c3 = 100 * lcg(&seed);
            for (i = 0; i < M; ++i) Y[b * M + i] = c1 * exp(-t[i] / t1) + c2 * exp(-t[i] / t2) + c3 + 0.05 * (lcg(&seed) - 0.5);
            a0[b * 2] = t1 * (0.8 + 0.4 * lcg(&seed));
            a0[b * 2 + 1] = t2 * (0.8 + 0.4 * lcg(&seed));
        }
        CHECK_HIP(hipMalloc((void **)&d_t, sizeof t));
        CHECK_HIP(hipMalloc((void **)&d_Y, sizeof Y));
        CHECK_HIP(hipMalloc((void **)&d_a, sizeof a0));
        CHECK_HIP(hipMalloc((void **)&d_sum4, 4 * sizeof(double)));
        CHECK_HIP(hipMemcpy(d_t, t, sizeof t, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_Y, Y, sizeof Y, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_a, a0, sizeof a0, hipMemcpyHostToDevice));
        CHECK_VP(vp_batch_create(&h, &m3, VP_F64, M, 1, B, d_t, d_Y, NULL, -1.0, VP_FLAG_DEVICE_PTRS, 0, stream));
        CHECK_VP(vp_fit(h, NULL, d_a, NULL, NULL));
        /* everything below is enqueued on `stream` with no host synchronisation in between */
        CHECK_VP(vp_summary_device(h, d_sum4));
        CHECK_NCCL(ncclAllReduce(d_sum4, d_sum4, 4, ncclDouble, ncclSum, comm, stream));
        CHECK_HIP(hipMemcpyAsync(red, d_sum4, sizeof red, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_VP(vp_summary(h, loc));
        CHECK_VP(vp_reduce_cost(h, comm, pk));
        printf("independent fits: sum cost %.9e  ok %.0f  failed %.0f  evaluations %.0f  (RCCL-reduced over 1 rank)\n", red[0], red[1], red[2], red[3]);
        for (i = 0; i < 4; ++i) {
            if (red[i] != loc[i]) { printf("  summary_device + ncclAllReduce [%d]: %.17g vs vp_summary %.17g\n", i, red[i], loc[i]); ++failures; }
            if (pk[i] != loc[i]) { printf("  vp_reduce_cost [%d]: %.17g vs vp_summary %.17g\n", i, pk[i], loc[i]); ++failures; }
        }
        if (!(red[1] + red[2] == B && red[1] >= 0.9 * B && red[3] >= B)) { printf("  implausible aggregates\n"); ++failures; }
        vp_batch_destroy(h);
        (void)hipFree(d_t); (void)hipFree(d_Y); (void)hipFree(d_a); (void)hipFree(d_sum4);
    }

    /* ---- (2) a global fit with right-hand-side sharding: the per-evaluation exchange is ncclAllReduce ---- */
    {
        static double t[M2], Y[S * M2];
        double *d_t, *d_Y, *d_a1, *d_a2, a0[2] = {2.0, 6.5}, a1[2], a2[2];
        vp_report r1, r2, *d_r1, *d_r2;
        vp_batch *h1 = 0, *h2 = 0;
        ar_ctx ctx;
        for (i = 0; i < M2; ++i) t[i] = 12.5 * i / (M2 - 1);
        for (b = 0; b < S; ++b) {
            const double c1 = 100 * lcg(&seed), c2 = 100 * lcg(&seed);
            for (i = 0; i < M2; ++i) Y[b * M2 + i] = c1 * exp(-t[i] / 1.0) + c2 * exp(-t[i] / 3.0) + 0.01 * (lcg(&seed) - 0.5);
        }
        CHECK_HIP(hipMalloc((void **)&d_t, sizeof t));
        CHECK_HIP(hipMalloc((void **)&d_Y, sizeof Y));
        CHECK_HIP(hipMalloc((void **)&d_a1, sizeof a0));
        CHECK_HIP(hipMalloc((void **)&d_a2, sizeof a0));
        CHECK_HIP(hipMalloc((void **)&d_r1, sizeof r1));
        CHECK_HIP(hipMalloc((void **)&d_r2, sizeof r2));
        CHECK_HIP(hipMemcpy(d_t, t, sizeof t, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_Y, Y, sizeof Y, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_a1, a0, sizeof a0, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_a2, a0, sizeof a0, hipMemcpyHostToDevice));
        CHECK_VP(vp_batch_create(&h1, &m2, VP_F64, M2, S, 1, d_t, d_Y, NULL, -1.0, VP_FLAG_DEVICE_PTRS, 0, stream));
        CHECK_VP(vp_batch_create(&h2, &m2, VP_F64, M2, S, 1, d_t, d_Y, NULL, -1.0, VP_FLAG_DEVICE_PTRS, 0, stream));
        CHECK_VP(vp_fit(h1, NULL, d_a1, NULL, d_r1)); /* unsharded */
        ctx.comm = comm;
        ctx.calls = 0;
        CHECK_VP(vp_set_rhs_allreduce(h2, allreduce_cb, &ctx, S)); /* this rank holds all S columns of a 1-rank job */
        CHECK_VP(vp_fit(h2, NULL, d_a2, NULL, d_r2));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(a1, d_a1, sizeof a1, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(a2, d_a2, sizeof a2, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(&r1, d_r1, sizeof r1, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(&r2, d_r2, sizeof r2, hipMemcpyDeviceToHost));
        printf("global fit, S=%d: unsharded alpha (%.12f, %.12f) term %d evals %d obj %.12e | RCCL-reduced alpha (%.12f, %.12f) term %d evals %d obj %.12e | %d all-reduces\n",
               S, a1[0], a1[1], r1.termination, r1.n_evals, r1.objective, a2[0], a2[1], r2.termination, r2.n_evals, r2.objective, ctx.calls);
        if (r1.termination <= 0 || r2.termination <= 0) ++failures;
        if (fabs(a1[0] - a2[0]) > 1e-7 * fabs(a1[0]) || fabs(a1[1] - a2[1]) > 1e-7 * fabs(a1[1])) ++failures;
        if (fabs(r1.objective - r2.objective) > 1e-10 * r1.objective) ++failures;
        if (ctx.calls < r2.n_evals) { printf("  expected one all-reduce per evaluation\n"); ++failures; }
        vp_batch_destroy(h1);
        vp_batch_destroy(h2);
        (void)hipFree(d_t); (void)hipFree(d_Y); (void)hipFree(d_a1); (void)hipFree(d_a2); (void)hipFree(d_r1); (void)hipFree(d_r2);
    }
    (void)hipStreamDestroy(stream);
    ncclCommDestroy(comm);
    printf("%d failure(s)\n", failures);
    return failures != 0;
}
