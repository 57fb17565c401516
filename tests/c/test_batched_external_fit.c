/* A BATCH of models the descriptor language cannot express, FITTED through the C ABI by reverse communication
 * (VERDICT round 4, row J3).
 *
 * The reference's `LevMarSolver::fit` (/root/reference/src/solvers/levmar/mod.rs:238-254) runs
 * levenberg_marquardt::LevenbergMarquardt::minimize (:247) over ANY `SeparableNonlinearModel`
 * (/root/reference/src/model/mod.rs:239-363).  Here B such problems are fitted at once: the model is a pair of plain C
 * callbacks on the HOST (eval() -> Phi, the non-zero columns of eval_partial_deriv(k) -> dPhi), the LM driver of every
 * problem lives on the device:
 *     vp_fit_begin(h, opts, alpha0, flags)
 *     repeat:  Phi / dPhi at alpha_trial for every problem that wants them  ->  vp_fit_step_with_basis  ->  alpha_trial, want
 *     vp_fit_end(h, alpha, C, report)
 * Only alpha_trial [B][q] and one word per problem come back per step; J [B][q][m] never exists in memory.
 * Checked against the oracle's own fit of every problem GIVEN THE SAME CALLBACKS (vpo_problem_set_external_model):
 * the same termination class on every problem, the same minimum (objective to 1e-9 relative, alpha to 1e-6 of max|alpha|),
 * evaluation counts within 3; both protocols (derivatives with every step / only at accepted points) give bit-identical
 * results, and the second asks for derivative columns exactly as often as the reference's driver calls jacobian().
 * Model: c1 Gauss(mu1, s1) + c2 Lorentz(mu2, g2) + c3  (n = 3, q = 4, 4 dependency pairs), m = 600, B = 64, unit and
 * per-row weights.
 * usage: test_batched_external_fit       (prints "no device" and exits 0 without a GPU) */
#include "varpro_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define M 600
#define NB 3
#define NQ 4
#define NP 4
#define B 64
static const int32_t PAIR_BASIS[NP] = {0, 0, 1, 1};
static const int32_t PAIR_PARAM[NP] = {0, 1, 2, 3};
static double X[M];

/* == SeparableNonlinearModel::eval (src/model/mod.rs:308): Phi m x n, column-major */
static void model_eval(void *user, const double *a, double *Phi) {
    int i;
    (void)user;
    for (i = 0; i < M; ++i) {
        const double u = (X[i] - a[0]) / a[1], d = X[i] - a[2];
        Phi[i] = exp(-0.5 * u * u);
        Phi[M + i] = a[3] * a[3] / (d * d + a[3] * a[3]);
        Phi[2 * M + i] = 1.0;
    }
}
/* the non-zero columns of eval_partial_deriv(k) (src/model/mod.rs:359-362) in pair order */
static void model_pairs(const double *a, double *dPhi) {
    int i;
    for (i = 0; i < M; ++i) {
        const double dx = X[i] - a[0], u = dx / a[1], g = exp(-0.5 * u * u);
        const double d = X[i] - a[2], den = d * d + a[3] * a[3];
        dPhi[i] = g * dx / (a[1] * a[1]);
        dPhi[M + i] = g * dx * dx / (a[1] * a[1] * a[1]);
        dPhi[2 * M + i] = 2.0 * a[3] * a[3] * d / (den * den);
        dPhi[3 * M + i] = 2.0 * a[3] * d * d / (den * den);
    }
}
static void model_dphi(void *user, const double *a, int k, double *Dk) {
    static double dPhi[NP * M];
    int p;
    (void)user;
    model_pairs(a, dPhi);
    memset(Dk, 0, sizeof(double) * M * NB);
    for (p = 0; p < NP; ++p)
        if (PAIR_PARAM[p] == k) memcpy(Dk + (size_t)PAIR_BASIS[p] * M, dPhi + (size_t)p * M, sizeof(double) * M);
}

static double lcg(unsigned long long *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) / 9007199254740992.0;
}

static double Y[B * M], W[M], PHI[B * NB * M], DPHI[B * NP * M];

/* one stepped fit of the whole batch; returns the number of steps (<0: error); n_deriv[b] counts the steps in which problem
 * b was asked for derivative columns */
static int stepped_fit(vp_batch *h, const double *alpha0, int flags, double *alpha, double *C, vp_report *rep, int *n_deriv) {
    static double trial[B * NQ];
    static int32_t want[B];
    int64_t nact = B;
    int steps = 0, b;
    vp_lm_opts o;
    vp_lm_opts_default(&o, VP_F64);
    if (vp_fit_begin(h, &o, alpha0, flags) != 0) return -1;
    memcpy(trial, alpha0, sizeof trial);
    for (b = 0; b < B; ++b) {
        want[b] = VP_WANT_BASIS | VP_WANT_DERIVATIVES;
        n_deriv[b] = 0;
    }
    while (nact > 0 && steps < 2000) {
        for (b = 0; b < B; ++b) { /* the caller's model, only where asked for */
            if (want[b] & VP_WANT_BASIS) model_eval(NULL, trial + b * NQ, PHI + (size_t)b * NB * M);
            if (want[b] & VP_WANT_DERIVATIVES) {
                model_pairs(trial + b * NQ, DPHI + (size_t)b * NP * M);
                n_deriv[b]++;
            }
        }
        if (vp_fit_step_with_basis(h, PHI, DPHI, trial, want, &nact) != 0) return -1;
        ++steps;
    }
    if (vp_fit_end(h, alpha, C, rep) != 0) return -1;
    return steps;
}

static int run_case(const char *name, const double *w, const double *guess) {
    static double a1[B * NQ], a2[B * NQ], C1[B * NB], C2[B * NB];
    static vp_report r1[B], r2[B];
    static int nd1[B], nd2[B];
    vp_model_desc shape;
    vp_lm_opts o;
    vp_batch *h = 0;
    int failures = 0, b, k, steps1, steps2, same_evals = 0, build_err = 0;
    long sum_dev = 0, sum_orc = 0;
    vp_lm_opts_default(&o, VP_F64);
    if (vp_batch_create_external(&h, NB, NQ, NP, PAIR_BASIS, PAIR_PARAM, VP_F64, M, 1, B, Y, w, -1.0, VP_FLAG_OWN_STREAM, 0, 0) != 0) {
        printf("%s: vp_batch_create_external failed: %s\n", name, vp_last_error());
        return 1;
    }
    steps1 = stepped_fit(h, guess, 0, a1, C1, r1, nd1);
    steps2 = stepped_fit(h, guess, VP_FIT_DERIVATIVES_ON_ACCEPT, a2, C2, r2, nd2);
    if (steps1 < 0 || steps2 < 0) {
        printf("%s: stepped fit failed: %s\n", name, vp_last_error());
        return 1;
    }
    if (memcmp(a1, a2, sizeof a1) || memcmp(C1, C2, sizeof C1) || memcmp(r1, r2, sizeof r1)) {
        printf("  %s: the two protocols disagree\n", name);
        ++failures;
    }
    memset(&shape, 0, sizeof shape);
    shape.n_basis = NB;
    shape.n_params = NQ;
    for (b = 0; b < B; ++b) {
        vp_report ro;
        double amax = 0.0;
        vpo_problem *op = vpo_problem_create(&shape, M, 1, NULL, Y + (size_t)b * M, w, -1.0, &build_err);
        if (!op) return 1;
        vpo_problem_set_external_model(op, model_eval, model_dphi, NULL);
        vpo_set_params(op, guess + b * NQ);
        vpo_fit(op, &o, &ro);
        sum_dev += r1[b].n_evals;
        sum_orc += ro.n_evals;
        same_evals += r1[b].n_evals == ro.n_evals;
        if ((r1[b].termination > 0) != (ro.termination > 0)) {
            printf("  %s: problem %d termination %d vs oracle %d\n", name, b, r1[b].termination, ro.termination);
            ++failures;
        } else if (ro.termination > 0) {
            if (abs(r1[b].n_evals - ro.n_evals) > 3) {
                printf("  %s: problem %d evaluations %d vs oracle %d\n", name, b, r1[b].n_evals, ro.n_evals);
                ++failures;
            }
            if (fabs(r1[b].objective - ro.objective) > 1e-9 * ro.objective) {
                printf("  %s: problem %d objective %.15g vs oracle %.15g\n", name, b, r1[b].objective, ro.objective);
                ++failures;
            }
            for (k = 0; k < NQ; ++k) amax = fmax(amax, fabs(op->alpha[k]));
            for (k = 0; k < NQ; ++k)
                if (fabs(a1[b * NQ + k] - op->alpha[k]) > 1e-6 * amax) {
                    printf("  %s: problem %d alpha[%d] %.15g vs oracle %.15g\n", name, b, k, a1[b * NQ + k], op->alpha[k]);
                    ++failures;
                }
            for (k = 0; k < NB; ++k)
                if (fabs(C1[b * NB + k] - op->C[k]) > 1e-6 * fmax(fabs(op->C[0]), fabs(op->C[1]))) {
                    printf("  %s: problem %d c[%d] %.15g vs oracle %.15g\n", name, b, k, C1[b * NB + k], op->C[k]);
                    ++failures;
                }
            /* the driver's own call order: derivative columns where the reference calls jacobian().  Equal evaluation counts do
             * not make two trajectories identical -- near the minimum an accept / reject decision hangs on the last digits of
             * ||r|| -- so the counts may differ by one accepted point; that the device asks for derivatives at ITS accepted
             * points and nowhere else is asserted exactly in tests/test_gpu_extfit.py */
            if (r1[b].n_evals == ro.n_evals && abs(nd2[b] - (int)op->n_jacobians) > 1) {
                printf("  %s: problem %d derivative requests %d vs the oracle's jacobian() calls %ld\n", name, b, nd2[b], op->n_jacobians);
                ++failures;
            }
        }
        vpo_problem_destroy(op);
    }
    printf("%-28s batched over the C ABI: %d problems, %d steps (eager) / %d steps (derivatives on accept), evaluations %ld vs oracle %ld, "
           "equal counts on %d\n", name, B, steps1, steps2, sum_dev, sum_orc, same_evals);
    vp_batch_destroy(h);
    if (failures) printf("  %s: %d failure(s)\n", name, failures);
    return failures;
}

int main(void) {
    static double guess[B * NQ], Phi[NB * M];
    unsigned long long seed = 777;
    int failures = 0, i, b, k;
    if (vp_device_count() <= 0) {
        printf("no device: the batched external fit needs a GPU (the C ABI has no CPU path)\n");
        return 0;
    }
    for (i = 0; i < M; ++i) {
        X[i] = 10.0 * (double)i / (double)(M - 1);
        W[i] = 0.5 + lcg(&seed);
    }
    for (b = 0; b < B; ++b) {
        double truth[NQ];
        const double c1 = 5.0 + 45.0 * lcg(&seed), c2 = 5.0 + 45.0 * lcg(&seed), c3 = 5.0 * lcg(&seed);
        truth[0] = 2.5 + lcg(&seed);
        truth[1] = 0.4 + 0.5 * lcg(&seed);
        truth[2] = 6.0 + lcg(&seed);
        truth[3] = 0.5 + 0.7 * lcg(&seed);
        model_eval(NULL, truth, Phi);
        for (i = 0; i < M; ++i)
            Y[(size_t)b * M + i] = c1 * Phi[i] + c2 * Phi[M + i] + c3 + 0.3 * (lcg(&seed) + lcg(&seed) + lcg(&seed) - 1.5);
        for (k = 0; k < NQ; ++k) guess[b * NQ + k] = truth[k] * (0.9 + 0.2 * lcg(&seed));
    }
    failures += run_case("unit weights", NULL, guess);
    failures += run_case("per-row weights", W, guess);
    printf("%d failure(s)\n", failures);
    return failures != 0;
}
