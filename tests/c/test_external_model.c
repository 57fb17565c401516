/* A model the descriptor language cannot express, end to end through the C ABI.
 *
 * The reference's solver works with ANY `SeparableNonlinearModel` (/root/reference/src/model/mod.rs:239-363); its
 * closure-based `SeparableModel` (:441-512) is how users bring a Gaussian or a Lorentzian.  Here such a model is a pair of
 * plain C callbacks on the HOST -- eval() -> Phi, eval_partial_deriv(k) -> its non-zero columns -- and the handle made by
 * vp_batch_create_external does everything downstream of them on the device.  An EXTERNAL MINPACK-style driver (the
 * oracle's lmder loop, vpo_lm_minimize: compiled into this test only) plays the unchanged
 * levenberg_marquardt::LevenbergMarquardt::minimize (/root/reference/src/solvers/levmar/mod.rs:247) and calls the trait:
 *     set_params(x)  ->  model callbacks at x, vp_set_params_with_basis        (src/solvers/levmar/mod.rs:42-73)
 *     residuals()    ->  vp_residuals                                          (:91-95)
 *     jacobian()     ->  derivative callbacks, vp_jacobian_with_derivatives    (:101-201)
 * Checked against the oracle's own fit GIVEN THE SAME CALLBACKS (vpo_problem_set_external_model):
 *   (1) at the initial point c, r and every Jacobian column to 1e-10 (north_star's tolerance; J with the rounding floor
 *       relative to the un-projected column), cost to 1e-10;
 *   (2) the same termination class, the leading trial points of the trajectory to 1e-8, the same minimum (alpha to 1e-7 of
 *       max|alpha|, objective to 1e-9 relative).
 * Model: c1 Gauss(mu1, s1) + c2 Lorentz(mu2, g2) + c3  (n = 3, q = 4, 4 dependency pairs) on m = 600 samples; cases S = 1,
 * S = 1 weighted, S = 3 (Jacobian branch S <= q), S = 6 (branch S > q), and a model that returns a non-finite basis
 * (set_params latches status != 0 <=> residuals() == None -> TerminationReason::User).
 * usage: test_external_model       (prints "no device" and exits 0 without a GPU) */
#include "varpro_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define M 600
#define NB 3
#define NQ 4
#define NP 4
static const int32_t PAIR_BASIS[NP] = {0, 0, 1, 1};
static const int32_t PAIR_PARAM[NP] = {0, 1, 2, 3};

typedef struct {
    double x[M];
    int poison; /* 1: the model "fails" (non-finite basis) when mu1 < 0 */
} model_t;

/* == SeparableNonlinearModel::eval (src/model/mod.rs:308): Phi m x n, column-major */
static void model_eval(void *user, const double *a, double *Phi) {
    const model_t *md = (const model_t *)user;
    int i;
    for (i = 0; i < M; ++i) {
        const double u = (md->x[i] - a[0]) / a[1], d = md->x[i] - a[2];
        Phi[i] = exp(-0.5 * u * u);
        Phi[M + i] = a[3] * a[3] / (d * d + a[3] * a[3]);
        Phi[2 * M + i] = 1.0;
    }
    if (md->poison && a[0] < 0.0) Phi[7] = NAN;
}
/* the non-zero columns of eval_partial_deriv(k) (src/model/mod.rs:359-362) in pair order */
static void model_pairs(const model_t *md, const double *a, double *dPhi) {
    int i;
    for (i = 0; i < M; ++i) {
        const double dx = md->x[i] - a[0], u = dx / a[1], g = exp(-0.5 * u * u);
        const double d = md->x[i] - a[2], den = d * d + a[3] * a[3];
        dPhi[i] = g * dx / (a[1] * a[1]);
        dPhi[M + i] = g * dx * dx / (a[1] * a[1] * a[1]);
        dPhi[2 * M + i] = 2.0 * a[3] * a[3] * d / (den * den);
        dPhi[3 * M + i] = 2.0 * a[3] * d * d / (den * den);
    }
}
/* == eval_partial_deriv(k): m x n with zero columns kept (what the oracle's trait call expects) */
static void model_dphi(void *user, const double *a, int k, double *Dk) {
    static double dPhi[NP * M];
    int p;
    model_pairs((const model_t *)user, a, dPhi);
    memset(Dk, 0, sizeof(double) * M * NB);
    for (p = 0; p < NP; ++p)
        if (PAIR_PARAM[p] == k) memcpy(Dk + (size_t)PAIR_BASIS[p] * M, dPhi + (size_t)p * M, sizeof(double) * M);
}

/* ---- the LeastSquaresProblem the external driver sees: the device handle behind the C ABI -------------------------- */
typedef struct {
    vp_batch *h;
    model_t *md;
    double alpha[NQ];
    int calls_set, calls_res, calls_jac;
} dev_problem;

static void d_set_params(void *u, const double *x) {
    static double Phi[NB * M];
    dev_problem *p = (dev_problem *)u;
    p->calls_set++;
    memcpy(p->alpha, x, sizeof(double) * NQ);
    model_eval(p->md, x, Phi); /* model.set_params(x); model.eval() on the HOST */
    if (vp_set_params_with_basis(p->h, x, Phi, NULL) != 0) {
        printf("vp_set_params_with_basis: %s\n", vp_last_error());
        exit(2);
    }
}
static void d_params(void *u, double *x) {
    if (vp_params(((dev_problem *)u)->h, x) != 0) exit(2);
}
static int d_residuals(void *u, double *r) {
    dev_problem *p = (dev_problem *)u;
    int32_t st = -1;
    p->calls_res++;
    if (vp_residuals(p->h, r, &st) != 0) {
        printf("vp_residuals: %s\n", vp_last_error());
        exit(2);
    }
    return st == VP_ST_OK;
}
static int d_jacobian(void *u, double *J) {
    static double dPhi[NP * M];
    dev_problem *p = (dev_problem *)u;
    int32_t st = -1;
    p->calls_jac++;
    model_pairs(p->md, p->alpha, dPhi); /* model.eval_partial_deriv(k) for all k on the HOST */
    if (vp_jacobian_with_derivatives(p->h, dPhi, J, &st) != 0) {
        printf("vp_jacobian_with_derivatives: %s\n", vp_last_error());
        exit(2);
    }
    return st == VP_ST_OK;
}

#define ROWS 64
static int run_case(const char *name, model_t *md, int S, const double *Y, const double *w, const double *alpha0, int expect_user) {
    const int mr = M * S, W = NQ + 4;
    int failures = 0, i, k, s, rows_ext, rows_orc, build_err = 0, lead;
    vp_lm_opts o;
    vp_report rep_ext, rep_orc;
    vp_model_desc shape;
    double a_ext[NQ], a_orc[NQ], amax = 0.0;
    double *tr_ext = (double *)malloc(sizeof(double) * ROWS * W), *tr_orc = (double *)malloc(sizeof(double) * ROWS * W);
    double *fvec = (double *)malloc(sizeof(double) * mr), *fwork = (double *)malloc(sizeof(double) * mr);
    double *fjac = (double *)malloc(sizeof(double) * mr * NQ), *J0 = (double *)malloc(sizeof(double) * mr * NQ);
    double *Jo = (double *)malloc(sizeof(double) * mr * NQ), *r0 = (double *)malloc(sizeof(double) * mr), *ro = (double *)malloc(sizeof(double) * mr);
    double C0[NB * 8], cost0 = 0.0;
    static double Phi[NB * M], dPhi[NP * M];
    vp_batch *h = 0;
    dev_problem dp;
    vpo_lsq P;
    vpo_problem *op;
    vp_lm_opts_default(&o, VP_F64);
    for (i = 0; i < ROWS * W; ++i) tr_ext[i] = tr_orc[i] = NAN;

    /* the oracle with the same callbacks */
    memset(&shape, 0, sizeof shape);
    shape.n_basis = NB;
    shape.n_params = NQ;
    op = vpo_problem_create(&shape, M, S, NULL, Y, w, -1.0, &build_err);
    if (!op) return 1;
    vpo_problem_set_external_model(op, model_eval, model_dphi, md);

    /* the device handle: shape + pair table, no grid, no descriptor */
    if (vp_batch_create_external(&h, NB, NQ, NP, PAIR_BASIS, PAIR_PARAM, VP_F64, M, S, 1, Y, w, -1.0, VP_FLAG_OWN_STREAM, 0, 0) != 0) {
        printf("%s: vp_batch_create_external failed: %s\n", name, vp_last_error());
        return 1;
    }
    /* (1) fixed-point parity at the initial parameters through the fused call */
    if (!expect_user) {
        int32_t st = -1;
        double ywmax = 0.0;
        model_eval(md, alpha0, Phi);
        model_pairs(md, alpha0, dPhi);
        if (vp_evaluate_with_basis(h, alpha0, Phi, dPhi, r0, J0, C0, &cost0, &st) != 0 || st != VP_ST_OK) {
            printf("%s: vp_evaluate_with_basis failed (%d): %s\n", name, (int)st, vp_last_error());
            return 1;
        }
        vpo_set_params(op, alpha0);
        if (!vpo_residuals(op, ro) || !vpo_jacobian(op, Jo)) return 1;
        for (i = 0; i < mr; ++i) ywmax = fmax(ywmax, fabs(Y[i] * (w ? w[i % M] : 1.0)));
        {
            double cmax = 0.0, dc = 0.0, dr = 0.0, co = 0.0;
            for (i = 0; i < NB * S; ++i) { cmax = fmax(cmax, fabs(op->C[i])); dc = fmax(dc, fabs(op->C[i] - C0[i])); }
            for (i = 0; i < mr; ++i) { dr = fmax(dr, fabs(ro[i] - r0[i])); co += 0.5 * ro[i] * ro[i]; }
            if (dc > 1e-10 * cmax) { printf("  %s: c differs by %.3e (max|c| %.3e)\n", name, dc, cmax); ++failures; }
            if (dr > 1e-10 * ywmax) { printf("  %s: r differs by %.3e (max|y_w| %.3e)\n", name, dr, ywmax); ++failures; }
            if (fabs(co - cost0) > 1e-10 * co) { printf("  %s: cost %.15g vs %.15g\n", name, cost0, co); ++failures; }
            for (k = 0; k < NQ; ++k) {
                double jmax = 0.0, dj = 0.0, unproj = 0.0;
                int p;
                for (i = 0; i < mr; ++i) { jmax = fmax(jmax, fabs(Jo[k * mr + i])); dj = fmax(dj, fabs(Jo[k * mr + i] - J0[k * mr + i])); }
                for (p = 0; p < NP; ++p)
                    if (PAIR_PARAM[p] == k)
                        for (s = 0; s < S; ++s)
                            for (i = 0; i < M; ++i)
                                unproj = fmax(unproj, fabs(dPhi[p * M + i] * op->C[s * NB + PAIR_BASIS[p]] * (w ? w[i] : 1.0)));
                if (dj > 1e-10 * jmax + 1e-13 * unproj) { printf("  %s: J[%d] differs by %.3e (max|J_k| %.3e)\n", name, k, dj, jmax); ++failures; }
            }
        }
    }

    /* (2) the external driver over the trait: build (= create + the builder's initial set_params), then minimize */
    dp.h = h; dp.md = md; dp.calls_set = dp.calls_res = dp.calls_jac = 0;
    d_set_params(&dp, alpha0); /* SeparableProblemBuilder::build, src/problem/builder.rs:321 */
    dp.calls_set = 0;
    P.n = NQ; P.mr = mr; P.user = &dp;
    P.set_params = d_set_params; P.params = d_params; P.residuals = d_residuals; P.jacobian = d_jacobian;
    rows_ext = vpo_lm_minimize(&P, &o, &rep_ext, tr_ext, ROWS, fvec, fwork, fjac);
    vp_params(h, a_ext);

    vpo_set_params(op, alpha0);
    rows_orc = vpo_fit_trace(op, &o, &rep_orc, tr_orc, ROWS);
    memcpy(a_orc, op->alpha, sizeof a_orc);

    printf("%-34s external over the C ABI: term %2d evals %3d | oracle, same callbacks: term %2d evals %3d | trait calls set/res/jac %d/%d/%d\n",
           name, rep_ext.termination, rep_ext.n_evals, rep_orc.termination, rep_orc.n_evals, dp.calls_set, dp.calls_res, dp.calls_jac);
    if (expect_user) {
        if (rep_ext.termination != VP_TERM_USER || rep_orc.termination != VP_TERM_USER || rep_ext.n_evals != rep_orc.n_evals) ++failures;
    } else {
        if (!(rep_ext.termination > 0 && rep_orc.termination > 0)) ++failures;
        if (abs(rep_ext.n_evals - rep_orc.n_evals) > 8) ++failures;
        lead = rows_ext < rows_orc ? rows_ext : rows_orc;
        if (lead > 6) lead = 6;
        if (lead < 2) ++failures;
        for (i = 0; i < lead; ++i)
            for (k = 0; k <= NQ; ++k) {
                const double e = tr_ext[i * W + k], g = tr_orc[i * W + k], sc = fabs(g) > 1e-300 ? fabs(g) : 1.0;
                if (fabs(e - g) > 1e-8 * sc + 1e-12) {
                    printf("  trajectory row %d col %d: external %.15g oracle %.15g\n", i, k, e, g);
                    ++failures;
                }
            }
        for (k = 0; k < NQ; ++k) amax = fmax(amax, fabs(a_orc[k]));
        for (k = 0; k < NQ; ++k)
            if (fabs(a_ext[k] - a_orc[k]) > 1e-7 * amax) {
                printf("  alpha[%d]: external %.15g oracle %.15g\n", k, a_ext[k], a_orc[k]);
                ++failures;
            }
        if (fabs(rep_ext.objective - rep_orc.objective) > 1e-9 * rep_orc.objective) {
            printf("  objective: external %.15g oracle %.15g\n", rep_ext.objective, rep_orc.objective);
            ++failures;
        }
    }
    vpo_problem_destroy(op);
    vp_batch_destroy(h);
    free(tr_ext); free(tr_orc); free(fvec); free(fwork); free(fjac); free(J0); free(Jo); free(r0); free(ro);
    if (failures) printf("  %s: %d failure(s)\n", name, failures);
    return failures;
}

static double lcg(unsigned long long *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(*s >> 11) / 9007199254740992.0;
}

int main(void) {
    static model_t md;
    static double Y[6 * M], w[M], Phi[NB * M];
    const double truth[NQ] = {3.0, 0.6, 6.5, 0.9}, guess[NQ] = {3.2, 0.7, 6.2, 1.05};
    unsigned long long seed = 4242;
    int failures = 0, i, s;
    if (vp_device_count() <= 0) {
        printf("no device: the external-model test needs a GPU (the C ABI has no CPU path)\n");
        return 0;
    }
    for (i = 0; i < M; ++i) {
        md.x[i] = 10.0 * (double)i / (double)(M - 1);
        w[i] = 0.5 + lcg(&seed);
    }
    md.poison = 0;
    model_eval(&md, truth, Phi);
    for (s = 0; s < 6; ++s) {
        const double c1 = 20.0 + 7.0 * s, c2 = 35.0 - 4.0 * s, c3 = 1.0 + 0.5 * s;
        for (i = 0; i < M; ++i)
            Y[s * M + i] = c1 * Phi[i] + c2 * Phi[M + i] + c3 + 0.02 * (lcg(&seed) + lcg(&seed) + lcg(&seed) - 1.5);
    }
    failures += run_case("S=1 Gauss+Lorentz+offset", &md, 1, Y, NULL, guess, 0);
    failures += run_case("S=1 weighted", &md, 1, Y, w, guess, 0);
    failures += run_case("S=3 (branch S<=q)", &md, 3, Y, NULL, guess, 0);
    failures += run_case("S=6 (branch S>q) weighted", &md, 6, Y, w, guess, 0);
    {
        const double bad[NQ] = {-1.0, 0.7, 6.2, 1.05};
        md.poison = 1;
        failures += run_case("model returns a non-finite basis", &md, 1, Y, NULL, bad, 1);
        md.poison = 0;
    }
    printf("%d failure(s)\n", failures);
    return failures != 0;
}
