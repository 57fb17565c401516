/*
 * varpro_oracle.h -- CPU restatement of the reference algorithm (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle of SURVEY.md section 8(c): a plain-C, fp64, single-problem
 * restatement of geo-ant/varpro 0.13.3's variable-projection hot path, following the
 * reference file:line cited at each function.  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the timed CPU baseline.  The product (varpro_amd/, include/) never
 * links, imports or falls back to anything in this directory.
 *
 * Pinning status ("how far is the oracle itself trusted"):
 *   - The reference is Rust; no rustc/cargo in the build image, so the reference cannot
 *     be compiled or run here.  Its arithmetic lives in third-party
 *     crates that are NOT under /root/reference: nalgebra "0.33" (thin SVD, solve, gemm)
 *     and levenberg-marquardt "0.14" (MINPACK lmder port).  Their published algorithms
 *     are restated here: thin SVD (any backward-stable thin SVD gives the same C, R and
 *     projector U U^T at full rank; one-sided Jacobi is used), SVD::solve with absolute
 *     singular-value threshold, and MINPACK lmder/lmpar/qrfac/qrsolv with the crate's
 *     termination semantics.
 *   - Pinned against every known-answer vector the reference's own tests hold for this
 *     path (tests/test_oracle_reference_vectors.py): Octave residuals at tau=(0.5,6.5)
 *     unweighted 1e-4 / weighted 1e-3 (src/solvers/levmar/test.rs:111-208), Jacobian
 *     relations (:21-108), end-to-end fits to 1e-8 (tests/integration_tests/main.rs:93-227,
 *     399-551), O'Leary/MATLAB 1e-5 (:713-778), lmfit fixtures 1e-5 (:554-668).
 *   - "parity unpinned at 1e-10 against the third-party crates": the reference asserts
 *     nothing tighter than 1e-8 and never stores Jacobian values or LM trajectories.  The
 *     1e-10 contract of BASELINE.json is pinned instead against an independent 50-digit
 *     mpmath evaluation of C, R, J (tests/golden/make_golden.py -> tests/golden/ npz files).
 */
#ifndef VARPRO_ORACLE_H
#define VARPRO_ORACLE_H

#include <stdint.h>

#include "../include/varpro_hip.h" /* vp_model_desc, vp_lm_opts, vp_report, enums (descriptor types only) */

#ifdef __cplusplus
extern "C" {
#endif

/* One separable problem == SeparableProblem (src/problem.rs:57-83) incl. its
 * CachedCalculations (src/problem.rs:88-107).  All matrices column-major. */
typedef struct vpo_problem {
    vp_model_desc model;
    int m, S;
    double *t;   /* m */
    double *w;   /* m or NULL (Weights::Unit) */
    double *Yw;  /* m x S  weighted data */
    double eps;  /* svd_epsilon */
    double *alpha; /* q  == model.params() */
    /* cache */
    int cached;    /* 1 <=> cached = Some(..) */
    double *U;     /* m x n */
    double *sigma; /* n */
    double *V;     /* n x n  (V, not V^T) */
    double *C;     /* n x S */
    double *R;     /* m x S */
    long n_set_params; /* counters for the CPU-baseline report */
    long n_jacobians;
    /* workspace owned by the problem (no allocation per evaluation): Phi_w, QR scratch, D_k, LM vectors */
    double *ws_phi, *ws_qr, *ws_dk, *ws_fvec, *ws_fwork, *ws_fjac;
    /* A model outside the descriptor language: the reference's solver works with ANY SeparableNonlinearModel
     * (src/model/mod.rs:239-363), e.g. the closure-based SeparableModel (:441-512).  When ext_eval is set the two model
     * calls of the path -- eval() at src/solvers/levmar/mod.rs:45 and eval_partial_deriv(k) at :141 -- go to these
     * callbacks instead of the descriptor formulas; model.n_basis / n_params give the shape, the kinds are ignored. */
    void (*ext_eval)(void *user, const double *alpha, double *Phi /* m x n, col-major */);
    void (*ext_dphi)(void *user, const double *alpha, int k, double *Dk /* m x n, zero columns kept */);
    void *ext_user;
} vpo_problem;

/* == SeparableProblemBuilder::new(model) with a user model (any impl of the trait): callbacks for eval() and
 * eval_partial_deriv(k); `t` of vpo_problem_create may then be NULL */
void vpo_problem_set_external_model(vpo_problem *p, void (*eval)(void *, const double *, double *),
                                    void (*dphi)(void *, const double *, int, double *), void *user);

/* == SeparableProblemBuilder::build (src/problem/builder.rs:278-324) WITHOUT the initial
 * set_params; returns NULL and sets *build_err (VP_BUILD_*) on validation failure. */
vpo_problem *vpo_problem_create(const vp_model_desc *model, int m, int S, const double *t, const double *Y,
                                const double *w, double svd_epsilon, int *build_err);
void vpo_problem_destroy(vpo_problem *p);
/* new observations Y (m x S) for an existing problem: same model / grid / weights, no re-allocation */
void vpo_problem_reset(vpo_problem *p, const double *Y);

/* == SeparableNonlinearModel::eval (src/model/mod.rs:308; closure impl :441-471) : Phi m x n */
void vpo_eval_phi(const vp_model_desc *model, int m, const double *t, const double *alpha, double *Phi);
/* == eval_partial_deriv(k) (src/model/mod.rs:359-362; :473-512) : D_k m x n, zero columns kept */
void vpo_eval_dphi(const vp_model_desc *model, int m, const double *t, const double *alpha, int k, double *Dk);

/* == SeparableProblem::set_params (src/solvers/levmar/mod.rs:42-73) */
void vpo_set_params(vpo_problem *p, const double *alpha);
/* == residuals (src/solvers/levmar/mod.rs:91-95); returns 0 if cached is None */
int vpo_residuals(const vpo_problem *p, double *r_out);
/* == jacobian (src/solvers/levmar/mod.rs:101-201); returns 0 if cached is None */
int vpo_jacobian(vpo_problem *p, double *J_out);
/* == FitResult::best_fit (src/fit.rs:55-59,87-91) */
int vpo_best_fit(const vpo_problem *p, double *fit_out);

/* == FitStatistics::try_calculate (src/statistics/mod.rs:352-441): covariance (n+q)^2 column-major,
 * reduced chi^2, per-row unscaled confidence sigma.  1 = ok, 0 = underdetermined / singular / not cached */
int vpo_statistics(vpo_problem *p, double *cov, double *reduced_chi2, double *conf_sigma);

/* == LevMarSolver::fit -> LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:238-254) */
void vpo_fit(vpo_problem *p, const vp_lm_opts *opts, vp_report *rep);
/* vpo_fit + per-evaluation trace rows [x_trial(q), ||r||, ratio, delta, par]; returns rows written */
int vpo_fit_trace(vpo_problem *p, const vp_lm_opts *opts, vp_report *rep, double *trace, int max_rows);

/* A LeastSquaresProblem as the levenberg-marquardt crate sees it (src/solvers/levmar/mod.rs:22-202): the driver below
 * touches the problem through these four calls only.  residuals / jacobian return 0 for the trait's `None`. */
typedef struct vpo_lsq {
    int n;  /* number of parameters */
    int mr; /* number of residuals (m * S) */
    void *user;
    void (*set_params)(void *user, const double *x);
    void (*params)(void *user, double *x);
    int (*residuals)(void *user, double *r);  /* mr */
    int (*jacobian)(void *user, double *J);   /* mr x n, column-major */
} vpo_lsq;
/* == LevenbergMarquardt::minimize over callbacks; vpo_fit_trace is this driver on the CPU restatement.  Workspace:
 * fvec, fwork mr doubles each, fjac mr*n doubles.  Returns the trace rows written. */
int vpo_lm_minimize(const vpo_lsq *P, const vp_lm_opts *opts, vp_report *rep, double *trace, int max_rows,
                    double *fvec, double *fwork, double *fjac);

/* thin SVD A (m x n, col-major, m >= n) = U diag(sigma) V^T, sigma descending */
void vpo_thin_svd(int m, int n, const double *A, double *U, double *sigma, double *V);

/* MINPACK enorm as used by the levenberg-marquardt crate */
double vpo_enorm(int n, const double *x);
/* MINPACK lmpar (== the levenberg-marquardt crate's determine_lambda_and_parameter_update) on a caller's pivoted QR
 * factor: r n x n column-major (upper triangle used), ipvt, diag, qtb = first n of Q^T f; returns par, x = step */
double vpo_lmpar(int n, const double *r, const int *ipvt, const double *diag, const double *qtb, double delta, double par,
                 double *x, double *dxnorm_out);

/*
 * Batched convenience used by the tests and by bench.py's cpu_baseline leg: B independent
 * single-RHS problems, shared grid t and weights w (may be NULL), Y [B][m], alpha [B][q]
 * in/out, C_out [B][n] (may be NULL), rep [B] (may be NULL).  n_threads > 1 splits the
 * batch statically over OpenMP threads ("single-socket CPU throughput").
 * Returns the wall time in seconds spent inside the fits (problem construction incl. the
 * initial set_params is excluded, as in benches/double_exponential_without_noise.rs:159-169).
 */
double vpo_fit_batch(const vp_model_desc *model, int m, int64_t B, const double *t, const double *Y,
                     const double *w, double svd_epsilon, const vp_lm_opts *opts, double *alpha_inout,
                     double *C_out, vp_report *rep, int n_threads);

/* batched set_params + residuals + jacobian + coefficients at given alpha (S = 1) */
void vpo_evaluate_batch(const vp_model_desc *model, int m, int64_t B, const double *t, const double *Y,
                        const double *w, double svd_epsilon, const double *alpha, double *r_out, double *J_out,
                        double *C_out, double *cost_out, int32_t *status, int n_threads);

void vpo_lm_opts_default(vp_lm_opts *o);
int vpo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
