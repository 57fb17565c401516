/*
 * varpro_hip.h -- C ABI of the MI355X-native batched variable-projection hot path.
 *
 * This header is the drop-in boundary.  The reference
 * (geo-ant/varpro 0.13.3) has no FFI; its boundary is two Rust traits plus the
 * builder/solver facade.  Every entry point below cites the reference item whose
 * observable contract it replaces (paths relative to the reference repository).
 * A Rust shim (bindings/rust/) implements
 * `SeparableNonlinearModel` / `LeastSquaresProblem` on top of these calls.
 *
 * Conventions (the reference's own, src/lib.rs:42-87):
 *   m  number of observations                  model.output_len()          src/model/mod.rs:263
 *   n  number of basis functions               base_function_count()       src/model/mod.rs:259
 *   q  number of nonlinear parameters alpha    parameter_count()           src/model/mod.rs:256
 *   S  number of right-hand sides              Y_w.ncols()                 src/solvers/levmar/mod.rs:111
 *   B  batch of independent problems (new; the reference solves one problem per call)
 *
 * Memory layouts (row index fastest == nalgebra column-major per problem):
 *   t      [m]  (shared)   or [B][m]  with VP_FLAG_T_PER_PROBLEM
 *   w      [m]  (shared)   or [B][m]  with VP_FLAG_W_PER_PROBLEM, NULL => Weights::Unit
 *   Y, R   [B][S][m]       residual vector of problem b == vec(R_b) (src/util/mod.rs:101-106)
 *   alpha  [B][q]
 *   C      [B][S][n]
 *   J      [B][q][S][m]    problem b: (m*S) x q column-major, RHS s at rows s*m.. (src/solvers/levmar/mod.rs:147-153)
 *   Phi    [B][n][m]       dPhi [B][p][m], p = number of (basis,param) dependency pairs in model order
 *
 * All data pointers of one handle live in one address space chosen at creation:
 * host (default; the library stages through hipMemcpyAsync) or device
 * (VP_FLAG_DEVICE_PTRS; zero copies, everything enqueued on the handle's stream).
 * One handle <-> one HIP stream; handles are independent and re-entrant, there is
 * no global state (the reference is single-threaded and Clone, src/problem.rs:55).
 *
 * Return value of every call: 0 = ok, <0 = call-level error (vp_last_error()).
 * Per-problem failures never abort a batch; they are latched in status[b]
 * exactly like `cached = None` in the reference (src/solvers/levmar/mod.rs:61-72):
 * status[b] != 0  <=>  residuals()/jacobian() would return None for problem b.
 */
#ifndef VARPRO_HIP_H
#define VARPRO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP_MAX_BASIS 8        /* n <= 8 */
#define VP_MAX_PARAMS 8       /* q <= 8 */
#define VP_MAX_BASIS_PARAMS 2 /* parameters one basis function may depend on */
#define VP_MAX_PAIRS 16       /* p <= 16 non-zero (basis,param) derivative columns */

/* scalar type of the problem == SeparableNonlinearModel::ScalarType (src/model/mod.rs:246) */
enum { VP_F64 = 0, VP_F32 = 1 };

/*
 * Closed descriptor language for basis functions.  The reference
 * accepts opaque Rust closures (src/model/model_basis_function.rs:11-12) which
 * cannot run on a GPU; these kinds cover every model the reference's tests and
 * benches use.  kind -> f(t, p0[, p1]) and its partial derivatives:
 */
enum {
    VP_BASIS_CONST = 0,     /* 1                       invariant_function, shared_test_code/src/lib.rs:123     */
    VP_BASIS_EXP_DECAY = 1, /* exp(-t/p0)              d/dp0 = exp(-t/p0)*t/p0^2   shared_test_code/src/lib.rs:101-114 */
    VP_BASIS_EXP_RATE = 2,  /* exp(-p0*t)              d/dp0 = -t*exp(-p0*t)                                    */
    VP_BASIS_EXP_COS = 3,   /* exp(-p0*t)*cos(p1*t)    d/dp0 = -t*f ; d/dp1 = -t*exp(-p0*t)*sin(p1*t)
                               shared_test_code/src/models.rs:310-372 (O'Leary example)                         */
    VP_BASIS_SIN_PHASE = 4, /* sin(p0*t+p1)            d/dp0 = t*cos(p0*t+p1) ; d/dp1 = cos(p0*t+p1)
                               src/test_helpers/mod.rs:28-52                                                     */
    VP_BASIS_EXTERNAL = 5   /* evaluated by the CALLER: any SeparableNonlinearModel (src/model/mod.rs:239-363), in
                               particular the closure-based SeparableModel (:441-512).  Only produced by
                               vp_batch_create_external; see "models outside the descriptor language" below          */
};

/*
 * Model descriptor == what SeparableModelBuilder::build() produces
 * (src/model/builder/mod.rs:338-525): basis functions in insertion order
 * (column j of Phi == basis j, src/model/builder/mod.rs:512-515) and, per basis
 * function, which entries of alpha it depends on (the "Ind" table of
 * matlab/examples/adaex.m:33-34; src/model/detail.rs:60-127).
 * param[j][a] = index into alpha of argument a of basis j, or -1 if unused.
 */
typedef struct vp_model_desc {
    int32_t n_basis;  /* n */
    int32_t n_params; /* q */
    int32_t kind[VP_MAX_BASIS];
    int32_t param[VP_MAX_BASIS][VP_MAX_BASIS_PARAMS];
} vp_model_desc;

/* creation flags */
enum {
    VP_FLAG_DEVICE_PTRS = 1 << 0,   /* every data pointer passed for this handle is a device pointer */
    VP_FLAG_T_PER_PROBLEM = 1 << 1, /* t is [B][m] instead of [m] */
    VP_FLAG_W_PER_PROBLEM = 1 << 2, /* w is [B][m] instead of [m] */
    VP_FLAG_OWN_STREAM = 1 << 3,    /* ignore hip_stream and run on a private non-blocking stream */
    /* fp64 kernels evaluate exp(-t/tau) on a grid that is uniform to rounding (checked once at creation:
     * |t_i - (t_0 + i dt)| <= 4 eps |t_i - t_0|) by a per-lane recurrence instead of one exponential per row;
     * results then agree with the per-row form to ~1e-15 relative instead of 1 ulp.  This flag keeps the per-row
     * exponential (== the reference's evaluation, shared_test_code/src/lib.rs:101-114) regardless of the grid. */
    VP_FLAG_NO_GRID_RECURRENCE = 1 << 4,
    /* Single-RHS fits of this handle run on the LENGTH-AGNOSTIC kernels (rows streamed in blocks through a TSQR-style
     * update of an (n+1+p)^2 triangle; any m) even where a register-resident kernel set covers m.  The library selects them
     * by itself beyond the largest resident set; the flag exists for memory-lean handles and for the parity tests.  Ignored
     * for models without such a set (they run on the generic kernels, as before). */
    VP_FLAG_STREAM_ROWS = 1 << 5
};

/* per-problem status word (0 == the reference's `cached = Some(..)`) */
enum {
    VP_ST_OK = 0,
    VP_ST_NONFINITE = 1,    /* Phi, C or R contains inf/nan (model error / SVD solve error path) */
    VP_ST_NOT_EVALUATED = 2 /* residuals()/jacobian() before any set_params() */
};

/* call-level error codes */
enum {
    VP_ERR_OK = 0,
    VP_ERR_INVALID = -1,     /* bad handle / argument / shape (builder errors, src/problem/builder.rs:15-46) */
    VP_ERR_UNSUPPORTED = -2, /* operation not available for this handle.  Every MODEL the descriptor can express is
                              * accepted for every entry point -- shapes without a specialised kernel set run on generic
                              * kernels, global fits (S > 1) and any batch size included; m < n (an underdetermined
                              * linear sub-problem) takes the reference's minimum-norm solution; right-hand-side sharding
                              * (vp_set_rhs_allreduce) works on every shape.  What remains unsupported: fit statistics with
                              * S > 1 (as in the reference), vp_debug_gram_evaluate on handles without the Gram kernel */
    VP_ERR_HIP = -3,         /* HIP runtime failure */
    VP_ERR_NO_DEVICE = -4    /* no gfx950 device / library built without device code */
};

/* builder validation errors mirror SeparableProblemBuilderError (src/problem/builder.rs:15-46);
 * returned by vp_batch_create as VP_ERR_INVALID with one of these in vp_last_error_detail() */
enum {
    VP_BUILD_OK = 0,
    VP_BUILD_Y_DATA_MISSING = 1,
    VP_BUILD_INVALID_LENGTH_OF_DATA = 2,
    VP_BUILD_ZERO_LENGTH_VECTOR = 3,
    VP_BUILD_INVALID_PARAMETER_COUNT = 4,
    VP_BUILD_INVALID_LENGTH_OF_WEIGHTS = 5
};

/*
 * LM driver options == levenberg_marquardt::LevenbergMarquardt builder knobs
 * reached through LevMarSolver::with_solver (src/solvers/levmar/mod.rs:221-223).
 * Defaults (vp_lm_opts_default): ftol = xtol = gtol = 30*eps(dtype), stepbound = 100,
 * patience = 100 (max evaluations = patience*(q+1)), scale_diag = 1.
 */
typedef struct vp_lm_opts {
    double ftol;
    double xtol;
    double gtol;
    double stepbound;
    int32_t patience;
    int32_t scale_diag;
} vp_lm_opts;

/* termination == levenberg_marquardt::TerminationReason; >0 <=> was_successful()
 * (src/fit.rs:120-122, src/solvers/levmar/mod.rs:249-253) */
enum {
    VP_TERM_RESIDUALS_ZERO = 1,
    VP_TERM_ORTHOGONAL = 2,
    VP_TERM_CONVERGED_FTOL = 3,
    VP_TERM_CONVERGED_XTOL = 4,
    VP_TERM_CONVERGED_BOTH = 5,
    VP_TERM_NOT_RUN = 0,
    VP_TERM_USER = -1, /* residuals()/jacobian() returned None */
    VP_TERM_NUMERICAL = -2,
    VP_TERM_NO_IMPROVEMENT = -3,
    VP_TERM_LOST_PATIENCE = -4,
    VP_TERM_NO_PARAMETERS = -5,
    VP_TERM_NO_RESIDUALS = -6,
    VP_TERM_WRONG_DIMENSIONS = -7
};

/* == levenberg_marquardt::MinimizationReport (src/fit.rs:24-29) */
typedef struct vp_report {
    int32_t termination;
    int32_t n_evals;  /* number_of_evaluations */
    double objective; /* objective_function = 1/2 ||r||^2 */
} vp_report;

typedef struct vp_batch vp_batch;

/* ---- lifetime ------------------------------------------------------------------------- */

/*
 * == SeparableProblemBuilder::{new|mrhs, observations, weights, epsilon, build}
 * (src/problem/builder.rs:116,194,142,220,261,246,278-324) for B problems at once.
 * Validates shapes, stores Y_w = W*Y (src/problem/builder.rs:307), default epsilon
 * = machine epsilon of dtype when svd_epsilon < 0 (src/problem/builder.rs:282).
 * Unlike build() it does NOT run the initial set_params (there is no alpha yet);
 * vp_set_params / vp_fit do.  `hip_stream` is the hipStream_t all work of this handle is
 * enqueued on; NULL means the HIP null (default) stream -- which is what PyTorch's default
 * stream is -- unless VP_FLAG_OWN_STREAM asks for a private stream.
 */
int vp_batch_create(vp_batch **h, const vp_model_desc *model, int dtype, int64_t m, int64_t S, int64_t B,
                    const void *t, const void *Y, const void *w, double svd_epsilon, int flags, int device,
                    void *hip_stream);

/* == Drop of SeparableProblem */
void vp_batch_destroy(vp_batch *h);

/* ---- LeastSquaresProblem surface (src/solvers/levmar/mod.rs:22-202) --------------------- */

/*
 * == SeparableProblem::set_params (src/solvers/levmar/mod.rs:42-73) incl.
 * model.set_params/eval (src/model/mod.rs:266-267,308) and the weighting
 * (src/util/weights.rs:82-99): Phi_w = W Phi(alpha); solve min ||Y_w - Phi_w C||
 * with singular values <= epsilon truncated; R = Y_w - Phi_w C.  Caches C, R,
 * 1/2||R||^2 and status per problem (== CachedCalculations, src/problem.rs:88-107).
 */
int vp_set_params(vp_batch *h, const void *alpha);

/* == LeastSquaresProblem::params (src/solvers/levmar/mod.rs:80-82) */
int vp_params(vp_batch *h, void *alpha_out);

/* == LeastSquaresProblem::residuals (src/solvers/levmar/mod.rs:91-95): r_b = vec(R_b).
 * status may be NULL. */
int vp_residuals(vp_batch *h, void *r_out, int32_t *status);

/* == LeastSquaresProblem::jacobian (src/solvers/levmar/mod.rs:101-201): Kaufman
 * Jacobian J[:,k] = -vec(P_perp W dPhi/dalpha_k C), incl. eval_partial_deriv
 * (src/model/mod.rs:359-362).  status may be NULL. */
int vp_jacobian(vp_batch *h, void *J_out, int32_t *status);

/* == SeparableProblem::linear_coefficients (src/problem.rs:142-147,173-183) */
int vp_linear_coeffs(vp_batch *h, void *C_out, int32_t *status);

/* == SeparableProblem::weighted_data (src/problem.rs:154-156,189-196) */
int vp_weighted_data(vp_batch *h, void *Yw_out);

/*
 * Replace the observations of an existing handle (same B, S, m, model, grid and weights): Y_w = W*Y is recomputed on
 * the handle's stream and every cached result is invalidated.  The batch counterpart of building a new
 * SeparableProblem with SeparableProblemBuilder::observations (src/problem/builder.rs:219-232) for the next frame
 * of a stream of same-shaped data, without re-allocating the device state.  Y is [B][S][m] like at creation
 * (host or device pointer according to the handle's flags).
 */
int vp_set_observations(vp_batch *h, const void *Y);

/* 1/2 ||vec R_b||^2 per problem, always f64 [B] (== MinimizationReport::objective_function) */
int vp_cost(vp_batch *h, double *cost_out);

/*
 * Fused evaluation: set_params + any subset of {residuals, jacobian, coefficients,
 * cost} in ONE kernel launch with Phi never leaving the chip.  NULL outputs are
 * skipped.  Equivalent to the sequence of calls above (same cached state after).
 */
int vp_evaluate(vp_batch *h, const void *alpha, void *r_out, void *J_out, void *C_out, double *cost_out,
                int32_t *status);

/* ---- model surface -------------------------------------------------------------------- */

/*
 * == SeparableNonlinearModel::{set_params, eval, eval_partial_deriv}
 * (src/model/mod.rs:266-267,308,359-362) for the whole batch, UNWEIGHTED:
 * Phi_out[b][j][:] = basis j; dPhi_out[b][pair][:] = d basis_j / d alpha_k for the
 * dependency pairs in model order (basis-major).  Either output may be NULL.
 * flags: VP_BASIS_SKIP_INVARIANT omits VP_BASIS_CONST columns from Phi_out (then
 * Phi_out is [B][n_alpha][m]); this is the stand-alone Phi kernel whose HBM
 * roofline fraction BASELINE.json asks for.
 */
enum { VP_BASIS_SKIP_INVARIANT = 1 };
int vp_basis(vp_batch *h, const void *alpha, void *Phi_out, void *dPhi_out, int flags);

/* ---- models outside the descriptor language --------------------------------
 *
 * The reference's plugin boundary is a TRAIT: any `SeparableNonlinearModel` (src/model/mod.rs:239-363) works with its
 * solver -- the closure-based `SeparableModel` (src/model/mod.rs:441-512, src/model/model_basis_function.rs:11-28) and
 * every hand-written impl (shared_test_code/src/models.rs:40-150).  A model the closed descriptor language above cannot
 * express (a Gaussian, a Lorentzian, a basis function of three parameters, a table look-up, ...) crosses this ABI as
 * the VALUES the trait returns: the caller evaluates `eval()` -> Phi and `eval_partial_deriv(k)` -> its non-zero
 * columns; the device does everything downstream of them -- weighting (src/solvers/levmar/mod.rs:47,141), the
 * factorisation / truncated solve / residual (:51-59) and the Kaufman Jacobian (:101-201).
 *
 * vp_batch_create_external == SeparableProblemBuilder::{new|mrhs, observations, weights, epsilon, build}
 * (src/problem/builder.rs:116-324) for a model that is known by its SHAPE only:
 *   n_basis, n_params   base_function_count() / parameter_count()                   src/model/mod.rs:256-259
 *   n_pairs, pair_basis[], pair_param[]
 *                       the (basis j, parameter k) pairs with d phi_j / d alpha_k != 0, i.e. the columns
 *                       eval_partial_deriv(k) does not leave zero (src/model/mod.rs:473-512; the "Ind" table of
 *                       matlab/examples/adaex.m:33-34).  Any order, each pair once, n_pairs <= VP_MAX_PAIRS.  A basis
 *                       function may depend on ANY number of parameters (no VP_MAX_BASIS_PARAMS limit here), and a
 *                       caller that does not know the sparsity lists all n*q pairs.
 * There is no grid: the independent variable is the caller's business (src/model/mod.rs:441-471 captures it in the
 * closures).  Y, w, svd_epsilon, flags (VP_FLAG_DEVICE_PTRS, VP_FLAG_W_PER_PROBLEM, VP_FLAG_OWN_STREAM), device and
 * hip_stream as for vp_batch_create.
 *
 * On such a handle:
 *   vp_set_params_with_basis   == SeparableProblem::set_params (src/solvers/levmar/mod.rs:42-73) with the model call at
 *                              :43-45 (`model.set_params(alpha); model.eval()`) replaced by its result:
 *                                Phi   [B][n][m]         UNWEIGHTED basis matrix of every problem (column j = basis j)
 *                                dPhi  [B][n_pairs][m]   UNWEIGHTED d phi_j / d alpha_k for the pairs in table order, or
 *                                                        NULL (derivatives follow with vp_jacobian_with_derivatives --
 *                                                        the LM driver asks for a Jacobian only at accepted points)
 *                              alpha is stored (vp_params) but not interpreted.  Caches C, R, cost, status like
 *                              vp_set_params.  Host-pointer handles copy Phi / dPhi; device-pointer handles keep the
 *                              POINTERS: the arrays must stay valid and unchanged until the next vp_set_params_with_basis
 *                              / vp_evaluate_with_basis / vp_batch_destroy (the model owns its matrices in the
 *                              reference, too).
 *   vp_jacobian_with_derivatives == SeparableProblem::jacobian (:101-201) with `model.eval_partial_deriv(k)` (:141)
 *                              replaced by its non-zero columns dPhi [B][n_pairs][m] at the current parameters.
 *   vp_evaluate_with_basis     the fused form (== vp_evaluate): Phi, dPhi in -> any subset of r, J, C, cost, status out in
 *                              one pass over the columns.  dPhi may be NULL when J_out is.
 *   vp_residuals, vp_jacobian (needs the dPhi of the last vp_set_params_with_basis), vp_linear_coeffs, vp_cost,
 *   vp_params, vp_weighted_data, vp_set_observations, vp_best_fit, vp_statistics, vp_synchronize work as on any handle.
 *   vp_set_params / vp_evaluate / vp_basis / vp_fit: VP_ERR_UNSUPPORTED -- the device cannot evaluate the model.  A FIT
 *   of such a batch runs by reverse communication: vp_fit_begin / vp_fit_step_with_basis / vp_fit_end below (the
 *   device keeps the LM loop, the caller keeps the model); one LM driver per problem on the host over the trait-level
 *   entries above (tests/c/test_external_model.c drives one) remains possible.
 */
int vp_batch_create_external(vp_batch **h, int32_t n_basis, int32_t n_params, int32_t n_pairs, const int32_t *pair_basis,
                             const int32_t *pair_param, int dtype, int64_t m, int64_t S, int64_t B, const void *Y,
                             const void *w, double svd_epsilon, int flags, int device, void *hip_stream);
int vp_set_params_with_basis(vp_batch *h, const void *alpha, const void *Phi, const void *dPhi);
int vp_jacobian_with_derivatives(vp_batch *h, const void *dPhi, void *J_out, int32_t *status);
int vp_evaluate_with_basis(vp_batch *h, const void *alpha, const void *Phi, const void *dPhi, void *r_out, void *J_out,
                           void *C_out, double *cost_out, int32_t *status);

/* ---- solver surface ------------------------------------------------------------------- */

void vp_lm_opts_default(vp_lm_opts *opts, int dtype);

/*
 * == LevMarSolver::fit (src/solvers/levmar/mod.rs:238-254) -> LevenbergMarquardt::minimize (call site :247) for a BATCH of
 * problems whose model only the caller can evaluate -- any SeparableNonlinearModel (src/model/mod.rs:239-363) on a handle
 * made by vp_batch_create_external -- by REVERSE COMMUNICATION.  The reference's driver talks to the problem through three
 * trait calls: set_params(x_trial) (:42-73, which evaluates the model, :43-45), residuals() (:91-95) and, at accepted
 * points only, jacobian() (:101-201, which asks the model for eval_partial_deriv(k), :141).  Here the device keeps the
 * whole driver -- one LM record per problem: trust region, lmpar, accept / reject, the crate's termination tests -- and
 * the model's answers cross the boundary as arrays; what comes back per step is alpha_trial [B][q] and one word per
 * problem.  Neither the residuals nor the Jacobian J [B][q][m] are ever written to memory.
 *
 *   vp_fit_begin(h, opts, alpha0, flags)
 *       starts a fit of all B problems from alpha0 [B][q] (opts == NULL: vp_lm_opts_default).  The first step takes the
 *       columns at alpha0.  flags: 0, or VP_FIT_DERIVATIVES_ON_ACCEPT (below).
 *   vp_fit_step_with_basis(h, Phi, dPhi, alpha_trial_out, want_out, n_active_out)
 *       Phi  [B][n][m], dPhi [B][n_pairs][m]: UNWEIGHTED model columns at the trial points the previous step returned
 *       (alpha0 for the first step), laid out as for vp_set_params_with_basis.  One launch runs, for every problem still
 *       active, one LM iteration: the evaluation at the trial point, the accept / reject decision and termination tests,
 *       at an accepted point the pivoted QR of the Kaufman Jacobian, and the next trust-region step.  Outputs (any may
 *       be NULL; same address space as the handle's other arrays):
 *         alpha_trial_out [B][q]  where the columns are wanted next; for a finished problem its final parameters
 *         want_out        [B]     VP_WANT_BASIS | VP_WANT_DERIVATIVES bits, 0 = the problem has terminated.  Entries of
 *                                 Phi / dPhi of problems that do not want them are never read.
 *         n_active_out    host    number of problems still running (reading it synchronises the handle's stream; pass
 *                                 NULL to keep a device-pointer pipeline asynchronous and look every few steps)
 *       Default protocol: every step carries Phi AND dPhi (want is 3 or 0) and is one LM iteration of every problem.
 *       VP_FIT_DERIVATIVES_ON_ACCEPT: the driver's own call order -- trial points want Phi only (want = 1, dPhi is not
 *       read and may be NULL); a problem that ACCEPTS its trial point answers want = 3 with the same alpha_trial and
 *       forms its Jacobian from the columns of the next step.  eval_partial_deriv is then evaluated exactly where the
 *       reference evaluates it, at the price of a second pass over Phi per accepted point.  A problem whose want = 3
 *       request is answered with dPhi == NULL ends with VP_TERM_USER (the reference: eval_partial_deriv failed ->
 *       jacobian() == None); a handle WITHOUT dependency pairs (n_pairs == 0: every eval_partial_deriv is zero) never
 *       reads dPhi and its fits end `Orthogonal` at alpha0 -- the zero Jacobian's scaled gradient is 0 <= gtol.
 *   vp_fit_end(h, alpha_out, C_out, rep)
 *       FitResult::nonlinear_parameters / linear_coefficients (src/fit.rs:113-115) and the MinimizationReport of every
 *       problem (rep[b].termination == VP_TERM_NOT_RUN for a problem the caller stopped stepping before it terminated;
 *       its best point so far is returned).  Afterwards the handle holds alpha, C, cost and status of the fitted point
 *       (vp_params, vp_linear_coeffs, vp_cost, vp_summary*, vp_reduce_cost); entries that need the MODEL at that point
 *       (vp_residuals, vp_jacobian, vp_best_fit, vp_statistics) first need its columns: vp_set_params_with_basis.
 * Every problem terminates after at most patience*(q+1) evaluations (TerminationReason::LostPatience), so stepping until
 * n_active == 0 always ends.  Covered: EVERY shape vp_batch_create_external admits -- n <= VP_MAX_BASIS, q <= VP_MAX_PARAMS,
 * any pair table, any number of right-hand sides S (fit<Rhs>, src/problem/builder.rs:194-225: the S data columns share
 * alpha, the residual is the stacked one, C_out of vp_fit_end is [B][S][n]) -- at ANY m >= n: the (n, q, pairs) of the
 * compiled step kernels register-resident to 4 096 rows and streamed in row blocks beyond (shapes of up to ten columns
 * n + 1 + pairs), every other shape and every S > 1 on the generic step kernel.  m < n: VP_ERR_UNSUPPORTED.
 */
enum { VP_FIT_DERIVATIVES_ON_ACCEPT = 1 };
enum { VP_WANT_BASIS = 1, VP_WANT_DERIVATIVES = 2 };
int vp_fit_begin(vp_batch *h, const vp_lm_opts *opts, const void *alpha0, int flags);
int vp_fit_step_with_basis(vp_batch *h, const void *Phi, const void *dPhi, void *alpha_trial_out, int32_t *want_out,
                           int64_t *n_active_out);
int vp_fit_end(vp_batch *h, void *alpha_out, void *C_out, vp_report *rep);
/*
 * The ACTIVE SET of the running fit, compacted: after a step, index_out[0 .. *count_out) are the problems still running
 * (want != 0), in no particular order; entries beyond the count are valid problem indices of finished problems (stale, never
 * out of range).  A model that evaluates only these -- instead of scanning want_out [B] -- does work proportional to the
 * problems left: the tail of a batched fit is a few hundred problems for ~100 steps (the reference's driver runs each of
 * them alone, src/solvers/levmar/mod.rs:247).  The device's own evaluation launch already covers just this set.  Both
 * arrays follow the handle's address space (device pointers: two asynchronous copies, no synchronisation; index_out [B]).
 */
int vp_fit_active_set(vp_batch *h, int32_t *index_out, int32_t *count_out);

/*
 * == LevMarSolver::fit (src/solvers/levmar/mod.rs:238-254), i.e.
 * LevenbergMarquardt::minimize (third-party, call site :247) for every problem of
 * the batch, device-resident.  alpha_inout: in = initial guess (the problem's
 * current params), out = FitResult::nonlinear_parameters (src/fit.rs:113-115).
 * C_out (may be NULL) = FitResult::linear_coefficients.  rep[b] (may be NULL) =
 * MinimizationReport; rep[b].termination > 0 <=> fit returned Ok.  After the call
 * the handle's cached state corresponds to the final parameters.
 */
int vp_fit(vp_batch *h, const vp_lm_opts *opts, void *alpha_inout, void *C_out, vp_report *rep);

/*
 * Diagnostics for the per-iteration parity tests: vp_fit that additionally
 * records, per problem, one row [alpha_trial(q), ||r(alpha_trial)||, ratio, delta, par] for every
 * evaluation of the LM loop (row 0: the initial point, ratio = NaN) into
 * trace_out [B][trace_rows][q+4] (f64; rows never written are NaN).  Same address space as the
 * other pointers of the handle.
 */
int vp_fit_trace(vp_batch *h, const vp_lm_opts *opts, void *alpha_inout, void *C_out, vp_report *rep,
                 double *trace_out, int trace_rows);

/* == FitResult::best_fit (src/fit.rs:55-59,87-91): UNWEIGHTED Phi(alpha) * C, [B][S][m] */
int vp_best_fit(vp_batch *h, void *fit_out);

/*
 * == FitStatistics::try_calculate (src/statistics/mod.rs:352-441) for every problem of a single-RHS batch at
 * the handle's current parameters (after vp_fit / vp_set_params); what fit_with_statistics
 * (src/solvers/levmar/mod.rs:275-304) adds to fit.
 *   cov_out            [B][(n+q)][(n+q)]  sigma^2 (H^T H)^-1, H = W [Phi, (dPhi/dalpha_k c)_k]; ordering
 *                      [linear coefficients, nonlinear parameters] (covariance_matrix(), :129)
 *   reduced_chi2_out   [B] f64            ||r_w||^2 / (m - n - q)  (reduced_chi2(), :183)
 *   conf_sigma_out     [B][m] or NULL     sqrt(j_i^T Cov j_i): multiply by the Student-t quantile
 *                                         t_ppf((p+1)/2, m-n-q) to get confidence_band_radius(p) (:271-304)
 *   status             [B] or NULL        0 ok; 4 = Underdetermined / MatrixInversion (the reference's Err)
 */
int vp_statistics(vp_batch *h, void *cov_out, double *reduced_chi2_out, void *conf_sigma_out, int32_t *status);

/*
 * Local batch aggregates for the multi-GPU cost reduction:
 * out = { sum_b 1/2||r_b||^2 , #successful , #failed , sum_b n_evals } over this
 * handle's problems (after vp_fit) -- always HOST doubles.  The cross-rank sum is one
 * RCCL all-reduce of these 4 doubles, issued by the host layer.
 */
int vp_summary(vp_batch *h, double out[4]);

/* Conditioning of the GLOBAL fit's LM steps (handles with S > 1 on a specialised kernel set).  The reference's driver
 * QR-factors the tall (m S) x q Jacobian (src/solvers/levmar/mod.rs:172-186 + qrfac via :247); the device step works on
 * J^T J accumulated over the right-hand sides and is exact to ~10 cond(J)^2 eps (measured within that
 * bound to cond(J) = 3.9e4).  cond_out[b] = the largest estimate of cond(J D^-1) -- J with normalised columns, as the LM
 * driver sees it -- over the steps of the last vp_fit (ratio of the extreme diagonal entries of the pivoted factor): a
 * caller who needs the QR-grade step beyond cond ~ 1e5 can tell from it that a fit ran there.  [B] doubles, host or
 * device pointer like every other array of the handle. */
int vp_global_fit_condition(vp_batch *h, double *cond_out);

/* same aggregates written to 4 DEVICE doubles, enqueued on the handle's stream without any host
 * synchronisation: the buffer can be handed straight to ncclAllReduce (RCCL) */
int vp_summary_device(vp_batch *h, double *dev_out4);

/*
 * The scalar LM cost reduction of SURVEY.md 8(b) / 8(e) for hosts WITHOUT a collective layer of their own (plain C / C++ /
 * Rust; Python callers use torch.distributed on vp_summary_device): the handle's aggregates
 * { sum 1/2||r||^2, #successful, #failed, sum n_evals } summed over all ranks of `rccl_comm` (an ncclComm_t) by ONE
 * ncclAllReduce of 4 doubles on the handle's stream -- 32 bytes over xGMI -- and returned in HOST doubles.  The library
 * does not link RCCL: ncclAllReduce is resolved at the first call from the process (a host that links librccl) or, failing
 * that, from librccl.so.1 (dlopen); VP_ERR_UNSUPPORTED if neither is there.  rccl_comm == NULL: the local aggregates
 * (== vp_summary).  Every rank of the communicator must make the call (it is a collective).
 */
int vp_reduce_cost(vp_batch *h, void *rccl_comm, double out[4]);

/*
 * One global fit whose right-hand sides are SHARDED over ranks (second row).  The reference fits
 * all S columns with one shared alpha (src/solvers/levmar/mod.rs:42-73, 154-186); here every rank owns a handle
 * with its block of the S columns (same model, grid, weights; B problems each) and vp_fit needs ONE exchange per
 * LM evaluation: the reduced sums {sum ||r||^2, sum c c^T, sum c_j G_p^T r} of the local columns, B*(1+n*n+p)
 * doubles.  The library leaves the collective to the caller: `fn` must sum `count` DEVICE doubles in place over
 * all ranks, ordered on `hip_stream` (ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, hip_stream) on
 * RCCL), and return 0.  Every rank then runs the identical LM step on identical totals.
 *   global_rhs_count   S of the whole problem (the LM driver's residual count is m * global_rhs_count)
 *   fn == NULL         back to an unsharded handle
 * After vp_fit: alpha and the report (objective = global 1/2 sum ||r||^2) are identical on every rank; C, the
 * residual cache and vp_cost cover the local columns only.
 */
typedef int (*vp_allreduce_fn)(void *dev_doubles, int64_t count, void *hip_stream, void *user);
int vp_set_rhs_allreduce(vp_batch *h, vp_allreduce_fn fn, void *user, int64_t global_rhs_count);

/* Single-RHS fit kernel selection of a handle (explicit state; the library reads no environment variables).
 *   AUTO  : the persistent slot kernel (a wavefront owns several problems whose LM bookkeeping runs lane-parallel,
 *           problems are handed out from a device-side queue) once the batch exceeds the device's resident
 *           wavefront slots, otherwise one wavefront per problem; cases the slot kernel does not cover (weights,
 *           per-problem grids, models without a trailing constant basis) always run one wavefront per problem
 *   WAVE  : always one wavefront per problem
 *   SLOTS : the slot kernel whenever it covers the problem, regardless of the batch size
 * All three implement LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:247) with identical arithmetic per
 * problem; results do not depend on which wavefront or slot ran a problem.
 * VP_F32 handles whose model needs more than one wavefront's registers (five exponentials + offset beyond 128 rows,
 * BASELINE.json configs[4]) are fitted on the fp64 Gram matrix of [Phi | y | dPhi] (normal equations in double, any grid,
 * any weights) WHATEVER the selection: there is no fp32 Householder fit kernel for these shapes (it lost 15 % of the fits
 * to non-finite evaluations); vp_evaluate / vp_residuals / vp_jacobian / vp_statistics keep the fp32 Householder kernels. */
enum { VP_FIT_KERNEL_AUTO = 0, VP_FIT_KERNEL_WAVE = 1, VP_FIT_KERNEL_SLOTS = 2 };
int vp_set_fit_kernel(vp_batch *h, int which);

/* ---- introspection --------------------------------------------------------------------- */

/* duration in ms of the most recent launch of a kernel family on this handle, measured
 * with hipEvents on the handle's stream (enabled by vp_set_timing(h,1)). */
enum { VP_KERNEL_EVALUATE = 0, VP_KERNEL_BASIS = 1, VP_KERNEL_FIT = 2 };
int vp_set_timing(vp_batch *h, int enable);
int vp_last_kernel_ms(vp_batch *h, int which, float *ms);

int vp_synchronize(vp_batch *h);
const char *vp_last_error(void);
int vp_last_error_detail(void);
const char *vp_version(void);
/* number of visible HIP devices (0 if none); never fails */
int vp_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* VARPRO_HIP_H */
