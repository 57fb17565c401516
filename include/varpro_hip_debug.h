/*
 * varpro_hip_debug.h -- TEST HOOKS of libvarpro_hip.so.  Not part of the drop-in boundary (include/varpro_hip.h): these
 * two entries expose intermediate quantities of one kernel family to the parity tests (tests/test_gpu_gram_parity.py,
 * tests/test_gpu_lmpar_gram.py) and replace nothing in the reference.  A binding of the reference does not need them.
 */
#ifndef VARPRO_HIP_DEBUG_H
#define VARPRO_HIP_DEBUG_H

#include "varpro_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Diagnostics for the fixed-alpha parity tests of the fp64-Gram fit kernel (VP_F32 handles whose fit runs on the normal
 * equations in double, see vp_set_fit_kernel): evaluates that kernel's formulation ONCE at alpha [B][q] and writes, per
 * problem, out[b] = { 1/2 ||r||^2, c (n), J^T r (q), J^T J (q x q, row-major) } as f64 -- the quantities the kernel
 * hands to its LM step, i.e. set_params + residuals + jacobian (src/solvers/levmar/mod.rs:42-73, 101-201) contracted
 * the way the LM driver consumes them.  Does not touch the handle's cached state.  VP_ERR_UNSUPPORTED for handles whose
 * fit does not use the Gram kernel.  `out` follows the handle's address space.
 */
int vp_debug_gram_evaluate(vp_batch *h, const void *alpha, double *out);

/*
 * Diagnostics for the parity test of the Gram fit kernel's LM step: the trust-region sub-problem of
 * LevenbergMarquardt::minimize (levenberg-marquardt 0.14 == MINPACK lmpar; call site src/solvers/levmar/mod.rs:247) as that
 * kernel solves it -- on the Cholesky factor of R^T R + par D^2 instead of qrsolv's Givens rotations (lmpar_chol,
 * varpro_amd/csrc/vp_fit.hpp) -- run once per record on the device.  HOST pointers; q in {2, 3, 5}.
 *   Rj [B][q][q] row-major upper-triangular factor of the pivoted Jacobian, ipvt [B][q], diag [B][q], qtb [B][q] (first q
 *   entries of Q^T f), delta [B], par_in [B]  ->  out [B][q + 2] = { par, ||diag .* step||, step (q, unpivoted order) }.
 */
int vp_debug_lmpar_gram(int64_t B, int q, const double *Rj, const int32_t *ipvt, const double *diag, const double *qtb,
                        const double *delta, const double *par_in, double *out);

/*
 * Diagnostics for the flag-and-refit tests: enabled != 0 (the default) lets vp_fit re-fit, in a second launch with
 * power-of-two column scaling, the problems whose Jacobian factor is not representable column by column -- the reference
 * forms D_k c before it projects (src/solvers/levmar/mod.rs:156-171); 0 returns what the fit kernels themselves report for
 * them (VP_TERM_NUMERICAL, parameters = the initial guess).  Single-right-hand-side descriptor handles only.
 */
int vp_debug_set_refit(vp_batch *h, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* VARPRO_HIP_DEBUG_H */
