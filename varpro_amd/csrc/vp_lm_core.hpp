// vp_lm_core.hpp -- the Levenberg-Marquardt bookkeeping as two reusable device steps.
//
// == levenberg_marquardt::LevenbergMarquardt::minimize (call site src/solvers/levmar/mod.rs:247), MINPACK
// lmder semantics with the crate's termination rules, split around the point where
// the caller has to produce numbers from the data:
//
//   lm_after_eval : given ||r(x_trial)|| (and whether the evaluation succeeded) -> trust-region update,
//                   accept/reject, termination tests.  Returns whether a fresh Jacobian factor is needed.
//   lm_next_step  : given (if refreshed) the pivoted QR factor of the Jacobian (Rj, qtf, acnorm, ipvt) ->
//                   gradient test, diag scaling, lmpar, predicted reduction, next trial point.
//
// Used by the multiple-right-hand-side path (vp_mrhs.hpp); the single-RHS kernels (vp_fit.hpp,
// vp_fit2.hpp) carry the same logic inlined around their register-resident columns.
#pragma once
#include "vp_fit.hpp"

namespace vp {

template <typename T, int N, int Q> struct LmVars {
    T x[Q], xt[Q], diag[Q], qtf[Q], acnorm[Q];
    T Rj[Q][Q];
    T fnorm, delta, par, xnorm, gnorm, pnorm, prered, dirder, objective;
    int ipvt[Q];
    int first, first_tr, first_update;
    int nfev;
    int term;   // VP_TERM_*, 0 while running
    int status; // VP_ST_* of the best point
    int accepted; // the latest evaluation became the new best point (outputs of lm_after_eval)
};

template <typename T> struct LmOpts {
    T ftol, xtol, gtol, stepbound;
    int patience;
    int scale_diag;
};

template <typename T, int N, int Q> __device__ __forceinline__ void lm_init(LmVars<T, N, Q> &s, const T *alpha0) {
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        s.x[k] = s.xt[k] = alpha0[k];
        s.diag[k] = T(1);
        s.qtf[k] = s.acnorm[k] = T(0);
        s.ipvt[k] = k;
#pragma unroll
        for (int j = 0; j < Q; ++j) s.Rj[k][j] = T(0);
    }
    s.fnorm = s.delta = s.par = s.xnorm = s.gnorm = s.pnorm = s.prered = s.dirder = T(0);
    s.objective = T(0) / T(0);
    s.first = s.first_tr = s.first_update = 1;
    s.nfev = 0;
    s.term = VP_TERM_NOT_RUN;
    s.status = VP_ST_NOT_EVALUATED;
    s.accepted = 0;
}

// Returns true if the caller must refresh the Jacobian factor (first evaluation or accepted step) and then
// call lm_next_step; returns false if the run terminated (s.term != 0) or the step was rejected (then call
// lm_next_step with jac_refreshed = false).
template <typename T, int N, int Q, bool U>
__device__ __forceinline__ bool lm_after_eval(LmVars<T, N, Q> &s, const LmOpts<T> &o, const T fnorm1, const bool ok,
                                              const long mres) {
    s.accepted = 0;
    if (s.first) {
        s.first = 0;
        s.nfev = 1;
        s.status = ok ? VP_ST_OK : VP_ST_NONFINITE;
        if (!ok) {
            s.term = VP_TERM_USER;
            return false;
        }
        s.fnorm = fnorm1;
        s.objective = T(0.5) * fnorm1 * fnorm1;
        s.accepted = 1;
        if ((long)Q > mres) s.term = VP_TERM_WRONG_DIMENSIONS;
        else if (!is_finite(fnorm1)) s.term = VP_TERM_NUMERICAL;
        else if (fnorm1 <= num<T>::tiny) s.term = VP_TERM_RESIDUALS_ZERO;
        return s.term == 0;
    }
    s.nfev += 1;
    if (!ok) { // residuals() == None at the trial point: the problem keeps the trial parameters
        s.term = VP_TERM_USER;
#pragma unroll
        for (int k = 0; k < Q; ++k) s.x[k] = s.xt[k];
        s.accepted = 1;
        s.status = VP_ST_NONFINITE;
        return false;
    }
    const T q1 = fnorm1 * frcp(s.fnorm);
    const T actred = (fnorm1 * T(0.1) < s.fnorm) ? T(1) - q1 * q1 : T(-1);
    const T ratio = (s.prered == T(0)) ? T(0) : actred * frcp(s.prered);
    if (ratio <= T(0.25)) {
        T temp = !(actred < T(0)) ? T(0.5) : T(0.5) * s.dirder * frcp(s.dirder + T(0.5) * actred);
        if (fnorm1 * T(0.1) >= s.fnorm || temp < T(0.1)) temp = T(0.1);
        s.delta = temp * tmin(s.delta, s.pnorm * T(10));
        s.par = s.par * frcp(temp);
    } else if (s.par == T(0) || ratio >= T(0.75)) {
        s.delta = s.pnorm * T(2);
        s.par = s.par * T(0.5);
    }
    const bool good = pol<U>(ratio >= T(1.0e-4));
    if (good) {
#pragma unroll
        for (int k = 0; k < Q; ++k) s.x[k] = s.xt[k];
        s.accepted = 1;
        T tmpv[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) tmpv[k] = o.scale_diag ? s.diag[k] * s.x[k] : s.x[k];
        s.xnorm = enorm_small<T, Q, U>(tmpv);
        s.fnorm = fnorm1;
        s.objective = T(0.5) * fnorm1 * fnorm1;
        if (!is_finite(s.xnorm)) {
            s.term = VP_TERM_NUMERICAL;
            return false;
        }
    }
    int tcode = 0;
    if (s.fnorm <= num<T>::tiny) tcode = VP_TERM_RESIDUALS_ZERO;
    if (!tcode) {
        const bool ftol_check = tabs(actred) <= o.ftol && s.prered <= o.ftol && ratio * T(0.5) <= T(1);
        const bool xtol_check = s.delta <= o.xtol * s.xnorm;
        if (ftol_check || xtol_check)
            tcode = (ftol_check && xtol_check) ? VP_TERM_CONVERGED_BOTH
                                               : (ftol_check ? VP_TERM_CONVERGED_FTOL : VP_TERM_CONVERGED_XTOL);
    }
    if (!tcode && s.nfev >= o.patience * (Q + 1)) tcode = VP_TERM_LOST_PATIENCE;
    if (!tcode && tabs(actred) <= num<T>::eps && s.prered <= num<T>::eps && ratio * T(0.5) <= T(1))
        tcode = VP_TERM_NO_IMPROVEMENT;
    if (!tcode && s.delta <= num<T>::eps * s.xnorm) tcode = VP_TERM_NO_IMPROVEMENT;
    if (!tcode && s.gnorm <= num<T>::eps) tcode = VP_TERM_NO_IMPROVEMENT;
    s.term = pol<U>(tcode);
    return s.term == 0 && good;
}

// Returns true when a REFRESHED factor's column norms are not finite (the fit then ends `Numerical`): the one failure a re-fit
// with scaled columns repairs (vp_fit.hpp, jac_not_finite) -- callers with a rescue list flag the problem, the others ignore it.
template <typename T, int N, int Q, bool U>
__device__ __forceinline__ bool lm_next_step(LmVars<T, N, Q> &s, const LmOpts<T> &o, const bool jac_refreshed) {
    if (s.term != 0) return false;
    if (jac_refreshed) {
        T gmax = T(0);
        bool degenerate = false;
        if (pol<U>(jac_not_finite<T, Q>(s.acnorm))) {
            s.term = VP_TERM_NUMERICAL;
            return true;
        }
        const T ifn = frcp(s.fnorm);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const T an = dyn_get_o<Q, true>(s.acnorm, s.ipvt[j]);
            if (an != T(0)) {
                T sum = T(0);
#pragma unroll
                for (int i = 0; i <= j; ++i) sum = tfma(s.Rj[i][j], s.qtf[i], sum);
                const T temp = tabs(sum * frcp(an) * ifn);
                if (temp != temp) degenerate = true;
                gmax = tmax(gmax, temp);
            }
        }
        s.gnorm = gmax;
        if (pol<U>(degenerate)) {
            s.term = VP_TERM_NUMERICAL;
            return false;
        }
        if (pol<U>(s.gnorm <= o.gtol)) {
            s.term = VP_TERM_ORTHOGONAL;
            return false;
        }
        if (s.first_update) {
            T tmpv[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                if (o.scale_diag) s.diag[k] = (s.acnorm[k] == T(0)) ? T(1) : s.acnorm[k];
                tmpv[k] = o.scale_diag ? s.diag[k] * s.x[k] : s.x[k];
            }
            s.xnorm = enorm_small<T, Q, U>(tmpv);
            if (pol<U>(!is_finite(s.xnorm))) {
                s.term = VP_TERM_NUMERICAL;
                return false;
            }
            s.delta = (s.xnorm == T(0)) ? o.stepbound : o.stepbound * s.xnorm;
            s.first_update = 0;
        } else if (o.scale_diag) {
#pragma unroll
            for (int k = 0; k < Q; ++k) s.diag[k] = tmax(s.diag[k], s.acnorm[k]);
        }
    }
    T step[Q];
    T Rwork[Q][Q]; // lmpar scribbles on the lower triangle
#pragma unroll
    for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int j = 0; j < Q; ++j) Rwork[i][j] = s.Rj[i][j];
    s.par = lmpar_any<T, Q, U, true>(Rwork, s.ipvt, s.diag, s.qtf, s.delta, s.par, step, s.pnorm);
    if (pol<U>(!is_finite(s.pnorm))) {
        s.term = VP_TERM_NUMERICAL;
        return false;
    }
    T wa[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) wa[i] = T(0);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const T pj = dyn_get_o<Q, true>(step, s.ipvt[j]);
#pragma unroll
        for (int i = 0; i <= j; ++i) wa[i] = tfma(s.Rj[i][j], pj, wa[i]);
    }
    const T ifn = frcp(s.fnorm);
    const T t1 = enorm_small<T, Q, U>(wa) * ifn;
    const T temp1 = t1 * t1;
    const T t2 = (fsqrt(s.par) * s.pnorm) * ifn;
    const T temp2 = t2 * t2;
    if (pol<U>(!is_finite(temp1) || !is_finite(temp2))) {
        s.term = VP_TERM_NUMERICAL;
        return false;
    }
    s.prered = temp1 + temp2 * T(2);
    s.dirder = -(temp1 + temp2);
    if (s.first_tr && s.pnorm < s.delta) s.delta = s.pnorm;
    s.first_tr = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) s.xt[k] = s.x[k] - step[k];
    return false;
}

// Pivoted Cholesky of the Gram matrix A = J^T J (q x q) -> the quantities MINPACK's qrfac/lmder deliver from a
// pivoted QR of J:  acnorm_k = ||J_k||, permutation by largest remaining (downdated) column norm, upper
// triangular Rj with Rj^T Rj = P^T A P, and qtf = Rj^{-T} P^T (J^T r)  (== first q entries of Q_J^T r).
// Used where J is never materialised (multiple right-hand sides: J^T J, J^T r are streamed reductions).
template <typename T, int Q>
__device__ __forceinline__ void gram_to_qr(const T (&A)[Q][Q], const T (&b)[Q], T (&Rj)[Q][Q], T (&acnorm)[Q],
                                           int (&ipvt)[Q], T (&qtf)[Q]) {
    T W[Q][Q], bw[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        acnorm[i] = tsqrt(tmax(A[i][i], T(0)));
        ipvt[i] = i;
        bw[i] = b[i];
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            W[i][j] = A[i][j];
            Rj[i][j] = T(0);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        // pivot: largest remaining diagonal (== largest downdated column norm)
        int kmax = j;
        T dmax = W[j][j];
#pragma unroll
        for (int k = j + 1; k < Q; ++k)
            if (W[k][k] > dmax) {
                dmax = W[k][k];
                kmax = k;
            }
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            if (kmax == k) { // symmetric swap j <-> k of W, swap of bw, ipvt and the finished rows of Rj
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const T tmp = W[i][j];
                    W[i][j] = W[i][k];
                    W[i][k] = tmp;
                }
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const T tmp = W[j][i];
                    W[j][i] = W[k][i];
                    W[k][i] = tmp;
                }
#pragma unroll
                for (int i = 0; i < j; ++i) {
                    const T tmp = Rj[i][j];
                    Rj[i][j] = Rj[i][k];
                    Rj[i][k] = tmp;
                }
                const T tb = bw[j];
                bw[j] = bw[k];
                bw[k] = tb;
                const int ti = ipvt[j];
                ipvt[j] = ipvt[k];
                ipvt[k] = ti;
            }
        }
        const T d = W[j][j];
        if (!(d > T(0))) { // rank deficient from here on: zero row (MINPACK leaves rdiag = 0)
            qtf[j] = T(0);
            continue;
        }
        const T rjj = tsqrt(d);
        const T inv = T(1) / rjj;
        Rj[j][j] = rjj;
#pragma unroll
        for (int k = j + 1; k < Q; ++k) Rj[j][k] = W[j][k] * inv;
        qtf[j] = bw[j] * inv;
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            bw[k] = tfma(-Rj[j][k], qtf[j], bw[k]);
#pragma unroll
            for (int l = j + 1; l < Q; ++l) W[k][l] = tfma(-Rj[j][k], Rj[j][l], W[k][l]);
        }
    }
}

} // namespace vp
