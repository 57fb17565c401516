// vp_extfit.hpp -- a BATCHED Levenberg-Marquardt fit of CALLER-EVALUATED models by reverse communication.
//
// == LevMarSolver::fit -> levenberg_marquardt::LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:238-254, call
// site :247) over ANY `SeparableNonlinearModel` (src/model/mod.rs:239-363), for a batch.  The reference's driver calls
// the trait: set_params(x_trial) -> model.eval() (:43-45), residuals() (:91-95) and, at accepted points only,
// jacobian() -> model.eval_partial_deriv(k) (:141).  A model the device cannot evaluate crosses the C ABI as those VALUES;
// everything else of the loop stays on the device:
//
//   vp_fit_begin            one LM record per problem (LmVars, vp_lm_core.hpp) from alpha0
//   vp_fit_step_with_basis  ONE launch of ext_fit_step_kernel: per problem, with Phi (and dPhi) at the current trial point
//                             Phi_w = W Phi, Householder QR applied to [y_w | W dPhi], truncated solve, ||r||     :42-73
//                             trust-region update, accept / reject, termination tests            (lm_after_eval)
//                             at an accepted point: Kaufman columns in Q-coordinates, MINPACK qrfac of the m x q
//                             Jacobian with Q_J^T r alongside (jac_qrfac)                                      :101-201
//                             gradient test, diag, lmpar, predicted reduction, next trial point   (lm_next_step)
//                           and hands back ONLY alpha_trial [B][q] + what it wants next [B] (+ the active count)
//   vp_fit_end              parameters, coefficients and MinimizationReport of every problem
//
// The Jacobian J [B][q][m] and the residuals never leave the chip (the trait-level route -- vp_evaluate_with_basis and one
// host LM driver per problem -- writes and copies them at every accepted point: 1 GB per step at the headline shape).
// One wavefront per problem, columns in registers (R rows per lane) exactly as ext_evaluate_kernel (vp_ext.hpp); a step
// streams m (n + 1 + p) scalars per active problem in and q + 1 scalars out and is HBM-bound.
//
// Two protocols (vp_fit_begin flags):
//   eager (default)              every step carries Phi AND dPhi at the trial point; a step is one LM iteration
//   VP_FIT_DERIVATIVES_ON_ACCEPT the driver's own order: a trial point is evaluated with Phi alone; a problem that
//                                accepts it asks for the derivative columns AT THAT POINT (want = basis | derivatives, same
//                                alpha_trial) and forms its Jacobian in the next step -- eval_partial_deriv is then called
//                                exactly as often as the reference calls it, at the price of a second pass over Phi
// Evaluation counts (MinimizationReport::number_of_evaluations) count residual evaluations, as the reference's do; the
// Jacobian pass of the second protocol is not one.
#pragma once
#include <vector>

#include "vp_ext.hpp"
#include "vp_lm_core.hpp"

namespace vp {
namespace ext {

enum { EXTFIT_WANT_BASIS = 1, EXTFIT_WANT_DERIVS = 2 }; // == VP_WANT_* (include/varpro_hip.h)
enum { EXTFIT_PH_EVAL = 0, EXTFIT_PH_JAC = 1 };

template <typename T, int N, int Q> struct ExtFitRec {
    LmVars<T, N, Q> lm;
    T cbest[N];
    int want;  // what the caller was asked to provide for THIS step
    int phase; // EXTFIT_PH_*
};

// field-by-field copies of an LM record (a struct assignment is a memcpy through a stack object: the whole record would
// live in scratch memory for the duration of the kernel)
template <typename T, int N, int Q>
__device__ __forceinline__ void lm_copy(LmVars<T, N, Q> &d, const LmVars<T, N, Q> &s) {
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        d.x[k] = s.x[k];
        d.xt[k] = s.xt[k];
        d.diag[k] = s.diag[k];
        d.qtf[k] = s.qtf[k];
        d.acnorm[k] = s.acnorm[k];
        d.ipvt[k] = s.ipvt[k];
#pragma unroll
        for (int j = 0; j < Q; ++j) d.Rj[k][j] = s.Rj[k][j];
    }
    d.fnorm = s.fnorm;
    d.delta = s.delta;
    d.par = s.par;
    d.xnorm = s.xnorm;
    d.gnorm = s.gnorm;
    d.pnorm = s.pnorm;
    d.prered = s.prered;
    d.dirder = s.dirder;
    d.objective = s.objective;
    d.first = s.first;
    d.first_tr = s.first_tr;
    d.first_update = s.first_update;
    d.nfev = s.nfev;
    d.term = s.term;
    d.status = s.status;
    d.accepted = s.accepted;
}

template <typename T> struct ExtFitArgs {
    const T *phi;  // [B][N][m]   UNWEIGHTED, at alpha_trial of the previous step (alpha0 for the first)
    const T *dphi; // [B][np][m]  UNWEIGHTED, pair-table order (null: no problem may want derivatives)
    const T *w;
    const T *yw;   // [B][m]
    void *state;   // [B] ExtFitRec
    const T *alpha0; // [B][q]: read when init != 0
    T *alpha_best;   // [B][q] best point so far      (the handle's parameter array)
    T *C_best;       // [B][n] its coefficients       (the handle's coefficient array)
    double *cost;    // [B]
    int32_t *status; // [B]
    vp_report *report; // [B] termination == 0 while the problem is running
    T *alpha_trial;  // [B][q] out: where Phi (and dPhi) are wanted next; the final parameters once a problem is done
    int32_t *want;   // [B] out: EXTFIT_WANT_* bits, 0 = done
    int32_t *nactive; // out: number of problems still running after this step (atomic; zeroed by the host)
    int32_t pb[VP_MAX_PAIRS], pp[VP_MAX_PAIRS];
    int np;
    int m;
    int64_t B;
    int64_t w_stride;
    T eps;
    LmOpts<T> o;
    int init; // first step after vp_fit_begin: records are created from alpha0
    int lazy; // VP_FIT_DERIVATIVES_ON_ACCEPT
    int vec;
};

template <typename T, int R, int N, int P, int Q> constexpr int extfit_waves() {
    // resident: N + 1 + P columns during the sweep, 1 + P + Q afterwards, plus ~(2N^2 + 6N + Q^2 + 8Q) wave-uniform values
    constexpr int cols = (N + 1 + P) > (1 + P + Q) ? (N + 1 + P) : (1 + P + Q);
    return ((cols * R + 2 * N * N + 6 * N + 2 * Q * Q + 8 * Q) * (int)(sizeof(T) / 4) <= 200) ? 2 : 1;
}

template <typename T, int N, int P, int Q, int R>
__global__ void __launch_bounds__(64, (extfit_waves<T, R, N, P, Q>())) ext_fit_step_kernel(const ExtFitArgs<T> a) {
    constexpr int NC = N + 1 + P;
    using G = Grp<1>;
    using L = Layout<R, 1>;
    using Rec = ExtFitRec<T, N, Q>;
    G grp = G::make(nullptr);
    const int lane = grp.gl;
    const int64_t b = blockIdx.x;
    if (b >= a.B) return;
    const int m = a.m;
    const bool vec = a.vec != 0;
    Rec *rec = reinterpret_cast<Rec *>(a.state) + b;

    LmVars<T, N, Q> s;
    T cbest[N];
    int want, phase;
    if (a.init) {
        lm_init<T, N, Q>(s, a.alpha0 + b * Q);
#pragma unroll
        for (int k = 0; k < N; ++k) cbest[k] = T(0);
        want = EXTFIT_WANT_BASIS | EXTFIT_WANT_DERIVS;
        phase = EXTFIT_PH_EVAL;
    } else {
        if (uni(rec->lm.term) != 0) return; // finished in an earlier step: nothing is read, nothing changes
        lm_copy<T, N, Q>(s, rec->lm);
#pragma unroll
        for (int k = 0; k < Q; ++k) s.ipvt[k] = uni(s.ipvt[k]);
        s.first = uni(s.first);
        s.first_tr = uni(s.first_tr);
        s.first_update = uni(s.first_update);
        s.nfev = uni(s.nfev);
        s.term = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) cbest[k] = rec->cbest[k];
        want = uni(rec->want);
        phase = uni(rec->phase);
    }
    const bool with_d = (want & EXTFIT_WANT_DERIVS) != 0 && a.dphi != nullptr; // (uniform)

    T C[NC][R];
    {
        const T *ph = a.phi + b * (int64_t)N * m;
#pragma unroll
        for (int j = 0; j < N; ++j) load_rows<T, R, 1>(ph + (int64_t)j * m, m, lane, vec, C[j]);
        load_rows<T, R, 1>(a.yw + b * (int64_t)m, m, lane, vec, C[N]);
        const T *dp = a.dphi + b * (int64_t)a.np * m;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (with_d && p < a.np) {
                load_rows<T, R, 1>(dp + (int64_t)p * m, m, lane, vec, C[N + 1 + p]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) C[N + 1 + p][r] = T(0);
            }
        }
        if (a.w) { // `&self.weights * ...` (src/util/weights.rs:82-99)
            T wt[R];
            load_rows<T, R, 1>(a.w + b * a.w_stride, m, lane, vec, wt);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (j == N) continue; // y_w was weighted when the handle was made
#pragma unroll
                for (int r = 0; r < R; ++r) C[j][r] *= wt[r];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- set_params at the trial point: src/solvers/levmar/mod.rs:42-73 ----
    T g[N], Rm[N][N], qty[N], c[N], e[N];
    house_qr<T, R, N, NC, 0, true, G>(C, g, Rm, qty, grp);
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, a.eps, c, e, truncated);
    T sq = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const T v = (r >= L::VW || L::row_of(r, lane) >= N) ? C[N][r] : T(0);
        sq = tfma(v, v, sq);
    }
    T fn2 = group_sum(grp, sq);
#pragma unroll
    for (int k = 0; k < N; ++k) fn2 = tfma(e[k], e[k], fn2);
    bool ok = is_finite(fn2);
#pragma unroll
    for (int k = 0; k < N; ++k) ok = ok && is_finite(c[k]) && is_finite(Rm[k][k]);
    ok = uni(ok);

    // ---- the driver's bookkeeping around that evaluation ----
    bool need_jac;
    if (phase == EXTFIT_PH_JAC) {
        need_jac = true; // the columns are those of the point accepted in the previous step: no new evaluation
    } else {
        need_jac = lm_after_eval<T, N, Q, true>(s, a.o, usqrt(fn2), ok, (long)m);
        if (s.accepted) {
#pragma unroll
            for (int k = 0; k < N; ++k) cbest[k] = c[k];
        }
    }
    bool deferred = false;
    if (s.term == 0 && need_jac && !with_d) {
        // accepted without derivative columns at hand (second protocol): ask for them at this very point
        deferred = true;
    } else if (s.term == 0) {
        if (need_jac) {
            // Kaufman columns in Q-coordinates (rows < N: the P_perp), MINPACK qrfac with Q_J^T r alongside: :101-201
            residual_qcoords<T, R, N>(C[N], e, grp);
            T Zs[Q][R];
#pragma unroll
            for (int k = 0; k < Q; ++k) {
#pragma unroll
                for (int r = 0; r < R; ++r) Zs[k][r] = T(0);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (p < a.np && a.pp[p] == k) { // (uniform)
                        const T cj = -dyn_get<N>(c, a.pb[p]);
#pragma unroll
                        for (int r = 0; r < R; ++r) Zs[k][r] = tfma(cj, C[N + 1 + p][r], Zs[k][r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < L::VW && r < R; ++r)
                    if (L::row_of(r, lane) < N) Zs[k][r] = T(0);
            }
            // (factor into locals: handing the record's own arrays to the factorisation pins the whole record in scratch)
            T Rj[Q][Q], acn[Q], qtf[Q];
            int ipv[Q];
            jac_qrfac<T, R, Q, N>(Zs, C[N], Rj, acn, ipv, qtf, grp);
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                s.acnorm[k] = acn[k];
                s.qtf[k] = qtf[k];
                s.ipvt[k] = ipv[k];
#pragma unroll
                for (int j = 0; j < Q; ++j) s.Rj[k][j] = Rj[k][j];
            }
        }
        lm_next_step<T, N, Q, true>(s, a.o, need_jac);
#ifdef VP_EXTFIT_DEBUG
        if (s.term == VP_TERM_NUMERICAL && lane == 0) {
            printf("extfit b=%ld nfev=%d NUMERICAL after step: need_jac=%d fnorm=%g gnorm=%g xnorm=%g delta=%g par=%g pnorm=%g prered=%g\n",
                   (long)b, s.nfev, (int)need_jac, (double)s.fnorm, (double)s.gnorm, (double)s.xnorm, (double)s.delta, (double)s.par,
                   (double)s.pnorm, (double)s.prered);
            for (int k = 0; k < Q; ++k)
                printf("   k=%d x=%g acnorm=%g diag=%g qtf=%g ipvt=%d R=[%g %g %g %g] c=%g\n", k, (double)s.x[k], (double)s.acnorm[k],
                       (double)s.diag[k], (double)s.qtf[k], s.ipvt[k], (double)s.Rj[k][0], (double)s.Rj[k][Q > 1 ? 1 : 0],
                       (double)s.Rj[k][Q > 2 ? 2 : 0], (double)s.Rj[k][Q > 3 ? 3 : 0], (double)c[k < N ? k : 0]);
        }
#endif
    }

    // ---- hand back: the next request, the record, the best point so far ----
    int want_next;
    if (s.term != 0) {
        want_next = 0;
        phase = EXTFIT_PH_EVAL;
    } else if (deferred) {
        want_next = EXTFIT_WANT_BASIS | EXTFIT_WANT_DERIVS;
        phase = EXTFIT_PH_JAC;
    } else {
        want_next = a.lazy ? EXTFIT_WANT_BASIS : (EXTFIT_WANT_BASIS | EXTFIT_WANT_DERIVS);
        phase = EXTFIT_PH_EVAL;
    }
    if (lane == 0) {
        lm_copy<T, N, Q>(rec->lm, s);
#pragma unroll
        for (int k = 0; k < N; ++k) rec->cbest[k] = cbest[k];
        rec->want = want_next;
        rec->phase = phase;
        a.want[b] = want_next;
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            // a finished problem reports its final parameters; a deferred one repeats the accepted point (x == xt)
            a.alpha_trial[b * Q + k] = (s.term != 0 || deferred) ? s.x[k] : s.xt[k];
            a.alpha_best[b * Q + k] = s.x[k];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) a.C_best[b * N + k] = cbest[k];
        vp_report rep;
        rep.termination = s.term;
        rep.n_evals = s.nfev;
        rep.objective = (double)s.objective;
        a.report[b] = rep;
        a.cost[b] = (double)s.objective;
        a.status[b] = s.status;
        if (s.term == 0) atomicAdd(a.nactive, 1);
    }
}

template <typename T, int N, int P, int Q, int R> int launch_fit_step(const ExtFitArgs<T> &a, hipStream_t stream) {
    hipLaunchKernelGGL((ext_fit_step_kernel<T, N, P, Q, R>), dim3((unsigned)a.B), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// one row of the table of compiled shapes (vp_inst_extfit*.hip)
template <typename T> struct ExtFitEntry {
    int N, P, Q, R;
    size_t rec_bytes;
    int (*launch)(const ExtFitArgs<T> &, hipStream_t);
};
template <typename T> std::vector<ExtFitEntry<T>> &extfit_table() {
    static std::vector<ExtFitEntry<T>> t;
    return t;
}
template <typename T> struct ExtFitRegistrar {
    explicit ExtFitRegistrar(const ExtFitEntry<T> &e) { extfit_table<T>().push_back(e); }
};

// the step kernel that covers (n, np pairs, q, m): exact n and q, the fewest rows per lane, then the fewest spare pairs
template <typename T> const ExtFitEntry<T> *find_extfit(int n, int np, int q, int64_t m) {
    const ExtFitEntry<T> *best = nullptr;
    for (const ExtFitEntry<T> &e : extfit_table<T>()) {
        if (e.N != n || e.Q != q || e.P < np || 64 * (int64_t)e.R < m) continue;
        if (!best || e.R < best->R || (e.R == best->R && e.P < best->P)) best = &e;
    }
    return best;
}

} // namespace ext
} // namespace vp

#define VP_REGISTER_EXTFIT(T, NN, PP, QQ, RR)                                                                          \
    static ::vp::ext::ExtFitRegistrar<T> VP_EXT_CAT(vp_extfit_reg_, __COUNTER__)(::vp::ext::ExtFitEntry<T>{            \
        NN, PP, QQ, RR, sizeof(::vp::ext::ExtFitRec<T, NN, QQ>), &::vp::ext::launch_fit_step<T, NN, PP, QQ, RR>});
