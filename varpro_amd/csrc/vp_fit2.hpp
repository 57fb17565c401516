// vp_fit2.hpp -- device-resident Levenberg-Marquardt, PERSISTENT SLOT formulation (the default fit kernel for
// unit-weight problems on a shared grid whose model ends in a constant column).
//
// == LevMarSolver::fit -> levenberg_marquardt::LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:238-254, call
// site :247) for a batch.  Same arithmetic per problem as fit_kernel (vp_fit.hpp); what changes is who pays for the
// wave-uniform LM bookkeeping.  gfx950 has no scalar fp64 unit, so in fit_kernel every scalar operation of the trust
// region update / lmpar / qrsolv costs a full 64-lane VALU instruction -- a quarter of all instructions of an LM
// iteration, and the fit kernel is VALU-issue bound (profiles/r01_fit_kernel_valu_pmc.json).  Here a wavefront is
// PERSISTENT and owns G problem SLOTS whose weighted data columns stay in LDS for the whole fit:
//
//   VECTOR phase   for each occupied slot in turn: all 64 lanes run the fused Householder sweep of that problem at
//                  its trial point and, if the step is accepted, the pivoted QR of its Jacobian -- the code of
//                  fit_kernel -- and lane 0 posts the handful of wave-uniform results in the slot's LDS record;
//   SCALAR phase   lane s runs the LM bookkeeping of slot s: G problems' scalar work in ONE pass of instructions;
//   REFILL         a slot whose fit terminated writes its results and takes the next problem off a global queue
//                  (one atomic per fit), so waves never idle while problems remain and the slowest fits do not pin
//                  a workgroup's partners.
//
// G is bounded by LDS (one m-row column per slot, 8 waves per CU): 2 at m = 1024 fp64, more for shorter problems.
// Problems too long for one wave run on a GROUP of W waves (one workgroup); there the scalar phase additionally runs
// on ONE wave instead of being repeated by all W (fp32 five-exponential fits at m = 4096: BASELINE configs[4]).
// The constant column's reflector H_0 does not depend on alpha, so a slot stores H_0 y_w instead of y_w (applied once
// per fit when the slot is filled) and the data column skips reflector 0 in every evaluation.
#pragma once
#include "vp_fit.hpp"

namespace vp {

// Per-slot LM record in LDS: lane s owns record s in the scalar phase; lane 0 posts evaluation results into it.
template <typename T, int N, int Q> struct alignas(8) SlotRec {
    T xt[Q], x[Q], diag[Q], qtf[Q], acnorm[Q], cbest[N], cnew[N];
    T Rj[Q][Q];
    T fnorm, delta, par, xnorm, gnorm, pnorm, prered, dirder, objective;
    T fnorm1, actred, ratio; // outputs of the latest evaluation
    T qty0;                  // (H_0 y_w)[0]: first entry of Q^T y_w, fixed for the whole fit
    // wave-uniform scalars of the trial point xt, computed where xt is (scalar phase: ONE lane per slot; slot_fill): 1 / xt_i,
    // exp(-delta / xt_i) (the ratio of the uniform-grid recurrence), then 1 / fnorm and 1 / prered -- the vector phase used to
    // compute them on all 64 lanes once per slot and evaluation (two reciprocals, two refined quotients and two exponentials
    // per exponential column).  Same functions, same inputs: bit-identical (build_columns PRE)
    T pre[2 * Q + 2];
    int ipvt[Q];
    int flags; // bit0 first, bit1 first_tr, bit2 first_update, bit3 eval ok, bit4 jacobian refreshed, bit5 good
    int nfev;
    int term;
    int status;
    int prob; // problem index of the slot, -1 = empty
    int trow; // trace rows written so far
    int pend[2]; // (self-rescue) problems this slot FLAGGED (vp_fit.hpp jac_not_finite), re-fitted by the wave before it ends; -1 = none.
                 // Not touched by slot_fill: the entries outlive the slot's refills
};

// Slots per group from the LDS budget: a workgroup holds ONE copy of the grid and NG groups x GS columns of 64*R*W
// scalars; `blocks_per_cu` workgroups are resident per CU (160 KiB of LDS).  Capped at 8 (lanes 0..GS-1 of ONE wave
// run the scalar phase; beyond 8 the divergent branches of lmpar eat the gain).
#ifndef VP_FIT2_PRE
#define VP_FIT2_PRE 1 /* the trial point's wave-uniform scalars come from the scalar phase (SlotRec::pre): bit-identical, 1 606 -> 1 582 VALU instructions per evaluation, 36 -> 32 spilled VGPRs, and NO time (2.0825 -> 2.079 ms per step: the moved instructions run for two lanes on the scalar phase's dependent chain) */
#endif
#ifndef VP_SLOT_CAP
#define VP_SLOT_CAP 8
#endif
// slots per group: what the LDS of a CU holds next to the shared grid -- per slot one data column and one LM record (REC
// bytes: a wide model's records are not small change at 24 rows per lane), plus 3 KiB of constants / exchange area
template <typename T, int R, int W, int NG, int BLOCKS_PER_CU, int REC = 0> constexpr int fit2_slots() {
    constexpr int col = 64 * R * W * (int)sizeof(T);
    constexpr int budget = (160 * 1024) / BLOCKS_PER_CU - col - 3 * 1024;
    constexpr int gs = budget / (NG * (col + REC));
    return gs < 1 ? 1 : (gs > VP_SLOT_CAP ? VP_SLOT_CAP : gs);
}

#ifndef VP_FITG_TIMELINE
#define VP_FITG_TIMELINE 0 // (vp_fitg.hpp, debug builds)
#endif
template <typename T, class M> struct Fit2Args {
    FitArgs<T, M> f;
    int *queue;        // next problem index to hand out (pre-set to the number of statically assigned problems)
    int waves_total;   // persistent waves in the grid
};


// Launch-wide constants of the slot kernel, staged ONCE per workgroup in LDS: the scalar phase and the refill path
// (both out of line, see below) read them from there, so they do not occupy SGPRs -- i.e. spill slots -- across the
// register-critical vector phase.
//   T = arithmetic type of the LM bookkeeping, TO = storage type of the handle's arrays (they differ in the Gram kernel,
//   vp_fitg.hpp: fp32 data, fp64 bookkeeping)
template <typename T, typename TO = T> struct SlotConsts {
    T ftol, xtol, gtol, stepbound;
    TO *alpha;         // in: initial guesses, out: final parameters  [B][q]
    TO *C_out;         // [B][n] or null
    double *cost_out;  // [B] or null
    int32_t *status;   // [B] or null
    vp_report *report; // [B]
    double *trace;     // [B][trace_rows][q+4] or null
    const TO *yw;      // [B][m]
    int *queue;
    int32_t *rescue;   // flag-and-refit list (vp_fit.hpp jac_not_finite; LaunchParams::rescue) or null
    int rescue_slot;
    int self_rescue;   // 1: flagged problems are parked in the slot's record (SlotRec::pend) and re-fitted by this wave at its end
    int64_t B;
    int trace_rows, scale_diag, max_fev, m;
    T eps;             // (Gram kernel) rank threshold of the linear solve
    T pre_delta;       // grid distance between a lane's consecutive row pairs (RowSource::delta) on a uniform grid, else 0
#ifdef VP_FIT2_CLOCKS
    long long sck[6];  // (debug) section clocks of the scalar phase, accumulated by wave 0 of workgroup 0
#endif
};

// Fill a slot: y' = H_0 y_w into the slot's LDS column (row order, zero padded); returns (H_0 y_w)[0].
template <typename T, int R, class Src, class G>
__device__ __forceinline__ T fill_slot_column(const T *__restrict__ yp, const int m, T *s_col, const Src &src,
                                              const ConstReflector<T> &h0, G &grp) {
    using L = Layout<R, G::W>;
    constexpr int VW = L::VW;
    const int lane = grp.gl;
    T y[R];
    load_rows<T, R, G::W>(yp, m, lane, vec_aligned<T>(yp, m), y);
    T acc = T(0);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int e = 0; e < VW; ++e) acc = tfma(sc[e], y[r0 + e], acc);
    }
    const T d = group_sum(grp, acc);
    const T top = group_row<R>(grp, y, 0);
    const T tau = h0.g * tfma(-h0.beta, top, d);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const T v = (r0 + e < VW && L::row_of(r0 + e, lane) == 0) ? h0.u : sc[e];
            y[r0 + e] = tfma(tau, v, y[r0 + e]);
        }
    }
    store_rows<T, R, G::W>(s_col, 64 * R * G::W, lane, true, y);
    return tfma(tau, h0.u, top);
}

// (Re)fill slot `s` of a wave with problem `prob` (wave-uniform; < 0 marks the slot empty).  Out of line: executed once
// per fit, its register needs must not shape the allocation of the LM loop.
template <typename T, int N, int Q, int R, int W, int PADM>
__device__ __noinline__ void slot_fill(VP_LDS SlotRec<T, N, Q> *rec, VP_LDS T *s_col, VP_LDS const T *s_t,
                                       VP_LDS const SlotConsts<T> *k, VP_LDS unsigned char *xch, const int prob,
                                       const T h0_beta, const T h0_u, const T h0_g) {
    using G = Grp<W>;
    G grp = G::make((unsigned char *)xch);
    const int lane = grp.gl; // group lane
    // W > 1, barrier on entry: (i) every wave decides from the slot's record whether to call this function -- the
    // record must not change before ALL waves have read it (a wave that saw prob = -1 early would skip the call and
    // its barriers: deadlock); (ii) the group's exchange phase restarts at 0 here, so every wave must be done with
    // the caller's last exchange before the buffers are reused
    if constexpr (W > 1) __syncthreads();
    if (prob < 0) {
        if (lane == 0) {
            rec->prob = -1;
            rec->term = VP_TERM_NOT_RUN;
        }
        if constexpr (W > 1) __syncthreads();
        return;
    }
    using Src = RowSource<T, R, true, 0, 1, W, true, PADM>;
    Src src;
    src.t = (const T *)s_t;
    src.w = nullptr;
    src.m = k->m;
    src.lane = lane;
    src.vec = true;
    ConstReflector<T> h0;
    h0.beta = h0_beta;
    h0.u = h0_u;
    h0.g = h0_g;
    h0.live = true;
    const T qty0 = fill_slot_column<T, R, Src, G>(k->yw + (int64_t)prob * k->m, k->m, (T *)s_col, src, h0, grp);
    if (lane == 0) {
        const T *a0 = k->alpha + (int64_t)prob * Q;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const T v = a0[i];
            rec->xt[i] = v;
            rec->x[i] = v;
            rec->diag[i] = T(1);
            rec->qtf[i] = T(0);
            rec->acnorm[i] = T(0);
            rec->ipvt[i] = i;
#pragma unroll
            for (int j = 0; j < Q; ++j) rec->Rj[i][j] = T(0);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            rec->cbest[i] = T(0);
            rec->cnew[i] = T(0);
        }
        rec->fnorm = rec->delta = rec->par = rec->xnorm = rec->gnorm = rec->pnorm = rec->prered = rec->dirder = T(0);
        rec->objective = T(0) / T(0);
        rec->fnorm1 = rec->actred = rec->ratio = T(0);
        rec->qty0 = qty0;
        if constexpr (VP_FIT2_PRE != 0) {
            const T pd = k->pre_delta;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                const T rti = frcp(a0[i]);
                rec->pre[2 * i] = rti;
                rec->pre[2 * i + 1] = texp(-div_refined(pd, a0[i], rti));
            }
            rec->pre[2 * Q] = rec->pre[2 * Q + 1] = T(0); // (first evaluation: no ratio test)
        }
        rec->flags = 1 | 2 | 4;
        rec->nfev = 0;
        rec->term = VP_TERM_NOT_RUN;
        rec->status = VP_ST_NOT_EVALUATED;
        rec->prob = prob;
        rec->trow = 0;
    }
    if constexpr (W > 1) __syncthreads(); // column + record visible to all waves; exchange buffers quiescent again
}

// SCALAR phase: lane s runs the LM bookkeeping of slot s on its LDS record (trust-region update, accept / reject,
// termination tests, gradient test, lmpar, predicted reduction, next trial point) and writes the results of a fit that
// terminated.  == the body of LevenbergMarquardt::minimize between two evaluations.  Out of line (see slot_fill).
// GRAM: the Jacobian factor in the record is the Cholesky factor of a Gram matrix (vp_fitg.hpp) -> lmpar_chol.
// _inl: the body, inlined into its caller (vp_fitg.hpp: one call site per wave role; out of line its 81 callee-saved
// VGPRs were stored and reloaded around every call -- 4 % of configs[4]'s launch).
template <typename T, int N, int Q, int GS, typename TO = T, bool GRAM = false>
__device__ __forceinline__ void slot_scalar_phase_inl(VP_LDS SlotRec<T, N, Q> *recs, VP_LDS const SlotConsts<T, TO> *k, const bool act = true) {
    const int lane = lane_id();
#ifdef VP_FIT2_CLOCKS
    long long sc0_ = __builtin_amdgcn_s_memtime();
    const bool sck_on_ = blockIdx.x == 0 && (threadIdx.x >> 6) == 0;
#define VP_SCK(i)                                                                                                      \
    do {                                                                                                               \
        const long long c1_ = __builtin_amdgcn_s_memtime();                                                           \
        if (sck_on_ && lane == 0) const_cast<SlotConsts<T, TO> *>((const SlotConsts<T, TO> *)k)->sck[i] += c1_ - sc0_;   \
        sc0_ = c1_;                                                                                                    \
    } while (0)
#else
#define VP_SCK(i)
#endif
    if (!(act && lane < GS && recs[lane].prob >= 0)) return;
    VP_LDS SlotRec<T, N, Q> *s = recs + lane;
    const T ftol = k->ftol, xtol = k->xtol, gtol = k->gtol, stepbound = k->stepbound;
    const int scale_diag = k->scale_diag, max_fev = k->max_fev, m = k->m;
    T x[Q], xt[Q], diag[Q], qtf[Q], acnorm[Q], step[Q];
    T Rj[Q][Q];
    int ipvt[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        x[i] = s->x[i];
        xt[i] = s->xt[i];
        diag[i] = s->diag[i];
        qtf[i] = s->qtf[i];
        acnorm[i] = s->acnorm[i];
        ipvt[i] = s->ipvt[i];
        step[i] = T(0);
#pragma unroll
        for (int j = 0; j < Q; ++j) Rj[i][j] = s->Rj[i][j];
    }
    T fnorm = s->fnorm, delta = s->delta, par = s->par, xnorm = s->xnorm, gnorm = s->gnorm;
    T pnorm = s->pnorm, prered = s->prered, dirder = s->dirder, objective = s->objective;
    const T fnorm1 = s->fnorm1, actred = s->actred, ratio = s->ratio;
    const int fl = s->flags;
    bool first = (fl & 1) != 0, first_tr = (fl & 2) != 0, first_update = (fl & 4) != 0;
    const bool ok = (fl & 8) != 0, jac_done = (fl & 16) != 0, good_v = (fl & 32) != 0;
    int nfev = s->nfev, term = 0, status = s->status, trow = s->trow;
    const int64_t prob = s->prob;
    bool accept_c = false; // cbest <- cnew

    auto trace_row = [&](T rt) {
        double *trace = k->trace;
        const int trace_rows = k->trace_rows;
        if (trace && trow < trace_rows) {
            double *tr = trace + ((size_t)prob * trace_rows + trow) * (Q + 4);
#pragma unroll
            for (int i = 0; i < Q; ++i) tr[i] = (double)xt[i];
            tr[Q] = (double)fnorm1;
            tr[Q + 1] = (double)rt;
            tr[Q + 2] = (double)delta;
            tr[Q + 3] = (double)par;
        }
        ++trow;
    };

    bool need_step = false;
    VP_SCK(0);
    if (first) {
        first = false;
        nfev = 1;
        status = ok ? VP_ST_OK : VP_ST_NONFINITE;
        if (!ok) { // residuals() == None
            term = VP_TERM_USER;
        } else {
            fnorm = fnorm1;
            objective = T(0.5) * fnorm * fnorm;
            trace_row(T(0) / T(0));
            accept_c = true;
            if (Q > m) term = VP_TERM_WRONG_DIMENSIONS;
            else if (!is_finite(fnorm)) term = VP_TERM_NUMERICAL;
            else if (fnorm <= num<T>::tiny) term = VP_TERM_RESIDUALS_ZERO;
            else need_step = true;
        }
    } else {
        nfev += 1;
        if (!ok) { // residuals() == None at the trial point: the problem keeps the trial parameters
            term = VP_TERM_USER;
#pragma unroll
            for (int i = 0; i < Q; ++i) x[i] = xt[i];
            accept_c = true;
            status = VP_ST_NONFINITE;
        } else {
            // STRAIGHT-LINE from here (round 5): lane s runs its own problem, so every `if` of the bookkeeping is a divergent
            // branch -- an exec-mask save / restore and a scalar branch whose latency nothing hides (23 of them in this block,
            // 31 % of the scalar phase by the section clocks of tools/fit2_clocks.py).  The same expressions, evaluated
            // unconditionally and SELECTED: identical values wherever the branchy form computed them.
            const bool lo = ratio <= T(0.25);
            const bool hi = !lo && (par == T(0) || ratio >= T(0.75));
            T temp = !(actred < T(0)) ? T(0.5) : T(0.5) * dirder * frcp(dirder + T(0.5) * actred);
            temp = (fnorm1 * T(0.1) >= fnorm || temp < T(0.1)) ? T(0.1) : temp;
            const T d_lo = temp * tmin(delta, pnorm * T(10)), p_lo = par * frcp(temp);
            delta = lo ? d_lo : (hi ? pnorm * T(2) : delta);
            par = lo ? p_lo : (hi ? par * T(0.5) : par);
            trace_row(ratio);
            const bool g = good_v;
#pragma unroll
            for (int i = 0; i < Q; ++i) x[i] = g ? xt[i] : x[i];
            accept_c = g;
            T tmpv[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) tmpv[i] = scale_diag ? diag[i] * x[i] : x[i];
            const T xn_new = enorm_small<T, Q, false>(tmpv);
            xnorm = g ? xn_new : xnorm;
            fnorm = g ? fnorm1 : fnorm;
            objective = g ? T(0.5) * fnorm1 * fnorm1 : objective;
            const bool xbad = g && !is_finite(xnorm);
            const bool ftol_check = tabs(actred) <= ftol && prered <= ftol && ratio * T(0.5) <= T(1);
            const bool xtol_check = delta <= xtol * xnorm;
            const int conv = ftol_check ? (xtol_check ? VP_TERM_CONVERGED_BOTH : VP_TERM_CONVERGED_FTOL)
                                        : (xtol_check ? VP_TERM_CONVERGED_XTOL : 0);
            int tcode = (fnorm <= num<T>::tiny) ? VP_TERM_RESIDUALS_ZERO : conv;
            tcode = (tcode == 0 && nfev >= max_fev) ? VP_TERM_LOST_PATIENCE : tcode;
            const bool stuck = (tabs(actred) <= num<T>::eps && prered <= num<T>::eps && ratio * T(0.5) <= T(1)) ||
                               delta <= num<T>::eps * xnorm || gnorm <= num<T>::eps;
            tcode = (tcode == 0 && stuck) ? VP_TERM_NO_IMPROVEMENT : tcode;
            term = xbad ? VP_TERM_NUMERICAL : tcode;
            // (a rejected step keeps the old Jacobian factor and only re-solves the trust-region problem)
            need_step = (term == 0);
        }
    }
    VP_SCK(1);
    bool flagged = false;
    if (need_step && jac_done) {
        // the vector phase refreshed (Rj, qtf, acnorm, ipvt) at the accepted point
        T gmax = T(0);
        // (rare) a factor whose column norms are not finite after an evaluation that was ok: the fit ends here and -- Householder
        // kernels with a rescue list -- is re-fitted by vp_fit's second launch (vp_fit.hpp, jac_not_finite)
        bool degenerate = false;
        if constexpr (!GRAM) {
            bool jb = false;
#pragma unroll
            for (int i = 0; i < Q; ++i) jb = jb || !is_finite(acnorm[i]);
            degenerate = jb;
            flagged = jb;
        }
        const T ifn = frcp(fnorm);
#pragma unroll
        for (int j = 0; j < Q; ++j) { // (selected, not branched: see above)
            const T an = dyn_get_o<Q, (Q > 3)>(acnorm, ipvt[j]);
            const bool nz = an != T(0);
            T sum = T(0);
#pragma unroll
            for (int i = 0; i <= j; ++i) sum = tfma(Rj[i][j], qtf[i], sum);
            const T temp = tabs(sum * frcp(an) * ifn);
            degenerate = degenerate || (nz && temp != temp);
            gmax = nz ? tmax(gmax, temp) : gmax;
        }
        gnorm = gmax;
        if (degenerate) {
            term = VP_TERM_NUMERICAL;
        } else if (gnorm <= gtol) {
            term = VP_TERM_ORTHOGONAL;
        } else if (first_update) {
            T tmpv[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                if (scale_diag) diag[i] = (acnorm[i] == T(0)) ? T(1) : acnorm[i];
                tmpv[i] = scale_diag ? diag[i] * x[i] : x[i];
            }
            xnorm = enorm_small<T, Q, false>(tmpv);
            if (!is_finite(xnorm)) term = VP_TERM_NUMERICAL;
            delta = (xnorm == T(0)) ? stepbound : stepbound * xnorm;
            first_update = false;
        } else if (scale_diag) {
#pragma unroll
            for (int i = 0; i < Q; ++i) diag[i] = tmax(diag[i], acnorm[i]);
        }
        if (term) need_step = false;
    }

    VP_SCK(2);
    if (need_step) {
        T Rw[Q][Q]; // lmpar scribbles on the strict lower triangle
#pragma unroll
        for (int i = 0; i < Q; ++i)
#pragma unroll
            for (int j = 0; j < Q; ++j) Rw[i][j] = Rj[i][j];
        if constexpr (GRAM) par = lmpar_chol<T, Q, false, (Q > 3)>(Rj, ipvt, diag, qtf, delta, par, step, pnorm);
        else par = lmpar_any<T, Q, false, (Q > 3)>(Rw, ipvt, diag, qtf, delta, par, step, pnorm);
        VP_SCK(3);
        {
            T wa[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) wa[i] = T(0);
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const T pj = dyn_get_o<Q, (Q > 3)>(step, ipvt[j]);
#pragma unroll
                for (int i = 0; i <= j; ++i) wa[i] = tfma(Rj[i][j], pj, wa[i]);
            }
            const T ifn = frcp(fnorm);
            const T t1 = enorm_small<T, Q, false>(wa) * ifn;
            const T temp1 = t1 * t1;
            const T t2 = (usqrt(par) * pnorm) * ifn;
            const T temp2 = t2 * t2;
            // (a non-finite step length makes both terms non-finite: one test, selected results)
            const bool bad = !is_finite(pnorm) || !is_finite(temp1) || !is_finite(temp2);
            term = bad ? VP_TERM_NUMERICAL : term;
            prered = bad ? prered : temp1 + temp2 * T(2);
            dirder = bad ? dirder : -(temp1 + temp2);
            delta = (!bad && first_tr && pnorm < delta) ? pnorm : delta;
            first_tr = bad ? first_tr : false;
#pragma unroll
            for (int i = 0; i < Q; ++i) xt[i] = bad ? xt[i] : x[i] - step[i];
        }
    }

    VP_SCK(4);
    // write the record back
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        s->x[i] = x[i];
        s->xt[i] = xt[i];
        s->diag[i] = diag[i];
    }
    if (accept_c) {
#pragma unroll
        for (int i = 0; i < N; ++i) s->cbest[i] = s->cnew[i];
    }
    if constexpr (!GRAM && VP_FIT2_PRE != 0) {
        const T pd = k->pre_delta;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const T rti = frcp(xt[i]);
            s->pre[2 * i] = rti;
            s->pre[2 * i + 1] = texp(-div_refined(pd, xt[i], rti));
        }
        s->pre[2 * Q] = frcp(fnorm);
        s->pre[2 * Q + 1] = frcp(prered);
    }
    s->fnorm = fnorm;
    s->delta = delta;
    s->par = par;
    s->xnorm = xnorm;
    s->gnorm = gnorm;
    s->pnorm = pnorm;
    s->prered = prered;
    s->dirder = dirder;
    s->objective = objective;
    s->flags = (first ? 1 : 0) | (first_tr ? 2 : 0) | (first_update ? 4 : 0);
    s->nfev = nfev;
    s->term = term;
    s->status = status;
    s->trow = trow;

    if (term != 0) {
        // results of a finished problem (per-lane scattered stores; a few dozen bytes each)
        vp_report rep;
        rep.termination = term;
        rep.n_evals = nfev;
        rep.objective = (double)objective;
        k->report[prob] = rep;
        double *cost_out = k->cost_out;
        int32_t *status_out = k->status;
        TO *alpha_out = k->alpha, *C_out = k->C_out;
        if (cost_out) cost_out[prob] = (double)objective;
        if (status_out && !VP_FITG_TIMELINE) status_out[prob] = status;
        int32_t *rescue = k->rescue;
        const int pslot = (s->pend[0] < 0) ? 0 : ((s->pend[1] < 0) ? 1 : -1);
        if (flagged && k->self_rescue != 0 && pslot >= 0) { // alpha[prob] keeps the initial guess; this wave re-fits it before it ends
            s->pend[pslot] = (int)prob;
        } else if (flagged && k->self_rescue == 0 && rescue) { // ... or the re-fit launch does
            rescue_push(rescue, k->rescue_slot, prob);
        } else {
#pragma unroll
            for (int i = 0; i < Q; ++i) alpha_out[prob * Q + i] = (TO)x[i];
        }
        if (C_out) {
#pragma unroll
            for (int i = 0; i < N; ++i) C_out[prob * N + i] = (TO)s->cbest[i];
        }
    }
    VP_SCK(5);
}

template <typename T, int N, int Q, int GS, typename TO = T, bool GRAM = false>
__device__ __noinline__ void slot_scalar_phase(VP_LDS SlotRec<T, N, Q> *recs, VP_LDS const SlotConsts<T, TO> *k, const bool act = true) {
    slot_scalar_phase_inl<T, N, Q, GS, TO, GRAM>(recs, k, act);
}

#ifndef VP_LONE_TAIL_LINKAGE
#define VP_LONE_TAIL_LINKAGE __forceinline__
#endif
// LONE TAIL (W = 1).  Once the problem queue is dry a wave that holds ONE unfinished fit has nobody to share the scalar
// phase with: the lane-parallel bookkeeping (one dependent fp64 chain issued for a single lane, an LDS record round
// trip and a call per evaluation) is then pure latency -- 5.6 us per evaluation against the 3.9 us of fit_kernel, whose
// bookkeeping is inline and wave-uniform -- and the end of a launch IS a few hundred waves each
// finishing one long fit.  This function takes the slot's record and runs the REST of that fit the fit_kernel way:
// same evaluation (the slot's H_0 y column), same arithmetic, results bit-identical to either kernel.  Out of line: its
// register allocation must not touch the slot loop's.
template <typename T, class M, int R, int PADM>
__device__ VP_LONE_TAIL_LINKAGE void fit2_lone_tail(VP_LDS SlotRec<T, M::N, M::Q> *rec, VP_LDS const T *s_col, VP_LDS const T *s_t,
                                            VP_LDS const SlotConsts<T> *kc, const M mdl, const T eps, const int uniform,
                                            const T h0_beta, const T h0_u, const T h0_g) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    constexpr int NC = N + P, YC = N - 1, DC = YC + 1;
    using G = Grp<1>;
    G grp = G::make(nullptr);
    const int lane = grp.gl;
    using Src = RowSource<T, R, true, 0, 1, 1, true, PADM>;
    Src src;
    src.t = (const T *)s_t;
    src.w = nullptr;
    src.m = kc->m;
    src.lane = lane;
    src.vec = true;
    src.set_uniform(uniform != 0);
    ConstReflector<T> h0;
    h0.beta = h0_beta;
    h0.u = h0_u;
    h0.g = h0_g;
    h0.live = true;
    const T ftol = kc->ftol, xtol = kc->xtol, gtol = kc->gtol, stepbound = kc->stepbound;
    const int scale_diag = uni(kc->scale_diag), max_fev = uni(kc->max_fev), mres = uni(kc->m);
    const int64_t prob = uni(rec->prob);
    const T qty0 = rec->qty0;
    T x[Q], xt[Q], diag[Q], qtf[Q], step[Q], acnorm[Q], cbest[N];
    T Rj[Q][Q];
    int ipvt[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        xt[k] = rec->xt[k];
        step[k] = T(0);
    }
    T fnorm, delta, par, xnorm, gnorm, pnorm, prered, dirder, objective;
    bool first, first_tr, first_update;
    int nfev, term = VP_TERM_NOT_RUN, st_best = uni(rec->status), trow = uni(rec->trow);
    bool flagged = false;
    auto trace_row = [&](const T(&xx)[Q], T fn, T ratio) {
        double *trace = kc->trace;
        const int trace_rows = kc->trace_rows;
        if (trace && trow < trace_rows && lane == 0) {
            double *tr = trace + ((size_t)prob * trace_rows + trow) * (Q + 4);
#pragma unroll
            for (int k = 0; k < Q; ++k) tr[k] = (double)xx[k];
            tr[Q] = (double)fn;
            tr[Q + 1] = (double)ratio;
            tr[Q + 2] = (double)delta;
            tr[Q + 3] = (double)par;
        }
        ++trow;
    };
    auto park = [&]() __attribute__((always_inline)) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                rec->x[k] = x[k];
                rec->diag[k] = diag[k];
                rec->qtf[k] = qtf[k];
                rec->acnorm[k] = acnorm[k];
                rec->ipvt[k] = ipvt[k];
#pragma unroll
                for (int j = 0; j < Q; ++j) rec->Rj[k][j] = Rj[k][j];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) rec->cbest[k] = cbest[k];
            rec->fnorm = fnorm;
            rec->delta = delta;
            rec->par = par;
            rec->xnorm = xnorm;
            rec->gnorm = gnorm;
            rec->pnorm = pnorm;
            rec->prered = prered;
            rec->dirder = dirder;
            rec->objective = objective;
            rec->flags = (first ? 1 : 0) | (first_tr ? 2 : 0) | (first_update ? 4 : 0);
            rec->nfev = nfev;
        }
    };
    auto unpark = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            x[k] = rec->x[k];
            diag[k] = rec->diag[k];
            qtf[k] = rec->qtf[k];
            acnorm[k] = rec->acnorm[k];
            ipvt[k] = uni(rec->ipvt[k]);
#pragma unroll
            for (int j = 0; j < Q; ++j) Rj[k][j] = rec->Rj[k][j];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) cbest[k] = rec->cbest[k];
        fnorm = rec->fnorm;
        delta = rec->delta;
        par = rec->par;
        xnorm = rec->xnorm;
        gnorm = rec->gnorm;
        pnorm = rec->pnorm;
        prered = rec->prered;
        dirder = rec->dirder;
        objective = rec->objective;
        {
            const int fl = uni(rec->flags);
            first = (fl & 1) != 0;
            first_tr = (fl & 2) != 0;
            first_update = (fl & 4) != 0;
            nfev = uni(rec->nfev);
        }
    };
    for (;;) {
        // the LM state stays parked in the slot's record during the sweep (the register-pressure peak); only the trial
        // parameters are live across it
        T C[NC][R];
        EvalUniform<T, N> u;
        load_rows_lds<T, R, 1>((const T *)s_col, lane, C[YC]);
        evaluate_core_const_first<T, M, R, NC, Src, G, true>(mdl, xt, src, eps, grp, h0, C, u, nullptr, qty0);
        asm volatile("" ::: "memory");
        unpark();
        const T fnorm1 = usqrt(u.fn2);
        bool need_jac = false;
        if (first) {
            first = false;
            nfev = 1;
            st_best = u.ok ? VP_ST_OK : VP_ST_NONFINITE;
            if (!u.ok) {
                term = VP_TERM_USER;
                break;
            }
            fnorm = fnorm1;
            objective = T(0.5) * fnorm * fnorm;
            trace_row(xt, fnorm1, T(0) / T(0));
#pragma unroll
            for (int k = 0; k < N; ++k) cbest[k] = u.c[k];
            if (Q > mres) {
                term = VP_TERM_WRONG_DIMENSIONS;
                break;
            }
            if (uni(!is_finite(fnorm))) {
                term = VP_TERM_NUMERICAL;
                break;
            }
            if (uni(fnorm <= num<T>::tiny)) {
                term = VP_TERM_RESIDUALS_ZERO;
                break;
            }
            need_jac = true;
        } else {
            nfev += 1;
            if (!u.ok) {
                term = VP_TERM_USER;
#pragma unroll
                for (int k = 0; k < Q; ++k) x[k] = xt[k];
#pragma unroll
                for (int k = 0; k < N; ++k) cbest[k] = u.c[k];
                st_best = VP_ST_NONFINITE;
                break;
            }
            const T q1 = fnorm1 * frcp(fnorm);
            const T actred = (fnorm1 * T(0.1) < fnorm) ? T(1) - q1 * q1 : T(-1);
            const T ratio = (prered == T(0)) ? T(0) : actred * frcp(prered);
            if (ratio <= T(0.25)) {
                T temp = !(actred < T(0)) ? T(0.5) : T(0.5) * dirder * frcp(dirder + T(0.5) * actred);
                if (fnorm1 * T(0.1) >= fnorm || temp < T(0.1)) temp = T(0.1);
                delta = temp * tmin(delta, pnorm * T(10));
                par = par * frcp(temp);
            } else if (par == T(0) || ratio >= T(0.75)) {
                delta = pnorm * T(2);
                par = par * T(0.5);
            }
            const bool good = uni(ratio >= T(1.0e-4));
            trace_row(xt, fnorm1, ratio);
            if (good) {
#pragma unroll
                for (int k = 0; k < Q; ++k) x[k] = xt[k];
#pragma unroll
                for (int k = 0; k < N; ++k) cbest[k] = u.c[k];
                T tmpv[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) tmpv[k] = scale_diag ? diag[k] * x[k] : x[k];
                xnorm = enorm_small<T, Q>(tmpv);
                fnorm = fnorm1;
                objective = T(0.5) * fnorm1 * fnorm1;
                if (uni(!is_finite(xnorm))) {
                    term = VP_TERM_NUMERICAL;
                    break;
                }
            }
            int tcode = 0;
            if (fnorm <= num<T>::tiny) tcode = VP_TERM_RESIDUALS_ZERO;
            if (!tcode) {
                const bool ftol_check = tabs(actred) <= ftol && prered <= ftol && ratio * T(0.5) <= T(1);
                const bool xtol_check = delta <= xtol * xnorm;
                if (ftol_check || xtol_check)
                    tcode = (ftol_check && xtol_check) ? VP_TERM_CONVERGED_BOTH
                                                       : (ftol_check ? VP_TERM_CONVERGED_FTOL : VP_TERM_CONVERGED_XTOL);
            }
            if (!tcode && nfev >= max_fev) tcode = VP_TERM_LOST_PATIENCE;
            if (!tcode && tabs(actred) <= num<T>::eps && prered <= num<T>::eps && ratio * T(0.5) <= T(1))
                tcode = VP_TERM_NO_IMPROVEMENT;
            if (!tcode && delta <= num<T>::eps * xnorm) tcode = VP_TERM_NO_IMPROVEMENT;
            if (!tcode && gnorm <= num<T>::eps) tcode = VP_TERM_NO_IMPROVEMENT;
            tcode = uni(tcode);
            if (tcode) {
                term = tcode;
                break;
            }
            need_jac = good;
        }
        if (need_jac) {
            residual_qcoords<T, R, N>(C[YC], u.e, grp);
            T zs[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) zs[k] = -u.c[k];
            jac_qrfac_scaled<T, R, Q, N>(reinterpret_cast<T(&)[Q][R]>(C[DC]), C[YC], zs, Rj, acnorm, ipvt, qtf, grp);
            if (uni(jac_not_finite<T, Q>(acnorm))) { // (rare) flag and re-fit: vp_fit.hpp, jac_not_finite
                term = VP_TERM_NUMERICAL;
                flagged = true;
                break;
            }
            T gmax = T(0);
            bool degenerate = false;
            const T ifn = frcp(fnorm);
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const T an = dyn_get<Q>(acnorm, ipvt[j]);
                if (an != T(0)) {
                    T sum = T(0);
#pragma unroll
                    for (int i = 0; i <= j; ++i) sum = tfma(Rj[i][j], qtf[i], sum);
                    const T temp = tabs(sum * frcp(an) * ifn);
                    if (temp != temp) degenerate = true;
                    gmax = tmax(gmax, temp);
                }
            }
            gnorm = gmax;
            if (uni(degenerate)) {
                term = VP_TERM_NUMERICAL;
                break;
            }
            if (uni(gnorm <= gtol)) {
                term = VP_TERM_ORTHOGONAL;
                break;
            }
            if (first_update) {
                T tmpv[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    if (scale_diag) diag[k] = (acnorm[k] == T(0)) ? T(1) : acnorm[k];
                    tmpv[k] = scale_diag ? diag[k] * x[k] : x[k];
                }
                xnorm = enorm_small<T, Q>(tmpv);
                if (uni(!is_finite(xnorm))) {
                    term = VP_TERM_NUMERICAL;
                    break;
                }
                delta = (xnorm == T(0)) ? stepbound : stepbound * xnorm;
                first_update = false;
            } else if (scale_diag) {
#pragma unroll
                for (int k = 0; k < Q; ++k) diag[k] = tmax(diag[k], acnorm[k]);
            }
        }
        par = lmpar_any<T, Q>(Rj, ipvt, diag, qtf, delta, par, step, pnorm);
        if (uni(!is_finite(pnorm))) {
            term = VP_TERM_NUMERICAL;
            break;
        }
        {
            T wa[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) wa[i] = T(0);
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const T pj = dyn_get<Q>(step, ipvt[j]);
#pragma unroll
                for (int i = 0; i <= j; ++i) wa[i] = tfma(Rj[i][j], pj, wa[i]);
            }
            const T ifn = frcp(fnorm);
            const T t1 = enorm_small<T, Q>(wa) * ifn;
            const T temp1 = t1 * t1;
            const T t2 = (usqrt(par) * pnorm) * ifn;
            const T temp2 = t2 * t2;
            if (uni(!is_finite(temp1) || !is_finite(temp2))) {
                term = VP_TERM_NUMERICAL;
                break;
            }
            prered = temp1 + temp2 * T(2);
            dirder = -(temp1 + temp2);
        }
        if (first_tr && pnorm < delta) delta = pnorm;
        first_tr = false;
#pragma unroll
        for (int k = 0; k < Q; ++k) xt[k] = x[k] - step[k];
        // park for the next sweep
        park();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // results of the finished problem
    if (lane == 0) {
        vp_report rep;
        rep.termination = term;
        rep.n_evals = nfev;
        rep.objective = (double)objective;
        kc->report[prob] = rep;
        double *cost_out = kc->cost_out;
        int32_t *status_out = kc->status;
        T *alpha_out = kc->alpha, *C_out = kc->C_out;
        if (cost_out) cost_out[prob] = (double)objective;
        if (status_out) status_out[prob] = st_best;
        const int pslot = (rec->pend[0] < 0) ? 0 : ((rec->pend[1] < 0) ? 1 : -1);
        if (flagged && kc->self_rescue != 0 && pslot >= 0) {
            rec->pend[pslot] = (int)prob;
        } else if (flagged && kc->self_rescue == 0 && kc->rescue) {
            rescue_push(kc->rescue, kc->rescue_slot, prob);
        } else {
#pragma unroll
            for (int k = 0; k < Q; ++k) alpha_out[prob * Q + k] = x[k];
        }
        if (C_out) {
#pragma unroll
            for (int k = 0; k < N; ++k) C_out[prob * N + k] = cbest[k];
        }
        rec->term = term;
    }
}

#ifndef VP_SELF_RESCUE
#define VP_SELF_RESCUE 1 // (A/B switch: 0 = flagged problems go to the handle's list and the re-fit launches behind the fit)
#endif
template <typename T, class M, int W> inline constexpr bool fit2_self_rescue_v = VP_SELF_RESCUE && W == 1 && fit_rescue_v<T, M>;

// the scaled re-fit of one flagged problem by the wave that flagged it (fit2_kernel's exit path).  OUT OF LINE: inlined, its
// 160-register column array and spill slots became part of the slot kernel's own frame and the hot loop ran 2 % slower
// (113 instead of 42 spilled VGPRs in the kernel's metadata, 2.10 instead of 2.065 ms per step with two batches in flight)
template <typename T, class M, int R>
__device__ __noinline__ void fit2_refit_flagged(const FitArgs<T, M> *a, const int64_t b, T *s_t, T *s_y, LmState<T, M::N, M::Q> *st) {
    (void)fit_problem<T, M, R, 1, false, 0, true, false>(*a, b, s_t, s_y, nullptr, nullptr, st, false);
}

// A workgroup = NG groups of W waves (W = 1: NG = 4 independent waves; W > 1: ONE group, whose reductions use
// workgroup barriers).  The groups share one LDS copy of the grid; each owns GS slots.  With W > 1 the scalar phase
// runs on wave 0 of the group only -- the one-problem-per-group kernel has all W waves repeat the bookkeeping.
template <typename T, class M, int R, int W, int PADM, int GS, int NG, int WPS>
__global__ void __launch_bounds__(64 * W * NG, (WPS * W * NG) / 4 > 0 ? (WPS * W * NG) / 4 : 1) fit2_kernel(const Fit2Args<T, M> args) {
    static_assert(M::kConstLast && M::kDiagonalPairs, "slot kernel: multi-exponential + offset models");
    static_assert(W == 1 || NG == 1, "multi-wave groups synchronise with workgroup barriers: one group per workgroup");
    constexpr int N = M::N, P = M::P, Q = M::Q;
    constexpr int NC = N + P;  // register columns: the constant column is never materialised
    constexpr int YC = N - 1;  // data column
    constexpr int DC = YC + 1; // first derivative column
    constexpr int MP = 64 * R * W;
    using Rec = SlotRec<T, N, Q>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s_t = reinterpret_cast<T *>(smem_raw);
    const int gi = (W == 1) ? (int)(threadIdx.x >> 6) : 0; // group within the workgroup
    T *s_y = s_t + MP + (size_t)gi * GS * MP;
    Rec *recs_all = reinterpret_cast<Rec *>(s_t + MP + (size_t)NG * GS * MP);
    Rec *recs = recs_all + (size_t)gi * GS;
    SlotConsts<T> *kc = reinterpret_cast<SlotConsts<T> *>(recs_all + (size_t)NG * GS);
    int *s_pop = reinterpret_cast<int *>(kc + 1);                                   // [GS] queue pops (W > 1)
    unsigned char *s_xch = reinterpret_cast<unsigned char *>(s_pop + ((GS + 3) / 4) * 4); // group exchange area (W > 1)
    using G = Grp<W>;
    G grp = G::make(s_xch);
    const int lane = grp.gl; // group lane: row ownership
    const int gw = (int)blockIdx.x * NG + gi; // persistent group index
    auto group_sync = [&]() __attribute__((always_inline)) {
        if constexpr (W > 1) {
            __syncthreads();
        } else {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    T eps_;
    int m_, uniform_;
    int64_t B_;
    {
        const FitArgs<T, M> &a = args.f;
        m_ = a.m;
        eps_ = a.eps;
        uniform_ = a.grid_uniform;
        B_ = a.B;
        // ---- the shared grid (every group writes the same values: no ownership split needed) + the constants ----
        T tmp[R];
        load_rows<T, R, W>(a.t, m_, lane, vec_aligned<T>(a.t, m_), tmp);
        store_rows<T, R, W>(s_t, MP, lane, true, tmp);
        if (threadIdx.x == 0) {
            kc->ftol = a.ftol;
            kc->xtol = a.xtol;
            kc->gtol = a.gtol;
            kc->stepbound = a.stepbound;
            kc->alpha = a.alpha;
            kc->rescue = a.rescue;
            kc->rescue_slot = a.rescue_slot;
            kc->self_rescue = (fit2_self_rescue_v<T, M, W> && a.rescue != nullptr) ? 1 : 0; // (a.rescue == null: re-fits are switched off)
            kc->C_out = a.C_out;
            kc->cost_out = a.cost_out;
            kc->status = a.status;
            kc->report = a.report;
            kc->trace = a.trace;
            kc->yw = a.yw;
            kc->queue = args.queue;
            kc->B = a.B;
            kc->trace_rows = a.trace_rows;
            kc->scale_diag = a.scale_diag;
            kc->max_fev = a.patience * (Q + 1);
            kc->m = a.m;
#ifdef VP_FIT2_CLOCKS
            for (int i = 0; i < 6; ++i) kc->sck[i] = 0;
#endif
        }
    }
    __syncthreads();
    const int m = m_;
    using Src = RowSource<T, R, true, 0, 1, W, true, PADM>;
    Src src;
    src.t = s_t;
    src.w = nullptr;
    src.m = m;
    src.lane = lane;
    src.vec = true;
    src.set_uniform(uniform_ != 0);
    const ConstReflector<T> h0 = make_const_reflector<T, R, Src, G>(src, grp);
    const M mdl = args.f.mdl;
    if (threadIdx.x == 0) kc->pre_delta = src.uniform ? src.delta : T(0);
    __syncthreads();

    auto fill = [&](int s, int prob) __attribute__((always_inline)) {
        slot_fill<T, N, Q, R, W, PADM>((VP_LDS Rec *)(recs + s), (VP_LDS T *)(s_y + (size_t)s * MP), (VP_LDS const T *)s_t,
                                       (VP_LDS const SlotConsts<T> *)kc, (VP_LDS unsigned char *)s_xch, prob, h0.beta, h0.u,
                                       h0.g);
        if constexpr (W > 1) grp.phase = 0; // slot_fill leaves the exchange buffers quiescent (barrier at its end)
    };

    if (lane == 0) {
#pragma nounroll
        for (int s = 0; s < GS; ++s) recs[s].pend[0] = recs[s].pend[1] = -1;
    }
    // ---- initial, static assignment: group gw takes problems gw*GS .. gw*GS+GS-1 ----
    int nactive = 0;
    bool queue_dry = false;
#pragma nounroll
    for (int s = 0; s < GS; ++s) {
        const int64_t prob = (int64_t)gw * GS + s;
        const bool have = prob < B_ && h0.live;
        fill(s, have ? (int)prob : -1);
        nactive += have ? 1 : 0;
    }
    group_sync();

#ifdef VP_FIT2_CLOCKS
    long long ck[4] = {0, 0, 0, 0};
    long long c0 = __builtin_amdgcn_s_memtime();
#define VP_CK2(i)                                                                                                      \
    do {                                                                                                               \
        const long long c1 = __builtin_amdgcn_s_memtime();                                                             \
        ck[i] += c1 - c0;                                                                                              \
        c0 = c1;                                                                                                       \
    } while (0)
#else
#define VP_CK2(i)
#endif
    while (nactive > 0) {
        // =============================== VECTOR phase ===============================
#pragma nounroll
        for (int s = 0; s < GS; ++s) {
            Rec *rec = recs + s;
            if (uni(rec->prob) < 0) continue;
            T alpha[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) alpha[k] = rec->xt[k];
            const int fl_in = uni(rec->flags);
            const bool first = (fl_in & 1) != 0;
            const T qty0 = rec->qty0;
            const T fnorm = rec->fnorm, prered = rec->prered;
            constexpr bool kPre = VP_FIT2_PRE != 0 && M::kStatic && sizeof(T) == 8;

            T C[NC][R];
            EvalUniform<T, N> u;
            load_rows_lds<T, R, W>(s_y + (size_t)s * MP, W == 1 ? lane_fresh() : lane, C[YC]);
            evaluate_core_const_first<T, M, R, NC, Src, G, true, false, false, kPre>(mdl, alpha, src, eps_, grp, h0, C, u, nullptr, qty0, nullptr,
                                                                                      rec->pre); // (read from LDS where they are used)

            const T fnorm1 = usqrt(u.fn2);
            T actred = T(0), ratio = T(0);
            bool good = false;
            if (!first) {
                const T q1 = fnorm1 * (kPre ? rec->pre[2 * Q] : frcp(fnorm));
                actred = (fnorm1 * T(0.1) < fnorm) ? T(1) - q1 * q1 : T(-1);
                ratio = (prered == T(0)) ? T(0) : actred * (kPre ? rec->pre[2 * Q + 1] : frcp(prered));
                good = uni(ratio >= T(1.0e-4));
            }
            const bool need_jac = u.ok && (first || good);
            T Rj[Q][Q], acnorm[Q], qtf[Q];
            int ipvt[Q];
            if (need_jac) {
                // z_k = -c_k Q^T D_k: factor the unscaled columns in place, the coefficients enter as column scales
                residual_qcoords<T, R, N>(C[YC], u.e, grp);
                T zs[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) zs[k] = -u.c[k];
                jac_qrfac_scaled<T, R, Q, N>(reinterpret_cast<T(&)[Q][R]>(C[DC]), C[YC], zs, Rj, acnorm, ipvt, qtf, grp);
            }
            if (lane == 0) { // group lane 0
                rec->fnorm1 = fnorm1;
                rec->actred = actred;
                rec->ratio = ratio;
#pragma unroll
                for (int k = 0; k < N; ++k) rec->cnew[k] = u.c[k];
                int fl = fl_in & 7;
                if (u.ok) fl |= 8;
                if (need_jac) {
                    fl |= 16;
#pragma unroll
                    for (int k = 0; k < Q; ++k) {
                        rec->acnorm[k] = acnorm[k];
                        rec->qtf[k] = qtf[k];
                        rec->ipvt[k] = ipvt[k];
#pragma unroll
                        for (int j = 0; j < Q; ++j) rec->Rj[k][j] = Rj[k][j];
                    }
                }
                if (good) fl |= 32;
                rec->flags = fl;
            }
        }
        group_sync();
        VP_CK2(0);

        // =============================== SCALAR phase: lane s of wave 0 <-> slot s ===============================
        if (grp.wave == 0) slot_scalar_phase<T, N, Q, GS>((VP_LDS Rec *)recs, (VP_LDS const SlotConsts<T> *)kc);
        group_sync();
        VP_CK2(1);

        // =============================== REFILL finished slots from the queue ===============================
        if constexpr (W > 1) {
            // one pop per finished slot by group lane 0, published through LDS (all waves must agree on the index)
            if (lane == 0) {
                for (int s = 0; s < GS; ++s)
                    s_pop[s] = (recs[s].prob >= 0 && recs[s].term != 0)
                                   ? __hip_atomic_fetch_add(kc->queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : -1;
            }
            __syncthreads();
        }
#pragma nounroll
        for (int s = 0; s < GS; ++s) {
            if (uni(recs[s].prob) < 0 || uni(recs[s].term) == 0) continue;
            int next = 0;
            if constexpr (W > 1) {
                next = uni(s_pop[s]);
            } else {
                if (lane == 0) next = __hip_atomic_fetch_add(kc->queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                next = uni(next);
            }
            const bool have = (int64_t)next < B_;
            fill(s, have ? next : -1);
            if (!have) {
                nactive -= 1;
                queue_dry = true; // learnt for free: the pop this wave had to make anyway came back empty
            }
        }
        group_sync();
        VP_CK2(2);
#ifndef VP_NO_LONE_TAIL
        if constexpr (W == 1) {
            if (queue_dry && nactive == 1) break; // ONE fit left and nothing to refill from: finish it below
        }
#endif
    }
#ifndef VP_NO_LONE_TAIL
    if constexpr (W == 1) {
        // the wave's last fit runs the fit_kernel way (fit2_lone_tail) -- as the kernel's exit path: nothing of the slot
        // loop is live across the call
        if (nactive == 1) {
#pragma nounroll
            for (int s = 0; s < GS; ++s) {
                if (uni(recs[s].prob) < 0) continue;
                fit2_lone_tail<T, M, R, PADM>((VP_LDS Rec *)(recs + s), (VP_LDS const T *)(s_y + (size_t)s * MP),
                                              (VP_LDS const T *)s_t, (VP_LDS const SlotConsts<T> *)kc, mdl, eps_, uniform_,
                                              h0.beta, h0.u, h0.g);
            }
        }
    }
#endif
    // SELF-RESCUE (round 6): the problems this wave flagged -- a Jacobian factor that is not finite after an evaluation that
    // was ok, vp_fit.hpp jac_not_finite -- are re-fitted HERE, from their initial guesses, with scaled derivative columns
    // (fit_problem<..., RESCUE>): no second launch behind the fit (two mostly empty launches cost the pipelined headline 2 %:
    // tools/refit_cost_probe.py), and as the kernel's exit path nothing of the slot loop is live across it.  The slot's data
    // column and records are free by now: they hold the re-fit's data copy and parked LM state.
    if constexpr (fit2_self_rescue_v<T, M, W>) {
        group_sync();
        int pend[2 * GS];
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            pend[2 * s] = uni(recs[s].pend[0]);
            pend[2 * s + 1] = uni(recs[s].pend[1]);
        }
        group_sync();
        static_assert(sizeof(LmState<T, N, Q>) <= sizeof(Rec) * GS, "the parked LM state of the re-fit lives in the group's records");
#pragma nounroll
        for (int i = 0; i < 2 * GS; ++i) {
            const int pb_ = dyn_get<2 * GS>(pend, i);
            if (pb_ < 0) continue;
            fit2_refit_flagged<T, M, R>(&args.f, (int64_t)pb_, s_t, s_y, reinterpret_cast<LmState<T, N, Q> *>(recs));
        }
    }
#ifdef VP_FIT2_CLOCKS
    if (args.f.trace && blockIdx.x == 0 && threadIdx.x == 0) {
        double *tr = args.f.trace + (size_t)(args.f.trace_rows - 1) * (Q + 4);
        for (int i = 0; i < 3; ++i) tr[i] = (double)ck[i];
        // (row trace_rows - 2: the scalar phase's sections -- record load | update + termination tests | gradient test, diag |
        // lmpar | predicted reduction, trial point | write-back, results)
        double *tr2 = args.f.trace + (size_t)(args.f.trace_rows - 2) * (Q + 4);
        for (int i = 0; i < 6 && i < Q + 4; ++i) tr2[i] = (double)kc->sck[i];
    }
#endif
}

template <typename T, class M, int R, int W = 1> int launch_fit2(const LaunchParams &p) {
    // the slot kernel covers: unit weights, one shared grid, model ending in a constant column
    if constexpr (!(M::kConstLast && M::kDiagonalPairs)) {
        return launch_fit<T, M, R, W>(p);
    } else {
        constexpr int NG = (W == 1) ? 4 : 1;                 // groups per workgroup
        constexpr int WPS = waves_for<T, R, M::N + M::P>();  // resident waves per SIMD
        constexpr int BPC = (4 * WPS) / (W * NG) > 0 ? (4 * WPS) / (W * NG) : 1; // workgroups per CU
        constexpr int GS = fit2_slots<T, R, W, NG, BPC, (int)sizeof(SlotRec<T, M::N, M::Q>)>();
        // fit_group: 0 = automatic, 1 = one problem per wave(-group) (fit_kernel), 2 = slots regardless of B.
        // Automatic, W = 1: a launch costs (work / throughput) + the latency of its slowest fit (~0.5 ms at m = 1024:
        // >100 LM evaluations of one problem, nothing to overlap them with).  The slot kernel has the higher
        // throughput (fewer instructions per evaluation) but a slot shares its wave with G-1 others, i.e. the slowest
        // fit advances more slowly while its partners are busy: measured cross-over at ~16x the device's resident
        // waves (32768 problems on MI355X, tools/slot_probe.py).  W > 1: the slot kernel runs the LM bookkeeping on
        // one wave instead of all W and is never slower.
        const int64_t cap_groups = (int64_t)p.num_cus * BPC * NG;
        const bool small = (W == 1) ? (p.B <= 16 * cap_groups) : false;
        if (p.w || p.t_stride != 0 || !p.queue || p.fit_group == 1 || (p.fit_group != 2 && small))
            return launch_fit<T, M, R, W>(p);
        Fit2Args<T, M> args;
        FitArgs<T, M> &a = args.f;
        if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
        a.t = (const T *)p.t;
        a.w = nullptr;
        a.yw = (const T *)p.yw;
        a.alpha = (T *)p.alpha_out;
        a.C_out = (T *)p.C_out;
        a.cost_out = p.cost_out;
        a.status = p.status;
        a.report = p.report;
        a.m = p.m;
        a.B = p.B;
        a.t_stride = 0;
        a.w_stride = 0;
        a.eps = (T)p.eps;
        a.ftol = (T)p.opts->ftol;
        a.xtol = (T)p.opts->xtol;
        a.gtol = (T)p.opts->gtol;
        a.stepbound = (T)p.opts->stepbound;
        a.patience = p.opts->patience;
        a.scale_diag = p.opts->scale_diag;
        a.trace = p.trace;
        a.trace_rows = p.trace_rows;
        a.grid_uniform = p.grid_uniform;
        a.rescue = p.rescue;
        a.rescue_slot = p.rescue_slot;
        if (a.B <= 0) return VP_ERR_OK;
        // (W == 1 slot kernels of fit_rescue_v models re-fit what they flag themselves: no list, no second launch)
        if (p.rescue_used && !fit2_self_rescue_v<T, M, W>) *p.rescue_used = 1;
        // persistent grid: every resident workgroup slot of the device, or fewer when the batch is smaller
        const int64_t cap_blocks = (int64_t)p.num_cus * BPC;
        const int64_t need_blocks = (a.B + (int64_t)GS * NG - 1) / ((int64_t)GS * NG);
        const int64_t blocks = need_blocks < cap_blocks ? need_blocks : cap_blocks;
        args.queue = p.queue;
        args.waves_total = (int)(blocks * NG * W);
        if (hipMemsetD32Async((hipDeviceptr_t)p.queue, (int)(blocks * NG * GS), 1, p.stream) != hipSuccess) return VP_ERR_HIP;
        const size_t lds = (size_t)64 * R * W * sizeof(T) * (1 + (size_t)NG * GS) + (size_t)NG * GS * sizeof(SlotRec<T, M::N, M::Q>) +
                           sizeof(SlotConsts<T>) + (size_t)((GS + 3) / 4) * 4 * sizeof(int) + (size_t)group_xch_bytes<W>() + 16;
        if (lds > (size_t)(160 * 1024) / BPC) return launch_fit<T, M, R, W>(p); // (never with the slot count above; a guard)
#define VP_F2(PADM_)                                                                                                   \
    hipLaunchKernelGGL((fit2_kernel<T, M, R, W, PADM_, GS, NG, WPS>), dim3((unsigned)blocks), dim3(64 * W * NG), lds,   \
                       p.stream, args)
        if (p.m == 64 * R * W) VP_F2(1);
        else if (R > 2 && p.m > 64 * (R - 2) * W) VP_F2(2);
        else VP_F2(0);
#undef VP_F2
        return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
    }
}

} // namespace vp
