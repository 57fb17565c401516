// vp_fit_mp.hpp -- device-resident Levenberg-Marquardt, multi-problem-per-wave formulation.
//
// gfx950 has no scalar fp64 unit, so the wave-uniform LM bookkeeping of ONE problem (trust-region
// update, termination tests, lmpar/qrsolv) costs a full 64-lane vector instruction per scalar
// operation -- as much as the vector work of the evaluation itself (vp_fit.hpp measures ~40 % of the
// time there).  This kernel gives every wavefront G problems and alternates two phases:
//
//   VECTOR phase   for each still-active problem p of the wave: all 64 lanes cooperate on ONE fused
//                  Householder sweep (evaluate_core) and, if the step is accepted, on the pivoted QR of
//                  its Jacobian (jac_qrfac) -- exactly the code of the one-problem kernel; the
//                  wave-uniform results are written by lane 0 into the problem's LDS state record.
//   SCALAR phase   lane p runs the LM bookkeeping of problem p on its own record: 64 problems'
//                  scalar work in ONE pass of vector instructions (divergence only in lmpar's
//                  iteration count).
//
// The scalar work per evaluation drops by ~G; the vector work is unchanged.  y_w of a problem is
// re-read from HBM/L2 per evaluation (8 KiB, coalesced 16 B/lane, issued before the exp work that
// hides its latency); the shared grid t (and weights) stay in LDS.
// Semantics are identical to fit_kernel: == LevMarSolver::fit (src/solvers/levmar/mod.rs:238-254).
#pragma once
#include "vp_fit.hpp"

namespace vp {

// Per-problem LM state record in LDS.  Lane p touches record p in the scalar phase, so the record
// stride (in 8-byte words) is odd to spread the lanes over the LDS banks.
template <typename T, int N, int Q> struct alignas(8) MpState {
    T xt[Q], x[Q], diag[Q], qtf[Q], acnorm[Q], cbest[N], cnew[N];
    T Rj[Q][Q];
    T fnorm, delta, par, xnorm, gnorm, pnorm, prered, dirder, objective;
    T fnorm1, actred, ratio; // outputs of the latest evaluation
    int ipvt[Q];
    int flags;  // bit0 first, bit1 first_tr, bit2 first_update, bit3 eval ok, bit4 jacobian refreshed, bit5 good
    int nfev;
    int term;
    int status;
};

template <typename T, int N, int Q> constexpr int mp_state_words() {
    int w = (int)((sizeof(MpState<T, N, Q>) + 7) / 8);
    return (w % 2 == 0) ? w + 1 : w;
}

template <typename T, class M> struct FitMpArgs {
    FitArgs<T, M> f;
    int G; // problems per wave
};

template <typename T, class M, int R>
__global__ void __launch_bounds__(64, (waves_for<T, R, M::N + 1 + M::P>())) fit_mp_kernel(const FitMpArgs<T, M> args) {
    constexpr int N = M::N, P = M::P, Q = M::Q, NC = N + 1 + P;
    constexpr int MP = 64 * R;
    constexpr int SW = mp_state_words<T, N, Q>();
    using State = MpState<T, N, Q>;
    const FitArgs<T, M> &a = args.f;
    const int G = args.G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s_t = reinterpret_cast<T *>(smem_raw);
    T *s_w = a.w ? s_t + MP : nullptr;
    double *s_state = reinterpret_cast<double *>(s_t + MP + (a.w ? MP : 0));
    auto rec = [&](int p) -> State * { return reinterpret_cast<State *>(s_state + (size_t)p * SW); };

    const int lane = lane_id();
    Grp<1> grp = Grp<1>::make(nullptr);
    const int64_t p0 = (int64_t)blockIdx.x * G;
    if (p0 >= a.B) return;
    const int count = (int)((a.B - p0) < G ? (a.B - p0) : G);
    const int m = a.m;
    const bool shared_t = (a.t_stride == 0), shared_w = (a.w_stride == 0);

    // stage the shared grid / weights in LDS (per-problem grids are read from global instead)
    {
        T tmp[R];
        if (shared_t) {
            load_rows<T, R>(a.t, m, lane, vec_aligned<T>(a.t, m), tmp);
            store_rows<T, R>(s_t, MP, lane, true, tmp);
        }
        if (a.w && shared_w) {
            load_rows<T, R>(a.w, m, lane, vec_aligned<T>(a.w, m), tmp);
            store_rows<T, R>(s_w, MP, lane, true, tmp);
        }
    }
    // initial state records
    if (lane < count) {
        State *s = rec(lane);
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const T v = a.alpha[(p0 + lane) * Q + k];
            s->xt[k] = v;
            s->x[k] = v;
            s->diag[k] = T(1);
            s->qtf[k] = T(0);
            s->acnorm[k] = T(0);
            s->ipvt[k] = k;
#pragma unroll
            for (int j = 0; j < Q; ++j) s->Rj[k][j] = T(0);
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            s->cbest[k] = T(0);
            s->cnew[k] = T(0);
        }
        s->fnorm = s->delta = s->par = s->xnorm = s->gnorm = s->pnorm = s->prered = s->dirder = T(0);
        s->objective = T(0) / T(0);
        s->fnorm1 = s->actred = s->ratio = T(0);
        s->flags = 1 | 2 | 4;
        s->nfev = 0;
        s->term = VP_TERM_NOT_RUN;
        s->status = VP_ST_NOT_EVALUATED;
    }
    __syncthreads();

    const int max_fev = a.patience * (Q + 1);
    unsigned long long active = (count >= 64) ? ~0ull : ((1ull << count) - 1ull);
    int trow = 0; // per-lane trace row counter

    while (active != 0ull) {
        // =============================== VECTOR phase ===============================
        unsigned long long todo = active;
        while (todo != 0ull) {
            const int p = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            State *s = rec(p);
            const int64_t prob = p0 + p;
            T alpha[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) alpha[k] = s->xt[k];
            const int fl_in = uni(s->flags);
            const bool first = (fl_in & 1) != 0;

            RowSource<T, R> src;
            src.t = shared_t ? s_t : a.t + prob * a.t_stride;
            src.w = a.w ? (shared_w ? s_w : a.w + prob * a.w_stride) : nullptr;
            src.m = m;
            src.lane = lane;
            src.vec = shared_t ? (((m & 1) == 0) && (a.w == nullptr || shared_w || vec_aligned<T>(src.w, m)))
                               : (vec_aligned<T>(src.t, m) && (a.w == nullptr || shared_w || vec_aligned<T>(src.w, m)));

            T C[NC][R];
            EvalUniform<T, N> u;
            {
                const T *yp = a.yw + prob * (int64_t)m;
                load_rows<T, R>(yp, m, lane, vec_aligned<T>(yp, m), C[N]);
            }
            evaluate_core<T, M, R, NC, RowSource<T, R>, Grp<1>>(a.mdl, alpha, src, a.eps, grp, C, u);

            const T fnorm1 = tsqrt(u.fn2);
            T actred = T(0), ratio = T(0);
            bool good = false;
            if (!first) {
                const T fnorm = s->fnorm, prered = s->prered;
                const T q1 = fnorm1 * frcp(fnorm);
                actred = (fnorm1 * T(0.1) < fnorm) ? T(1) - q1 * q1 : T(-1);
                ratio = (prered == T(0)) ? T(0) : actred * frcp(prered);
                good = uni(ratio >= T(1.0e-4));
            }
            const bool need_jac = u.ok && (first || good);
            T Rj[Q][Q], acnorm[Q], qtf[Q];
            int ipvt[Q];
            if (need_jac) {
                T Zs[M::kDiagonalPairs ? 1 : Q][R];
                jacobian_qcoords<T, M, R, NC>(a.mdl, C, u.c, Zs, grp);
                residual_qcoords<T, R, N>(C[N], u.e, grp);
                if constexpr (M::kDiagonalPairs) {
                    jac_qrfac<T, R, Q, N>(reinterpret_cast<T(&)[Q][R]>(C[N + 1]), C[N], Rj, acnorm, ipvt, qtf, grp);
                } else {
                    jac_qrfac<T, R, Q, N>(Zs, C[N], Rj, acnorm, ipvt, qtf, grp);
                }
            }
            if (lane == 0) {
                s->fnorm1 = fnorm1;
                s->actred = actred;
                s->ratio = ratio;
#pragma unroll
                for (int k = 0; k < N; ++k) s->cnew[k] = u.c[k];
                int fl = fl_in & 7;
                if (u.ok) fl |= 8;
                if (need_jac) {
                    fl |= 16;
#pragma unroll
                    for (int k = 0; k < Q; ++k) {
                        s->acnorm[k] = acnorm[k];
                        s->qtf[k] = qtf[k];
                        s->ipvt[k] = ipvt[k];
#pragma unroll
                        for (int j = 0; j < Q; ++j) s->Rj[k][j] = Rj[k][j];
                    }
                }
                if (good) fl |= 32;
                s->flags = fl;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // =============================== SCALAR phase: lane p <-> problem p ===============================
        if ((active >> lane) & 1ull) {
            State *s = rec(lane);
            T x[Q], xt[Q], diag[Q], qtf[Q], acnorm[Q], step[Q];
            T Rj[Q][Q];
            int ipvt[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                x[k] = s->x[k];
                xt[k] = s->xt[k];
                diag[k] = s->diag[k];
                qtf[k] = s->qtf[k];
                acnorm[k] = s->acnorm[k];
                ipvt[k] = s->ipvt[k];
                step[k] = T(0);
#pragma unroll
                for (int j = 0; j < Q; ++j) Rj[k][j] = s->Rj[k][j];
            }
            T fnorm = s->fnorm, delta = s->delta, par = s->par, xnorm = s->xnorm, gnorm = s->gnorm;
            T pnorm = s->pnorm, prered = s->prered, dirder = s->dirder, objective = s->objective;
            const T fnorm1 = s->fnorm1, actred = s->actred, ratio = s->ratio;
            const int fl = s->flags;
            bool first = (fl & 1) != 0, first_tr = (fl & 2) != 0, first_update = (fl & 4) != 0;
            const bool ok = (fl & 8) != 0, jac_done = (fl & 16) != 0, good_v = (fl & 32) != 0;
            int nfev = s->nfev, term = 0, status = s->status;
            bool accept_c = false; // cbest <- cnew

            auto trace_row = [&](T rt) {
                if (a.trace && trow < a.trace_rows) {
                    double *tr = a.trace + ((size_t)(p0 + lane) * a.trace_rows + trow) * (Q + 4);
#pragma unroll
                    for (int k = 0; k < Q; ++k) tr[k] = (double)xt[k];
                    tr[Q] = (double)fnorm1;
                    tr[Q + 1] = (double)rt;
                    tr[Q + 2] = (double)delta;
                    tr[Q + 3] = (double)par;
                }
                ++trow;
            };

            bool need_step = false;
            if (first) {
                first = false;
                nfev = 1;
                status = ok ? VP_ST_OK : VP_ST_NONFINITE;
                if (!ok) {
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
                    for (int k = 0; k < Q; ++k) x[k] = xt[k];
                    accept_c = true;
                    status = VP_ST_NONFINITE;
                } else {
                    if (ratio <= T(0.25)) {
                        T temp = !(actred < T(0)) ? T(0.5) : T(0.5) * dirder * frcp(dirder + T(0.5) * actred);
                        if (fnorm1 * T(0.1) >= fnorm || temp < T(0.1)) temp = T(0.1);
                        delta = temp * tmin(delta, pnorm * T(10));
                        par = par * frcp(temp);
                    } else if (par == T(0) || ratio >= T(0.75)) {
                        delta = pnorm * T(2);
                        par = par * T(0.5);
                    }
                    trace_row(ratio);
                    if (good_v) {
#pragma unroll
                        for (int k = 0; k < Q; ++k) x[k] = xt[k];
                        accept_c = true;
                        T tmpv[Q];
#pragma unroll
                        for (int k = 0; k < Q; ++k) tmpv[k] = a.scale_diag ? diag[k] * x[k] : x[k];
                        xnorm = enorm_small<T, Q, false>(tmpv);
                        fnorm = fnorm1;
                        objective = T(0.5) * fnorm1 * fnorm1;
                        if (!is_finite(xnorm)) term = VP_TERM_NUMERICAL;
                    }
                    if (!term) {
                        int tcode = 0;
                        if (fnorm <= num<T>::tiny) tcode = VP_TERM_RESIDUALS_ZERO;
                        if (!tcode) {
                            const bool ftol_check =
                                tabs(actred) <= a.ftol && prered <= a.ftol && ratio * T(0.5) <= T(1);
                            const bool xtol_check = delta <= a.xtol * xnorm;
                            if (ftol_check || xtol_check)
                                tcode = (ftol_check && xtol_check)
                                            ? VP_TERM_CONVERGED_BOTH
                                            : (ftol_check ? VP_TERM_CONVERGED_FTOL : VP_TERM_CONVERGED_XTOL);
                        }
                        if (!tcode && nfev >= max_fev) tcode = VP_TERM_LOST_PATIENCE;
                        if (!tcode && tabs(actred) <= num<T>::eps && prered <= num<T>::eps && ratio * T(0.5) <= T(1))
                            tcode = VP_TERM_NO_IMPROVEMENT;
                        if (!tcode && delta <= num<T>::eps * xnorm) tcode = VP_TERM_NO_IMPROVEMENT;
                        if (!tcode && gnorm <= num<T>::eps) tcode = VP_TERM_NO_IMPROVEMENT;
                        term = tcode;
                        need_step = (tcode == 0);
                    }
                }
            }

            if (need_step && jac_done) {
                // the vector phase refreshed (Rj, qtf, acnorm, ipvt) at the accepted point
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
                if (degenerate) {
                    term = VP_TERM_NUMERICAL;
                } else if (gnorm <= a.gtol) {
                    term = VP_TERM_ORTHOGONAL;
                } else if (first_update) {
                    T tmpv[Q];
#pragma unroll
                    for (int k = 0; k < Q; ++k) {
                        if (a.scale_diag) diag[k] = (acnorm[k] == T(0)) ? T(1) : acnorm[k];
                        tmpv[k] = a.scale_diag ? diag[k] * x[k] : x[k];
                    }
                    xnorm = enorm_small<T, Q, false>(tmpv);
                    if (!is_finite(xnorm)) term = VP_TERM_NUMERICAL;
                    delta = (xnorm == T(0)) ? a.stepbound : a.stepbound * xnorm;
                    first_update = false;
                } else if (a.scale_diag) {
#pragma unroll
                    for (int k = 0; k < Q; ++k) diag[k] = tmax(diag[k], acnorm[k]);
                }
                if (term) need_step = false;
            }

            if (need_step) {
                par = lmpar<T, Q, false>(Rj, ipvt, diag, qtf, delta, par, step, pnorm);
                if (!is_finite(pnorm)) {
                    term = VP_TERM_NUMERICAL;
                } else {
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
                    const T t1 = enorm_small<T, Q, false>(wa) * ifn;
                    const T temp1 = t1 * t1;
                    const T t2 = (fsqrt(par) * pnorm) * ifn;
                    const T temp2 = t2 * t2;
                    if (!is_finite(temp1) || !is_finite(temp2)) {
                        term = VP_TERM_NUMERICAL;
                    } else {
                        prered = temp1 + temp2 * T(2);
                        dirder = -(temp1 + temp2);
                        if (first_tr && pnorm < delta) delta = pnorm;
                        first_tr = false;
#pragma unroll
                        for (int k = 0; k < Q; ++k) xt[k] = x[k] - step[k];
                    }
                }
            }

            // write the record back (Rj's lower triangle is lmpar scratch: restore the upper part only)
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                s->x[k] = x[k];
                s->xt[k] = xt[k];
                s->diag[k] = diag[k];
            }
            if (accept_c) {
#pragma unroll
                for (int k = 0; k < N; ++k) s->cbest[k] = s->cnew[k];
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

            if (term != 0) {
                // results of a finished problem (per-lane scattered stores; a few dozen bytes each)
                const int64_t prob = p0 + lane;
                vp_report rep;
                rep.termination = term;
                rep.n_evals = nfev;
                rep.objective = (double)objective;
                a.report[prob] = rep;
                if (a.cost_out) a.cost_out[prob] = (double)objective;
                if (a.status) a.status[prob] = status;
#pragma unroll
                for (int k = 0; k < Q; ++k) a.alpha[prob * Q + k] = x[k];
                if (a.C_out) {
#pragma unroll
                    for (int k = 0; k < N; ++k) a.C_out[prob * N + k] = s->cbest[k];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // lanes whose problem terminated drop out
        {
            int t = 0;
            if ((active >> lane) & 1ull) t = rec(lane)->term;
            active &= ~__ballot(t != 0);
        }
    }
}

template <typename T, class M, int R> int launch_fit_mp(const LaunchParams &p) {
    FitMpArgs<T, M> args;
    FitArgs<T, M> &a = args.f;
    if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
    a.t = (const T *)p.t;
    a.w = (const T *)p.w;
    a.yw = (const T *)p.yw;
    a.alpha = (T *)p.alpha_out;
    a.C_out = (T *)p.C_out;
    a.cost_out = p.cost_out;
    a.status = p.status;
    a.report = p.report;
    a.m = p.m;
    a.B = p.B;
    a.t_stride = p.t_stride;
    a.w_stride = p.w_stride;
    a.eps = (T)p.eps;
    a.ftol = (T)p.opts->ftol;
    a.xtol = (T)p.opts->xtol;
    a.gtol = (T)p.opts->gtol;
    a.stepbound = (T)p.opts->stepbound;
    a.patience = p.opts->patience;
    a.scale_diag = p.opts->scale_diag;
    a.trace = p.trace;
    a.trace_rows = p.trace_rows;
    if (a.B <= 0) return VP_ERR_OK;
    // problems per wave: enough waves to fill 256 CUs x 4 SIMDs x resident waves first, then grow G
    // (the scalar LM phase is amortised over G problems)
    int G = p.fit_group > 0 ? p.fit_group : (int)(a.B / (1024 * waves_for<T, R, M::N + 1 + M::P>()));
    if (G < 1) G = 1;
    if (G > 32) G = 32;
    args.G = G;
    const int64_t waves = (a.B + G - 1) / G;
    const size_t lds = (size_t)(p.w ? 2 : 1) * 64 * R * sizeof(T) + (size_t)G * mp_state_words<T, M::N, M::Q>() * 8;
    hipLaunchKernelGGL((fit_mp_kernel<T, M, R>), dim3((unsigned)waves), dim3(64), lds, p.stream, args);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
