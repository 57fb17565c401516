// vp_fit.hpp -- device-resident Levenberg-Marquardt over the variable-projection functional.
//
// == LevMarSolver::fit -> levenberg_marquardt::LevenbergMarquardt::minimize
//    (src/solvers/levmar/mod.rs:238-254, call site :247) for a batch: one wavefront runs the whole
//    MINPACK lmder-style iteration of one problem without ever leaving the chip.
//
// Key structural fact used here: the LM step needs only ||r||, the column norms of J, the
// triangular factor of J's pivoted QR and the first q entries of Q_J^T r.  All of these are
// invariant under an orthogonal change of basis of the residual space, so the loop works entirely
// in the Q-coordinates of Phi's Householder factorisation (r~ = Q^T r, J~ = Q^T J, both supported
// on rows >= n) and never back-transforms: per trial point it costs ONE fused QR sweep
// (vp_core.hpp), per accepted point additionally one pivoted QR of the q Jacobian columns.
// The grid t, the row scale (weights) and the weighted data y_w of the problem stay in LDS for the
// whole fit: HBM traffic per fit is m scalars in, q + n + report out.
#pragma once
#include "vp_kernels.hpp"

namespace vp {

// MINPACK qrsolv on wave-uniform registers.  r[row][col]: upper triangle incl. diagonal = R of the
// pivoted QR; the strict lower triangle is scratch (receives S^T).  Solves
// min || [R P^T; D] x - [qtb; 0] ||.
template <typename T, int Q>
__device__ __forceinline__ void qrsolv(T (&r)[Q][Q], const int (&ipvt)[Q], const T (&diag)[Q], const T (&qtb)[Q],
                                       T (&x)[Q], T (&sdiag)[Q]) {
    T wa[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
#pragma unroll
        for (int i = j; i < Q; ++i) r[i][j] = r[j][i];
        x[j] = r[j][j];
        wa[j] = qtb[j];
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const T dl = dyn_get<Q>(diag, ipvt[j]);
        if (uni(dl != T(0))) {
#pragma unroll
            for (int k = j; k < Q; ++k) sdiag[k] = T(0);
            sdiag[j] = dl;
            T qtbpj = T(0);
#pragma unroll
            for (int k = j; k < Q; ++k) {
                if (uni(sdiag[k] == T(0))) continue;
                T c, s;
                if (uni(tabs(r[k][k]) < tabs(sdiag[k]))) {
                    const T cotan = r[k][k] / sdiag[k];
                    s = T(0.5) / tsqrt(T(0.25) + T(0.25) * (cotan * cotan));
                    c = s * cotan;
                } else {
                    const T tn = sdiag[k] / r[k][k];
                    c = T(0.5) / tsqrt(T(0.25) + T(0.25) * (tn * tn));
                    s = c * tn;
                }
                r[k][k] = c * r[k][k] + s * sdiag[k];
                const T temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
#pragma unroll
                for (int i = k + 1; i < Q; ++i) {
                    const T t2 = c * r[i][k] + s * sdiag[i];
                    sdiag[i] = -s * r[i][k] + c * sdiag[i];
                    r[i][k] = t2;
                }
            }
        }
        sdiag[j] = r[j][j];
        r[j][j] = x[j];
    }
    int nsing = Q;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        if (sdiag[j] == T(0) && nsing == Q) nsing = j;
        if (nsing < Q) wa[j] = T(0);
    }
    nsing = uni(nsing);
#pragma unroll
    for (int k = 1; k <= Q; ++k) {
        const int j = Q - k; // only rows j < nsing participate
        if (j < nsing) {
            T sum = T(0);
#pragma unroll
            for (int i = j + 1; i < Q; ++i)
                if (i < nsing) sum = tfma(r[i][j], wa[i], sum);
            wa[j] = (wa[j] - sum) / sdiag[j];
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) dyn_set<Q>(x, ipvt[j], wa[j]);
}

template <typename T, int Q> __device__ __forceinline__ T enorm_small(const T (&v)[Q]) {
    // uniform q-vector norm with a scale guard (the MINPACK enorm protects against overflow)
    T mx = T(0);
#pragma unroll
    for (int j = 0; j < Q; ++j) mx = tmax(mx, tabs(v[j]));
    if (!(mx > T(0)) || !is_finite(mx)) return mx; // 0, inf or nan
    T s = T(0);
    const T inv = T(1) / mx;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const T u = v[j] * inv;
        s = tfma(u, u, s);
    }
    return mx * tsqrt(s);
}

// MINPACK lmpar.  Returns par; step = p (new point is x - p); dxnorm = ||diag .* p||.
template <typename T, int Q>
__device__ __forceinline__ T lmpar(T (&r)[Q][Q], const int (&ipvt)[Q], const T (&diag)[Q], const T (&qtb)[Q],
                                   const T delta, T par, T (&x)[Q], T &dxnorm_out) {
    const T p1 = T(0.1), p001 = T(0.001), dwarf = num<T>::tiny;
    T wa1[Q], wa2[Q], sdiag[Q];
    int nsing = Q;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == T(0) && nsing == Q) nsing = j;
        if (nsing < Q) wa1[j] = T(0);
    }
    nsing = uni(nsing);
#pragma unroll
    for (int k = 1; k <= Q; ++k) {
        const int j = Q - k;
        if (j < nsing) {
            wa1[j] = wa1[j] / r[j][j];
            const T temp = wa1[j];
#pragma unroll
            for (int i = 0; i < j; ++i) wa1[i] = tfma(-r[i][j], temp, wa1[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) dyn_set<Q>(x, ipvt[j], wa1[j]);
#pragma unroll
    for (int j = 0; j < Q; ++j) wa2[j] = diag[j] * x[j];
    T dxnorm = enorm_small<T, Q>(wa2);
    T fp = dxnorm - delta;
    if (uni(fp <= p1 * delta)) {
        dxnorm_out = dxnorm;
        return T(0);
    }
    T parl = T(0);
    if (nsing >= Q) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int l = ipvt[j];
            wa1[j] = dyn_get<Q>(diag, l) * (dyn_get<Q>(wa2, l) / dxnorm);
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            T sum = T(0);
#pragma unroll
            for (int i = 0; i < j; ++i) sum = tfma(r[i][j], wa1[i], sum);
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const T temp = enorm_small<T, Q>(wa1);
        parl = ((fp / delta) / temp) / temp;
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        T sum = T(0);
#pragma unroll
        for (int i = 0; i <= j; ++i) sum = tfma(r[i][j], qtb[i], sum);
        wa1[j] = sum / dyn_get<Q>(diag, ipvt[j]);
    }
    const T gnorm = enorm_small<T, Q>(wa1);
    T paru = gnorm / delta;
    if (paru == T(0)) paru = dwarf / tmin(delta, p1);
    par = tmax(par, parl);
    par = tmin(par, paru);
    if (par == T(0)) par = gnorm / dxnorm;
    for (int iter = 1;; ++iter) {
        if (par == T(0)) par = tmax(dwarf, p001 * paru);
        const T sq = tsqrt(par);
#pragma unroll
        for (int j = 0; j < Q; ++j) wa1[j] = sq * diag[j];
        qrsolv<T, Q>(r, ipvt, wa1, qtb, x, sdiag);
#pragma unroll
        for (int j = 0; j < Q; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm_small<T, Q>(wa2);
        const T temp = fp;
        fp = dxnorm - delta;
        if (uni(tabs(fp) <= p1 * delta || (parl == T(0) && fp <= temp && temp < T(0)) || iter == 10)) break;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int l = ipvt[j];
            wa1[j] = dyn_get<Q>(diag, l) * (dyn_get<Q>(wa2, l) / dxnorm);
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            wa1[j] = wa1[j] / sdiag[j];
            const T tj = wa1[j];
#pragma unroll
            for (int i = j + 1; i < Q; ++i) wa1[i] = tfma(-r[i][j], tj, wa1[i]);
        }
        const T tn = enorm_small<T, Q>(wa1);
        const T parc = ((fp / delta) / tn) / tn;
        if (fp > T(0)) parl = tmax(parl, par);
        if (fp < T(0)) paru = tmin(paru, par);
        par = tmax(parl, par + parc);
    }
    dxnorm_out = dxnorm;
    return par;
}

// MINPACK qrfac (column pivoting, partial-norm downdating) of the Q Jacobian columns Z living in
// rows >= ROW0, applied simultaneously to the residual column rv (-> qtf), as lmder does.
template <typename T, int R, int Q, int ROW0>
__device__ __forceinline__ void jac_qrfac(T (&Z)[Q][R], T (&rv)[R], T (&Rj)[Q][Q], T (&acnorm)[Q], int (&ipvt)[Q],
                                          T (&qtf)[Q], const int lane) {
    using L = Layout<R>;
    T rdiag[Q], wa[Q];
    {
        T s[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], Z[j][r], acc);
            s[j] = acc;
        }
        wave_allreduce(s);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            acnorm[j] = tsqrt(s[j]);
            rdiag[j] = acnorm[j];
            wa[j] = acnorm[j];
            ipvt[j] = j;
        }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int j = 0; j < Q; ++j) Rj[i][j] = T(0);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int prow = ROW0 + j;
        // bring the column of largest (downdated) norm into the pivot position
        int kmax = j;
#pragma unroll
        for (int k = j + 1; k < Q; ++k)
            if (dyn_get<Q>(rdiag, k) > dyn_get<Q>(rdiag, kmax)) kmax = k;
        kmax = uni(kmax);
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            if (kmax == k) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T tmp = Z[j][r];
                    Z[j][r] = Z[k][r];
                    Z[k][r] = tmp;
                }
                // rows < j of R already extracted also swap with their columns
#pragma unroll
                for (int i = 0; i < j; ++i) {
                    const T tmp = Rj[i][j];
                    Rj[i][j] = Rj[i][k];
                    Rj[i][k] = tmp;
                }
                rdiag[k] = rdiag[j];
                wa[k] = wa[j];
                const int ti = ipvt[j];
                ipvt[j] = ipvt[k];
                ipvt[k] = ti;
            }
        }
        // Householder vector for column j (rows >= prow)
        T s = T(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const T v = (L::row_of(r, lane) >= prow) ? Z[j][r] : T(0);
            s = tfma(v, v, s);
        }
        T ajnorm = tsqrt(wave_sum(s));
        if (uni(ajnorm == T(0))) {
            rdiag[j] = T(0);
            // remaining columns untouched: their row-prow entries are the R entries
#pragma unroll
            for (int k = j + 1; k < Q; ++k) Rj[j][k] = bcast_row<R>(Z[k], prow);
            qtf[j] = bcast_row<R>(rv, prow);
            continue;
        }
        const T piv = bcast_row<R>(Z[j], prow);
        if (piv < T(0)) ajnorm = -ajnorm;
        const T inv = T(1) / ajnorm;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = L::row_of(r, lane);
            const T v = Z[j][r] * inv;
            Z[j][r] = (i > prow) ? v : ((i == prow) ? v + T(1) : T(0));
        }
        const T vp = piv * inv + T(1); // v[prow]
        // dots with the remaining columns and with the residual column: one reduction round
        T w[Q]; // w[0..Q-j-2]: columns k > j ; w[Q-1]: residual
#pragma unroll
        for (int k = 0; k < Q; ++k) w[k] = T(0);
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], Z[k][r], acc);
            w[k - j - 1] = acc;
        }
        {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], rv[r], acc);
            w[Q - 1] = acc;
        }
        wave_allreduce(w);
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            const T temp = w[k - j - 1] / vp;
#pragma unroll
            for (int r = 0; r < R; ++r) Z[k][r] = tfma(-temp, Z[j][r], Z[k][r]);
            const T akj = bcast_row<R>(Z[k], prow);
            Rj[j][k] = akj;
            if (uni(rdiag[k] != T(0))) {
                const T tq = akj / rdiag[k];
                rdiag[k] = rdiag[k] * tsqrt(tmax(T(0), T(1) - tq * tq));
                const T rr = rdiag[k] / wa[k];
                if (uni(T(0.05) * (rr * rr) <= num<T>::eps)) {
                    T s2 = T(0);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const T v = (L::row_of(r, lane) > prow) ? Z[k][r] : T(0);
                        s2 = tfma(v, v, s2);
                    }
                    rdiag[k] = tsqrt(wave_sum(s2));
                    wa[k] = rdiag[k];
                }
            }
        }
        {
            const T temp = -w[Q - 1] / vp;
#pragma unroll
            for (int r = 0; r < R; ++r) rv[r] = tfma(temp, Z[j][r], rv[r]);
            qtf[j] = bcast_row<R>(rv, prow);
        }
        rdiag[j] = -ajnorm;
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) Rj[j][j] = rdiag[j];
}

template <typename T, class M> struct FitArgs {
    M mdl;
    const T *t;
    const T *w;
    const T *yw;
    T *alpha;      // in: initial guess, out: final parameters  [B][q]
    T *C_out;      // [B][n]
    double *cost_out;
    int32_t *status;
    vp_report *report;
    int m;
    int64_t B;
    int64_t t_stride, w_stride;
    T eps;
    T ftol, xtol, gtol, stepbound;
    int patience;
    int scale_diag;
    double *trace;  // diagnostics (vp_fit_trace): [B][trace_rows][q+4] or null
    int trace_rows;
};

template <typename T, class M, int R> __global__ void __launch_bounds__(64) fit_kernel(const FitArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    using L = Layout<R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s_t = reinterpret_cast<T *>(smem_raw);
    T *s_scale = s_t + 64 * R;
    T *s_y = s_scale + 64 * R;
    const int lane = lane_id();
    const int64_t b = blockIdx.x;
    if (b >= a.B) return;
    const int m = a.m;
    constexpr int MP = 64 * R;

    // stage the problem's grid, row scale and weighted data in LDS (row order, padding rows zero)
    {
        T tmp[R];
        const T *tp = a.t + b * a.t_stride;
        load_rows<T, R>(tp, m, lane, vec_aligned<T>(tp, m), tmp);
        store_rows<T, R>(s_t, MP, lane, true, tmp);
        load_scale<T, R>(a.w ? a.w + b * a.w_stride : nullptr, m, lane, tmp);
        store_rows<T, R>(s_scale, MP, lane, true, tmp);
        const T *yp = a.yw + b * (int64_t)m;
        load_rows<T, R>(yp, m, lane, vec_aligned<T>(yp, m), tmp);
        store_rows<T, R>(s_y, MP, lane, true, tmp);
    }
    __syncthreads(); // single wave: orders the LDS writes before the reads below

    // ---- LM state (wave-uniform) ----
    T x[Q], xt[Q], diag[Q], qtf[Q], step[Q], acnorm[Q], cbest[N];
    T Rj[Q][Q];
    int ipvt[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        xt[k] = a.alpha[b * Q + k];
        x[k] = xt[k];
        diag[k] = T(1);
        qtf[k] = T(0);
        step[k] = T(0);
        acnorm[k] = T(0);
        ipvt[k] = k;
#pragma unroll
        for (int j = 0; j < Q; ++j) Rj[k][j] = T(0);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) cbest[k] = T(0);
    T fnorm = T(0), delta = T(0), par = T(0), xnorm = T(0), gnorm = T(0);
    T pnorm = T(0), prered = T(0), dirder = T(0);
    T objective = T(0) / T(0); // NaN until the first evaluation succeeded
    bool first = true, first_tr = true, first_update = true;
    int nfev = 0, term = VP_TERM_NOT_RUN;
    int st_best = VP_ST_NOT_EVALUATED;
    const int max_fev = a.patience * (Q + 1);
    const int mres = m; // number of residuals (S == 1)
    int trow = 0;
    auto trace_row = [&](const T(&xx)[Q], T fn, T ratio) {
        if (a.trace && trow < a.trace_rows && lane == 0) {
            double *tr = a.trace + ((size_t)b * a.trace_rows + trow) * (Q + 4);
#pragma unroll
            for (int k = 0; k < Q; ++k) tr[k] = (double)xx[k];
            tr[Q] = (double)fn;
            tr[Q + 1] = (double)ratio;
            tr[Q + 2] = (double)delta;
            tr[Q + 3] = (double)par;
        }
        ++trow;
    };

    for (;;) {
        // ================= evaluate the VarPro functional at xt =================
        T A[N][R], X[1 + P][R];
        EvalUniform<T, N> u;
        {
            T t[R], scale[R], yw[R];
            load_rows<T, R>(s_t, MP, lane, true, t);
            load_rows<T, R>(s_scale, MP, lane, true, scale);
            load_rows<T, R>(s_y, MP, lane, true, yw);
            evaluate_core<T, M, R>(a.mdl, xt, t, scale, yw, a.eps, lane, A, X, u);
        }
        const T fnorm1 = tsqrt(u.fn2);
        bool need_jac = false;
        if (first) {
            first = false;
            nfev = 1;
            st_best = u.ok ? VP_ST_OK : VP_ST_NONFINITE;
            if (!u.ok) { // residuals() == None
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
            if (!is_finite(fnorm)) {
                term = VP_TERM_NUMERICAL;
                break;
            }
            if (fnorm <= num<T>::tiny) {
                term = VP_TERM_RESIDUALS_ZERO;
                break;
            }
            need_jac = true;
        } else {
            nfev += 1;
            if (!u.ok) { // residuals() == None at the trial point
                term = VP_TERM_USER;
                // the problem keeps the trial parameters (the reference's target holds them too)
#pragma unroll
                for (int k = 0; k < Q; ++k) x[k] = xt[k];
#pragma unroll
                for (int k = 0; k < N; ++k) cbest[k] = u.c[k];
                st_best = VP_ST_NONFINITE;
                break;
            }
            const T q1 = fnorm1 / fnorm;
            const T actred = (fnorm1 * T(0.1) < fnorm) ? T(1) - q1 * q1 : T(-1);
            const T ratio = (prered == T(0)) ? T(0) : actred / prered;
            if (ratio <= T(0.25)) {
                T temp = !(actred < T(0)) ? T(0.5) : T(0.5) * dirder / (dirder + T(0.5) * actred);
                if (fnorm1 * T(0.1) >= fnorm || temp < T(0.1)) temp = T(0.1);
                delta = temp * tmin(delta, pnorm * T(10));
                par = par / temp;
            } else if (par == T(0) || ratio >= T(0.75)) {
                delta = pnorm / T(0.5);
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
                for (int k = 0; k < Q; ++k) tmpv[k] = a.scale_diag ? diag[k] * x[k] : x[k];
                xnorm = enorm_small<T, Q>(tmpv);
                fnorm = fnorm1;
                objective = T(0.5) * fnorm1 * fnorm1;
                if (!is_finite(xnorm)) {
                    term = VP_TERM_NUMERICAL;
                    break;
                }
            }
            int tcode = 0;
            if (fnorm <= num<T>::tiny) tcode = VP_TERM_RESIDUALS_ZERO;
            if (!tcode) {
                const bool ftol_check = tabs(actred) <= a.ftol && prered <= a.ftol && ratio * T(0.5) <= T(1);
                const bool xtol_check = delta <= a.xtol * xnorm;
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
                // reset_params_if(!good): x / cbest already hold the best point, nothing to recompute
                term = tcode;
                break;
            }
            need_jac = good;
        }

        if (need_jac) {
            // ================= Jacobian in Q-coordinates, pivoted QR, Q_J^T r =================
            T Z[Q][R];
            jacobian_qcoords<T, M, R>(a.mdl, X, u.c, Z, lane);
            residual_qcoords<T, R, N>(X[0], u.e, lane);
            jac_qrfac<T, R, Q, N>(Z, X[0], Rj, acnorm, ipvt, qtf, lane);
            // norm of the scaled gradient
            T g = T(0);
            bool degenerate = false;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const T an = dyn_get<Q>(acnorm, ipvt[j]);
                if (an != T(0)) {
                    T sum = T(0);
#pragma unroll
                    for (int i = 0; i <= j; ++i) sum = tfma(Rj[i][j], qtf[i], sum);
                    const T temp = tabs(sum / (an * fnorm));
                    if (temp != temp) degenerate = true;
                    g = tmax(g, temp);
                }
            }
            gnorm = g;
            if (uni(degenerate)) {
                term = VP_TERM_NUMERICAL;
                break;
            }
            if (uni(gnorm <= a.gtol)) {
                term = VP_TERM_ORTHOGONAL;
                break;
            }
            if (first_update) {
                T tmpv[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    if (a.scale_diag) diag[k] = (acnorm[k] == T(0)) ? T(1) : acnorm[k];
                    tmpv[k] = a.scale_diag ? diag[k] * x[k] : x[k];
                }
                xnorm = enorm_small<T, Q>(tmpv);
                if (uni(!is_finite(xnorm))) {
                    term = VP_TERM_NUMERICAL;
                    break;
                }
                delta = (xnorm == T(0)) ? a.stepbound : a.stepbound * xnorm;
                first_update = false;
            } else if (a.scale_diag) {
#pragma unroll
                for (int k = 0; k < Q; ++k) diag[k] = tmax(diag[k], acnorm[k]);
            }
        }

        // ================= trust-region step =================
        par = lmpar<T, Q>(Rj, ipvt, diag, qtf, delta, par, step, pnorm);
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
            const T t1 = enorm_small<T, Q>(wa) / fnorm;
            const T temp1 = t1 * t1;
            const T t2 = (tsqrt(par) * pnorm) / fnorm;
            const T temp2 = t2 * t2;
            if (uni(!is_finite(temp1) || !is_finite(temp2))) {
                term = VP_TERM_NUMERICAL;
                break;
            }
            prered = temp1 + temp2 / T(0.5);
            dirder = -(temp1 + temp2);
        }
        if (first_tr && pnorm < delta) delta = pnorm;
        first_tr = false;
#pragma unroll
        for (int k = 0; k < Q; ++k) xt[k] = x[k] - step[k];
    }

    // ================= results =================
    if (lane == 0) {
        vp_report rep;
        rep.termination = term;
        rep.n_evals = nfev;
        rep.objective = (double)objective;
        a.report[b] = rep;
        if (a.cost_out) a.cost_out[b] = (double)objective;
        if (a.status) a.status[b] = st_best;
    }
    if (lane < Q) a.alpha[b * Q + lane] = dyn_get<Q>(x, lane);
    if (a.C_out && lane < N) a.C_out[b * N + lane] = dyn_get<N>(cbest, lane);
}

template <typename T, class M, int R> int launch_fit(const LaunchParams &p) {
    FitArgs<T, M> a;
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
    const size_t lds = (size_t)3 * 64 * R * sizeof(T);
    hipLaunchKernelGGL((fit_kernel<T, M, R>), dim3((unsigned)a.B), dim3(64), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// == FitResult::best_fit (src/fit.rs:55-59, 87-91): UNWEIGHTED Phi(alpha) * C
template <typename T, class M> struct BestFitArgs {
    M mdl;
    const T *t;
    const T *alpha;
    const T *C;
    T *out;
    int m, S;
    int64_t nprob;
    int64_t t_stride;
};

template <typename T, class M, int R> __global__ void __launch_bounds__(64) best_fit_kernel(const BestFitArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    using L = Layout<R>;
    const int lane = lane_id();
    const int64_t prob = blockIdx.x;
    if (prob >= a.nprob) return;
    const int64_t b = prob / a.S;
    const int m = a.m;
    T alpha[Q], c[N];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = a.C[prob * N + k];
    T t[R], scale[R];
    const T *tp = a.t + b * a.t_stride;
    load_rows<T, R>(tp, m, lane, vec_aligned<T>(tp, m), t);
#pragma unroll
    for (int r = 0; r < R; ++r) scale[r] = (L::row_of(r, lane) < m) ? T(1) : T(0);
    T A[N][R], D[P > 0 ? P : 1][R];
    build_columns<T, M, R>(a.mdl, alpha, t, scale, A, D);
    T f[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) acc = tfma(A[j][r], c[j], acc);
        f[r] = acc;
    }
    T *op = a.out + prob * (int64_t)m;
    store_rows<T, R>(op, m, lane, vec_aligned<T>(op, m), f);
}

template <typename T, class M, int R> int launch_best_fit(const LaunchParams &p) {
    BestFitArgs<T, M> a;
    if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
    a.t = (const T *)p.t;
    a.alpha = (const T *)p.alpha;
    a.C = (const T *)p.C_out;
    a.out = (T *)p.r_out;
    a.m = p.m;
    a.S = p.S;
    a.nprob = p.B * p.S;
    a.t_stride = p.t_stride;
    if (a.nprob <= 0) return VP_ERR_OK;
    hipLaunchKernelGGL((best_fit_kernel<T, M, R>), dim3((unsigned)a.nprob), dim3(64), 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
