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
//
// The wave-uniform bookkeeping (lmpar, qrsolv, trust-region update) runs on the vector ALU -- gfx950
// has no scalar fp64 -- so every instruction of it costs as much as a 64-row vector instruction.
// It is therefore written division-free where MINPACK divides repeatedly by the same quantity
// (hoisted Newton-refined reciprocals, rsqrt-based Givens rotations).
#pragma once
#include "vp_kernels.hpp"

namespace vp {

// branch policy: U == true  -> the operands are wave-uniform, make the branch scalar (s_cbranch);
//                U == false -> every lane runs its own problem (lane-parallel LM bookkeeping)
template <bool U> __device__ __forceinline__ bool pol(bool c) {
    if constexpr (U) return uni(c);
    else return c;
}
template <bool U> __device__ __forceinline__ int pol(int v) {
    if constexpr (U) return uni(v);
    else return v;
}

// plain Euclidean norm of a wave-uniform q-vector; falls back to a scaled accumulation (the point of
// MINPACK's enorm) only when the plain sum of squares over/underflows
template <typename T, int Q, bool U = true> __device__ __forceinline__ T enorm_small(const T (&v)[Q]) {
    T s = T(0);
#pragma unroll
    for (int j = 0; j < Q; ++j) s = tfma(v[j], v[j], s);
    if (pol<U>(s > T(1e-280) && s < T(1e280))) return usqrt(s);
    T mx = T(0);
#pragma unroll
    for (int j = 0; j < Q; ++j) mx = tmax(mx, tabs(v[j]));
    if (!(mx > T(0)) || !is_finite(mx)) return (s != s) ? s : mx; // 0, inf or nan
    s = T(0);
    const T inv = T(1) / mx;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const T u = v[j] * inv;
        s = tfma(u, u, s);
    }
    return mx * tsqrt(s);
}

// MINPACK qrsolv on wave-uniform registers.  r[row][col]: upper triangle incl. diagonal = R of the
// pivoted QR; the strict lower triangle is scratch (receives S^T).  Solves
// min || [R P^T; D] x - [qtb; 0] ||.
template <typename T, int Q, bool U = true, bool O = false>
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
        const T dl = dyn_get_o<Q, O>(diag, ipvt[j]);
        if (pol<U>(dl != T(0))) {
#pragma unroll
            for (int k = j; k < Q; ++k) sdiag[k] = T(0);
            sdiag[j] = dl;
            T qtbpj = T(0);
#pragma unroll
            for (int k = j; k < Q; ++k) {
                if (pol<U>(sdiag[k] == T(0))) continue;
                // Givens rotation (c, s) = sg * (r_kk, sdiag_k) / hypot, sg as in MINPACK's two branches
                const T rk = r[k][k], sk = sdiag[k];
                const T ih = frsqrt(tfma(rk, rk, sk * sk));
                const T sg = (tabs(rk) < tabs(sk)) ? tcopysign(T(1), sk) : tcopysign(T(1), rk);
                const T c = sg * rk * ih;
                const T s = sg * sk * ih;
                r[k][k] = c * rk + s * sk;
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
    nsing = pol<U>(nsing);
#pragma unroll
    for (int k = 1; k <= Q; ++k) {
        const int j = Q - k; // only rows j < nsing participate
        if (j < nsing) {
            T sum = T(0);
#pragma unroll
            for (int i = j + 1; i < Q; ++i)
                if (i < nsing) sum = tfma(r[i][j], wa[i], sum);
            wa[j] = (wa[j] - sum) * frcp(sdiag[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) dyn_set_o<Q, O>(x, ipvt[j], wa[j]);
}

// ((fp/delta)/temp)/temp of MINPACK's lmpar with temp = ||w||, given the plain sum of squares t2 = sum w_j^2.  MINPACK
// takes the norm with enorm (scaled: no over/underflow) and divides twice; the fast form x * rcp(t2) is the same number
// whenever t2 is an ordinary one.  With a nearly singular factor (a Jacobian column that underflowed: diagonal entries of
// 1e-189) w reaches 1e189, t2 overflows and rcp(inf) refined by Newton steps is NaN -- the step, and with it the fit,
// ended `Numerical` where the reference's lmpar carries on with parc = 0.  Out-of-range sums take MINPACK's own route.
template <typename T, int Q, bool U>
__device__ __forceinline__ T lmpar_ratio(const T x, const T t2, const T (&w)[Q]) {
    if (pol<U>(t2 > T(1e-280) && t2 < T(1e280))) return x * frcp(t2);
    const T tn = enorm_small<T, Q, U>(w);
    return (x / tn) / tn;
}

// MINPACK lmpar.  Returns par; step = p (new point is x - p); dxnorm = ||diag .* p||.
template <typename T, int Q, bool U = true, bool O = false>
__device__ __forceinline__ T lmpar(T (&r)[Q][Q], const int (&ipvt)[Q], const T (&diag)[Q], const T (&qtb)[Q],
                                   const T delta, T par, T (&x)[Q], T &dxnorm_out) {
    const T p1 = T(0.1), p001 = T(0.001), dwarf = num<T>::tiny;
    T wa1[Q], wa2[Q], sdiag[Q], rinv[Q], dperm[Q];
    int nsing = Q;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == T(0) && nsing == Q) nsing = j;
        if (nsing < Q) wa1[j] = T(0);
        rinv[j] = (r[j][j] != T(0)) ? frcp(r[j][j]) : T(0);
        dperm[j] = dyn_get_o<Q, O>(diag, ipvt[j]);
    }
    nsing = pol<U>(nsing);
#pragma unroll
    for (int k = 1; k <= Q; ++k) {
        const int j = Q - k;
        if (j < nsing) {
            wa1[j] = wa1[j] * rinv[j];
            const T temp = wa1[j];
#pragma unroll
            for (int i = 0; i < j; ++i) wa1[i] = tfma(-r[i][j], temp, wa1[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) dyn_set_o<Q, O>(x, ipvt[j], wa1[j]);
#pragma unroll
    for (int j = 0; j < Q; ++j) wa2[j] = diag[j] * x[j];
    T dxnorm = enorm_small<T, Q, U>(wa2);
    T fp = dxnorm - delta;
    if (pol<U>(fp <= p1 * delta)) {
        dxnorm_out = dxnorm;
        return T(0);
    }
    const T idelta = frcp(delta);
    T parl = T(0);
    if (nsing >= Q) {
        const T idx = frcp(dxnorm);
#pragma unroll
        for (int j = 0; j < Q; ++j) wa1[j] = dperm[j] * (dyn_get_o<Q, O>(wa2, ipvt[j]) * idx);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            T sum = T(0);
#pragma unroll
            for (int i = 0; i < j; ++i) sum = tfma(r[i][j], wa1[i], sum);
            wa1[j] = (wa1[j] - sum) * rinv[j];
        }
        T t2 = T(0);
#pragma unroll
        for (int j = 0; j < Q; ++j) t2 = tfma(wa1[j], wa1[j], t2);
        parl = lmpar_ratio<T, Q, U>(fp * idelta, t2, wa1); // ((fp/delta)/temp)/temp
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        T sum = T(0);
#pragma unroll
        for (int i = 0; i <= j; ++i) sum = tfma(r[i][j], qtb[i], sum);
        wa1[j] = sum * frcp(dperm[j]);
    }
    const T gnorm = enorm_small<T, Q, U>(wa1);
    T paru = gnorm * idelta;
    if (paru == T(0)) paru = dwarf / tmin(delta, p1);
    par = tmax(par, parl);
    par = tmin(par, paru);
    if (par == T(0)) par = gnorm * frcp(dxnorm);
    for (int iter = 1;; ++iter) {
        if (par == T(0)) par = tmax(dwarf, p001 * paru);
        const T sq = usqrt(par);
#pragma unroll
        for (int j = 0; j < Q; ++j) wa1[j] = sq * diag[j];
        qrsolv<T, Q, U, O>(r, ipvt, wa1, qtb, x, sdiag);
#pragma unroll
        for (int j = 0; j < Q; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm_small<T, Q, U>(wa2);
        const T temp = fp;
        fp = dxnorm - delta;
        if (pol<U>(tabs(fp) <= p1 * delta || (parl == T(0) && fp <= temp && temp < T(0)) || iter == 10)) break;
        const T idx = frcp(dxnorm);
#pragma unroll
        for (int j = 0; j < Q; ++j) wa1[j] = dperm[j] * (dyn_get_o<Q, O>(wa2, ipvt[j]) * idx);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            wa1[j] = wa1[j] * frcp(sdiag[j]);
            const T tj = wa1[j];
#pragma unroll
            for (int i = j + 1; i < Q; ++i) wa1[i] = tfma(-r[i][j], tj, wa1[i]);
        }
        T t2 = T(0);
#pragma unroll
        for (int j = 0; j < Q; ++j) t2 = tfma(wa1[j], wa1[j], t2);
        const T parc = lmpar_ratio<T, Q, U>(fp * idelta, t2, wa1);
        if (fp > T(0)) parl = tmax(parl, par);
        if (fp < T(0)) paru = tmin(paru, par);
        par = tmax(parl, par + parc);
    }
    dxnorm_out = dxnorm;
    return par;
}

// lmpar for TWO nonlinear parameters (round 5): MINPACK's lmpar statement by statement -- Gauss-Newton shortcut, the bounds
// parl / paru, the Newton iteration on par and its exits -- with qrsolv's three Givens rotations per iteration (a chain of
// three reciprocal square roots with their sign selects and zero branches, ~60 dependent instructions) replaced by the
// 2 x 2 triangular factor S of  R^T R + par D^2  written out:  s00^2 = r00^2 + par d0^2,  s01 = r00 r01 / s00,
// s11^2 = r11^2 + par d1^2 + r01^2 (par d0^2 / s00^2)  -- the Schur complement as a sum of non-negative terms, so every
// entry of S carries a few ulp of RELATIVE error whatever the conditioning of R (the subtraction c - s01^2 of a generic
// Cholesky is what loses digits).  S is the matrix qrsolv produces (it is unique up to row signs, which cancel in the
// step).  One function for every fp64 Householder kernel (fit_kernel, the slot kernel's scalar phase and its lone tail):
// their reports stay bit-identical to each other.
#ifndef VP_LMPAR_Q2
#define VP_LMPAR_Q2 1
#endif
template <typename T, bool U = true>
__device__ __forceinline__ T lmpar_q2(const T (&r)[2][2], const int (&ipvt)[2], const T (&diag)[2], const T (&qtb)[2],
                                      const T delta, T par, T (&x)[2], T &dxnorm_out) {
    const T p1 = T(0.1), p001 = T(0.001), dwarf = num<T>::tiny;
    const T r00 = r[0][0], r01 = r[0][1], r11 = r[1][1];
    const bool swapped = ipvt[0] != 0;
    const T d0 = swapped ? diag[1] : diag[0], d1 = swapped ? diag[0] : diag[1]; // D in pivoted order
    const T b0 = qtb[0], b1 = qtb[1];
    const bool full = (r00 != T(0)) && (r11 != T(0)); // nsing == 2
    const T i00 = (r00 != T(0)) ? frcp(r00) : T(0), i11 = full ? frcp(r11) : T(0);
    // Gauss-Newton direction, zero beyond the numerical rank
    T q1 = b1 * i11;
    T q0 = tfma(-r01, q1, b0) * i00;
    T w[2] = {d0 * q0, d1 * q1};
    T dxnorm = enorm_small<T, 2, U>(w);
    T fp = dxnorm - delta;
    if (pol<U>(fp <= p1 * delta)) {
        x[0] = swapped ? q1 : q0;
        x[1] = swapped ? q0 : q1;
        dxnorm_out = dxnorm;
        return T(0);
    }
    const T idelta = frcp(delta);
    T parl = T(0);
    if (full) {
        const T idx = frcp(dxnorm);
        T z[2];
        z[0] = (d0 * (w[0] * idx)) * i00;                      // R^T z = D^2 p / ||D p||
        z[1] = tfma(-r01, z[0], d1 * (w[1] * idx)) * i11;
        parl = lmpar_ratio<T, 2, U>(fp * idelta, tfma(z[0], z[0], z[1] * z[1]), z);
    }
    const T g0 = r00 * b0, g1 = tfma(r01, b0, r11 * b1); // R^T (Q^T f) = P^T J^T f
    T u[2] = {g0 * frcp(d0), g1 * frcp(d1)};
    const T gnorm = enorm_small<T, 2, U>(u);
    T paru = gnorm * idelta;
    if (paru == T(0)) paru = dwarf / tmin(delta, p1);
    par = tmax(par, parl);
    par = tmin(par, paru);
    if (par == T(0)) par = gnorm * frcp(dxnorm);
    const T d02 = d0 * d0, d12 = d1 * d1, r00s = r00 * r00, r01s = r01 * r01, r11s = r11 * r11, r0001 = r00 * r01;
    for (int iter = 1;; ++iter) {
        if (par == T(0)) par = tmax(dwarf, p001 * paru);
        const T pd0 = par * d02, pd1 = par * d12;
        const T a = r00s + pd0;
        const T is0 = (a > T(0)) ? frsqrt(a) : T(0); // 1 / s00
        const T s01 = r0001 * is0;
        const T c = tfma(r01s, pd0 * (is0 * is0), r11s + pd1);
        const T is1 = (c > T(0)) ? frsqrt(c) : T(0); // 1 / s11
        const T w0 = g0 * is0;                        // S^T w = g
        const T w1 = tfma(-s01, w0, g1) * is1;
        q1 = w1 * is1;                                // S p = w
        q0 = tfma(-s01, q1, w0) * is0;
        w[0] = d0 * q0;
        w[1] = d1 * q1;
        dxnorm = enorm_small<T, 2, U>(w);
        const T temp = fp;
        fp = dxnorm - delta;
        if (pol<U>(tabs(fp) <= p1 * delta || (parl == T(0) && fp <= temp && temp < T(0)) || iter == 10)) break;
        const T idx = frcp(dxnorm);
        T z[2];
        z[0] = (d0 * (w[0] * idx)) * is0;             // S^T z = D^2 p / ||D p||
        z[1] = tfma(-s01, z[0], d1 * (w[1] * idx)) * is1;
        const T parc = lmpar_ratio<T, 2, U>(fp * idelta, tfma(z[0], z[0], z[1] * z[1]), z);
        if (fp > T(0)) parl = tmax(parl, par);
        if (fp < T(0)) paru = tmin(paru, par);
        par = tmax(parl, par + parc);
    }
    x[0] = swapped ? q1 : q0;
    x[1] = swapped ? q0 : q1;
    dxnorm_out = dxnorm;
    return par;
}
// the trust-region sub-problem of the fp64 Householder kernels: MINPACK's lmpar, or its two-parameter form
template <typename T, int Q, bool U = true, bool O = false>
__device__ __forceinline__ T lmpar_any(T (&r)[Q][Q], const int (&ipvt)[Q], const T (&diag)[Q], const T (&qtb)[Q],
                                       const T delta, T par, T (&x)[Q], T &dxnorm_out) {
    if constexpr (Q == 2 && VP_LMPAR_Q2 && sizeof(T) == 8) return lmpar_q2<T, U>(r, ipvt, diag, qtb, delta, par, x, dxnorm_out);
    else return lmpar<T, Q, U, O>(r, ipvt, diag, qtb, delta, par, x, dxnorm_out);
}

// lmpar for callers whose R is itself the Cholesky factor of a Gram matrix J^T J (vp_fitg.hpp: J is never materialised).
// Same trust-region sub-problem, same Newton iteration on par, same exits as lmpar above; what differs is how the
// regularised factor S  (S^T S = R^T R + par D_p^2, what qrsolv's Q(Q+1)/2 Givens rotations produce) is obtained: as the
// Cholesky factor of  G + par D_p^2  with  G = R^T R  formed once per call.  Each rotation of qrsolv needs the row the
// previous one left (a chain of Q(Q+1)/2 reciprocal square roots with their selects and branches: ~2 500 instructions per
// lmpar iteration at Q = 5 on one lane); the Cholesky form is Q dependent pivots and ~Q^3/3 multiply-adds (~250).  R came
// from the Gram matrix in the first place, so nothing is lost against the data: G is what the caller measured, R its
// factor.  A pivot is clamped from below by  par d_j^2  -- its exact lower bound (the Schur complement of a positive
// semi-definite matrix plus par D^2) -- so that rounding in an ill-conditioned G cannot drive it through zero.
// The step and the norms are formed in pivoted order (a permutation does not change a Euclidean norm).
template <typename T, int Q, bool U = true, bool O = false>
__device__ __forceinline__ T lmpar_chol(const T (&r)[Q][Q], const int (&ipvt)[Q], const T (&diag)[Q], const T (&qtb)[Q],
                                        const T delta, T par, T (&x)[Q], T &dxnorm_out) {
    const T p1 = T(0.1), p001 = T(0.001), dwarf = num<T>::tiny;
    T wa1[Q], wa2[Q], rinv[Q], dperm[Q];
    int nsing = Q;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        wa1[j] = qtb[j];
        if (r[j][j] == T(0) && nsing == Q) nsing = j;
        if (nsing < Q) wa1[j] = T(0);
        rinv[j] = (r[j][j] != T(0)) ? frcp(r[j][j]) : T(0);
        dperm[j] = dyn_get_o<Q, O>(diag, ipvt[j]);
    }
    nsing = pol<U>(nsing);
    // Gauss-Newton direction (pivoted order): R_11 p = (Q^T f)_1, zero beyond the numerical rank
#pragma unroll
    for (int k = 1; k <= Q; ++k) {
        const int j = Q - k;
        if (j < nsing) {
            wa1[j] = wa1[j] * rinv[j];
            const T temp = wa1[j];
#pragma unroll
            for (int i = 0; i < j; ++i) wa1[i] = tfma(-r[i][j], temp, wa1[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) wa2[j] = dperm[j] * wa1[j];
    T dxnorm = enorm_small<T, Q, U>(wa2);
    T fp = dxnorm - delta;
    if (pol<U>(fp <= p1 * delta)) {
#pragma unroll
        for (int j = 0; j < Q; ++j) dyn_set_o<Q, O>(x, ipvt[j], wa1[j]);
        dxnorm_out = dxnorm;
        return T(0);
    }
    const T idelta = frcp(delta);
    T parl = T(0);
    if (nsing >= Q) {
        const T idx = frcp(dxnorm);
        T z[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) z[j] = dperm[j] * (wa2[j] * idx);
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            T sum = T(0);
#pragma unroll
            for (int i = 0; i < j; ++i) sum = tfma(r[i][j], z[i], sum);
            z[j] = (z[j] - sum) * rinv[j];
        }
        T t2 = T(0);
#pragma unroll
        for (int j = 0; j < Q; ++j) t2 = tfma(z[j], z[j], t2);
        parl = lmpar_ratio<T, Q, U>(fp * idelta, t2, z);
    }
    // G = R^T R (upper triangle) and b = R^T (Q^T f) = P^T J^T f
    T G[Q][Q], bp[Q], dp2[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
#pragma unroll
        for (int i = 0; i <= j; ++i) {
            T sum = T(0);
#pragma unroll
            for (int k = 0; k <= i; ++k) sum = tfma(r[k][i], r[k][j], sum);
            G[i][j] = sum;
        }
        T sum = T(0);
#pragma unroll
        for (int i = 0; i <= j; ++i) sum = tfma(r[i][j], qtb[i], sum);
        bp[j] = sum;
        wa1[j] = sum * frcp(dperm[j]);
        dp2[j] = dperm[j] * dperm[j];
    }
    const T gnorm = enorm_small<T, Q, U>(wa1);
    T paru = gnorm * idelta;
    if (paru == T(0)) paru = dwarf / tmin(delta, p1);
    par = tmax(par, parl);
    par = tmin(par, paru);
    if (par == T(0)) par = gnorm * frcp(dxnorm);
    T xp[Q];
    for (int iter = 1;; ++iter) {
        if (par == T(0)) par = tmax(dwarf, p001 * paru);
        // S^T S = G + par D_p^2  (S upper triangular, is[j] = 1 / S_jj)
        T S[Q][Q], is[Q], w[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const T floor_j = par * dp2[j];
            T d = G[j][j] + floor_j;
#pragma unroll
            for (int k = 0; k < j; ++k) d = tfma(-S[k][j], S[k][j], d);
            d = tmax(d, floor_j);
            is[j] = (d > T(0)) ? frsqrt(d) : T(0);
#pragma unroll
            for (int l = j + 1; l < Q; ++l) {
                T a = G[j][l];
#pragma unroll
                for (int k = 0; k < j; ++k) a = tfma(-S[k][j], S[k][l], a);
                S[j][l] = a * is[j];
            }
            // S^T w = b alongside
            T a = bp[j];
#pragma unroll
            for (int k = 0; k < j; ++k) a = tfma(-S[k][j], w[k], a);
            w[j] = a * is[j];
        }
        // S p = w
#pragma unroll
        for (int k = 1; k <= Q; ++k) {
            const int j = Q - k;
            T a = w[j];
#pragma unroll
            for (int i = j + 1; i < Q; ++i) a = tfma(-S[j][i], xp[i], a);
            xp[j] = a * is[j];
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) wa2[j] = dperm[j] * xp[j];
        dxnorm = enorm_small<T, Q, U>(wa2);
        const T temp = fp;
        fp = dxnorm - delta;
        if (pol<U>(tabs(fp) <= p1 * delta || (parl == T(0) && fp <= temp && temp < T(0)) || iter == 10)) break;
        const T idx = frcp(dxnorm);
        T t2 = T(0);
#pragma unroll
        for (int j = 0; j < Q; ++j) { // S^T z = D_p^2 p / ||D p||
            T a = dperm[j] * (wa2[j] * idx);
#pragma unroll
            for (int k = 0; k < j; ++k) a = tfma(-S[k][j], wa1[k], a);
            wa1[j] = a * is[j];
            t2 = tfma(wa1[j], wa1[j], t2);
        }
        const T parc = lmpar_ratio<T, Q, U>(fp * idelta, t2, wa1);
        if (fp > T(0)) parl = tmax(parl, par);
        if (fp < T(0)) paru = tmin(paru, par);
        par = tmax(parl, par + parc);
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) dyn_set_o<Q, O>(x, ipvt[j], xp[j]);
    dxnorm_out = dxnorm;
    return par;
}

// MINPACK qrfac (column pivoting, partial-norm downdating) of the Q Jacobian columns Z living in
// rows >= ROW0 (rows < ROW0 of Z are zero), applied simultaneously to the residual column rv (-> qtf), as lmder
// does.  MINPACK's reflector  v = a/ajnorm + e_p,  H = I - v v^T / v_p  is applied in the equivalent
// unnormalised form  H = I + g v' v'^T,  v' = a + ajnorm e_p,  g = -1/(ajnorm v'_p).
// Reduction rounds: ONE for the whole first step -- the Gram matrix of [Z | rv] gives the column norms (pivot
// choice, acnorm), the pivot column's exact norm and its raw dot products  a^T z_k, a^T rv  (v'^T z = a^T z +
// ajnorm z[p]) -- and one per further step (raw dots of the new pivot column, its norm among them): Q rounds
// instead of 1 + 2Q.
template <typename T, int R, int Q, int ROW0, class G>
__device__ __forceinline__ void jac_qrfac(T (&Z)[Q][R], T (&rv)[R], T (&Rj)[Q][Q], T (&acnorm)[Q], int (&ipvt)[Q],
                                          T (&qtf)[Q], G &grp) {
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
    T rdiag[Q], wa[Q];
    T Gm[Q][Q], bz[Q]; // Gram matrix of the columns (symmetric, both triangles kept) and Z^T rv
    {
        constexpr int NG = Q * (Q + 1) / 2 + Q;
        T gr[NG];
        int idx = 0;
#pragma unroll
        for (int a = 0; a < Q; ++a)
#pragma unroll
            for (int b = a; b < Q; ++b) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc = tfma(Z[a][r], Z[b][r], acc);
                gr[idx++] = acc;
            }
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(Z[a][r], rv[r], acc);
            gr[idx++] = acc;
        }
        group_allreduce(grp, gr);
        idx = 0;
#pragma unroll
        for (int a = 0; a < Q; ++a)
#pragma unroll
            for (int b = a; b < Q; ++b) {
                Gm[a][b] = gr[idx];
                Gm[b][a] = gr[idx];
                ++idx;
            }
#pragma unroll
        for (int a = 0; a < Q; ++a) bz[a] = gr[idx++];
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            acnorm[a] = usqrt(Gm[a][a]);
            rdiag[a] = acnorm[a];
            wa[a] = acnorm[a];
            ipvt[a] = a;
        }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int k = 0; k < Q; ++k) Rj[i][k] = T(0);
    static_for<0, Q>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int prow = ROW0 + j;
        constexpr int NREM = Q - j; // columns j .. Q-1
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
                if constexpr (j == 0) { // the Gram bookkeeping follows the columns (only step 0 uses it)
#pragma unroll
                    for (int a = 0; a < Q; ++a) {
                        const T tmp = Gm[a][0];
                        Gm[a][0] = Gm[a][k];
                        Gm[a][k] = tmp;
                    }
#pragma unroll
                    for (int a = 0; a < Q; ++a) {
                        const T tmp = Gm[0][a];
                        Gm[0][a] = Gm[k][a];
                        Gm[k][a] = tmp;
                    }
                    const T tb = bz[0];
                    bz[0] = bz[k];
                    bz[k] = tb;
                }
            }
        }
        // raw dot products of the pivot column a (rows >= prow) with itself, the remaining columns and rv
        T dz[NREM], dr;
        if constexpr (j == 0) {
#pragma unroll
            for (int k = 0; k < NREM; ++k) dz[k] = Gm[0][k];
            dr = bz[0];
        } else {
            // rows ROW0 .. prow-1 of the pivot column hold R entries that were already extracted: clear them
#pragma unroll
            for (int r = 0; r < L::VW && r < R; ++r) {
                const int i = L::row_of(r, lane);
                Z[j][r] = (i >= prow) ? Z[j][r] : T(0);
            }
            T w[NREM + 1];
#pragma unroll
            for (int k = j; k < Q; ++k) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], Z[k][r], acc);
                w[k - j] = acc;
            }
            {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], rv[r], acc);
                w[NREM] = acc;
            }
            group_allreduce(grp, w);
#pragma unroll
            for (int k = 0; k < NREM; ++k) dz[k] = w[k];
            dr = w[NREM];
        }
        // pivot-row entries of the same columns and of rv: one broadcast
        T top[NREM + 1];
#pragma unroll
        for (int k = j; k < Q; ++k) top[k - j] = Z[k][L::reg_of_row(prow)];
        top[NREM] = rv[L::reg_of_row(prow)];
        group_bcast<NREM + 1>(grp, top, L::lane_of_row(prow));
        T ajnorm = usqrt(dz[0]);
        // (a column below norm2_min counts as a zero column: MINPACK normalises the pivot column before it forms the
        // reflector; the unnormalised form here would overflow in 1/(ajnorm v_p))
        if (uni(dz[0] <= num<T>::norm2_min)) {
            rdiag[j] = T(0);
            // remaining columns untouched: their row-prow entries are the R entries
#pragma unroll
            for (int k = j + 1; k < Q; ++k) Rj[j][k] = top[k - j];
            qtf[j] = top[NREM];
            return;
        }
        const T piv = top[0];
        if (piv < T(0)) ajnorm = -ajnorm;
        const T vp = piv + ajnorm;           // v'_p
        const T gj = -frcp(ajnorm * vp);     // H = I + gj v' v'^T
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r) {
            const int i = L::row_of(r, lane);
            Z[j][r] = (i == prow) ? vp : Z[j][r];
        }
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            const T f = gj * tfma(ajnorm, top[k - j], dz[k - j]); // gj * v'^T z_k
#pragma unroll
            for (int r = 0; r < R; ++r) Z[k][r] = tfma(f, Z[j][r], Z[k][r]);
            const T akj = tfma(f, vp, top[k - j]); // row prow of the updated column
            Rj[j][k] = akj;
            if (uni(rdiag[k] != T(0))) {
                const T tq = akj * frcp(rdiag[k]);
                rdiag[k] = rdiag[k] * usqrt(tmax(T(0), T(1) - tq * tq));
                const T rr = rdiag[k] * frcp(wa[k]);
                if (uni(T(0.05) * (rr * rr) <= num<T>::eps)) {
                    T s2 = T(0);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const T v = (r >= L::VW || L::row_of(r, lane) > prow) ? Z[k][r] : T(0);
                        s2 = tfma(v, v, s2);
                    }
                    rdiag[k] = tsqrt(group_sum(grp, s2));
                    wa[k] = rdiag[k];
                }
            }
        }
        {
            const T f = gj * tfma(ajnorm, top[NREM], dr);
#pragma unroll
            for (int r = 0; r < R; ++r) rv[r] = tfma(f, Z[j][r], rv[r]);
            qtf[j] = tfma(f, vp, top[NREM]);
        }
        rdiag[j] = -ajnorm;
    });
#pragma unroll
    for (int j = 0; j < Q; ++j) Rj[j][j] = rdiag[j];
}

// jac_qrfac_scaled for TWO columns with the pivot order as a compile-time constant (round 5): the generic routine brings the
// pivot column into position 0 by exchanging the register columns (2 R moves per exchange, taken by half of all accepted
// steps); here the two orders are two instantiations of the same statements and the choice is one scalar branch.  Same
// operations on the same operands in the same order as the generic routine: results bit-identical.
#ifndef VP_JAC_Q2_NOSWAP
#define VP_JAC_Q2_NOSWAP 1
#endif
template <typename T, int R, int ROW0, bool SWAP, class G>
__device__ __forceinline__ void jac_q2_ordered(T (&Z)[2][R], T (&rv)[R], const T (&s_in)[2], const T (&Gm)[2][2], const T (&bz)[2],
                                               T (&Rj)[2][2], int (&ipvt)[2], T (&qtf)[2], G &grp) {
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
    constexpr int PC = SWAP ? 1 : 0, OC = SWAP ? 0 : 1; // pivot column, other column
    ipvt[0] = PC;
    ipvt[1] = OC;
    const T sc0 = s_in[PC], sc1 = s_in[OC];
    T rdiag0, rdiag1;
    Rj[1][0] = T(0);
    { // ---- step 0: raw dots from the Gram round
        constexpr int prow = ROW0;
        T top[3] = {Z[PC][L::reg_of_row(prow)], Z[OC][L::reg_of_row(prow)], rv[L::reg_of_row(prow)]};
        group_bcast<3>(grp, top, L::lane_of_row(prow));
        const T dz0 = Gm[PC][PC], dz1 = Gm[PC][OC], dr = bz[PC];
        T an = usqrt(dz0);
        if (uni(tabs(sc0) * an == T(0) || dz0 <= num<T>::norm2_min)) {
            rdiag0 = T(0);
            Rj[0][1] = sc1 * top[1];
            qtf[0] = top[2];
        } else {
            const T piv = top[0];
            if (piv < T(0)) an = -an;
            const T vp = piv + an;
            const T gj = -frcp(an * vp);
#pragma unroll
            for (int r = 0; r < L::VW && r < R; ++r) {
                const int i = L::row_of(r, lane);
                Z[PC][r] = (i == prow) ? vp : Z[PC][r];
            }
            {
                const T f = gj * tfma(an, top[1], dz1);
#pragma unroll
                for (int r = 0; r < R; ++r) Z[OC][r] = tfma(f, Z[PC][r], Z[OC][r]);
                Rj[0][1] = sc1 * tfma(f, vp, top[1]);
            }
            {
                const T f = gj * tfma(an, top[2], dr);
#pragma unroll
                for (int r = 0; r < R; ++r) rv[r] = tfma(f, Z[PC][r], rv[r]);
                qtf[0] = tfma(f, vp, top[2]);
            }
            rdiag0 = -an;
        }
    }
    { // ---- step 1: one reduction round (the remaining column's norm below the pivot row and its dot with rv)
        constexpr int prow = ROW0 + 1;
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r) {
            const int i = L::row_of(r, lane);
            Z[OC][r] = (i >= prow) ? Z[OC][r] : T(0);
        }
        T w[2];
        {
            T a0 = T(0), a1 = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) a0 = tfma(Z[OC][r], Z[OC][r], a0);
#pragma unroll
            for (int r = 0; r < R; ++r) a1 = tfma(Z[OC][r], rv[r], a1);
            w[0] = a0;
            w[1] = a1;
        }
        group_allreduce(grp, w);
        T top[2] = {Z[OC][L::reg_of_row(prow)], rv[L::reg_of_row(prow)]};
        group_bcast<2>(grp, top, L::lane_of_row(prow));
        T an = usqrt(w[0]);
        if (uni(tabs(sc1) * an == T(0) || w[0] <= num<T>::norm2_min)) {
            rdiag1 = T(0);
            qtf[1] = top[1];
        } else {
            const T piv = top[0];
            if (piv < T(0)) an = -an;
            const T vp = piv + an;
            const T gj = -frcp(an * vp);
            const T f = gj * tfma(an, top[1], w[1]);
            // (rv is not used after this routine: only its pivot-row entry is formed)
            qtf[1] = tfma(f, vp, top[1]);
            rdiag1 = -an;
        }
    }
    Rj[0][0] = sc0 * rdiag0;
    Rj[1][1] = sc1 * rdiag1;
}

// The same factorisation for a Jacobian given as COLUMN-SCALED columns  z_k = s_k g_k  (Kaufman: z_k = -c_k Q^T D_k for
// models whose pair p is (basis p, parameter p)):  the reflectors depend only on the directions g_k, so all vector
// work runs on the unscaled columns G in place -- no scaling pass over the q columns -- and the scales enter as wave-
// uniform factors:  R_Z = R_G diag(s_perm),  acnorm_k = |s_k| ||g_k||,  pivoting by |s_k| * (downdated norm of g_k),
// qtf unchanged.  Rows < ROW0 of G (the range(Q) part, not in P_perp) are cleared here.  MINPACK's "zero pivot column"
// branch is taken on the SCALED norm (s_k == 0 makes z_k a zero column whatever g_k is).
template <typename T, int R, int Q, int ROW0, class G>
__device__ __forceinline__ void jac_qrfac_scaled(T (&Z)[Q][R], T (&rv)[R], const T (&s_in)[Q], T (&Rj)[Q][Q],
                                                 T (&acnorm)[Q], int (&ipvt)[Q], T (&qtf)[Q], G &grp) {
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
    T rdiag[Q], wa[Q], sa[Q], sc[Q]; // rdiag/wa: norms of the UNSCALED columns; sa = |s|; sc = s (follow the columns)
    T Gm[Q][Q], bz[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k)
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r) Z[k][r] = (L::row_of(r, lane) < ROW0) ? T(0) : Z[k][r];
    {
        constexpr int NG = Q * (Q + 1) / 2 + Q;
        T gr[NG];
        int idx = 0;
#pragma unroll
        for (int a = 0; a < Q; ++a)
#pragma unroll
            for (int b = a; b < Q; ++b) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc = tfma(Z[a][r], Z[b][r], acc);
                gr[idx++] = acc;
            }
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(Z[a][r], rv[r], acc);
            gr[idx++] = acc;
        }
        group_allreduce(grp, gr);
        idx = 0;
#pragma unroll
        for (int a = 0; a < Q; ++a)
#pragma unroll
            for (int b = a; b < Q; ++b) {
                Gm[a][b] = gr[idx];
                Gm[b][a] = gr[idx];
                ++idx;
            }
#pragma unroll
        for (int a = 0; a < Q; ++a) bz[a] = gr[idx++];
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            sc[a] = s_in[a];
            sa[a] = tabs(s_in[a]);
            rdiag[a] = usqrt(Gm[a][a]);
            wa[a] = rdiag[a];
            acnorm[a] = sa[a] * rdiag[a];
            ipvt[a] = a;
        }
    }
    if constexpr (Q == 2 && VP_JAC_Q2_NOSWAP) {
        // (acnorm is already set, in model order)
        if (uni(sa[1] * rdiag[1] > sa[0] * rdiag[0])) jac_q2_ordered<T, R, ROW0, true, G>(Z, rv, s_in, Gm, bz, Rj, ipvt, qtf, grp);
        else jac_q2_ordered<T, R, ROW0, false, G>(Z, rv, s_in, Gm, bz, Rj, ipvt, qtf, grp);
        return;
    }
#pragma unroll
    for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int k = 0; k < Q; ++k) Rj[i][k] = T(0);
    static_for<0, Q>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int prow = ROW0 + j;
        constexpr int NREM = Q - j; // columns j .. Q-1
        // bring the column of largest (downdated, SCALED) norm into the pivot position
        int kmax = j;
#pragma unroll
        for (int k = j + 1; k < Q; ++k)
            if (dyn_get<Q>(sa, k) * dyn_get<Q>(rdiag, k) > dyn_get<Q>(sa, kmax) * dyn_get<Q>(rdiag, kmax)) kmax = k;
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
#pragma unroll
                for (int i = 0; i < j; ++i) {
                    const T tmp = Rj[i][j];
                    Rj[i][j] = Rj[i][k];
                    Rj[i][k] = tmp;
                }
                T tv = rdiag[j];
                rdiag[j] = rdiag[k];
                rdiag[k] = tv;
                tv = wa[j];
                wa[j] = wa[k];
                wa[k] = tv;
                tv = sa[j];
                sa[j] = sa[k];
                sa[k] = tv;
                tv = sc[j];
                sc[j] = sc[k];
                sc[k] = tv;
                const int ti = ipvt[j];
                ipvt[j] = ipvt[k];
                ipvt[k] = ti;
                if constexpr (j == 0) { // the Gram bookkeeping follows the columns (only step 0 uses it)
#pragma unroll
                    for (int a = 0; a < Q; ++a) {
                        const T tmp = Gm[a][0];
                        Gm[a][0] = Gm[a][k];
                        Gm[a][k] = tmp;
                    }
#pragma unroll
                    for (int a = 0; a < Q; ++a) {
                        const T tmp = Gm[0][a];
                        Gm[0][a] = Gm[k][a];
                        Gm[k][a] = tmp;
                    }
                    const T tb = bz[0];
                    bz[0] = bz[k];
                    bz[k] = tb;
                }
            }
        }
        // raw dot products of the pivot column g (rows >= prow) with itself, the remaining columns and rv
        T dz[NREM], dr;
        if constexpr (j == 0) {
#pragma unroll
            for (int k = 0; k < NREM; ++k) dz[k] = Gm[0][k];
            dr = bz[0];
        } else {
#pragma unroll
            for (int r = 0; r < L::VW && r < R; ++r) {
                const int i = L::row_of(r, lane);
                Z[j][r] = (i >= prow) ? Z[j][r] : T(0);
            }
            T w[NREM + 1];
#pragma unroll
            for (int k = j; k < Q; ++k) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], Z[k][r], acc);
                w[k - j] = acc;
            }
            {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) acc = tfma(Z[j][r], rv[r], acc);
                w[NREM] = acc;
            }
            group_allreduce(grp, w);
#pragma unroll
            for (int k = 0; k < NREM; ++k) dz[k] = w[k];
            dr = w[NREM];
        }
        T top[NREM + 1];
#pragma unroll
        for (int k = j; k < Q; ++k) top[k - j] = Z[k][L::reg_of_row(prow)];
        top[NREM] = rv[L::reg_of_row(prow)];
        group_bcast<NREM + 1>(grp, top, L::lane_of_row(prow));
        T an = usqrt(dz[0]); // norm of the unscaled pivot column
        if (uni(sa[j] * an == T(0) || dz[0] <= num<T>::norm2_min)) {
            // zero (scaled) pivot column: no reflector; the remaining columns' row-prow entries are the R entries
            rdiag[j] = T(0);
#pragma unroll
            for (int k = j + 1; k < Q; ++k) Rj[j][k] = sc[k] * top[k - j];
            qtf[j] = top[NREM];
            return;
        }
        const T piv = top[0];
        if (piv < T(0)) an = -an;
        const T vp = piv + an;         // v'_p of the unscaled column
        const T gj = -frcp(an * vp);   // H = I + gj v' v'^T
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r) {
            const int i = L::row_of(r, lane);
            Z[j][r] = (i == prow) ? vp : Z[j][r];
        }
#pragma unroll
        for (int k = j + 1; k < Q; ++k) {
            const T f = gj * tfma(an, top[k - j], dz[k - j]); // gj * v'^T g_k
#pragma unroll
            for (int r = 0; r < R; ++r) Z[k][r] = tfma(f, Z[j][r], Z[k][r]);
            const T akj = tfma(f, vp, top[k - j]); // row prow of the updated (unscaled) column
            Rj[j][k] = sc[k] * akj;
            // MINPACK downdates the partial column norms only to choose LATER pivots: when a single column remains after
            // this step there is no choice left (its rdiag is overwritten by its exact norm in the next step)
            if constexpr (j < Q - 2)
            if (uni(rdiag[k] != T(0))) {
                const T tq = akj * frcp(rdiag[k]);
                rdiag[k] = rdiag[k] * usqrt(tmax(T(0), T(1) - tq * tq));
                const T rr = rdiag[k] * frcp(wa[k]);
                if (uni(T(0.05) * (rr * rr) <= num<T>::eps)) {
                    T s2 = T(0);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const T v = (r >= L::VW || L::row_of(r, lane) > prow) ? Z[k][r] : T(0);
                        s2 = tfma(v, v, s2);
                    }
                    rdiag[k] = tsqrt(group_sum(grp, s2));
                    wa[k] = rdiag[k];
                }
            }
        }
        {
            const T f = gj * tfma(an, top[NREM], dr);
#pragma unroll
            for (int r = 0; r < R; ++r) rv[r] = tfma(f, Z[j][r], rv[r]);
            qtf[j] = tfma(f, vp, top[NREM]);
        }
        // MINPACK: rdiag_z = -ajnorm_z = -(s an); kept unscaled here for the downdating, scaled when R is assembled
        rdiag[j] = -an;
    });
#pragma unroll
    for (int j = 0; j < Q; ++j) Rj[j][j] = sc[j] * rdiag[j];
}

// ---- a Jacobian that is not representable column by column: flag and re-fit (round 6) -------------------------------
// The fit kernels carry the UNSCALED derivative columns dPhi_k through the sweep and let the coefficient c_k enter the
// Jacobian QR as a wave-uniform factor (jac_qrfac_scaled).  The reference forms D_k c first and projects afterwards
// (src/solvers/levmar/mod.rs:156-171), so a trial point at which a basis column is huge and its coefficient tiny -- a decay
// time stepping through zero: exp(+t/0.035) = 1e153, c = 1e-152 -- is an ordinary evaluation there (D_k c = 1e5), while
// here the dot product of the 1e153 column with its 1e157 derivative column overflows (one problem in the 524 288 of
// BASELINE configs[3]).  Round 5 redid such a Jacobian out of line INSIDE the hot kernels (rescue_jacobian: 29 more spilled
// VGPRs in every wave of the headline kernel for a 2e-6 event, and only in the full-length unweighted static kernels).
// Round 6: every fit kernel -- any length, weights, run-time descriptors, streamed rows -- only TESTS the column norms of
// the Jacobian factor it just made; a problem whose factor is not finite after an evaluation that was itself ok ends its
// fit, appends its index to the handle's rescue list (rescue_push, vp_kernels.hpp) and leaves its initial guess in place.
// vp_fit then re-fits the listed problems from alpha0 in a second, tiny launch of the generic kernel with every huge basis
// column and its derivative columns scaled by the same power of two (gen::evaluate, vp_generic.hpp: c_k D_k is unchanged,
// power-of-two factors are exact, nothing overflows).  A fit is a pure function of its inputs: every problem that is not
// flagged is bit for bit what it was.
template <typename T, int Q> __device__ __forceinline__ bool jac_not_finite(const T (&acnorm)[Q]) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < Q; ++k) bad = bad || !is_finite(acnorm[k]);
    return bad;
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
    int grid_uniform; // the handle's grids passed grid_check_kernel -> exp recurrence allowed
    int32_t *rescue;  // flag-and-refit list (LaunchParams::rescue) or null
    int rescue_slot;
};

// Wave-uniform LM state.  It is PARKED in LDS while the fused QR sweep runs (lane 0 writes, every
// lane re-reads afterwards): during the sweep -- the register-pressure peak, 2R VGPRs per column --
// only the trial parameters stay in registers.
template <typename T, int N, int Q> struct LmState {
    T x[Q], diag[Q], qtf[Q], acnorm[Q], cbest[N];
    T Rj[Q][Q];
    T fnorm, delta, par, xnorm, gnorm, pnorm, prered, dirder, objective;
    int ipvt[Q];
    int flags; // bit0 first, bit1 first_tr, bit2 first_update
    int nfev;
};

// W = waves per problem (the workgroup is one group): W > 1 spreads the rows of one problem over W*64 lanes.
// Every wave of a group runs the identical wave-uniform LM bookkeeping on bit-identical inputs (the group
// all-reduce delivers the same totals to all waves), so no LM state is ever exchanged between waves.
// PADM (unit weights only): 0 = general m; 1 = m == 64*R*W, no padding rows at all; 2 = only the last register pair
// can hold padding rows -- the row-validity masks (32 SGPRs of hoisted lane masks, two selects per element) disappear
// from the column build entirely or from all pairs but the last
// RESCUE (round 6): the launch that re-fits the problems the fit kernels FLAGGED (jac_not_finite above) -- workgroup i takes
// problem list[2 + i], i < list[slot] -- with every derivative column built as 2^-ks times its value, ks = the binary exponent
// of its basis column's largest entry (from the two ends of the grid), and the coefficient entering the Jacobian QR as
// c_k 2^ks: c_k D_k is unchanged, the basis part (R, c, the residual) is bit for bit the unscaled evaluation, nothing
// overflows.  One wavefront(-group) per problem at the lone-wave rate (~4 us per evaluation): a flagged problem must not
// outlast the batch it came from.  Models with a trailing constant and diagonal pairs (MultiExpModel<.., true>); every
// other flagged problem goes to the generic kernel (vp_api.hip rescue_refit).
template <typename T, class M> inline constexpr bool fit_rescue_v = M::kStatic && M::kConstLast && M::kDiagonalPairs;

// The fit of ONE problem by one wavefront(-group): the body of fit_kernel, also the exit path of the slot kernel's self-rescue
// (vp_fit2.hpp: a wave re-fits the problems IT flagged before it ends -- no second launch).  LDS: s_t / s_y / s_w = zero-padded
// row-order copies of grid, data and weights (MP = 64 R W values each), xch = the group's exchange area, st = this wave's
// parked LM state.  stage_grid = false: s_t already holds the problem's grid (the slot kernel's shared grid).
// Returns true when the fit ended on a Jacobian factor that is not finite after an evaluation that was ok (jac_not_finite)
// and RESCUE is off -- the caller re-fits with RESCUE on (SELF) or the problem is on the handle's list (a.rescue).
//   SELF: the caller re-fits a flagged problem itself: it is neither pushed to the list nor are its parameters stored
template <typename T, class M, int R, int W, bool WEIGHTED, int PADM, bool RESCUE, bool SELF>
__device__ __forceinline__ bool fit_problem(const FitArgs<T, M> &a, const int64_t b, T *s_t, T *s_y, T *s_w, unsigned char *xch,
                                            LmState<T, M::N, M::Q> *st, const bool stage_grid) {
    static_assert(!(WEIGHTED && PADM != 0), "PADM is a unit-weight specialisation");
    static_assert(!RESCUE || fit_rescue_v<T, M>, "the scaled re-fit needs the constant-first sweep and diagonal pairs");
    constexpr int N = M::N, P = M::P, Q = M::Q;
    // CF: the constant column leads the factorisation and is never materialised (evaluate_core_const_first)
    constexpr bool CF = M::kConstLast;
    constexpr int NC = CF ? N + P : N + 1 + P; // register columns
    constexpr int YC = CF ? N - 1 : N;         // data column
    constexpr int DC = YC + 1;                 // first derivative column
    constexpr int MP = 64 * R * W;
    using G = Grp<W>;
    G grp = G::make(xch);
    const int lane = grp.gl; // group lane: row ownership; LDS park / result writes use grp.lane / grp.wave
    const int m = a.m;

    // stage the problem's grid, weights and weighted data in LDS (row order, padding rows zero)
    {
        T tmp[R];
        if (stage_grid) {
            const T *tp = a.t + b * a.t_stride;
            load_rows<T, R, W>(tp, m, lane, vec_aligned<T>(tp, m), tmp);
            store_rows<T, R, W>(s_t, MP, lane, true, tmp);
        }
        if constexpr (WEIGHTED) {
            const T *wp = a.w + b * a.w_stride;
            load_rows<T, R, W>(wp, m, lane, vec_aligned<T>(wp, m), tmp);
            store_rows<T, R, W>(s_w, MP, lane, true, tmp);
        }
        const T *yp = a.yw + b * (int64_t)m;
        load_rows<T, R, W>(yp, m, lane, vec_aligned<T>(yp, m), tmp);
        store_rows<T, R, W>(s_y, MP, lane, true, tmp);
    }
    // orders the LDS writes before the reads below (every lane re-reads only its own rows; one wave: no workgroup barrier -- the
    // slot kernel's other waves are not here)
    if constexpr (W > 1) {
        __syncthreads();
    } else {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the LDS copies are zero-padded to MP rows; valid rows are still i < m (scale 0 beyond)
    using Src = RowSource<T, R, true, WEIGHTED ? 1 : 0, 1, W, true, PADM>;
    Src src;
    src.t = s_t;
    src.w = s_w;
    src.m = m;
    src.lane = lane;
    src.vec = true;
    src.set_uniform(a.grid_uniform != 0);
    ConstReflector<T> h0;
    if constexpr (CF) h0 = make_const_reflector<T, R, Src, G>(src, grp); // alpha-independent: once per fit

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
    bool flagged = false; // handed to the re-fit launch (jac_not_finite)
    const int max_fev = a.patience * (Q + 1);
    const int mres = m; // number of residuals (S == 1)
    int trow = 0;
    auto trace_row = [&](const T(&xx)[Q], T fn, T ratio) {
        if (a.trace && trow < a.trace_rows && lane == 0) { // group lane 0 only
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

#ifdef VP_FIT_CLOCKS
    SectionClock clk_, *clk = &clk_;
    clk_.start();
#else
    SectionClock *clk = nullptr;
#endif
    auto park = [&]() __attribute__((always_inline)) {
        if (grp.lane == 0) {
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                st->x[k] = x[k];
                st->diag[k] = diag[k];
                st->qtf[k] = qtf[k];
                st->acnorm[k] = acnorm[k];
                st->ipvt[k] = ipvt[k];
#pragma unroll
                for (int j = 0; j < Q; ++j) st->Rj[k][j] = Rj[k][j];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) st->cbest[k] = cbest[k];
            st->fnorm = fnorm;
            st->delta = delta;
            st->par = par;
            st->xnorm = xnorm;
            st->gnorm = gnorm;
            st->pnorm = pnorm;
            st->prered = prered;
            st->dirder = dirder;
            st->objective = objective;
            st->flags = (first ? 1 : 0) | (first_tr ? 2 : 0) | (first_update ? 4 : 0);
            st->nfev = nfev;
        }
    };
    auto unpark = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            x[k] = st->x[k];
            diag[k] = st->diag[k];
            qtf[k] = st->qtf[k];
            acnorm[k] = st->acnorm[k];
            ipvt[k] = uni(st->ipvt[k]);
#pragma unroll
            for (int j = 0; j < Q; ++j) Rj[k][j] = st->Rj[k][j];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) cbest[k] = st->cbest[k];
        fnorm = st->fnorm;
        delta = st->delta;
        par = st->par;
        xnorm = st->xnorm;
        gnorm = st->gnorm;
        pnorm = st->pnorm;
        prered = st->prered;
        dirder = st->dirder;
        objective = st->objective;
        {
            const int fl = uni(st->flags);
            first = (fl & 1) != 0;
            first_tr = (fl & 2) != 0;
            first_update = (fl & 4) != 0;
            nfev = uni(st->nfev);
        }
    };
    for (;;) {
        // ---- park the LM state in LDS for the duration of the sweep (each wave parks its own copy) ----
        park();
        asm volatile("" ::: "memory");

        // ================= evaluate the VarPro functional at xt =================
        T C[NC][R];
        EvalUniform<T, N> u;
        if constexpr (R >= 2) load_rows_lds<T, R, W>(s_y, lane, C[YC]);
        else load_rows<T, R, W>(s_y, MP, lane, true, C[YC]);
        VP_TICK(clk, 0);
        int ks[N]; // (RESCUE) binary exponents the derivative columns are scaled down by
#pragma unroll
        for (int j = 0; j < N; ++j) ks[j] = 0;
        if constexpr (RESCUE) {
            // largest exponent of exp(-t/tau_j) over the grid, from its two ends (grids are sorted; an unsorted one keeps ks = 0
            // and with it the unscaled result)
            const T t_first = s_t[0], t_last = s_t[m - 1];
#pragma unroll
            for (int j = 0; j < N - 1; ++j) {
                const T rt = T(1) / xt[j];
                const T e2 = tmax(-t_first * rt, -t_last * rt) * T(1.4426950408889634);
                ks[j] = uni((e2 > T(64) && e2 < T(1100)) ? (int)e2 : 0);
            }
            evaluate_core_const_first<T, M, R, NC, Src, G, false, false, true>(a.mdl, xt, src, a.eps, grp, h0, C, u, clk, T(0), ks);
        } else if constexpr (CF) evaluate_core_const_first<T, M, R, NC, Src, G>(a.mdl, xt, src, a.eps, grp, h0, C, u, clk);
        else evaluate_core<T, M, R, NC, Src, G>(a.mdl, xt, src, a.eps, grp, C, u, clk);
        VP_TICK(clk, 3);

        asm volatile("" ::: "memory");
        // ---- un-park ----
        unpark();

        const T fnorm1 = usqrt(u.fn2);
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
                for (int k = 0; k < Q; ++k) tmpv[k] = a.scale_diag ? diag[k] * x[k] : x[k];
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

        VP_TICK(clk, 4);
        if (need_jac) {
            // ================= Jacobian in Q-coordinates, pivoted QR, Q_J^T r =================
            residual_qcoords<T, R, N>(C[YC], u.e, grp);
            if constexpr (M::kDiagonalPairs) {
                // z_k = -c_k Q^T D_k: factor the unscaled columns in place, the coefficients enter as column scales
                T zs[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) zs[k] = RESCUE ? -tldexp(u.c[k], ks[k]) : -u.c[k];
                jac_qrfac_scaled<T, R, Q, N>(reinterpret_cast<T(&)[Q][R]>(C[DC]), C[YC], zs, Rj, acnorm, ipvt, qtf, grp);
            } else {
                T Zs[Q][R];
                jacobian_qcoords<T, M, R, NC, G, DC>(a.mdl, C, u.c, Zs, grp);
                jac_qrfac<T, R, Q, N>(Zs, C[YC], Rj, acnorm, ipvt, qtf, grp);
            }
            if (uni(jac_not_finite<T, Q>(acnorm))) { // (rare) flag and re-fit: see jac_not_finite
                term = VP_TERM_NUMERICAL;
                flagged = !RESCUE;
                break;
            }
            VP_TICK(clk, 5);
            // norm of the scaled gradient
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
        VP_TICK(clk, 6);
        par = lmpar_any<T, Q>(Rj, ipvt, diag, qtf, delta, par, step, pnorm);
        VP_TICK(clk, 7);
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
        VP_TICK(clk, 8);
    }
#ifdef VP_FIT_CLOCKS
    // section cycle sums of this fit overwrite the head of its trace record (diagnostic build only)
    if (a.trace && lane == 0 && a.trace_rows * (Q + 4) >= 12) {
        double *tr = a.trace + (size_t)b * a.trace_rows * (Q + 4);
        for (int k = 0; k < 12; ++k) tr[k] = (double)clk_.acc[k];
    }
#endif

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
    // lane 0 stores the (uniform) results one by one: a lane-indexed gather would turn x[] into a scratch array
    if (lane == 0) {
        if (flagged && (SELF || a.rescue != nullptr)) { // alpha[b] keeps the initial guess: the re-fit starts from it and overwrites every output
            if (!SELF) rescue_push(a.rescue, a.rescue_slot, b);
        } else {
#pragma unroll
            for (int k = 0; k < Q; ++k) a.alpha[b * Q + k] = x[k];
        }
        if (a.C_out) {
#pragma unroll
            for (int k = 0; k < N; ++k) a.C_out[b * N + a.mdl.out_index(k)] = cbest[k];
        }
    }
    return flagged;
}

template <typename T, class M, int R, int W, bool WEIGHTED, int PADM = 0, bool RESCUE = false>
__global__ void __launch_bounds__(64 * W, (model_waves_for<T, M, R, M::N + 1 + M::P>())) fit_kernel(const FitArgs<T, M> a) {
    constexpr int MP = 64 * R * W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s_t = reinterpret_cast<T *>(smem_raw);
    T *s_y = s_t + MP;
    T *s_w = WEIGHTED ? s_y + MP : nullptr;
    unsigned char *s_after = reinterpret_cast<unsigned char *>(s_y + MP + (WEIGHTED ? MP : 0));
    // exchange area first (8-byte aligned), then one LmState per wave
    LmState<T, M::N, M::Q> *st = reinterpret_cast<LmState<T, M::N, M::Q> *>(s_after + ((group_xch_bytes<W>() + 15) / 16) * 16) +
                                 (W > 1 ? (int)(threadIdx.x >> 6) : 0);
    int64_t b = blockIdx.x;
    if constexpr (RESCUE) {
        if (b >= (int64_t)uni(a.rescue[a.rescue_slot])) return;
        b = (int64_t)uni(a.rescue[2 + b]);
    } else {
        if (b >= a.B) return;
    }
    (void)fit_problem<T, M, R, W, WEIGHTED, PADM, RESCUE, false>(a, b, s_t, s_y, s_w, s_after, st, true);
}

// dynamic LDS of fit_kernel: zero-padded copies of the grid, the data column and (weighted problems) the weights, the group
// exchange area, one parked LM state per wave.  vp_batch_create checks it against the CU's 160 KiB (KernelEntry::fit_lds_w)
template <typename T, class M, int R, int W> constexpr size_t fit_lds_bytes(bool weighted) {
    return (size_t)(weighted ? 3 : 2) * 64 * R * W * sizeof(T) + ((group_xch_bytes<W>() + 15) / 16) * 16 +
           (size_t)W * sizeof(LmState<T, M::N, M::Q>);
}
template <typename T, class M, int R, int W = 1> int launch_fit(const LaunchParams &p) {
    FitArgs<T, M> a;
    if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
    sweep_invariant_first(a.mdl); // (run-time-descriptor models: constant columns first in the sweep, vp_model.hpp)
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
    a.grid_uniform = p.grid_uniform;
    a.rescue = p.rescue;
    a.rescue_slot = p.rescue_slot;
    if (a.B <= 0) return VP_ERR_OK;
    if (p.rescue_used) *p.rescue_used = 1;
    const size_t lds = fit_lds_bytes<T, M, R, W>(p.w != nullptr);
    if (p.w) hipLaunchKernelGGL((fit_kernel<T, M, R, W, true>), dim3((unsigned)a.B), dim3(64 * W), lds, p.stream, a);
    else if (p.m == 64 * R * W)
        hipLaunchKernelGGL((fit_kernel<T, M, R, W, false, 1>), dim3((unsigned)a.B), dim3(64 * W), lds, p.stream, a);
    else if (R > 2 && p.m > 64 * (R - 2) * W)
        hipLaunchKernelGGL((fit_kernel<T, M, R, W, false, 2>), dim3((unsigned)a.B), dim3(64 * W), lds, p.stream, a);
    else hipLaunchKernelGGL((fit_kernel<T, M, R, W, false>), dim3((unsigned)a.B), dim3(64 * W), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// the re-fit launch of the flagged problems (fit_kernel<..., RESCUE>): up to kFitRescueGrid of them, one workgroup each; entries
// beyond that (and weighted problems, and every model outside fit_rescue_v) are the generic kernel's (vp_api.hip rescue_refit)
template <typename T, class M, int R, int W = 1> int launch_fit_rescue(const LaunchParams &p) {
    if constexpr (!fit_rescue_v<T, M>) {
        return VP_ERR_UNSUPPORTED;
    } else {
        if (p.w || p.t_stride != 0 || !p.rescue) return VP_ERR_UNSUPPORTED;
        FitArgs<T, M> a;
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
        const size_t lds = fit_lds_bytes<T, M, R, W>(false);
        hipLaunchKernelGGL((fit_kernel<T, M, R, W, false, 0, true>), dim3(kFitRescueGrid), dim3(64 * W), lds, p.stream, a);
        return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
    }
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

template <typename T, class M, int R, int W>
__global__ void __launch_bounds__(64 * W) best_fit_kernel(const BestFitArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q, NC = N + 1 + P;
    const int lane = (int)threadIdx.x; // group lane
    const int64_t prob = blockIdx.x;
    if (prob >= a.nprob) return;
    const int64_t b = prob / a.S;
    const int m = a.m;
    T alpha[Q], c[N];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = a.C[prob * N + k];
    using Src = RowSource<T, R, false, 0, 0, W>;
    Src src;
    src.t = a.t + b * a.t_stride;
    src.w = nullptr;
    src.m = m;
    src.lane = lane;
    src.vec = false;
    T C[NC][R];
    build_columns<T, M, R, NC, Src>(a.mdl, alpha, src, C);
    T f[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) acc = tfma(C[j][r], c[j], acc);
        f[r] = acc;
    }
    T *op = a.out + prob * (int64_t)m;
    store_rows<T, R, W>(op, m, lane, vec_aligned<T>(op, m), f);
}

template <typename T, class M, int R, int W = 1> int launch_best_fit(const LaunchParams &p) {
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
    hipLaunchKernelGGL((best_fit_kernel<T, M, R, W>), dim3((unsigned)a.nprob), dim3(64 * W), 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
