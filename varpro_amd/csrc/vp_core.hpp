// vp_core.hpp -- register-resident Householder QR, linear solve, projected residual and Kaufman
// Jacobian of ONE separable problem owned by ONE wavefront.
//
// Reference semantics (src/solvers/levmar/mod.rs):
//   set_params :42-73   Phi_w = W Phi;  C = pinv_eps(Phi_w) Y_w;  R = Y_w - Phi_w C
//   jacobian   :101-201 J[:,k] = U (U^T D_k C) - D_k C = -P_perp (W dPhi/dalpha_k) C
// The reference factors Phi_w with a thin SVD; only the projector U U^T = Q Q^T and the
// minimum-norm solution enter, so a Householder QR gives identical C, R, J at full column rank
// (SURVEY.md fact 1).  The SVD's absolute singular-value threshold is honoured by a guarded slow
// path: if a cheap lower bound on sigma_min(R) does not clear epsilon, the n x n triangular
// factor is decomposed with a one-sided Jacobi SVD and the truncated solve is applied.
//
// One fused sweep does all of it:  [Phi_w | y_w | D_1 .. D_P]  --Householder-->  R, Q^T y, Q^T D
// (the dot products of a reflector with all remaining columns share ONE wave reduction round),
// so per evaluation there are exactly 2n reduction rounds, independent of q.
#pragma once
#include "vp_device.hpp"
#include "vp_model.hpp"

namespace vp {

// Householder QR of A (N columns) applied simultaneously to NX extra columns X.
//   ROW0: first row the factorisation acts on (0 for Phi; N for the Jacobian living in rows >= N)
// On return: A[k] holds reflector v_k (1 at row ROW0+k, 0 above); tau[k]; Rm = upper triangle
// (row-major Rm[i][j], i <= j); xt[x][k] = (Q^T X_x)[ROW0+k]; X holds Q^T X in all rows.
template <typename T, int R, int N, int NX, int ROW0>
__device__ __forceinline__ void house_qr(T (&A)[N][R], T (&X)[NX][R], T (&tau)[N], T (&Rm)[N][N], T (&xt)[NX][N],
                                         const int lane) {
    using L = Layout<R>;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int prow = ROW0 + k;
        // squared norm of the pivot column below the pivot
        T s = T(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const T v = (L::row_of(r, lane) > prow) ? A[k][r] : T(0);
            s = tfma(v, v, s);
        }
        const T xn2 = wave_sum(s);
        const T alpha = bcast_row<R>(A[k], prow);
        T beta = alpha, tk = T(0), scal = T(0);
        if (uni(xn2 != T(0))) {
            beta = -tcopysign(tsqrt(tfma(alpha, alpha, xn2)), alpha);
            tk = (beta - alpha) / beta;
            scal = T(1) / (alpha - beta);
        }
        tau[k] = tk;
        Rm[k][k] = beta;
        // v_k in place: rows > prow scaled, row prow = 1, rows < prow = 0 (makes the loops below mask-free)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = L::row_of(r, lane);
            A[k][r] = (i > prow) ? A[k][r] * scal : ((i == prow) ? T(1) : T(0));
        }
        // w_c = v^T c for every remaining column: ONE reduction round for all of them
        constexpr int NREM_MAX = (N - 1) + NX;
        T w[NREM_MAX > 0 ? NREM_MAX : 1];
#pragma unroll
        for (int c = 0; c < NREM_MAX; ++c) w[c] = T(0);
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(A[k][r], A[j][r], acc);
            w[j - k - 1] = acc;
        }
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(A[k][r], X[x][r], acc);
            w[(N - 1 - k) + x] = acc;
        }
        wave_allreduce(w);
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
            const T f = -tk * w[j - k - 1];
#pragma unroll
            for (int r = 0; r < R; ++r) A[j][r] = tfma(f, A[k][r], A[j][r]);
            Rm[k][j] = bcast_row<R>(A[j], prow);
        }
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            const T f = -tk * w[(N - 1 - k) + x];
#pragma unroll
            for (int r = 0; r < R; ++r) X[x][r] = tfma(f, A[k][r], X[x][r]);
            xt[x][k] = bcast_row<R>(X[x], prow);
        }
    }
}

// z <- Q z for NZ columns (Q = H_0 ... H_{N-1} from house_qr with the same ROW0)
template <typename T, int R, int N, int NZ>
__device__ __forceinline__ void apply_q(const T (&A)[N][R], const T (&tau)[N], T (&Z)[NZ][R]) {
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        T w[NZ];
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(A[k][r], Z[z][r], acc);
            w[z] = acc;
        }
        wave_allreduce(w);
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const T f = -tau[k] * w[z];
#pragma unroll
            for (int r = 0; r < R; ++r) Z[z][r] = tfma(f, A[k][r], Z[z][r]);
        }
    }
}

// Slow path of the linear solve: truncated SVD of the N x N triangular factor (absolute threshold
// eps, as nalgebra's SVD::solve at src/solvers/levmar/mod.rs:52-54).  All arithmetic wave-uniform.
// Returns the minimum-norm c and e = qty - Rm c (the part of the residual that lives in range(Q)).
template <typename T, int N>
__device__ __noinline__ void truncated_solve(const T (&Rm)[N][N], const T (&qty)[N], T eps, T (&c)[N], T (&e)[N]) {
    T W[N][N], V[N][N]; // W[col][row]
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            W[j][i] = (i <= j) ? Rm[i][j] : T(0);
            V[j][i] = (i == j) ? T(1) : T(0);
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                T a = 0, b = 0, g = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    a = tfma(W[p][i], W[p][i], a);
                    b = tfma(W[q][i], W[q][i], b);
                    g = tfma(W[p][i], W[q][i], g);
                }
                if (g != T(0) && tabs(g) > num<T>::eps * T(0.25) * tsqrt(a * b)) {
                    rotated = true;
                    const T zeta = (b - a) / (T(2) * g);
                    const T tt = tcopysign(T(1), zeta) / (tabs(zeta) + tsqrt(T(1) + zeta * zeta));
                    const T cs = T(1) / tsqrt(T(1) + tt * tt), sn = cs * tt;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        T x = W[p][i], y = W[q][i];
                        W[p][i] = cs * x - sn * y;
                        W[q][i] = sn * x + cs * y;
                        x = V[p][i];
                        y = V[q][i];
                        V[p][i] = cs * x - sn * y;
                        V[q][i] = sn * x + cs * y;
                    }
                }
            }
        if (!uni(rotated)) break;
    }
    // W[:,j] = sigma_j u_j ;  c = sum_j (sigma_j > eps) v_j (u_j^T qty) / sigma_j
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T s2 = 0, d = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            s2 = tfma(W[j][i], W[j][i], s2);
            d = tfma(W[j][i], qty[i], d);
        }
        const T sg = tsqrt(s2);
        const T coef = (sg > eps) ? d / s2 : T(0); // (u^T qty)/sigma = (W^T qty)/sigma^2
#pragma unroll
        for (int i = 0; i < N; ++i) c[i] = tfma(coef, V[j][i], c[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T acc = qty[i];
#pragma unroll
        for (int j = i; j < N; ++j) acc = tfma(-Rm[i][j], c[j], acc);
        e[i] = acc;
    }
}

// c = R^{-1} qty by back substitution, guarded by a rank test equivalent to the reference's
// "singular value <= eps" rule.  truncated != 0 => e (top part of the residual) is non-zero.
template <typename T, int N>
__device__ __forceinline__ void solve_coeffs(const T (&Rm)[N][N], const T (&qty)[N], T eps, T (&c)[N], T (&e)[N],
                                             bool &truncated) {
    // sigma_min(R) >= 1/||R^{-1}||_F ; compute R^{-1} column by column (upper triangular)
    bool zero_diag = false;
#pragma unroll
    for (int i = 0; i < N; ++i) zero_diag = zero_diag || (Rm[i][i] == T(0));
    T inv_f2 = T(0);
    if (!uni(zero_diag)) {
        T Ri[N][N]; // inverse, upper triangular
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                if (i > j) {
                    Ri[i][j] = T(0);
                    continue;
                }
                T acc = (i == j) ? T(1) : T(0);
#pragma unroll
                for (int l = i + 1; l <= j; ++l) acc = tfma(-Rm[i][l], Ri[l][j], acc);
                Ri[i][j] = acc / Rm[i][i];
                inv_f2 = tfma(Ri[i][j], Ri[i][j], inv_f2);
            }
        }
        // fast path: all singular values certainly above eps -> plain triangular solve
        if (uni(inv_f2 * eps * eps < T(1))) { // 1/||R^-1||_F > eps
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                T acc = qty[i];
#pragma unroll
                for (int j = i + 1; j < N; ++j) acc = tfma(-Rm[i][j], c[j], acc);
                c[i] = acc / Rm[i][i];
                e[i] = T(0);
            }
            truncated = false;
            return;
        }
    }
    truncated_solve<T, N>(Rm, qty, eps, c, e);
    truncated = true;
}

// Everything one evaluation produces that is wave-uniform.
template <typename T, int N> struct EvalUniform {
    T c[N];      // linear coefficients
    T e[N];      // range(Q)-part of the residual (non-zero only on the truncated path)
    T tau[N];    // Householder scalars
    T fn2;       // ||R||^2 (squared norm of the weighted residual)
    bool ok;     // all finite
};

// One full evaluation at `alpha`:
//   A  <- Householder vectors of Phi_w
//   X0 <- Q^T y_w  (rows >= N: the projected residual in Q-coordinates)
//   D  <- Q^T (W dPhi_p)  for every dependency pair p
template <typename T, class M, int R>
__device__ __forceinline__ void evaluate_core(const M &mdl, const T (&alpha)[M::Q], const T (&t)[R],
                                              const T (&scale)[R], const T (&yw)[R], T eps, const int lane,
                                              T (&A)[M::N][R], T (&X)[1 + M::P][R], EvalUniform<T, M::N> &u) {
    constexpr int N = M::N, P = M::P;
    using L = Layout<R>;
    {
        T D[P > 0 ? P : 1][R];
        build_columns<T, M, R>(mdl, alpha, t, scale, A, D);
#pragma unroll
        for (int r = 0; r < R; ++r) X[0][r] = yw[r];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < R; ++r) X[1 + p][r] = D[p][r];
    }
    T Rm[N][N], xt[1 + P][N];
    house_qr<T, R, N, 1 + P, 0>(A, X, u.tau, Rm, xt, lane);
    T qty[N];
#pragma unroll
    for (int k = 0; k < N; ++k) qty[k] = xt[0][k];
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, eps, u.c, u.e, truncated);
    // ||r||^2 = ||e||^2 + sum_{rows >= N} (Q^T y)^2
    T s = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const T v = (L::row_of(r, lane) >= N) ? X[0][r] : T(0);
        s = tfma(v, v, s);
    }
    T fn2 = wave_sum(s);
#pragma unroll
    for (int k = 0; k < N; ++k) fn2 = tfma(u.e[k], u.e[k], fn2);
    u.fn2 = fn2;
    bool ok = is_finite(fn2);
#pragma unroll
    for (int k = 0; k < N; ++k) ok = ok && is_finite(u.c[k]);
    u.ok = uni(ok);
}

// Projected residual in Q-coordinates: rows < N <- e, rows >= N keep Q^T y.  (in place on X0)
template <typename T, int R, int N>
__device__ __forceinline__ void residual_qcoords(T (&x0)[R], const T (&e)[N], const int lane) {
    using L = Layout<R>;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = L::row_of(r, lane);
        if (r < L::VW) { // pivot rows live in the first VW registers
            T v = x0[r];
#pragma unroll
            for (int k = 0; k < N; ++k) v = (i == k) ? e[k] : v;
            x0[r] = v;
        }
    }
}

// Kaufman Jacobian columns in Q-coordinates:  Z_k = -(sum over pairs p of param k) c_{basis(p)} (Q^T D_p),
// rows < N zeroed (that is the P_perp).  Z must not alias X.
template <typename T, class M, int R>
__device__ __forceinline__ void jacobian_qcoords(const M &mdl, const T (&X)[1 + M::P][R], const T (&c)[M::N],
                                                 T (&Z)[M::Q][R], const int lane) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    using L = Layout<R>;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
#pragma unroll
        for (int r = 0; r < R; ++r) Z[k][r] = T(0);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (mdl.pair_param(p) == k) {
                const T cj = -dyn_get<N>(c, mdl.pair_basis(p));
#pragma unroll
                for (int r = 0; r < R; ++r) Z[k][r] = tfma(cj, X[1 + p][r], Z[k][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r)
            if (L::row_of(r, lane) < N) Z[k][r] = T(0);
    }
}

} // namespace vp
