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
// All columns of one problem live in ONE register array  C[NC][R], NC = N + 1 + P:
//     C[0..N)      W Phi          (overwritten by the Householder vectors)
//     C[N]         y_w            (overwritten by Q^T y_w: rows >= N are the projected residual)
//     C[N+1+p]     W dPhi_pair_p  (overwritten by Q^T W dPhi_p)
// One fused sweep does everything: reflector k is applied to ALL remaining columns, and its norm and all its dot
// products come out of ONE wave reduction round (raw dots a_k^T a_j), so an evaluation costs N rounds regardless
// of Q.  The fit kernel of a model with a constant basis uses evaluate_core_const_first instead: the constant
// column leads the factorisation, its reflector is computed once per fit and the column is never materialised.
#pragma once
#include "vp_device.hpp"
#include "vp_model.hpp"

namespace vp {

// Householder QR of the first N columns of C applied simultaneously to columns N..NC-1.
//   ROW0: first row the factorisation acts on.
// Reflector k is kept UNNORMALISED:  H_k = I + g_k v_k v_k^T  with v_k = a_k[prow:] except
// v_k[prow] = alpha - beta, and g_k = 1/(beta (alpha - beta)) -- one reciprocal per reflector and no
// scaling pass over the column.
// On return: C[k] (k < N) holds v_k (0 above row ROW0+k); g[k]; Rm = upper triangle (Rm[i][j], i <= j);
// qty[k] = (Q^T C[N])[ROW0+k]; columns >= N hold Q^T (.) in all rows.
//   HAS_Y: column N is a data column whose top entries are returned in qty
template <int I, int E, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

template <typename T, int R, int N, int NC, int ROW0, bool HAS_Y = true, class G>
__device__ __forceinline__ void house_qr(T (&C)[NC][R], T (&g)[N], T (&Rm)[N][N], T (&qty)[N], G &grp) {
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
    static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int prow = ROW0 + k;
        constexpr int NREM = NC - k; // columns k .. NC-1 take part in round k
        // rows above the pivot of column k hold R entries that were already extracted: clear them, so that the
        // dot products below run over all registers without masks
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r) {
            const int i = L::row_of(r, lane);
            C[k][r] = (i >= prow) ? C[k][r] : T(0);
        }
        // ONE reduction round per reflector: the RAW dot products d_j = a_k^T a_j over rows >= prow, j = k..NC-1
        // (d_k is the squared norm).  With v = a_k - beta e_prow:  v^T a_j = d_j - beta a_j[prow].
        T d[NREM], top[NREM];
#pragma unroll
        for (int j = k; j < NC; ++j) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(C[k][r], C[j][r], acc);
            d[j - k] = acc;
            top[j - k] = C[j][L::reg_of_row(prow)]; // pivot-row entry a_j[prow] (valid in the owning lane)
        }
        group_allreduce(grp, d);
        group_bcast<NREM>(grp, top, L::lane_of_row(prow));
        // beta = -sign(alpha) sigma, sigma = ||a_k||;  u = alpha - beta = sign(alpha)(|alpha| + sigma);
        // g = 1/(beta u) = -1/(sigma (|alpha| + sigma)) = -rsqrt(d_k) / (|alpha| + sigma): one rsq, one rcp, no
        // division / sqrt expansion on the critical path.  A zero column (d_k == 0) leaves H = I; a non-finite
        // norm is passed on into R so that the evaluation is flagged.
        const T alpha = top[0], nrm2 = d[0];
        const bool live = nrm2 > num<T>::norm2_min && is_finite(nrm2);
        const T y = live ? frsqrt(nrm2) : T(0);
        const T s0 = nrm2 * y;
        const T sigma = tfma(tfma(-s0, s0, nrm2), T(0.5) * y, s0);
        const T beta = live ? -tcopysign(sigma, alpha) : ((nrm2 <= num<T>::norm2_min) ? alpha : nrm2);
        const T u = live ? alpha - beta : T(0);
        const T gk = live ? -y * frcp(tabs(alpha) + sigma) : T(0);
        g[k] = gk;
        Rm[k][k] = beta;
        // v_k in place: row prow := u (rows above are already 0)
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r) {
            const int i = L::row_of(r, lane);
            C[k][r] = (i == prow) ? u : C[k][r];
        }
#pragma unroll
        for (int j = k + 1; j < NC; ++j) {
            const T f = gk * tfma(-beta, top[j - k], d[j - k]); // g * v^T a_j
#pragma unroll
            for (int r = 0; r < R; ++r) C[j][r] = tfma(f, C[k][r], C[j][r]);
            // row prow of the updated column (the same fma the vector update performs on that row):
            // R entries of the factor columns, Q^T y entry of the data column
            const T tj = tfma(f, u, top[j - k]);
            if (j < N) Rm[k][j] = tj;
            if (HAS_Y && j == N) qty[k] = tj;
        }
    });
}

// z <- Q z for NZ columns (Q = H_0 ... H_{N-1}; V = the first N columns left by house_qr)
template <typename T, int R, int N, int NC, int NZ, class G>
__device__ __forceinline__ void apply_q(const T (&V)[NC][R], const T (&g)[N], T (&Z)[NZ][R], G &grp) {
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        T w[NZ];
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(V[k][r], Z[z][r], acc);
            w[z] = acc;
        }
        group_allreduce(grp, w);
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const T f = g[k] * w[z];
#pragma unroll
            for (int r = 0; r < R; ++r) Z[z][r] = tfma(f, V[k][r], Z[z][r]);
        }
    }
}

// in-place variant: columns [C0, C1) of the unified array <- Q (.)   (C0 >= N)
template <typename T, int R, int N, int NC, int C0, int C1, class G>
__device__ __forceinline__ void apply_q_cols(T (&C)[NC][R], const T (&g)[N], G &grp) {
    constexpr int NZ = C1 - C0;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        T w[NZ];
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(C[k][r], C[C0 + z][r], acc);
            w[z] = acc;
        }
        group_allreduce(grp, w);
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const T f = g[k] * w[z];
#pragma unroll
            for (int r = 0; r < R; ++r) C[C0 + z][r] = tfma(f, C[k][r], C[C0 + z][r]);
        }
    }
}

// both at once: columns [C0, C1) of the unified array <- Q (.) and Z <- Q Z, ONE reduction round per reflector
template <typename T, int R, int N, int NC, int C0, int C1, int NZ, class G>
__device__ __forceinline__ void apply_q_cols_and_z(T (&C)[NC][R], const T (&g)[N], T (&Z)[NZ][R], G &grp) {
    constexpr int NCZ = C1 - C0, NW = NCZ + NZ;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        T w[NW];
#pragma unroll
        for (int z = 0; z < NW; ++z) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(C[k][r], z < NCZ ? C[C0 + (z < NCZ ? z : 0)][r] : Z[z >= NCZ ? z - NCZ : 0][r], acc);
            w[z] = acc;
        }
        group_allreduce(grp, w);
#pragma unroll
        for (int z = 0; z < NCZ; ++z) {
            const T f = g[k] * w[z];
#pragma unroll
            for (int r = 0; r < R; ++r) C[C0 + z][r] = tfma(f, C[k][r], C[C0 + z][r]);
        }
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const T f = g[k] * w[NCZ + z];
#pragma unroll
            for (int r = 0; r < R; ++r) Z[z][r] = tfma(f, C[k][r], Z[z][r]);
        }
    }
}

// Slow path of the linear solve: truncated SVD of the N x N triangular factor (absolute threshold
// eps, as nalgebra's SVD::solve at src/solvers/levmar/mod.rs:52-54).  All arithmetic wave-uniform.
// Returns the minimum-norm c and e = qty - Rm c (the part of the residual that lives in range(Q)).
// Arguments and results travel BY VALUE: a by-reference signature would force the caller's R, Q^T y, c, e
// into scratch memory on every evaluation just for this rarely taken call.
template <typename T, int N> struct TruncIn {
    T Rm[N][N];
    T qty[N];
    T eps;
};
template <typename T, int N> struct TruncOut {
    T c[N];
    T e[N];
};
template <typename T, int N> __device__ __noinline__ TruncOut<T, N> truncated_solve(const TruncIn<T, N> in) {
    const T(&Rm)[N][N] = in.Rm;
    const T(&qty)[N] = in.qty;
    const T eps = in.eps;
    TruncOut<T, N> out;
    T(&c)[N] = out.c;
    T(&e)[N] = out.e;
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
                T a = 0, b = 0, gg = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    a = tfma(W[p][i], W[p][i], a);
                    b = tfma(W[q][i], W[q][i], b);
                    gg = tfma(W[p][i], W[q][i], gg);
                }
                if (gg != T(0) && tabs(gg) > num<T>::eps * T(0.25) * tsqrt(a * b)) {
                    rotated = true;
                    const T zeta = (b - a) / (T(2) * gg);
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
    return out;
}

// c = R^{-1} qty by back substitution, guarded by a rank test equivalent to the reference's
// "singular value <= eps" rule.  truncated => e (top part of the residual) is non-zero.
// Only N reciprocals (of the diagonal) are taken; everything else is multiply-add.
template <typename T, int N>
__device__ __forceinline__ void solve_coeffs(const T (&Rm)[N][N], const T (&qty)[N], T eps, T (&c)[N], T (&e)[N],
                                             bool &truncated) {
    bool zero_diag = false;
#pragma unroll
    for (int i = 0; i < N; ++i) zero_diag = zero_diag || (Rm[i][i] == T(0));
    if (!uni(zero_diag)) {
        // reciprocals of the diagonal (Newton-refined v_rcp, 1-2 ulp: c below is re-rounded by a residual step, and
        // R^-1 only feeds a bound) -- N IEEE division expansions less per evaluation
        T d[N];
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = frcp(Rm[i][i]);
        // sigma_min(R) >= 1/||R^{-1}||_F : R^{-1} column by column (upper triangular)
        T inv_f2 = T(0);
        T Ri[N][N];
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
                Ri[i][j] = acc * d[i];
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
                // (acc * d) corrected by one residual step == acc / R_ii to the last bit in practice
                const T q0 = acc * d[i];
                c[i] = tfma(tfma(-q0, Rm[i][i], acc), d[i], q0);
                e[i] = T(0);
            }
            truncated = false;
            return;
        }
    }
    {
        TruncIn<T, N> in;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            in.qty[i] = qty[i];
#pragma unroll
            for (int j = 0; j < N; ++j) in.Rm[i][j] = (j >= i) ? Rm[i][j] : T(0);
        }
        in.eps = eps;
        const TruncOut<T, N> out = truncated_solve<T, N>(in);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            c[i] = out.c[i];
            e[i] = out.e[i];
        }
    }
    truncated = true;
}

// Everything one evaluation produces that is wave-uniform.
template <typename T, int N> struct EvalUniform {
    T c[N];   // linear coefficients
    T e[N];   // range(Q)-part of the residual (non-zero only on the truncated path)
    T g[N];   // Householder scalars g_k = 1/(beta_k u_k)
    T fn2;    // ||R||^2 (squared norm of the weighted residual)
    bool ok;  // all finite
};

// One full evaluation at `alpha`: builds the columns (the data column C[N] must already hold y_w),
// runs the fused sweep, solves for c and forms ||r||^2.
//   NOD: no derivative columns (NC = N + 1: basis + data) -- phase 1 of the split evaluate kernels
template <typename T, class M, int R, int NC, class Src, class G, bool NOD = false>
__device__ __forceinline__ void evaluate_core(const M &mdl, const T (&alpha)[M::Q], const Src &src, T eps, G &grp,
                                              T (&C)[NC][R], EvalUniform<T, M::N> &u, SectionClock *clk = nullptr) {
    constexpr int N = M::N;
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
    static_assert(!NOD || NC == N + 1, "phase 1: basis + data");
    build_columns<T, M, R, NC, Src, N + 1, false, true, !NOD>(mdl, alpha, src, C);
    VP_TICK(clk, 1);
#ifndef VP_NO_SWEEP_FENCE
    // keep the scheduler from interleaving the tail of the column build with the first dot products: the
    // overlapping live ranges (grid values, unscaled exponentials) push a column into scratch otherwise
    __builtin_amdgcn_sched_barrier(0);
#endif
    T Rm[N][N], qty[N];
    house_qr<T, R, N, NC, 0, true, G>(C, u.g, Rm, qty, grp);
    VP_TICK(clk, 2);
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, eps, u.c, u.e, truncated);
    // ||r||^2 = ||e||^2 + sum_{rows >= N} (Q^T y)^2
    T s = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const T v = (r >= L::VW || L::row_of(r, lane) >= N) ? C[N][r] : T(0);
        s = tfma(v, v, s);
    }
    T fn2 = group_sum(grp, s);
#pragma unroll
    for (int k = 0; k < N; ++k) fn2 = tfma(u.e[k], u.e[k], fn2);
    u.fn2 = fn2;
    // a basis column that overflowed (inf / NaN norm) leaves a non-finite diagonal in R: the reference's SVD turns
    // that into NaN residuals (residuals() == None); c alone would not show it (1/inf = 0)
    bool ok = is_finite(fn2);
#pragma unroll
    for (int k = 0; k < N; ++k) ok = ok && is_finite(u.c[k]) && is_finite(Rm[k][k]);
    u.ok = uni(ok);
}

// ---- constant column first, never materialised -----------------------------------------------------------------
// For models whose LAST basis is the constant (MultiExpModel<NEXP, true>) the fit kernel factors Phi_w with the
// column order [scale | exp_1 .. exp_NEXP]: the first column is the row-scale vector s itself (1/0 validity mask or
// the weights), which does not depend on alpha.  Its reflector v_0 = s - beta_0 e_0 is computed ONCE PER FIT
// (ConstReflector) and applied implicitly -- d_j = sum_r s_r a_j[r], a_j += tau_j v_0 -- so the column is never
// held in registers: 2*NEXP+1 register columns instead of 2*NEXP+2, which is what lets the double-exponential
// sweep (6 x 32 VGPRs before) run without scratch spills.  QR of a column-permuted Phi spans the same range: c, r
// and J are the same up to rounding; c is returned in MODEL order, e in Q-coordinate ROW order.
template <typename T> struct ConstReflector {
    T beta, u, g; // H_0 = I + g v v^T, v = s - beta e_0, u = v[0]
    bool live;    // false: s == 0 (no valid row)
};

template <typename T, int R, class Src, class G>
__device__ __forceinline__ ConstReflector<T> make_const_reflector(const Src &src, G &grp) {
    using L = Layout<R, G::W>;
    constexpr int VW = L::VW;
    T acc = T(0), s0 = T(0);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            acc = tfma(sc[e], sc[e], acc);
            if (r0 + e == L::reg_of_row(0)) s0 = sc[e];
        }
    }
    const T nrm2 = group_sum(grp, acc);
    T top[1] = {s0};
    group_bcast<1>(grp, top, L::lane_of_row(0));
    ConstReflector<T> h;
    h.live = uni(nrm2 > num<T>::norm2_min && is_finite(nrm2));
    const T sigma = tsqrt(nrm2);
    h.beta = h.live ? -tcopysign(sigma, top[0]) : top[0];
    h.u = h.live ? top[0] - h.beta : T(0);
    h.g = h.live ? T(1) / (h.beta * h.u) : T(0);
    return h;
}

// Z <- H_0 Z for the implicit reflector of the scale column (H_0 is symmetric: the same call serves the forward sweep and
// the back-application).  One reduction round.
template <typename T, int R, int NZ, class Src, class G>
__device__ __forceinline__ void apply_const_reflector(T (&Z)[NZ][R], const ConstReflector<T> &h0, const Src &src, G &grp) {
    using L = Layout<R, G::W>;
    constexpr int VW = L::VW;
    const int lane = grp.gl;
    T d[NZ], top[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) d[j] = T(0);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int j = 0; j < NZ; ++j)
#pragma unroll
            for (int e = 0; e < VW; ++e) d[j] = tfma(sc[e], Z[j][r0 + e], d[j]);
    }
#pragma unroll
    for (int j = 0; j < NZ; ++j) top[j] = Z[j][L::reg_of_row(0)];
    group_allreduce(grp, d);
    group_bcast<NZ>(grp, top, L::lane_of_row(0));
    T tau[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) tau[j] = h0.g * tfma(-h0.beta, top[j], d[j]);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const T v = (r0 + e < VW && L::row_of(r0 + e, lane) == 0) ? h0.u : sc[e];
#pragma unroll
            for (int j = 0; j < NZ; ++j) Z[j][r0 + e] = tfma(tau[j], v, Z[j][r0 + e]);
        }
    }
}

// z <- Q^T z for NZ columns: the reflectors of apply_q in ascending order (V = the first N columns left by house_qr)
template <typename T, int R, int N, int NC, int NZ, class G>
__device__ __forceinline__ void apply_qt(const T (&V)[NC][R], const T (&g)[N], T (&Z)[NZ][R], G &grp) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T w[NZ];
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(V[k][r], Z[z][r], acc);
            w[z] = acc;
        }
        group_allreduce(grp, w);
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const T f = g[k] * w[z];
#pragma unroll
            for (int r = 0; r < R; ++r) Z[z][r] = tfma(f, V[k][r], Z[z][r]);
        }
    }
}

// C: [0, NEXP) exponential columns, [NEXP] data column (already loaded with y_w), [NEXP+1, 2 NEXP+1) derivative
// columns.  On return: columns in Q-coordinates (rows >= N of the data / derivative columns are what the LM uses).
//   YPRE: the data column arrives as H_0 y_w (the slot kernel applies the alpha-independent reflector once per fit,
//         vp_fit2.hpp) and skips reflector 0 here; qty0 = (H_0 y_w)[0]
//   NOD:  no derivative columns (NCX = N: exponentials + data) -- phase 1 of the split evaluate kernel
//   SHIFT: the derivative column of exponential j carries the factor 2^-ks[j] (build_columns); everything the basis
//          columns produce (R, c, the residual) is the unshifted evaluation's bit for bit
//   PRE:  1 / alpha_i and the recurrence ratios of the trial point come from the caller (build_columns)
template <typename T, class M, int R, int NCX, class Src, class G, bool YPRE = false, bool NOD = false, bool SHIFT = false, bool PRE = false>
__device__ __forceinline__ void evaluate_core_const_first(const M &mdl, const T (&alpha)[M::Q], const Src &src, T eps,
                                                          G &grp, const ConstReflector<T> &h0, T (&C)[NCX][R],
                                                          EvalUniform<T, M::N> &u, SectionClock *clk = nullptr,
                                                          const T qty0 = T(0), const int *ks = nullptr, const T *pre = nullptr) {
    constexpr int N = M::N, NE = M::N - 1;
    static_assert(M::kConstLast && NCX == M::N + (NOD ? 0 : M::P), "const-first sweep: N-1 exponentials + data + P derivatives");
    using L = Layout<R, G::W>;
    constexpr int VW = L::VW;
    const int lane = grp.gl;
    build_columns<T, M, R, NCX, Src, NE + 1, true, true, !NOD, -1, SHIFT, PRE>(mdl, alpha, src, C, ks, pre);
    VP_TICK(clk, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- reflector 0: the implicit scale column ----
    T d[NCX], top[NCX];
#pragma unroll
    for (int j = 0; j < NCX; ++j) d[j] = T(0);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int j = 0; j < NCX; ++j) {
            if (YPRE && j == NE) continue;
#pragma unroll
            for (int e = 0; e < VW; ++e) d[j] = tfma(sc[e], C[j][r0 + e], d[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NCX; ++j) top[j] = C[j][L::reg_of_row(0)];
    group_allreduce(grp, d);
    group_bcast<NCX>(grp, top, L::lane_of_row(0));
    T Rm[N][N], qty[N];
    T tau[NCX];
    // (opaque: 1 / beta_0 and its square in solve_coeffs are loop invariants of the caller's LM loop -- hoisted, they are two
    // more spilled registers whose reload is a VMEM wait in every evaluation)
#ifndef VP_OPAQUE_BETA0
#define VP_OPAQUE_BETA0 1
#endif
    Rm[0][0] = VP_OPAQUE_BETA0 ? dyn_opq(h0.beta) : h0.beta;
#pragma unroll
    for (int j = 0; j < NCX; ++j) {
        if (YPRE && j == NE) {
            tau[j] = T(0);
            qty[0] = qty0;
            continue;
        }
        tau[j] = h0.g * tfma(-h0.beta, top[j], d[j]);
        const T tj = tfma(tau[j], h0.u, top[j]);
        if (j < NE) Rm[0][1 + j] = tj;
        if (j == NE) qty[0] = tj;
    }
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += VW) {
        T tt[2], sc[2];
        src.get(r0, tt, sc);
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const T v = (r0 + e < VW && L::row_of(r0 + e, lane) == 0) ? h0.u : sc[e];
#pragma unroll
            for (int j = 0; j < NCX; ++j) {
                if (YPRE && j == NE) continue;
                C[j][r0 + e] = tfma(tau[j], v, C[j][r0 + e]);
            }
        }
    }
    // ---- reflectors 1..NE on the exponential columns, pivot rows 1..NE ----
    T g1[NE], R1[NE][NE], q1[NE];
    house_qr<T, R, NE, NCX, 1, true, G>(C, g1, R1, q1, grp);
    VP_TICK(clk, 2);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        Rm[1 + i][0] = T(0);
        qty[1 + i] = q1[i];
#pragma unroll
        for (int j = 0; j < NE; ++j) Rm[1 + i][1 + j] = R1[i][j];
    }
    T cp[N];
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, eps, cp, u.e, truncated);
#pragma unroll
    for (int j = 0; j < NE; ++j) u.c[j] = cp[1 + j];
    u.c[NE] = cp[0];
    u.g[0] = h0.g;
#pragma unroll
    for (int j = 0; j < NE; ++j) u.g[1 + j] = g1[j];
    // ||r||^2 = ||e||^2 + sum_{rows >= N} (Q^T y)^2
    T s = T(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const T v = (r >= VW || L::row_of(r, lane) >= N) ? C[NE][r] : T(0);
        s = tfma(v, v, s);
    }
    T fn2 = group_sum(grp, s);
#pragma unroll
    for (int k = 0; k < N; ++k) fn2 = tfma(u.e[k], u.e[k], fn2);
    u.fn2 = fn2;
    // a basis column that overflowed (inf / NaN norm) leaves a non-finite diagonal in R: the reference's SVD turns
    // that into NaN residuals (residuals() == None); c alone would not show it (1/inf = 0)
    bool ok = is_finite(fn2);
#pragma unroll
    for (int k = 0; k < N; ++k) ok = ok && is_finite(u.c[k]) && is_finite(Rm[k][k]);
    u.ok = uni(ok);
}

// Projected residual in Q-coordinates: rows < N <- e, rows >= N keep Q^T y.  (in place on the data column)
template <typename T, int R, int N, class G>
__device__ __forceinline__ void residual_qcoords(T (&x0)[R], const T (&e)[N], const G &grp) {
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
#pragma unroll
    for (int r = 0; r < L::VW && r < R; ++r) { // pivot rows live in the first VW registers
        const int i = L::row_of(r, lane);
        T v = x0[r];
#pragma unroll
        for (int k = 0; k < N; ++k) v = (i == k) ? e[k] : v;
        x0[r] = v;
    }
}

// Kaufman Jacobian columns in Q-coordinates:  Z_k = -(sum over pairs p of param k) c_{basis(p)} (Q^T D_p),
// rows < N zeroed (that is the P_perp).
//   diagonal models (pair p == (basis p, param p)): done IN PLACE, Z_k is C[N+1+k];
//   general models: written to the separate array Zs and the caller uses that.
template <typename T, class M, int R, int NC, class G, int DC = M::N + 1>
__device__ __forceinline__ void jacobian_qcoords(const M &mdl, T (&C)[NC][R], const T (&c)[M::N],
                                                 T (&Zs)[M::kDiagonalPairs ? 1 : M::Q][R], const G &grp) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    using L = Layout<R, G::W>;
    const int lane = grp.gl;
    if constexpr (M::kDiagonalPairs) {
        (void)Zs;
        (void)mdl;
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const T ck = -c[k];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool top = (r < L::VW) && (L::row_of(r, lane) < N);
                C[DC + k][r] = top ? T(0) : ck * C[DC + k][r];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
#pragma unroll
            for (int r = 0; r < R; ++r) Zs[k][r] = T(0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (mdl.pair_param(p) == k) {
                    const T cj = -dyn_get<N>(c, mdl.pair_basis(p));
#pragma unroll
                    for (int r = 0; r < R; ++r) Zs[k][r] = tfma(cj, C[DC + p][r], Zs[k][r]);
                }
            }
#pragma unroll
            for (int r = 0; r < L::VW && r < R; ++r)
                if (L::row_of(r, lane) < N) Zs[k][r] = T(0);
        }
    }
}

} // namespace vp
