// vp_stats.hpp -- batched post-fit statistics.
//
// == FitStatistics::try_calculate (src/statistics/mod.rs:352-441) for every problem of a single-RHS batch:
//      J = [Phi, (dPhi/dalpha_k c)_k]   (m x (n+q), UNweighted; model_function_jacobian :481-511)
//      H = W J;   sigma^2 = ||r_w||^2 / (m - n - q);   Cov = sigma^2 (H^T H)^{-1}
//      unscaled_confidence_sigma_i = sqrt(j_i^T Cov j_i)
// On the GPU: H is factored by the same register-resident Householder sweep as Phi (H = Q R, so
// (H^T H)^{-1} = R^{-1} R^{-T} without ever forming the normal equations), the n+q triangular inverse is
// wave-uniform arithmetic, and the confidence sigma is sigma * ||R^{-T} j_i|| row by row.
#pragma once
#include "vp_kernels.hpp"

namespace vp {

enum { VP_ST_STATS_FAILED = 4 }; // Underdetermined / MatrixInversion (src/statistics/mod.rs:15-25)

template <typename T, class M> struct StatsArgs {
    M mdl;
    const T *t;
    const T *w;
    const T *alpha;      // [B][q]
    const T *C;          // [B][n]
    const double *cost;  // [B]  1/2 ||r_w||^2
    const int32_t *status_in; // [B] status of the cached evaluation
    T *cov_out;          // [B][(n+q)^2]  column-major (symmetric)
    double *chi2_out;    // [B]
    T *sigma_out;        // [B][m] or null
    int32_t *status_out; // [B]
    int m;
    int64_t B;
    int64_t t_stride, w_stride;
};

template <typename T, class M, int R, int W>
__global__ void __launch_bounds__(64 * W) stats_kernel(const StatsArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q, K = N + Q, NB = N + P;
    __shared__ __attribute__((aligned(16))) unsigned char s_xch[group_xch_bytes<W>() > 0 ? group_xch_bytes<W>() : 16];
    using G = Grp<W>;
    using L = Layout<R, W>;
    G grp = G::make(s_xch);
    const int lane = grp.gl;
    const int64_t b = blockIdx.x;
    if (b >= a.B) return;
    const int m = a.m;
    T alpha[Q], c[N];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = a.C[b * N + k];
    const T *tp = a.t + b * a.t_stride;
    const T *wp = a.w ? a.w + b * a.w_stride : nullptr;

    auto build_j = [&](const T *weights, T(&Hc)[K][R]) {
        using Src = RowSource<T, R, false, 2, 0, W>;
        Src src;
        src.t = tp;
        src.w = weights;
        src.m = m;
        src.lane = lane;
        src.vec = false;
        T Cc[NB][R];
        build_columns<T, M, R, NB, Src, N>(a.mdl, alpha, src, Cc);
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) Hc[j][r] = Cc[j][r];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
#pragma unroll
            for (int r = 0; r < R; ++r) Hc[N + k][r] = T(0);
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (a.mdl.pair_param(p) == k) {
                    const T cj = dyn_get<N>(c, a.mdl.pair_basis(p));
#pragma unroll
                    for (int r = 0; r < R; ++r) Hc[N + k][r] = tfma(cj, Cc[N + p][r], Hc[N + k][r]);
                }
        }
    };

    // H = W J = Q R
    T Rm[K][K], gdummy[K], qdummy[K];
    {
        T H[K][R];
        build_j(wp, H);
        house_qr<T, R, K, K, 0, false, G>(H, gdummy, Rm, qdummy, grp);
    }
    const int dof = m - K;
    bool ok = (dof > 0) && (a.status_in[b] == VP_ST_OK);
#pragma unroll
    for (int i = 0; i < K; ++i) ok = ok && (Rm[i][i] != T(0)) && is_finite(Rm[i][i]);
    ok = uni(ok);
    T Ri[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) Ri[i][j] = T(0);
    const T chi2 = ok ? (T)(2.0 * a.cost[b] / (double)dof) : T(0) / T(0);
    if (ok) {
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                if (i > j) continue;
                T acc = (i == j) ? T(1) : T(0);
#pragma unroll
                for (int l = i + 1; l <= j; ++l) acc = tfma(-Rm[i][l], Ri[l][j], acc);
                Ri[i][j] = acc / Rm[i][i];
            }
    }
    // Cov = chi2 * R^{-1} R^{-T}  (wave-uniform arithmetic; group lane 0 stores)
    {
        const T nanv = T(0) / T(0);
#pragma unroll
        for (int bi = 0; bi < K; ++bi)
#pragma unroll
            for (int ai = 0; ai < K; ++ai) {
                T val = T(0);
#pragma unroll
                for (int l = (ai > bi ? ai : bi); l < K; ++l) val = tfma(Ri[ai][l], Ri[bi][l], val);
                if (lane == 0) a.cov_out[b * (K * K) + bi * K + ai] = ok ? val * chi2 : nanv;
            }
    }
    if (lane == 0) {
        a.chi2_out[b] = (double)chi2;
        a.status_out[b] = ok ? VP_ST_OK : VP_ST_STATS_FAILED;
    }
    if (a.sigma_out) {
        // sigma_i = sqrt(chi2) * || R^{-T} j_i ||  with the UNweighted rows j_i
        T Jc[K][R];
        build_j(nullptr, Jc);
        T sig[R];
        const T s0 = ok ? tsqrt(chi2) : T(0) / T(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T acc = T(0);
#pragma unroll
            for (int col = 0; col < K; ++col) {
                T v = T(0);
#pragma unroll
                for (int row = 0; row <= col; ++row) v = tfma(Ri[row][col], Jc[row][r], v);
                acc = tfma(v, v, acc);
            }
            sig[r] = s0 * tsqrt(acc);
        }
        T *op = a.sigma_out + b * (int64_t)m;
        store_rows<T, R, W>(op, m, lane, false, sig);
    }
    (void)sizeof(L);
}

template <typename T, class M, int R, int W = 1> int launch_stats(const LaunchParams &p) {
    StatsArgs<T, M> a;
    if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
    a.t = (const T *)p.t;
    a.w = (const T *)p.w;
    a.alpha = (const T *)p.alpha;
    a.C = (const T *)p.C_out;
    a.cost = p.cost_out;
    a.status_in = p.status;
    a.cov_out = (T *)p.Phi_out;
    a.chi2_out = (double *)p.dPhi_out;
    a.sigma_out = (T *)p.r_out;
    a.status_out = (int32_t *)p.J_out;
    a.m = p.m;
    a.B = p.B;
    a.t_stride = p.t_stride;
    a.w_stride = p.w_stride;
    if (a.B <= 0) return VP_ERR_OK;
    hipLaunchKernelGGL((stats_kernel<T, M, R, W>), dim3((unsigned)a.B), dim3(64 * W), 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
