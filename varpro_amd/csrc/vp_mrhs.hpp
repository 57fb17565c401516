// vp_mrhs.hpp -- multiple right-hand sides: one alpha shared by S data columns ("global fit").
//
// == SeparableProblemBuilder::mrhs + the S > 1 paths of set_params / residuals / jacobian / fit
//    (src/problem/builder.rs:194-225, src/solvers/levmar/mod.rs:42-73, 101-201 branch B :172-186).
//
// Phi depends on alpha only, so an evaluation splits into
//   FACTOR  (one wavefront per problem, negligible): Phi_w = Q R by the register-resident Householder sweep;
//           writes the explicit thin Q (m x n), R^{-1}, and  G_p = P_perp (W dPhi_p)  (m x P, original
//           coordinates) -- the reference's  A = U (U^T D_k) - D_k  of branch B, up to sign -- plus G^T G.
//   STREAM  (HBM-bound, one pass over Y): per column s, entirely in registers with Q and G staged in LDS:
//           T = Q^T y_s (one reduction round), c_s = R^{-1} T, r_s = y_s - Q T,
//           trait outputs:   R, C, and J_k[:, s] = - sum_{p in k} c_{j(p), s} G_p   (no reductions at all)
//           fit (reduced):   ||r_s||^2, u_s = G^T r_s -> accumulates  sum ||r||^2,  sum_s c_s c_s^T,  sum_s c_s o u_s,
//                            from which  J^T J = (sum c c^T) o (G^T G)  and  J^T r  follow WITHOUT ever
//                            materialising the (m S) x q Jacobian (768 MiB per evaluation at configs[2]).
//   LM STEP (one wavefront per problem): the MINPACK bookkeeping of vp_lm_core.hpp on J^T J / J^T r.
//
// Algorithmic HBM bytes per evaluation (T = 8): trait level  T m S (2 + q) + ... (read Y, write R and J);
// fit level  T m S  (read Y once).  Deviation from the reference, documented in DESIGN.md: the LM step uses
// the Gram matrix (normal equations of the q x q trust-region subproblem) instead of a QR of the tall J; the
// linear sub-problem itself stays Householder-based.  A rank-deficient Phi_w takes the reference's truncated-SVD
// branch here too: the factor kernel then stores the truncated pseudo-inverse R^+ in place of R^{-1} and the
// projector P_n = R R^+ onto the retained subspace, and the streaming kernel uses c = R^+ Q^T y,
// r = y - Q P_n Q^T y (the Jacobian keeps the full Q, like the reference keeps the full U).
#pragma once
#include "vp_lm_core.hpp"

namespace vp {

enum { VP_ST_SINGULAR = 3 }; // MRHS path: R has a zero / non-finite diagonal and no truncated solve exists
// per-problem small workspace: R^{-1} or R^+ (row-major) | G^T G | P_n (row-major) | truncated flag
template <int N, int P> __host__ __device__ constexpr int mrhs_small_stride() { return 2 * N * N + P * P + 1; }
inline int mrhs_small_stride_rt(int n, int p) { return 2 * n * n + p * p + 1; }

// device workspace of the MRHS path (owned by the handle)
struct MrhsWs {
    void *qthin;      // [B][N][m]  T
    void *g;          // [B][P][m]  T
    double *small;    // [B][mrhs_small_stride]   R^{-1} or R^+ (row-major), G^T G, P_n, truncated flag
    int32_t *statusA; // [B]
    double *acc;      // [B][1 + N*N + P][gx]  per-workgroup partials of sum ||r||^2, sum c c^T, sum c_{j(p)} u_p
    void *lm_state;   // [B] LmVars
    int32_t *nactive; // [2]  problems whose LM loop is still running | most evaluations any finished problem took
    int32_t *done;    // [B] 1 once the problem's LM loop has terminated: later factor / stream launches skip it
    void *alpha_trial; // [B][q] T
    // per-column results of the fit's passes, double-buffered: the pass at a trial point writes buffer widx[b]; when the LM
    // accepts the point that buffer becomes bidx[b] (the best point's) and the next pass writes the other one.  After the
    // loop the best buffer IS (C, cost, status) at the final parameters -- no extra evaluation.
    void *cbuf[2];        // [B][S][n] T
    double *costbuf[2];   // [B][S]
    int32_t *stbuf[2];    // [B][S]
    int32_t *widx, *bidx; // [B]
    // largest conditioning estimate of the column-scaled Jacobian any LM step of the current / last fit has seen (round 5):
    // the step works on J^T J, exact to ~10 cond(J)^2 eps -- vp_global_fit_condition hands it to the
    // caller, who can tell a fit that ran beyond the stated bound from one that did not
    double *jcond; // [B]
};

// where a captured whole-fit graph finds the CALLER's arrays of this call (device-pointer handles): a pinned, device-mapped
// record the host fills before every replay -- the graph itself holds only its address.  Null members: not wanted /
// host-pointer handle (the library's own staging copies are used instead).
struct MrhsIo {
    const void *alpha_in; // [B][q] initial parameters
    void *alpha_out;      // [B][q]
    void *C_out;          // [B][S][n]
    void *rep_out;        // [B] vp_report
};

template <typename T, class M> struct MrhsFactorArgs {
    M mdl;
    const T *t;
    const T *w;
    const T *alpha; // [B][q] trial parameters
    MrhsWs ws;
    int m;
    int64_t B;
    int64_t t_stride, w_stride;
    T eps;
    int skip_done; // fit loop: problems whose LM loop has terminated are skipped
    int grid_uniform; // every grid of the handle is uniform to rounding (grid_check_kernel)
};

// W waves per problem: W = 1 for batches that fill the GPU with one wave per problem; W = 4 (R/4 rows per lane) when
// there are few problems and the factorisation of ONE tall Phi is on the critical path of every LM iteration.
// the factorisation of problem b at `alpha` by the calling group of W waves (results -> a.ws)
// the rows of problem b as the group's lanes see them (uniform grids: the exp recurrence of build_columns).  Built by the
// caller BEFORE anything else it waits for: set_uniform reads the two ends of the grid, a dependent global round trip
// that otherwise sits at the head of the column build
template <typename T, int R, int W> using MrhsSrc = RowSource<T, R, false, 2, 2, W, true>;
template <typename T, class M, int R, int W>
__device__ __forceinline__ MrhsSrc<T, R, W> mrhs_make_src(const MrhsFactorArgs<T, M> &a, const int64_t b, const int gl) {
    MrhsSrc<T, R, W> src;
    src.t = a.t + b * a.t_stride;
    src.w = a.w ? a.w + b * a.w_stride : nullptr;
    src.m = a.m;
    src.lane = gl;
    // 2-element accesses where every grid / weight array starts 16-byte (8-byte for fp32) aligned and m is even
    src.vec = (a.m & 1) == 0 && ((reinterpret_cast<uintptr_t>(a.t) | reinterpret_cast<uintptr_t>(a.w)) & (2 * sizeof(T) - 1)) == 0;
    src.set_uniform(a.grid_uniform != 0);
    return src;
}
template <typename T, class M, int R, int W>
__device__ __forceinline__ void mrhs_factor_body(const MrhsFactorArgs<T, M> &a, const int64_t b, const T (&alpha)[M::Q], Grp<W> &grp,
                                                 const MrhsSrc<T, R, W> &src, long long *clk = nullptr) {
#ifdef VP_MRHS_STEP_CLOCKS
#define VP_MTICK(i, dep) do { if (clk) clk[i] = (long long)__builtin_amdgcn_s_memtime() + ((dep) == T(-1.2345) ? 1 : 0); } while (0)
#else
#define VP_MTICK(i, dep) do { } while (0)
#endif
    constexpr int N = M::N, P = M::P, NC = N + P;
    using L = Layout<R, W>;
    using G = Grp<W>;
    const int lane = grp.gl; // group lane
    const int m = a.m;
    using Src = MrhsSrc<T, R, W>;
    T C[NC][R];
    VP_MTICK(2, alpha[0]);
    build_columns<T, M, R, NC, Src, N>(a.mdl, alpha, src, C);
    VP_MTICK(3, C[0][0] + C[NC - 1][R - 1]);
    T g[N], Rm[N][N], qdummy[N];
    house_qr<T, R, N, NC, 0, false, G>(C, g, Rm, qdummy, grp);
    VP_MTICK(4, Rm[N - 1][N - 1]);
    // R^{-1} (upper triangular) and the rank test of solve_coeffs
    double *small = a.ws.small + b * mrhs_small_stride<N, P>();
    int st = VP_ST_OK;
    bool zero_diag = false;
#pragma unroll
    for (int i = 0; i < N; ++i) zero_diag = zero_diag || !(tabs(Rm[i][i]) > T(0)) || !is_finite(Rm[i][i]);
    T Ri[N][N];
    if (!uni(zero_diag)) {
        T inv_f2 = T(0);
        T rdiag[N]; // one division per diagonal entry (N instead of N (N + 1) / 2)
#pragma unroll
        for (int i = 0; i < N; ++i) rdiag[i] = T(1) / Rm[i][i];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                if (i > j) {
                    Ri[i][j] = T(0);
                    continue;
                }
                T acc = (i == j) ? T(1) : T(0);
#pragma unroll
                for (int l = i + 1; l <= j; ++l) acc = tfma(-Rm[i][l], Ri[l][j], acc);
                Ri[i][j] = acc * rdiag[i];
                inv_f2 = tfma(Ri[i][j], Ri[i][j], inv_f2);
            }
        if (!uni(inv_f2 * a.eps * a.eps < T(1))) st = VP_ST_SINGULAR;
        if (!uni(is_finite(inv_f2))) st = VP_ST_NONFINITE;
    } else {
        st = VP_ST_SINGULAR;
    }
    // rank deficient to the reference's rule (a singular value <= eps): truncated pseudo-inverse, column by column
    // (R^+ e_k and (I - R R^+) e_k from the same one-sided Jacobi SVD as the single-RHS kernels; rare path)
    T Pn[N][N];
    bool truncated = false;
    {
        bool finite_r = true;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = i; j < N; ++j) finite_r = finite_r && is_finite(Rm[i][j]);
        if (uni(st == VP_ST_SINGULAR && finite_r)) {
            truncated = true;
            st = VP_ST_OK;
#pragma unroll 1
            for (int k = 0; k < N; ++k) {
                TruncIn<T, N> in;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    in.qty[i] = (i == k) ? T(1) : T(0);
#pragma unroll
                    for (int j = 0; j < N; ++j) in.Rm[i][j] = (j >= i) ? Rm[i][j] : T(0);
                }
                in.eps = a.eps;
                const TruncOut<T, N> out = truncated_solve<T, N>(in);
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    dyn_set<N>(Ri[i], k, out.c[i]);
                    dyn_set<N>(Pn[i], k, ((i == k) ? T(1) : T(0)) - out.e[i]);
                }
            }
        } else if (st != VP_ST_OK) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) Ri[i][j] = T(0);
        }
    }
    if (lane == 0) {
        a.ws.statusA[b] = st;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                small[i * N + j] = (double)Ri[i][j];
                small[N * N + P * P + i * N + j] = truncated ? (double)Pn[i][j] : (i == j ? 1.0 : 0.0);
            }
        small[2 * N * N + P * P] = truncated ? 1.0 : 0.0;
    }
    // G_p = Q [0; (Q^T W dPhi_p)_{>= N}]  in place on the derivative columns
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r)
            if (L::row_of(r, lane) < N) C[N + p][r] = T(0);
    VP_MTICK(5, Ri[0][0]);
    // 16-byte stores of the row pairs where the columns allow it (m even: every column starts 16-byte aligned)
    const bool vst = sizeof(T) == 8 && (m & 1) == 0 && ((reinterpret_cast<uintptr_t>(a.ws.qthin) | reinterpret_cast<uintptr_t>(a.ws.g)) & 15) == 0;
    // G and the explicit thin Q (column j = Q e_j) from ONE back-sweep over the reflectors when the N unit vectors fit
    // the registers next to the columns (N rounds instead of 2 N), else G first, then Q (N rounds, or N*N one by one)
    T *qout = (T *)a.ws.qthin + b * (int64_t)N * m;
    constexpr bool Z_FITS = N * R * (int)(sizeof(T) / 4) <= 96;
    auto gram_and_store_g = [&]() __attribute__((always_inline)) {
        if constexpr (P > 0) {
            T gg[P * P];
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int p2 = 0; p2 < P; ++p2) {
                    T acc = T(0);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc = tfma(C[N + p][r], C[N + p2][r], acc);
                    gg[p * P + p2] = acc;
                }
            group_allreduce(grp, gg);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < P * P; ++i) small[N * N + i] = (double)gg[i];
            }
            T *gout = (T *)a.ws.g + b * (int64_t)P * m;
#pragma unroll
            for (int p = 0; p < P; ++p) store_rows<T, R, W>(gout + (int64_t)p * m, m, lane, vst, C[N + p]);
        }
    };
    if constexpr (Z_FITS && P > 0 && P + N <= VP_XV && P * P <= VP_XV) {
        T Z[N][R];
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) Z[j][r] = (L::row_of(r, lane) == j) ? T(1) : T(0);
        apply_q_cols_and_z<T, R, N, NC, N, NC, N>(C, g, Z, grp);
        VP_MTICK(6, Z[0][0]);
#pragma unroll
        for (int j = 0; j < N; ++j) store_rows<T, R, W>(qout + (int64_t)j * m, m, lane, vst, Z[j]);
        gram_and_store_g();
    } else {
        if constexpr (P > 0) apply_q_cols<T, R, N, NC, N, NC>(C, g, grp);
        gram_and_store_g();
        if constexpr (Z_FITS) {
            T Z[N][R];
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r) Z[j][r] = (L::row_of(r, lane) == j) ? T(1) : T(0);
            apply_q<T, R, N, NC, N>(C, g, Z, grp);
#pragma unroll
            for (int j = 0; j < N; ++j) store_rows<T, R, W>(qout + (int64_t)j * m, m, lane, vst, Z[j]);
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T Z[1][R];
#pragma unroll
                for (int r = 0; r < R; ++r) Z[0][r] = (L::row_of(r, lane) == j) ? T(1) : T(0);
                apply_q<T, R, N, NC, 1>(C, g, Z, grp);
                store_rows<T, R, W>(qout + (int64_t)j * m, m, lane, vst, Z[0]);
            }
        }
    }
}

template <typename T, class M, int R, int W = 1>
__global__ void __launch_bounds__(64 * W) mrhs_factor_kernel(const MrhsFactorArgs<T, M> a) {
    constexpr int Q = M::Q;
    using G = Grp<W>;
    __shared__ __attribute__((aligned(16))) unsigned char s_xch[group_xch_bytes<W>() > 0 ? group_xch_bytes<W>() : 16];
    G grp = G::make(s_xch);
    const int64_t b = blockIdx.x;
    if (b >= a.B) return;
    if (a.skip_done && uni(a.ws.done[b]) != 0) return;
    const MrhsSrc<T, R, W> src = mrhs_make_src<T, M, R, W>(a, b, grp.gl);
    T alpha[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];
    mrhs_factor_body<T, M, R, W>(a, b, alpha, grp, src);
}

// Packed wave reduction of V values whose totals are STORED by the lanes that end up holding them: dst[v] = total of
// value v (same packing as wave_allreduce, vp_device.hpp; no broadcast, i.e. no 2V SGPRs)
template <int V, typename T> __device__ __forceinline__ void wave_reduce_store_g(T (&x)[V], T *dst) {
    constexpr int V1 = (V + 1) / 2, V2 = (V1 + 1) / 2, V3 = (V2 + 1) / 2, V4 = (V3 + 1) / 2;
    T y1[V1], y2[V2], y3[V3], y4[V4];
    pack_level<0>(x, y1);
    pack_level<1>(y1, y2);
    pack_level<2>(y2, y3);
    pack_level<3>(y3, y4);
#pragma unroll
    for (int i = 0; i < V4; ++i) y4[i] += dpp<DPP_ROR4>(y4[i]);
#pragma unroll
    for (int i = 0; i < V4; ++i) y4[i] += dpp<DPP_ROR8>(y4[i]);
    const int L = lane_id();
    const int v = ((L >> 5) & 1) | (((L >> 4) & 1) << 1) | ((L & 1) << 2) | (((L >> 1) & 1) << 3);
    if ((L & 0xC) == 0) {
#pragma unroll
        for (int i = 0; i < V4; ++i)
            if (16 * i + v < V) dst[16 * i + v] = y4[i];
    }
}

template <typename T, int N, int P> struct MrhsStreamArgs {
    const T *yw;   // [B][S][m]
    MrhsWs ws;
    T *r_out;      // MODE 1: [B][S][m] or null
    T *J_out;      // MODE 1: [B][q][S][m] or null
    T *C_out;      // MODE 1: [B][S][n] or null
    double *cost_bs;  // MODE 1
    int32_t *status_bs;
    int pb[P > 0 ? P : 1], pp[P > 0 ? P : 1]; // pair -> basis, pair -> parameter
    int q;
    int m;
    int S;
    int64_t B;
    int gx; // workgroups per problem: the grid is ONE-dimensional, gx * B workgroups (no 65535 limit on B)
};

#ifndef VP_MRHS_WAVES
#define VP_MRHS_WAVES 8
#endif
#ifndef VP_MRHS_CH
#define VP_MRHS_CH 8
#endif
// MODE 0: reduced quantities for the LM loop; MODE 1: trait-level outputs
template <typename T, int N, int P, int R, int MODE>
__global__ void __launch_bounds__(64 * VP_MRHS_WAVES) mrhs_stream_kernel(const MrhsStreamArgs<T, N, P> a) {
    constexpr int MP = 64 * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *s_q = reinterpret_cast<T *>(smem_raw); // [N][MP]
    T *s_g = s_q + N * MP;                    // [P][MP]
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6), nwave = (int)(blockDim.x >> 6);
    const int64_t b = blockIdx.x / a.gx;
    const int wgi = (int)(blockIdx.x - b * a.gx); // workgroup within the problem
    const int m = a.m;
    if constexpr (MODE == 0) {
        if (a.ws.done[b] != 0) return; // the LM loop of this problem has terminated (uniform per workgroup)
    }
    const T *qsrc = (const T *)a.ws.qthin + b * (int64_t)N * m;
    const T *gsrc = (const T *)a.ws.g + b * (int64_t)P * m;
    // stage Q and G in LDS.  Fast path (row pairs divide evenly over the workgroup, m even): every thread requests ALL of
    // its 16-byte pieces before it stores the first one -- the element-by-element loop below is (N+P)*MP/threads dependent
    // global loads deep, ~6 us in front of a pass that streams for ~60
    auto stage_slow = [&]() __attribute__((always_inline)) {
        for (int idx = threadIdx.x; idx < (N + P) * MP; idx += blockDim.x) {
            const int col = idx / MP, row = idx - col * MP;
            T v = T(0);
            if (row < m) v = (col < N) ? qsrc[(int64_t)col * m + row] : gsrc[(int64_t)(col - N) * m + row];
            s_q[idx] = v;
        }
    };
    constexpr int NT = 64 * VP_MRHS_WAVES, PAIRS = MP / 2;
    if constexpr (PAIRS % NT == 0 && sizeof(T) == 8) {
        if ((m & 1) == 0 && blockDim.x == NT) {
            constexpr int KP = PAIRS / NT;
            double2 tmp[(N + P) * KP];
#pragma unroll
            for (int c = 0; c < N + P; ++c) {
                const T *src = (c < N) ? qsrc + (int64_t)c * m : gsrc + (int64_t)(c - N) * m;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const int row = 2 * ((int)threadIdx.x + k * NT);
                    tmp[c * KP + k] = (row < m) ? *reinterpret_cast<const double2 *>(src + row) : make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int c = 0; c < N + P; ++c)
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    *reinterpret_cast<double2 *>(s_q + (size_t)c * MP + 2 * ((int)threadIdx.x + k * NT)) = tmp[c * KP + k];
        } else {
            stage_slow();
        }
    } else {
        stage_slow();
    }
    const double *small = a.ws.small + b * mrhs_small_stride<N, P>();
    // R^{-1} (or R^+) sits in LDS behind the columns, not in 2 N^2 VGPRs: read (broadcast) once per right-hand side
    T *s_ri = s_q + (size_t)(N + P) * MP;
    if (threadIdx.x < N * N) s_ri[threadIdx.x] = (T)small[threadIdx.x];
    const int stA = a.ws.statusA[b];
    const int wsel = (MODE == 0) ? (uni(a.ws.widx[b]) & 1) : 0;
    const bool truncated = uni(small[2 * N * N + P * P] != 0.0); // rank-deficient Phi_w: R^+ and P_n (rare)
    __syncthreads();

    using L = Layout<R>;
    // the 1 + N^2 + P wave-uniform sums (cost | c c^T | c_{j(p)} u_p) live ONE PER LANE in a single register: lane k
    // accumulates sum k (same fma per sum as a register per sum would do; 2 VGPRs instead of 2 (1 + N^2 + P))
    constexpr int NACC = 1 + N * N + P;
    static_assert(NACC <= 64, "one lane per accumulator");
    T accv = T(0);
    const int ak = lane < NACC ? lane : 0;
    const int ai = (ak >= 1 && ak <= N * N) ? (ak - 1) / N : 0, aj = (ak >= 1 && ak <= N * N) ? (ak - 1) % N : 0;
    const int ap = ak > N * N ? ak - 1 - N * N : 0;

    const int64_t gw = (int64_t)wgi * nwave + wave, nw = (int64_t)a.gx * nwave;
    for (int64_t s = gw; s < a.S; s += nw) {
        const int64_t prob = b * a.S + s;
        const T *yp = a.yw + prob * (int64_t)m;
        const bool yvec = vec_aligned<T>(yp, m);
        T y[R];
        load_rows<T, R>(yp, m, lane, yvec, y);
        // All row loops are processed in chunks of CH rows separated by scheduling fences: without them the
        // scheduler hoists the LDS reads of every (column, row) ahead (7 columns x R rows) and spills.
        // (the chunk must divide R: 12 rows per lane take chunks of 4)
        constexpr int CH = (R > VP_MRHS_CH) ? ((R % VP_MRHS_CH == 0) ? VP_MRHS_CH : ((R % 4 == 0) ? 4 : 2)) : R;
        static_assert(R % CH == 0, "row chunks tile the lane's rows");
        // T = Q^T y
        T tq[N];
#pragma unroll
        for (int j = 0; j < N; ++j) tq[j] = T(0);
#pragma unroll
        for (int c0 = 0; c0 < R; c0 += CH) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
                for (int r0 = c0; r0 < c0 + CH; r0 += L::VW) {
                    const int i = L::row_of(r0, lane);
#pragma unroll
                    for (int e = 0; e < L::VW; ++e) tq[j] = tfma(s_q[j * MP + i + e], y[r0 + e], tq[j]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        wave_allreduce(tq);
        // c = R^{-1} T   (truncated: c = R^+ T with the full matrix, and T := P_n T for the residual)
        T c[N];
        if (truncated) {
            T tp[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                T acc = T(0), accp = T(0);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    acc = tfma(s_ri[i * N + j], tq[j], acc);
                    accp = tfma((T)small[N * N + P * P + i * N + j], tq[j], accp);
                }
                c[i] = acc;
                tp[i] = accp;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) tq[i] = tp[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                T acc = T(0);
#pragma unroll
                for (int j = i; j < N; ++j) acc = tfma(s_ri[i * N + j], tq[j], acc);
                c[i] = acc;
            }
        }
        // r = y - Q T  (in place)
#pragma unroll
        for (int c0 = 0; c0 < R; c0 += CH) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
                for (int r0 = c0; r0 < c0 + CH; r0 += L::VW) {
                    const int i = L::row_of(r0, lane);
#pragma unroll
                    for (int e = 0; e < L::VW; ++e) y[r0 + e] = tfma(-tq[j], s_q[j * MP + i + e], y[r0 + e]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ||r||^2 and u = G^T r
        T red[1 + P];
        {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < R; ++r) acc = tfma(y[r], y[r], acc);
            red[0] = acc;
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p) red[1 + p] = T(0);
#pragma unroll
            for (int c0 = 0; c0 < R; c0 += CH) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r0 = c0; r0 < c0 + CH; r0 += L::VW) {
                        const int i = L::row_of(r0, lane);
#pragma unroll
                        for (int e = 0; e < L::VW; ++e)
                            red[1 + p] = tfma(s_g[p * MP + i + e], y[r0 + e], red[1 + p]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            wave_allreduce(red);
            { // the column's coefficients / cost / status at this trial point (kept if the LM accepts it)
                bool ok = is_finite(red[0]) && stA == VP_ST_OK;
#pragma unroll
                for (int i = 0; i < N; ++i) ok = ok && is_finite(c[i]);
                if (lane == 0) {
                    a.ws.costbuf[wsel][prob] = 0.5 * (double)red[0];
                    a.ws.stbuf[wsel][prob] = ok ? VP_ST_OK : (stA != VP_ST_OK ? stA : VP_ST_NONFINITE);
                }
                if (lane < N) ((T *)a.ws.cbuf[wsel])[prob * N + lane] = dyn_get<N>(c, lane);
            }
            {
                T cpb = T(0), up = T(0); // c_{j(p)}, u_p of this lane's pair
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    cpb = (ap == p) ? dyn_get<N>(c, a.pb[p]) : cpb;
                    up = (ap == p) ? red[1 + p] : up;
                }
                const T fa_ = (ak == 0) ? red[0] : (ak <= N * N ? dyn_get<N>(c, ai) : cpb);
                const T fb_ = (ak == 0) ? T(1) : (ak <= N * N ? dyn_get<N>(c, aj) : up);
                accv = tfma(fa_, fb_, accv);
            }
        } else {
            T r1[1] = {red[0]};
            wave_allreduce(r1);
            bool ok = is_finite(r1[0]) && stA == VP_ST_OK;
#pragma unroll
            for (int i = 0; i < N; ++i) ok = ok && is_finite(c[i]);
            if (lane == 0) {
                if (a.cost_bs) a.cost_bs[prob] = 0.5 * (double)r1[0];
                if (a.status_bs) a.status_bs[prob] = ok ? VP_ST_OK : (stA != VP_ST_OK ? stA : VP_ST_NONFINITE);
            }
            if (a.C_out && lane < N) a.C_out[prob * N + lane] = dyn_get<N>(c, lane);
            if (a.r_out) store_rows_out<T, R>(a.r_out + prob * (int64_t)m, m, lane, yvec, y);
            if (a.J_out) {
                for (int k = 0; k < a.q; ++k) {
                    T jk[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) jk[r] = T(0);
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        if (a.pp[p] == k) {
                            const T cj = -dyn_get<N>(c, a.pb[p]);
#pragma unroll
                            for (int r0 = 0; r0 < R; r0 += L::VW) {
                                const int i = L::row_of(r0, lane);
#pragma unroll
                                for (int e = 0; e < L::VW; ++e) jk[r0 + e] = tfma(cj, s_g[p * MP + i + e], jk[r0 + e]);
                            }
                        }
                    }
                    T *jp = a.J_out + ((b * a.q + k) * (int64_t)a.S + s) * (int64_t)m;
                    store_rows_out<T, R>(jp, m, lane, vec_aligned<T>(jp, m), jk);
                }
            }
        }
    }
    if constexpr (MODE == 0) {
        // workgroup-level reduction of the per-wave partial sums, then ONE plain store per workgroup into its
        // own slot acc[b][blockIdx.x][:] (no atomics: 2048 waves x 20 same-address atomics cost ~0.5 ms)
        __syncthreads(); // everyone is done reading s_q / s_g: reuse the front of the LDS as scratch
        double *s_part = reinterpret_cast<double *>(smem_raw);
        if (lane < NACC) s_part[wave * NACC + lane] = (double)accv;
        __syncthreads();
        if (threadIdx.x < NACC) {
            double tot = 0.0;
            for (int w = 0; w < nwave; ++w) tot += s_part[w * NACC + threadIdx.x];
            a.ws.acc[((size_t)b * NACC + threadIdx.x) * a.gx + wgi] = tot; // [b][accumulator][workgroup]
        }
    }
}

// ---- MODE 0 (fit), workgroup-cooperative with LDS-DMA prefetch (round 3) ------------------------------------------------
// The one-wave-per-column kernel above re-reads Q and G from LDS for every column (176 ds_read_b128 per column: the LDS
// pipe alone is ~38 us of a pass at configs[2]) and waits for its 16 KiB column before it computes (SQ_WAIT_ANY 75 %,
// profiles/r02_mrhs_stream_pmc.json).  Here the NW waves of a workgroup share EVERY column: group lane gl owns
// RW = R / NW rows (Layout<RW, NW>: row pairs dealt round-robin over the 64 NW lanes, so each wave-level access is 1 KiB
// contiguous and a column is one 16 KiB burst per workgroup) and the lane's slice of Q lives in REGISTERS for the whole
// launch -- no LDS reads of Q or G in the loop.  y streams through a per-wave LDS ring of 2 batches of NB columns filled by
// global_load_lds_dwordx4 (asynchronous global -> LDS DMA: no VGPRs, lane-linear destination = this layout), i.e. batches
// k+1 and k+2 are in flight while batch k is computed.  Per batch ONE cross-wave reduction (packed wave reduction -> LDS
// -> one bare s_barrier): T = Q^T y for the NB columns, plus the squared residual norms of the PREVIOUS batch (they need T
// first), so the per-column cost / status trail one batch.  The fit's other sums need no per-column reduction:
//     sum_s c_{j(p),s} (G_p^T r_s) = G_p^T (sum_s c_{j(p),s} r_s):  P accumulator rows per lane, ONE dot with G at the end;
//     c = R^{-1} T and sum c c^T are wave-uniform, accumulated one per lane by wave 0.
// The DMA is issued from inline asm (the compiler would drain it with vmcnt(0) at every LDS read it cannot disambiguate),
// so the waits are counted by hand: a wave issues KL = NB * RW/2 = 8 DMA instructions per batch, in order, and NOTHING else
// that counts in vmcnt inside the loop -- the per-column results (c, cost, status) are staged in LDS and written by wave 0
// in bursts, after a full drain; selections out of the coefficient vector are one-hot dot products (a select chain would
// become a dynamically indexed vector in scratch, whose accesses count in vmcnt too).
// Measured at configs[2] (profiles/r03_mrhs_stream.json): the pass is bound by the length of each wave's instruction stream
// (VALU + SALU, in-order, 2 waves per SIMD), not by HBM: 82 us with 8 waves per column group, 66 us with 4 (half the
// redundant wave-uniform work per column), 55 us of it with the loads switched off; a pure read of the same 268 MB takes
// 41 us (tools/read_pattern.hip).
template <typename T, int N, int P, int RW, int NW, int NB>
__global__ void __launch_bounds__(64 * NW) mrhs_coop_dma_kernel(const MrhsStreamArgs<T, N, P> a) {
    static_assert(sizeof(T) == 8 && RW >= 2 && RW % 2 == 0, "fp64, row pairs");
    constexpr int NPAIR = RW / 2, KL = NB * NPAIR;
    constexpr int D = 2; // ring depth
    constexpr int NRED = N * NB + NB;                 // per batch: T of the NB columns | ||r||^2 of the previous batch
    constexpr int NFL = P * N + NB;                   // flush round: the P x N dots G_p^T W_j | ||r||^2 of the last batch
    constexpr int NX = NRED > NFL ? NRED : NFL;
    constexpr int NACC = 1 + N * N + P;
    constexpr int FB = 16;            // batches between two result bursts
    constexpr int FC = FB * NB;       // columns staged
    static_assert(NACC <= 64 && NX <= 64 && KL == 8, "one lane per value; the wait counts below are written for KL = 8");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // ring[NW][D][KL][64] double2 | s_x[2][NW][NX] | s_ri[2 N N] | s_t[FC][N] | s_cost[FC] | s_st[FC] | s_fin[N N + P N]
    double2 *ring = reinterpret_cast<double2 *>(smem_raw);
    double *s_x = reinterpret_cast<double *>(ring + (size_t)NW * D * KL * 64);
    double *s_ri = s_x + 2 * NW * NX;
    double *s_t = s_ri + 2 * N * N;
    double *s_cost = s_t + FC * N;
    double *s_fin = s_cost + FC;
    int *s_st = reinterpret_cast<int *>(s_fin + N * N + P * N);
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int gl = (int)threadIdx.x;
    const int64_t b = blockIdx.x / a.gx;
    const int wgi = (int)(blockIdx.x - b * a.gx); // workgroup within the problem
    const int m = a.m, S = a.S, gx = a.gx;
    if (a.ws.done[b] != 0) return; // (uniform per workgroup)
    const T *qsrc = (const T *)a.ws.qthin + b * (int64_t)N * m;
    const T *gsrc = (const T *)a.ws.g + b * (int64_t)P * m;
    const double *small = a.ws.small + b * mrhs_small_stride<N, P>();
    const int nbatch = (S + NB - 1) / NB;
    const int nloc = (nbatch > wgi) ? (nbatch - 1 - wgi) / gx + 1 : 0; // batches of this workgroup
    const bool ragged = (S % NB) != 0;                                  // the last batch has columns past S
    const bool fullrows = m == 64 * RW * NW;
    const unsigned ring_lds = (unsigned)(uintptr_t)(VP_LDS unsigned char *)smem_raw + (unsigned)wave * (D * KL * 1024u);
    // rows of this lane: pair k covers rows (k * 64 NW + gl) * 2, +1; rows >= m (m even) are clamped for the DMA and zeroed after
    bool rvalid[NPAIR];
    unsigned roffb[NPAIR]; // byte offset of the pair within a column
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
        const int row = (k * 64 * NW + gl) * 2;
        rvalid[k] = row < m;
        roffb[k] = (unsigned)(rvalid[k] ? row : 0) * (unsigned)sizeof(T);
    }
    const T *ybase = a.yw + b * (int64_t)S * m;
    // the KL DMA instructions of local batch i into ring slot i % D: wave-uniform column base in SGPRs, per-lane byte
    // offset in one VGPR (columns past S re-read column S - 1: the instruction count per batch must not depend on the data)
    auto issue = [&](const int i) __attribute__((always_inline)) {
        const int bt = wgi + i * gx;
        const unsigned slot = ring_lds + (unsigned)(i & (D - 1)) * (KL * 1024u);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            int s = bt * NB + c;
            s = s < S ? s : S - 1;
            const T *col = ybase + (int64_t)s * m;
#pragma unroll
            for (int k = 0; k < NPAIR; ++k) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)(c * NPAIR + k) * 1024u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(roffb[k]), "s"(col), "s"(dst)
                             : "memory");
            }
        }
    };
    // The first two batches of y are requested BEFORE the workgroup fetches its slices of Q (they depend on nothing the
    // factorisation wrote): one global round trip instead of two in front of the loop.  G is only needed by the flush at
    // the very end: one dword per cache line is touched here so that the flush finds it in this XCD's L2.
    if (nloc > 0) issue(0);
    if (nloc > 1) issue(1);
    if constexpr (P > 0) {
        const int glines = (int)(((int64_t)P * m * (int64_t)sizeof(T) + 127) / 128);
        for (int ln = gl; ln < glines; ln += 64 * NW) {
            unsigned dummy;
            asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(reinterpret_cast<const char *>(gsrc) + (size_t)ln * 128) : "memory");
        }
    }
    T q[N][RW];
#pragma unroll
    for (int j = 0; j < N; ++j) load_rows<T, RW, NW>(qsrc + (int64_t)j * m, m, gl, true, q[j]);
    // W_j = sum_s T_{j,s} r_s  (row space; c = R^-1 T is linear in T, so sum_s c_{i,s} r_s = sum_j Rinv[i][j] W_j at the end)
    T wacc[N][RW];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int r = 0; r < RW; ++r) wacc[j][r] = T(0);
    if (threadIdx.x < N * N) {
        s_ri[threadIdx.x] = small[threadIdx.x];                          // R^-1 (or R^+), row-major
        s_ri[N * N + threadIdx.x] = small[N * N + P * P + threadIdx.x];  // P_n
    }
    const int stA = a.ws.statusA[b];
    const int wsel = uni(a.ws.widx[b]) & 1;
    const bool truncated = uni(small[2 * N * N + P * P] != 0.0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // every compiler-issued load has landed: vmcnt is ours now
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // wave 0, one sum per lane: lane 0 sum ||r||^2, lane 1 + i N + j sum T_i T_j (one-hot selections: a select chain over
    // a small array is folded into a dynamically indexed vector kept in SCRATCH, whose accesses count in vmcnt)
    T accv = T(0);
    const int ak = lane < NACC ? lane : 0;
    const bool ak_tt = ak >= 1 && ak <= N * N;
    T ohi[N], ohj[N];
#pragma unroll
    for (int j2 = 0; j2 < N; ++j2) {
        ohi[j2] = (ak_tt && (ak - 1) / N == j2) ? T(1) : T(0);
        ohj[j2] = (ak_tt && (ak - 1) % N == j2) ? T(1) : T(0);
    }
    T r2prev[NB];
    unsigned badprev = 0u; // bit c: T of column c of the previous batch was not finite
#pragma unroll
    for (int c = 0; c < NB; ++c) r2prev[c] = T(0);
    int ph = 0;
    // cost / status of local batch i (its squared norms arrive one batch late) -> staging rows (i % FB) * NB + c
    auto emit_prev = [&](const T tot, const int i) __attribute__((always_inline)) {
        if (i < 0) return;
        const int bt = wgi + i * gx;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (bt * NB + c >= S) continue;
            const T r2 = readlane(tot, N * NB + c);
            const bool ok = is_finite(r2) && stA == VP_ST_OK && ((badprev >> c) & 1u) == 0u;
            if (wave == 0) {
                if (lane == 0) {
                    s_cost[(i & (FB - 1)) * NB + c] = 0.5 * (double)r2;
                    s_st[(i & (FB - 1)) * NB + c] = ok ? VP_ST_OK : (stA != VP_ST_OK ? stA : VP_ST_NONFINITE);
                }
                accv += (ak == 0) ? r2 : T(0);
            }
        }
    };
    // wave 0 writes the staged results of local batches [i0, i1): c = R^-1 T per column, cost, status (after draining its
    // DMA: stores and loads share vmcnt)
    auto burst = [&](const int i0, const int i1) __attribute__((always_inline)) {
        if (wave != 0 || i1 <= i0) return;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const int c = lane / N, jn = lane % N; // lanes 0 .. NB*N-1: coefficient jn of column c
        for (int i = i0; i < i1; ++i) {
            const int bt = wgi + i * gx;
            const int s = bt * NB + c;
            if (lane < NB * N && s < S) {
                const double *tt = s_t + ((i & (FB - 1)) * NB + c) * N;
                double cv = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) cv = tfma(s_ri[jn * N + j], tt[j], cv);
                ((T *)a.ws.cbuf[wsel])[(b * S + s) * N + jn] = (T)cv;
            }
            const int s2 = bt * NB + lane;
            if (lane < NB && s2 < S) {
                a.ws.costbuf[wsel][b * S + s2] = s_cost[(i & (FB - 1)) * NB + lane];
                a.ws.stbuf[wsel][b * S + s2] = s_st[(i & (FB - 1)) * NB + lane];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores are out before the next DMA is counted
    };
    int flushed = 0; // local batches whose cost / status / c have been written
    for (int i = 0; i < nloc; ++i) {
        const int bt = wgi + i * gx;
        // ---- batch i has landed once at most the KL instructions of batch i+1 are outstanding ----
#ifdef VP_MRHS_NOLOAD
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (i + 1 < nloc) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        T y[NB][RW];
        {
            const double2 *slot = ring + ((size_t)wave * D + (size_t)(i & (D - 1))) * KL * 64 + lane;
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int k = 0; k < NPAIR; ++k) {
                    const double2 v = slot[(c * NPAIR + k) * 64];
                    y[c][2 * k] = v.x;
                    y[c][2 * k + 1] = v.y;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot has been read: it may be refilled
#ifdef VP_MRHS_NOLOAD // developer A/B: compute + synchronisation alone (the ring keeps the first two batches)
        (void)issue;
#else
        if (i + D < nloc) issue(i + D);
#endif
        if (!fullrows) { // padding rows of a partially filled row block
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int k = 0; k < NPAIR; ++k) {
                    y[c][2 * k] = rvalid[k] ? y[c][2 * k] : T(0);
                    y[c][2 * k + 1] = rvalid[k] ? y[c][2 * k + 1] : T(0);
                }
        }
        if (ragged && bt == nbatch - 1) { // columns past S in the very last batch
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (bt * NB + c >= S) {
#pragma unroll
                    for (int r = 0; r < RW; ++r) y[c][r] = T(0);
                }
        }
        // ---- partial T = Q^T y | carried squared norms ----
        T red[NRED]; // (exactly the NRED values of a batch: padded to the flush round's NX = 14 the packed reduction merged
                     // two more register pairs per level, round 5)
#pragma unroll
        for (int c = 0; c < NB; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc = tfma(q[j][r], y[c][r], acc);
                red[c * N + j] = acc;
            }
            red[N * NB + c] = r2prev[c];
        }
        wave_reduce_store_g<NRED>(red, s_x + ((size_t)ph * NW + wave) * NX);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); // (a bare barrier: __syncthreads() would carry a full memory fence, i.e. drain the DMA)
        asm volatile("" ::: "memory");
        T tot = T(0);
        if (lane < NRED) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += s_x[((size_t)ph * NW + w) * NX + lane];
        }
        ph ^= 1;
        emit_prev(tot, i - 1);
        // a burst of FB batches is complete once the cost of its last batch has been emitted (one batch late)
        if (i > 0 && (i & (FB - 1)) == 0) {
            burst(flushed, i);
            flushed = i;
        }
        // non-finite T (lanes c N .. c N + N - 1 of tot) -> status of that column
        {
            const unsigned long long bad = __builtin_amdgcn_ballot_w64(lane < N * NB && !is_finite(tot));
            badprev = 0u;
#pragma unroll
            for (int c = 0; c < NB; ++c) badprev |= ((bad >> (c * N)) & ((1ull << N) - 1ull)) != 0ull ? (1u << c) : 0u;
        }
        if (wave == 0 && lane < N * NB) s_t[(i & (FB - 1)) * NB * N + lane] = (double)tot; // T of the batch, for the burst
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            T tq[N];
#pragma unroll
            for (int j = 0; j < N; ++j) tq[j] = readlane(tot, c * N + j);
            if (wave == 0) { // sum T T^T (columns past S carry T = 0)
                T fa_ = T(0), fb_ = T(0);
#pragma unroll
                for (int j2 = 0; j2 < N; ++j2) {
                    fa_ = tfma(ohi[j2], tq[j2], fa_);
                    fb_ = tfma(ohj[j2], tq[j2], fb_);
                }
                accv = tfma(fa_, fb_, accv);
            }
            T tr[N]; // what the residual subtracts: T itself, or P_n T for a rank-deficient Phi_w (rare)
#pragma unroll
            for (int j = 0; j < N; ++j) tr[j] = tq[j];
            if (truncated) {
#pragma unroll
                for (int i2 = 0; i2 < N; ++i2) {
                    T accp = T(0);
#pragma unroll
                    for (int j = 0; j < N; ++j) accp = tfma((T)s_ri[N * N + i2 * N + j], tq[j], accp);
                    tr[i2] = accp;
                }
            }
            T r2 = T(0);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                T v = y[c][r];
#pragma unroll
                for (int j = 0; j < N; ++j) v = tfma(-tr[j], q[j][r], v);
                r2 = tfma(v, v, r2);
#pragma unroll
                for (int j = 0; j < N; ++j) wacc[j][r] = tfma(tq[j], v, wacc[j][r]);
            }
            r2prev[c] = r2;
        }
    }
    // ---- flush: squared norms of the last batch, the P x N dots G_p^T W_j, the remaining staged results ----
    {
        T red[NX];
#pragma unroll
        for (int v = 0; v < NX; ++v) red[v] = T(0);
#pragma unroll
        for (int c = 0; c < NB; ++c) red[P * N + c] = r2prev[c];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            T gs[RW];
            load_rows<T, RW, NW>(gsrc + (int64_t)p * m, m, gl, true, gs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc = tfma(gs[r], wacc[j][r], acc);
                red[p * N + j] = acc;
            }
        }
        wave_reduce_store_g<NX>(red, s_x + ((size_t)ph * NW + wave) * NX);
        __syncthreads();
        T tot = T(0);
        if (lane < NFL) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += s_x[((size_t)ph * NW + w) * NX + lane];
        }
        // the squared norms of the last batch sit in lanes P N .. here: move them to where emit_prev expects them
        {
            T shifted = T(0);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const T v = readlane(tot, P * N + c);
                shifted = (lane == N * NB + c) ? v : shifted;
            }
            emit_prev(shifted, nloc - 1);
        }
        burst(flushed, nloc);
        if (wave == 0) {
            // back from T-space: sum c c^T = Rinv (sum T T^T) Rinv^T;  sum_s c_{j(p),s} u_{p,s} = sum_j Rinv[j(p)][j] (G_p^T W_j)
            if (ak_tt) s_fin[ak - 1] = (double)accv;
            if (lane < P * N) s_fin[N * N + lane] = (double)tot;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double outv = (double)accv; // lane 0: sum ||r||^2
            if (ak_tt) {
                const int ci = (ak - 1) / N, cj = (ak - 1) % N;
                double acc2 = 0.0;
                for (int i2 = 0; i2 < N; ++i2)
                    for (int j2 = 0; j2 < N; ++j2) acc2 = tfma(s_ri[ci * N + i2] * s_ri[cj * N + j2], s_fin[i2 * N + j2], acc2);
                outv = acc2;
            } else if (ak > N * N) {
                const int p = ak - 1 - N * N;
                int jb = 0;
#pragma unroll
                for (int pp = 0; pp < P; ++pp) jb = (pp == p) ? a.pb[pp] : jb;
                double acc2 = 0.0;
                for (int j2 = 0; j2 < N; ++j2) acc2 = tfma(s_ri[jb * N + j2], s_fin[N * N + p * N + j2], acc2);
                outv = acc2;
            }
            if (lane < NACC) a.ws.acc[((size_t)b * NACC + lane) * gx + wgi] = outv; // [b][accumulator][workgroup]
        }
    }
}

// ---- MODE 1 (trait-level outputs), workgroup-cooperative --------------------------------------------------------------
// The same column-sharing layout as mrhs_coop_dma_kernel for the pass that WRITES r and J (4 of its 5 streams are stores):
// slices of Q and G in registers, the next batch of y prefetched into registers by ordinary loads (stores and loads share
// vmcnt, so the hand-counted LDS-DMA of the fit pass is not an option here; the compiler's own counting is), one cross-wave
// reduction per batch, r = y - Q T and J_k = -sum_{p in k} c_{j(p)} G_p written as 1 KiB bursts per wave instruction.
// Two shapes are instantiated (A/B on the GPU, cfg2): batches of NB = 2 columns with one workgroup per CU (270 VGPRs) for
// the pass that writes r AND J (0.247 ms against 0.26-0.27 for the other shape), and NB = 1 with two workgroups per CU
// (<= 256 VGPRs) for the pass that writes r alone (0.106 ms against 0.142).
template <typename T, int N, int P, int RW, int NW, int NB, int WPE>
__global__ void __launch_bounds__(64 * NW, WPE) mrhs_coop_out_kernel(const MrhsStreamArgs<T, N, P> a) {
    static_assert(sizeof(T) == 8 && RW >= 2 && RW % 2 == 0, "fp64, row pairs");
    constexpr int NRED = N * NB + NB;
    static_assert(NRED <= 64, "one lane per value");
    __shared__ __attribute__((aligned(16))) double s_x[2][NW][NRED];
    __shared__ double s_ri[2 * N * N];
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int gl = (int)threadIdx.x;
    const int64_t b = blockIdx.x / a.gx;
    const int wgi = (int)(blockIdx.x - b * a.gx);
    const int m = a.m, S = a.S, gx = a.gx;
    const T *qsrc = (const T *)a.ws.qthin + b * (int64_t)N * m;
    const T *gsrc = (const T *)a.ws.g + b * (int64_t)P * m;
    const double *small = a.ws.small + b * mrhs_small_stride<N, P>();
    T q[N][RW], g[P > 0 ? P : 1][RW];
#pragma unroll
    for (int j = 0; j < N; ++j) load_rows<T, RW, NW>(qsrc + (int64_t)j * m, m, gl, true, q[j]);
#pragma unroll
    for (int p = 0; p < P; ++p) load_rows<T, RW, NW>(gsrc + (int64_t)p * m, m, gl, true, g[p]);
    if (threadIdx.x < N * N) {
        s_ri[threadIdx.x] = small[threadIdx.x];
        s_ri[N * N + threadIdx.x] = small[N * N + P * P + threadIdx.x];
    }
    const int stA = a.ws.statusA[b];
    const bool truncated = uni(small[2 * N * N + P * P] != 0.0);
    // which coefficient scales pair p, as a one-hot row (a select chain over the coefficient array would be folded into a
    // dynamically indexed vector kept in scratch)
    T ohp[P > 0 ? P : 1][N];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j2 = 0; j2 < N; ++j2) ohp[p][j2] = (a.pb[p] == j2) ? T(1) : T(0);
    __syncthreads();
    const int nbatch = (S + NB - 1) / NB;
    const T *ybase = a.yw + b * (int64_t)S * m;
    T ynext[NB][RW];
    auto load_batch = [&](const int bt) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int s = bt * NB + c;
            if (s < S) {
                load_rows<T, RW, NW>(ybase + (int64_t)s * m, m, gl, true, ynext[c]);
            } else {
#pragma unroll
                for (int r = 0; r < RW; ++r) ynext[c][r] = T(0);
            }
        }
    };
    T r2prev[NB];
    unsigned badprev = 0u;
#pragma unroll
    for (int c = 0; c < NB; ++c) r2prev[c] = T(0);
    int ph = 0, prev_bt = -1;
    auto emit_prev = [&](const T tot, const int bt) __attribute__((always_inline)) {
        if (bt < 0) return;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int s = bt * NB + c;
            if (s >= S) continue;
            const T r2 = readlane(tot, N * NB + c);
            const bool ok = is_finite(r2) && stA == VP_ST_OK && ((badprev >> c) & 1u) == 0u;
            if (wave == 0 && lane == 0) {
                const int64_t prob = b * S + s;
                if (a.cost_bs) a.cost_bs[prob] = 0.5 * (double)r2;
                if (a.status_bs) a.status_bs[prob] = ok ? VP_ST_OK : (stA != VP_ST_OK ? stA : VP_ST_NONFINITE);
            }
        }
    };
    int bt = wgi;
    if (bt < nbatch) load_batch(bt);
    for (; bt < nbatch; bt += gx) {
        T y[NB][RW];
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int r = 0; r < RW; ++r) y[c][r] = ynext[c][r];
        if (bt + gx < nbatch) load_batch(bt + gx); // in flight while this batch is computed and written
        T red[NRED];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T acc = T(0);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc = tfma(q[j][r], y[c][r], acc);
                red[c * N + j] = acc;
            }
            red[N * NB + c] = r2prev[c];
        }
        wave_reduce_store_g<NRED>(red, &s_x[ph][wave][0]);
        __syncthreads();
        T tot = T(0);
        if (lane < NRED) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += s_x[ph][w][lane];
        }
        ph ^= 1;
        emit_prev(tot, prev_bt);
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int s = bt * NB + c;
            const bool valid = s < S; // (uniform)
            T tq[N], cc[N], tr[N];
#pragma unroll
            for (int j = 0; j < N; ++j) tq[j] = readlane(tot, c * N + j);
#pragma unroll
            for (int i2 = 0; i2 < N; ++i2) {
                T acc = T(0), accp = T(0);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (truncated || j >= i2) acc = tfma((T)s_ri[i2 * N + j], tq[j], acc);
                    if (truncated) accp = tfma((T)s_ri[N * N + i2 * N + j], tq[j], accp);
                }
                cc[i2] = acc;
                tr[i2] = truncated ? accp : tq[i2];
            }
            bool cfin = true;
#pragma unroll
            for (int i2 = 0; i2 < N; ++i2) cfin = cfin && is_finite(cc[i2]);
            badprev = (badprev & ~(1u << c)) | (uni(cfin) ? 0u : (1u << c));
            T r2 = T(0);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                T v = y[c][r];
#pragma unroll
                for (int j = 0; j < N; ++j) v = tfma(-tr[j], q[j][r], v);
                y[c][r] = v;
                r2 = tfma(v, v, r2);
            }
            r2prev[c] = valid ? r2 : T(0);
            if (!valid) continue;
            const int64_t prob = b * S + s;
            if (a.C_out && wave == 0) {
#pragma unroll
                for (int j2 = 0; j2 < N; ++j2)
                    if (lane == j2) a.C_out[prob * N + j2] = cc[j2];
            }
// (non-temporal stores, vp_kernels.hpp store_rows<..., NT>: configs[2]'s trait evaluation -- 1.34 GB of R and J out -- 0.268 -> 0.222 ms,
// 0.63 -> 0.76 of 8 TB/s, 25 launches per build on one box, alternating: tools/mrhs_trait_probe.py)
#ifndef VP_MRHS_NT
#define VP_MRHS_NT 1
#endif
            if (a.r_out) store_rows<T, RW, NW, VP_MRHS_NT != 0>(a.r_out + prob * (int64_t)m, m, gl, true, y[c]);
            if (a.J_out) {
                T cp[P > 0 ? P : 1];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    T v = T(0);
#pragma unroll
                    for (int j2 = 0; j2 < N; ++j2) v = tfma(ohp[p][j2], cc[j2], v);
                    cp[p] = -v;
                }
                for (int k2 = 0; k2 < a.q; ++k2) {
                    T jk[RW];
#pragma unroll
                    for (int r = 0; r < RW; ++r) jk[r] = T(0);
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        if (a.pp[p] == k2) {
#pragma unroll
                            for (int r = 0; r < RW; ++r) jk[r] = tfma(cp[p], g[p][r], jk[r]);
                        }
                    }
                    store_rows<T, RW, NW, VP_MRHS_NT != 0>(a.J_out + ((b * a.q + k2) * (int64_t)S + s) * (int64_t)m, m, gl, true, jk);
                }
            }
        }
        prev_bt = bt;
    }
    { // flush: the squared norms of the last batch
        T red[NRED];
#pragma unroll
        for (int v = 0; v < NRED; ++v) red[v] = T(0);
#pragma unroll
        for (int c = 0; c < NB; ++c) red[N * NB + c] = r2prev[c];
        wave_reduce_store_g<NRED>(red, &s_x[ph][wave][0]);
        __syncthreads();
        T tot = T(0);
        if (lane < NRED) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += s_x[ph][w][lane];
        }
        emit_prev(tot, prev_bt);
    }
}

template <typename T, int N, int Q, int P> struct MrhsLmArgs {
    MrhsWs ws;
    LmOpts<T> opts;
    const T *alpha0;   // init only
    const MrhsIo *io;  // init only: != null and io->alpha_in != null: read the initial parameters from there instead
    int pb[P > 0 ? P : 1], pp[P > 0 ? P : 1];
    int m, S;
    int64_t S_global;  // right-hand sides of the whole problem (== S unless the columns are sharded over ranks)
    int64_t B;
    int init;          // 1: initialise the state from alpha0 and publish the first trial point
    int gx;            // workgroups per problem of the streaming kernel (partial-sum slots)
    double *trace;     // [B][trace_rows][q+4] or null
    int trace_rows;
};

// ONE kernel between two streaming passes of the fit: the LM step on the sums the pass left, then -- unless the loop has
// terminated -- the factorisation of the new trial point, by the same group of W waves (W = 4 when few problems make both
// pure latency on the critical path of every iteration; the LM arithmetic is wave-uniform and every wave carries it).
// Everything the step reads from global memory is REQUESTED before anything is waited for: the LM state, the partial sums
// (one record per thread in flight, the whole group reduces), G^T G, the status and the buffer index -- one round trip
// instead of five dependent ones (phase clocks at configs[2]: 4.6 us of a 10 us LM step were the serial loop over the 512
// partial records, 4.2 us the dependent loads of the accept / Gram phase).  `init`: set the state up and factorise alpha0.
template <typename T, class M, int R, int W>
__global__ void __launch_bounds__(64 * W) mrhs_step_kernel(const MrhsFactorArgs<T, M> fa, const MrhsLmArgs<T, M::N, M::Q, M::P> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    using G = Grp<W>;
    __shared__ __attribute__((aligned(16))) unsigned char s_xch[group_xch_bytes<W>() > 0 ? group_xch_bytes<W>() : 16];
    G grp = G::make(s_xch);
    const int64_t b = blockIdx.x;
    if (b >= a.B) return;
    const int gl = grp.gl;
    const MrhsSrc<T, R, W> src = mrhs_make_src<T, M, R, W>(fa, b, gl);
    using Vars = LmVars<T, N, Q>;
    Vars *gs = reinterpret_cast<Vars *>(a.ws.lm_state) + b;
    T *trial = (T *)a.ws.alpha_trial + b * Q;
    constexpr int NACC = 1 + N * N + P;
#ifdef VP_MRHS_STEP_CLOCKS
    long long clkv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    Vars s;
    if (a.init) {
        T a0[Q];
        const T *src0 = a.alpha0;
        if (a.io) {
            const T *user = (const T *)a.io->alpha_in;
            if (user) src0 = user;
        }
#pragma unroll
        for (int k = 0; k < Q; ++k) a0[k] = src0[b * Q + k];
        lm_init<T, N, Q>(s, a0);
        if (gl == 0) {
            *gs = s;
#pragma unroll
            for (int k = 0; k < Q; ++k) trial[k] = s.xt[k];
            if (b == 0) { // (every decrement happens in a later launch)
                a.ws.nactive[0] = (int32_t)a.B;
                a.ws.nactive[1] = 0;
            }
            a.ws.done[b] = 0;
            a.ws.widx[b] = 0;
            a.ws.bidx[b] = 0;
            if (a.ws.jcond) a.ws.jcond[b] = 0.0;
        }
    } else {
#ifdef VP_MRHS_STEP_CLOCKS
    clkv[0] = (long long)__builtin_amdgcn_s_memtime();
#endif
    // ---- every load first ----
    s = *gs;
    const int stA = a.ws.statusA[b];
    const int wold = a.ws.widx[b];
    const double *small = a.ws.small + b * mrhs_small_stride<N, P>();
    double gtg[P > 0 ? P * P : 1];
#pragma unroll
    for (int i = 0; i < P * P; ++i) gtg[i] = small[N * N + i];
    double acc[NACC];
    {
        // accumulator-major partials [NACC][gx]: group lane g reads workgroup g's (and g + 64 W's) value of every sum -- each
        // wave-level load is 512 contiguous bytes (record-major, 64 lanes x 160-byte stride = 64 cache lines per
        // instruction, made this phase 4 us); two values per lane and trip in flight, then one group all-reduce
        const double *part = a.ws.acc + (size_t)b * a.gx * NACC;
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        constexpr int NT = 64 * W;
        for (int g = gl; g < a.gx; g += 2 * NT) {
            const bool two = g + NT < a.gx;
            const int g1 = two ? g + NT : g;
            double v0[NACC], v1[NACC];
#pragma unroll
            for (int i = 0; i < NACC; ++i) v0[i] = part[(size_t)i * a.gx + g];
#pragma unroll
            for (int i = 0; i < NACC; ++i) v1[i] = part[(size_t)i * a.gx + g1];
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] += v0[i] + (two ? v1[i] : 0.0);
        }
    }
    if (uni(s.term != 0)) return; // finished earlier (uniform over the group)
    // group all-reduce of the NACC sums, VP_XV values per exchange
#pragma unroll
    for (int c0 = 0; c0 < NACC; c0 += VP_XV) {
        constexpr int CH = VP_XV;
        double part[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) part[i] = (c0 + i < NACC) ? acc[c0 + i < NACC ? c0 + i : 0] : 0.0;
        group_allreduce(grp, part);
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c0 + i < NACC) acc[c0 + i] = part[i];
    }
    const T cost2 = (T)acc[0];
#ifdef VP_MRHS_STEP_CLOCKS
    clkv[1] = (long long)__builtin_amdgcn_s_memtime() + (cost2 == T(-1.2345) ? 1 : 0);
#endif
    const bool ok = uni(stA == VP_ST_OK && is_finite(cost2));
    const T fnorm1 = tsqrt(cost2);
    const bool first_eval = s.first != 0;
    const bool need_jac = lm_after_eval<T, N, Q, true>(s, a.opts, fnorm1, ok, (long)a.m * (long)a.S_global);
    if (gl == 0 && (s.accepted || first_eval)) { // the pass that just ran evaluated the new best point: keep its buffer
        const int w = wold & 1;
        a.ws.bidx[b] = w;
        a.ws.widx[b] = w ^ 1;
    }
    if (a.trace && gl == 0 && s.nfev - 1 < a.trace_rows) {
        double *tr = a.trace + ((size_t)b * a.trace_rows + (s.nfev - 1)) * (Q + 4);
        for (int k = 0; k < Q; ++k) tr[k] = (double)s.xt[k];
        tr[Q] = (double)fnorm1;
        tr[Q + 1] = 0.0 / 0.0;
        tr[Q + 2] = (double)s.delta;
        tr[Q + 3] = (double)s.par;
    }
    if (need_jac) {
        // J^T J and J^T r are assembled and factored in DOUBLE for both dtypes (the accumulators are double): an fp32
        // handle would otherwise square the conditioning of J in fp32 and drop columns from cond(J) ~ 3e3 on
        double A[Q][Q], bv[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            bv[k] = 0.0;
#pragma unroll
            for (int l = 0; l < Q; ++l) A[k][l] = 0.0;
        }
        if constexpr (M::kDiagonalPairs) { // pair p <-> (basis p, parameter p): everything static
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                bv[k] = -acc[1 + N * N + k];
#pragma unroll
                for (int l = 0; l < Q; ++l) A[k][l] = acc[1 + k * N + l] * gtg[k * P + l];
            }
        } else {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int kp = a.pp[p], jp = a.pb[p];
            dyn_set_o<Q, true>(bv, kp, dyn_get_o<Q, true>(bv, kp) - acc[1 + N * N + p]);
#pragma unroll
            for (int p2 = 0; p2 < P; ++p2) {
                const int kp2 = a.pp[p2], jp2 = a.pb[p2];
                double ccv = acc[1]; // sum c_jp c_jp2 (a select chain: a run-time index would pin acc[] in scratch)
#pragma unroll
                for (int e = 1; e < N * N; ++e) ccv = (jp * N + jp2 == e) ? dyn_opq(acc[1 + e]) : ccv;
                const double contrib = ccv * gtg[p * P + p2];
#pragma unroll
                for (int k = 0; k < Q; ++k)
#pragma unroll
                    for (int l = 0; l < Q; ++l)
                        if (k == kp && l == kp2) A[k][l] += contrib;
            }
        }
        }
        double Rd[Q][Q], acd[Q], qd[Q];
        gram_to_qr<double, Q>(A, bv, Rd, acd, s.ipvt, qd);
        if (a.ws.jcond && gl == 0) {
            // pivoted factor of the column-NORMALISED Jacobian: |R_kk| / ||J_pivot(k)||; the ratio of the largest to the
            // smallest is cond(J D^-1) to within a small factor (rank-revealing pivoting); 1e300 for a dropped column
            double dmx = 0.0, dmn = 1e300;
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                const double an = dyn_get_o<Q, true>(acd, s.ipvt[k]);
                const double d = (an > 0.0) ? fabs(Rd[k][k]) / an : 0.0;
                dmx = d > dmx ? d : dmx;
                dmn = d < dmn ? d : dmn;
            }
            const double est = (dmn > 0.0) ? dmx / dmn : 1e300;
            if (est > a.ws.jcond[b]) a.ws.jcond[b] = est;
        }
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            s.acnorm[k] = (T)acd[k];
            s.qtf[k] = (T)qd[k];
#pragma unroll
            for (int l = 0; l < Q; ++l) s.Rj[k][l] = (T)Rd[k][l];
        }
    }
    lm_next_step<T, N, Q, true>(s, a.opts, need_jac);
    if (gl == 0) {
        *gs = s;
#pragma unroll
        for (int k = 0; k < Q; ++k) trial[k] = s.xt[k];
        if (s.term != 0) {
            atomicAdd(a.ws.nactive, -1);
            atomicMax(a.ws.nactive + 1, s.nfev);
            a.ws.done[b] = 1;
        }
    }
    if (uni(s.term != 0)) return;
    }
    T xt[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) xt[k] = s.xt[k];
#ifdef VP_MRHS_STEP_CLOCKS
    mrhs_factor_body<T, M, R, W>(fa, b, xt, grp, src, clkv);
    __syncthreads();
    if (a.trace && gl == 0 && s.nfev - 1 < a.trace_rows && !a.init) { // the row of this step: phase durations in ticks
        clkv[7] = (long long)__builtin_amdgcn_s_memtime();
        double *tr = a.trace + ((size_t)b * a.trace_rows + (s.nfev - 1)) * (Q + 4);
        for (int k = 0; k < 7 && k < Q + 4; ++k) tr[k] = (double)(clkv[k + 1] - clkv[k]);
    }
#else
    mrhs_factor_body<T, M, R, W>(fa, b, xt, grp, src);
#endif
}

} // namespace vp

// ---- host-side launchers ---------------------------------------------------------------------------
namespace vp {

template <class M> inline void pair_maps(const vp_model_desc &d, int (&pb)[M::P > 0 ? M::P : 1], int (&pp)[M::P > 0 ? M::P : 1]) {
    int p = 0;
    for (int j = 0; j < d.n_basis; ++j)
        for (int a = 0; a < VP_MAX_BASIS_PARAMS; ++a)
            if (d.param[j][a] >= 0 && p < M::P) {
                pb[p] = j;
                pp[p] = d.param[j][a];
                ++p;
            }
}

// workgroups per problem of the streaming kernel: one persistent 8-wave workgroup per CU at most (its LDS copy
// of Q and G is ~100 KiB)
constexpr int VP_MRHS_GX_MAX = 512; // partial-sum slots per problem the handle allocates
inline int mrhs_gx(int64_t S, int cap = 256) {
    int64_t gx = (S + 7) / 8;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return (int)gx;
}
// workgroups per problem of the MODE 0 streaming pass of a (scalar type, rows per lane) kernel set: the LDS-DMA kernel runs
// two 4-wave workgroups per CU, the one-wave-per-column kernel one 8-wave workgroup
template <typename T, int R> constexpr int mrhs_gx_cap() {
#ifndef VP_NO_MRHS_DMA
    return (sizeof(T) == 8 && R == 32) ? 512 : 256;
#else
    return 256;
#endif
}

template <typename T, class M> inline bool fill_factor_args(const LaunchParams &p, MrhsFactorArgs<T, M> &a) {
    if (!bind_model(*p.model, a.mdl)) return false;
    a.t = (const T *)p.t;
    a.w = (const T *)p.w;
    a.alpha = (const T *)p.alpha;
    a.ws = *reinterpret_cast<const MrhsWs *>(p.mrhs_fws ? p.mrhs_fws : p.mrhs_ws);
    a.m = p.m;
    a.B = p.B;
    a.t_stride = p.t_stride;
    a.w_stride = p.w_stride;
    a.eps = (T)p.eps;
    a.skip_done = (p.mrhs_mode == 0) ? 1 : 0; // fit loop (reduced sums) vs trait-level evaluation
    a.grid_uniform = p.grid_uniform;
    return true;
}

// the N + 1 + P columns of R rows per lane do not fit the 512 VGPRs of a wave running alone on its SIMD
template <typename T, class M, int R> constexpr bool mrhs_one_wave_spills() {
    return (M::N + 1 + M::P) * R * (int)(sizeof(T) / 4) > 400;
}

template <typename T, class M, int R> int launch_mrhs_factor(const LaunchParams &p) {
    MrhsFactorArgs<T, M> a;
    if (!fill_factor_args<T, M>(p, a)) return VP_ERR_UNSUPPORTED;
    // few problems: the factorisation is pure latency -> 4 waves per problem (R/4 rows per lane); ALWAYS four waves where
    // the columns of a problem do not fit the registers of one (the one-wave form of the triple exponential at 32 rows per
    // lane spilled 596 VGPRs)
    if constexpr (R % 8 == 0) { // (R / 4 rows per lane, in pairs)
        if (a.B <= 1024 || mrhs_one_wave_spills<T, M, R>()) {
            hipLaunchKernelGGL((mrhs_factor_kernel<T, M, R / 4, 4>), dim3((unsigned)a.B), dim3(256), 0, p.stream, a);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
    }
    if constexpr (!(R % 8 == 0 && mrhs_one_wave_spills<T, M, R>()))
        hipLaunchKernelGGL((mrhs_factor_kernel<T, M, R, 1>), dim3((unsigned)a.B), dim3(64), 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

template <typename T, class M, int R> int launch_mrhs_stream(const LaunchParams &p) {
    constexpr int N = M::N, P = M::P;
    MrhsStreamArgs<T, N, P> a;
    a.yw = (const T *)p.yw;
    a.ws = *reinterpret_cast<const MrhsWs *>(p.mrhs_ws);
    a.r_out = (T *)p.r_out;
    a.J_out = (T *)p.J_out;
    a.C_out = (T *)p.C_out;
    a.cost_bs = p.cost_out;
    a.status_bs = p.status;
    pair_maps<M>(*p.model, a.pb, a.pp);
    a.q = M::Q;
    a.m = p.m;
    a.S = p.S;
    a.B = p.B;
    // MODE 0 leaves one partial-sum record per workgroup for the LM step: the grid must be the (T, R) kernel set's own
    // mrhs_gx_cap; MODE 1 has no such coupling
    const int gx = (p.mrhs_mode == 0) ? mrhs_gx(p.S, mrhs_gx_cap<T, R>()) : mrhs_gx(p.S);
    a.gx = gx;
    if ((int64_t)gx * p.B > 0x7fffffffLL) return VP_ERR_UNSUPPORTED;
#ifndef VP_NO_MRHS_DMA
    if constexpr (sizeof(T) == 8 && R == 32) {
        // 16-byte row pairs (m even, aligned bases); NW waves share a column (R / NW rows per lane), NB columns per batch
        // (8 DMA instructions per batch and wave), ring of 2 batches per wave; two workgroups per CU
        const bool vec = (p.m & 1) == 0 && ((reinterpret_cast<uintptr_t>(a.yw) | reinterpret_cast<uintptr_t>(a.ws.qthin) |
                                             reinterpret_cast<uintptr_t>(a.ws.g)) & 15) == 0;
#ifndef VP_NO_MRHS_COOP_OUT
        if (p.mrhs_mode != 0 && vec && (a.r_out || a.J_out) && (reinterpret_cast<uintptr_t>(a.r_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.J_out) & 15) == 0) {
            constexpr int NWd = 4, RWd = R / NWd;
            const int gxo = mrhs_gx(p.S, 512);
            a.gx = gxo;
            if (a.J_out)
                hipLaunchKernelGGL((mrhs_coop_out_kernel<T, N, P, RWd, NWd, 2, 1>), dim3((unsigned)((int64_t)gxo * p.B)), dim3(64 * NWd), 0, p.stream, a);
            else
                hipLaunchKernelGGL((mrhs_coop_out_kernel<T, N, P, RWd, NWd, 1, 2>), dim3((unsigned)((int64_t)gxo * p.B)), dim3(64 * NWd), 0, p.stream, a);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
#endif
        if (p.mrhs_mode == 0 && vec) {
            constexpr int NWd = 4, RWd = R / NWd, NBd = 16 / RWd;
            constexpr int NXd = (N * NBd + NBd) > (P * N + NBd) ? (N * NBd + NBd) : (P * N + NBd);
            const size_t dlds = (size_t)NWd * 2 * NBd * (RWd / 2) * 1024 +
                                (size_t)(2 * NWd * NXd + 2 * N * N + 16 * NBd * N + 16 * NBd + N * N + P * N) * 8 + 16 * NBd * 4 + 64;
            if (hipFuncSetAttribute((const void *)mrhs_coop_dma_kernel<T, N, P, RWd, NWd, NBd>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds) != hipSuccess)
                return VP_ERR_HIP;
            hipLaunchKernelGGL((mrhs_coop_dma_kernel<T, N, P, RWd, NWd, NBd>), dim3((unsigned)((int64_t)gx * p.B)), dim3(64 * NWd), dlds,
                               p.stream, a);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
    }
#endif
    const size_t lds = (size_t)(N + P) * 64 * R * sizeof(T) + (size_t)N * N * sizeof(T);
    const int waves_per_wg = VP_MRHS_WAVES;
    dim3 grid((unsigned)((int64_t)gx * p.B)), block(64 * waves_per_wg);
    hipError_t e;
    if (p.mrhs_mode == 0) {
        e = hipFuncSetAttribute((const void *)mrhs_stream_kernel<T, N, P, R, 0>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return VP_ERR_HIP;
        hipLaunchKernelGGL((mrhs_stream_kernel<T, N, P, R, 0>), grid, block, lds, p.stream, a);
    } else {
        e = hipFuncSetAttribute((const void *)mrhs_stream_kernel<T, N, P, R, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return VP_ERR_HIP;
        hipLaunchKernelGGL((mrhs_stream_kernel<T, N, P, R, 1>), grid, block, lds, p.stream, a);
    }
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// the LM step + the factorisation of the next trial point (mrhs_step_kernel); p.mrhs_ws = the workspace the LM step reads
// its sums from, p.mrhs_fws (or p.mrhs_ws when null) the one the factorisation writes
template <typename T, class M, int R> int launch_mrhs_lm(const LaunchParams &p) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    MrhsFactorArgs<T, M> fa;
    if (!fill_factor_args<T, M>(p, fa)) return VP_ERR_UNSUPPORTED;
    fa.skip_done = 1;
    MrhsLmArgs<T, N, Q, P> a;
    a.ws = *reinterpret_cast<const MrhsWs *>(p.mrhs_ws);
    a.opts.ftol = (T)p.opts->ftol;
    a.opts.xtol = (T)p.opts->xtol;
    a.opts.gtol = (T)p.opts->gtol;
    a.opts.stepbound = (T)p.opts->stepbound;
    a.opts.patience = p.opts->patience;
    a.opts.scale_diag = p.opts->scale_diag;
    a.alpha0 = (const T *)p.alpha;
    a.io = (const MrhsIo *)p.mrhs_io;
    pair_maps<M>(*p.model, a.pb, a.pp);
    a.m = p.m;
    a.S = p.S;
    a.B = p.B;
    a.S_global = p.mrhs_S_global > 0 ? p.mrhs_S_global : p.S;
    a.init = p.mrhs_init;
    a.gx = p.mrhs_gx > 0 ? p.mrhs_gx : mrhs_gx(p.S, mrhs_gx_cap<T, R>());
    a.trace = p.trace;
    a.trace_rows = p.trace_rows;
    // few problems: LM step and factorisation are pure latency -> 4 waves per problem (R/4 rows per lane)
    if constexpr (R % 8 == 0) { // (R / 4 rows per lane, in pairs)
        if (a.B <= 1024 || mrhs_one_wave_spills<T, M, R>()) {
            hipLaunchKernelGGL((mrhs_step_kernel<T, M, R / 4, 4>), dim3((unsigned)a.B), dim3(256), 0, p.stream, fa, a);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
    }
    if constexpr (!(R % 8 == 0 && mrhs_one_wave_spills<T, M, R>()))
        hipLaunchKernelGGL((mrhs_step_kernel<T, M, R, 1>), dim3((unsigned)p.B), dim3(64), 0, p.stream, fa, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// bytes of one LmVars record (the handle allocates B of them) and final-state extraction
template <typename T, class M> size_t mrhs_state_bytes() { return sizeof(LmVars<T, M::N, M::Q>); }

template <typename T, int N, int Q>
__global__ void mrhs_finish_kernel(const LmVars<T, N, Q> *st, int64_t B, T *alpha_out, vp_report *rep, const int32_t *nactive,
                                   int32_t *hflag, const MrhsIo *io) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b == 0 && hflag) { // pinned host memory: what the host looks at after the stream has drained
        hflag[0] = nactive[0];
        hflag[1] = nactive[1];
    }
    if (b >= B) return;
    T *ua = nullptr;
    vp_report *ur = nullptr;
    if (io) {
        ua = (T *)io->alpha_out;
        ur = (vp_report *)io->rep_out;
    }
    const LmVars<T, N, Q> s = st[b];
    for (int k = 0; k < Q; ++k) {
        alpha_out[b * Q + k] = s.x[k];
        if (ua) ua[b * Q + k] = s.x[k];
    }
    vp_report r;
    r.termination = s.term;
    r.n_evals = s.nfev;
    r.objective = (double)s.objective;
    rep[b] = r;
    if (ur) ur[b] = r;
}

// (C, cost, status) of every column at the final parameters = the best point's buffer
template <typename T>
__global__ void mrhs_gather_kernel(const MrhsWs ws, int64_t B, int S, int n, T *C_out, double *cost_bs, int32_t *status_bs,
                                   const int gx, const MrhsIo *io) {
    const int64_t b = blockIdx.x / gx;
    const int wgi = (int)(blockIdx.x - b * gx), nwg = gx;
    if (b >= B) return;
    const int sel = ws.bidx[b] & 1;
    const T *cs = (const T *)ws.cbuf[sel] + b * (int64_t)S * n;
    T *uc = io ? (T *)io->C_out : nullptr; // the caller's array as well (device-pointer handles, whole-fit graph)
    for (int64_t i = wgi * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)S * n; i += (int64_t)nwg * blockDim.x) {
        const T v = cs[i];
        C_out[b * (int64_t)S * n + i] = v;
        if (uc) uc[b * (int64_t)S * n + i] = v;
    }
    for (int64_t i = wgi * (int64_t)blockDim.x + threadIdx.x; i < S; i += (int64_t)nwg * blockDim.x) {
        cost_bs[b * S + i] = ws.costbuf[sel][b * S + i];
        status_bs[b * S + i] = ws.stbuf[sel][b * S + i];
    }
}

template <typename T, class M, int R> int launch_mrhs_finish(const LaunchParams &p) {
    const MrhsWs &ws = *reinterpret_cast<const MrhsWs *>(p.mrhs_ws);
    const unsigned grid = (unsigned)((p.B + 63) / 64);
    hipLaunchKernelGGL((mrhs_finish_kernel<T, M::N, M::Q>), dim3(grid), dim3(64), 0, p.stream,
                       (const LmVars<T, M::N, M::Q> *)ws.lm_state, p.B, (T *)p.alpha_out, p.report, (const int32_t *)ws.nactive,
                       p.mrhs_hflag, (const MrhsIo *)p.mrhs_io);
    if (p.C_out && p.cost_out && p.status) {
        const int64_t per = (int64_t)p.S * M::N;
        unsigned gx = (unsigned)((per + 255) / 256);
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL((mrhs_gather_kernel<T>), dim3((unsigned)((int64_t)gx * p.B)), dim3(256), 0, p.stream, ws, p.B, p.S, (int)M::N,
                           (T *)p.C_out, p.cost_out, p.status, (int)gx, (const MrhsIo *)p.mrhs_io);
    }
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
