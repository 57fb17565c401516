// vp_ext.hpp -- register-resident evaluation for CALLER-EVALUATED models (vp_batch_create_external).
//
// The reference's plugin boundary is a trait: any `SeparableNonlinearModel` (src/model/mod.rs:239-363) -- the closure-based
// `SeparableModel` (:441-512) in particular -- works with its solver.  A model the descriptor language cannot express
// crosses the C ABI as the VALUES the trait returns: Phi = eval() and the non-zero columns of eval_partial_deriv(k).
// This kernel is `evaluate_kernel` (vp_kernels.hpp) with the column BUILD replaced by column LOADS:
//     Phi_w = W Phi, D_w = W dPhi     src/solvers/levmar/mod.rs:47, 141
//     Householder QR of Phi_w applied to [y_w | D_w], truncated solve, ||r||^2          :51-59  (house_qr, solve_coeffs)
//     r = Q [e; (Q^T y)_{>=n}],  J_k = -Q [0; sum_{pairs p of k} c_j(p) (Q^T D_p)_{>=n}]  :91-95, 101-201 (apply_q_cols)
// One wavefront per (problem, right-hand side); the columns live in registers (R rows per lane), every global access is
// 16 bytes per lane when the arrays allow it.  It streams T*m*(n + p + 1) bytes in and T*m*(1 + q) out per
// problem-evaluation and is HBM-bound.
//
// Shapes: compile-time (N, P, R); the parameter count q and the pair table are run-time (J_k is assembled from the P
// back-transformed derivative columns when it is stored), and a model with fewer pairs than P runs with zero columns that
// are never loaded.  Longer problems (round 5): W = 4 / 8 / 16 waves per problem, fp64 to 8 192 rows, fp32 to 16 384.
// Beyond those lengths: the streamed kernels of vp_blk_ext.hpp (any m, two passes over the caller's columns); shapes in
// neither table run on the generic kernels (vp_generic.hpp).
#pragma once
#include <vector>

#include "vp_kernels.hpp"

namespace vp {
namespace ext {

template <typename T> struct ExtArgs {
    const T *phi;   // [B][N][m]  UNWEIGHTED
    const T *dphi;  // [B][np][m] UNWEIGHTED, pair-table order (null: no derivative columns)
    const T *w;     // null, [m] or [B][m]
    const T *yw;    // [B][S][m] weighted data
    T *r_out, *J_out, *C_out;
    double *cost_out;
    int32_t *status;
    int32_t pb[VP_MAX_PAIRS], pp[VP_MAX_PAIRS];
    int np; // pairs of the model (<= P of the instantiation)
    int q;
    int m, S;
    int64_t nprob; // B*S
    int64_t w_stride;
    T eps;
    int vec; // every array 16-byte aligned and m even: 2-element accesses
};

// waves per SIMD the register allocator must leave room for: the NC resident columns plus the wave-uniform state of the
// factorisation (R, R^-1, Q^T y, c, e, the reflector's dot products: ~2N^2 + 6N values -- gfx950 has no scalar fp64 registers, they sit in VGPRs)
// W > 1 (round 5): the columns of ONE problem spread over the W waves of a workgroup (Layout<R, W>, reductions through the
// group's LDS exchange area) -- problems longer than one wave's registers stay ONE pass over the caller's columns: a
// workgroup of W waves must be resident on one CU, i.e. at least W / 4 waves per SIMD
template <typename T, int R, int N, int NC, int W = 1> constexpr int ext_waves() {
    constexpr int words = (NC * R + 2 * N * N + 6 * N + (W > 1 ? 24 : 0)) * (int)(sizeof(T) / 4);
    constexpr int fit = words <= 100 ? 4 : (words <= 200 ? 2 : 1);
    return fit > W / 4 ? fit : (W / 4 > 0 ? W / 4 : 1);
}

template <typename T, int N, int P, int R, bool WITH_D, int W = 1>
__global__ void __launch_bounds__(64 * W, (ext_waves<T, R, N, N + 1 + (WITH_D ? P : 0), W>())) ext_evaluate_kernel(const ExtArgs<T> a) {
    constexpr int NC = N + 1 + (WITH_D ? P : 0);
    using G = Grp<W>;
    using L = Layout<R, W>;
    extern __shared__ __attribute__((aligned(16))) unsigned char ext_smem[];
    G grp = G::make(W > 1 ? ext_smem : nullptr);
    const int lane = grp.gl; // GROUP lane: row ownership
    const int64_t prob = blockIdx.x; // problem * S + rhs
    if (prob >= a.nprob) return;
    const int64_t b = prob / a.S;
    const int s = (int)(prob - b * a.S);
    const int m = a.m;
    const bool vec = a.vec != 0;

    T C[NC][R];
    {
        const T *ph = a.phi + b * (int64_t)N * m;
#pragma unroll
        for (int j = 0; j < N; ++j) load_rows<T, R, W>(ph + (int64_t)j * m, m, lane, vec, C[j]);
        load_rows<T, R, W>(a.yw + prob * (int64_t)m, m, lane, vec, C[N]);
        if constexpr (WITH_D) {
            const T *dp = a.dphi + b * (int64_t)a.np * m;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (p < a.np) { // (uniform)
                    load_rows<T, R, W>(dp + (int64_t)p * m, m, lane, vec, C[N + 1 + p]);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) C[N + 1 + p][r] = T(0);
                }
            }
        }
        if (a.w) { // `&self.weights * ...` (src/util/weights.rs:82-99): row i of every model column times w_i
            T wt[R];
            load_rows<T, R, W>(a.w + b * a.w_stride, m, lane, vec, wt);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (j == N) continue; // y_w was weighted when the handle was made (src/problem/builder.rs:307)
#pragma unroll
                for (int r = 0; r < R; ++r) C[j][r] *= wt[r];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    T g[N], Rm[N][N], qty[N], c[N], e[N];
    house_qr<T, R, N, NC, 0, true, G>(C, g, Rm, qty, grp);
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, a.eps, c, e, truncated);
    // ||r||^2 = ||e||^2 + sum_{rows >= N} (Q^T y)^2
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

    if (lane == 0) {
        if (a.status) a.status[prob] = ok ? VP_ST_OK : VP_ST_NONFINITE;
        if (a.cost_out) a.cost_out[prob] = 0.5 * (double)fn2;
    }
    if (a.C_out && lane < N) a.C_out[prob * N + lane] = dyn_get<N>(c, lane);

    const bool want_j = WITH_D && a.J_out != nullptr;
    if (!a.r_out && !want_j) return;
    residual_qcoords<T, R, N>(C[N], e, grp);
    if constexpr (!WITH_D) {
        apply_q_cols<T, R, N, NC, N, N + 1>(C, g, grp);
        store_rows<T, R, W>(a.r_out + prob * (int64_t)m, m, lane, vec, C[N]);
    } else {
        // P_perp: the rows < N of the derivative columns in Q-coordinates do not enter the Kaufman columns
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < L::VW && r < R; ++r)
                if (L::row_of(r, lane) < N) C[N + 1 + p][r] = T(0);
        apply_q_cols<T, R, N, NC, N, NC>(C, g, grp); // [r~ | D~_1 .. D~_P] <- Q (.)  in ONE back-sweep
        if (a.r_out) store_rows<T, R, W>(a.r_out + prob * (int64_t)m, m, lane, vec, C[N]);
        if (a.J_out) {
            for (int k = 0; k < a.q; ++k) { // J[b][k][s][m] = -sum over the pairs p of parameter k of c_{basis(p)} Q D~_p
                T cj[P];
#pragma unroll
                for (int p = 0; p < P; ++p) cj[p] = (p < a.np && a.pp[p] == k) ? -dyn_get<N>(c, a.pb[p]) : T(0);
                T *jp = a.J_out + ((b * a.q + k) * (int64_t)a.S + s) * (int64_t)m;
                // assembled and stored row pair by row pair: no R-register accumulator column next to the resident ones
#pragma unroll
                for (int r0 = 0; r0 < R; r0 += L::VW) {
                    T v[2] = {T(0), T(0)};
#pragma unroll
                    for (int p = 0; p < P; ++p)
#pragma unroll
                        for (int x = 0; x < L::VW; ++x) v[x] = tfma(cj[p], C[N + 1 + p][r0 + x], v[x]);
                    const int i = L::row_of(r0, lane);
                    if (L::VW == 2 && vec) {
                        if (i < m) {
                            using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
                            V2 o;
                            o.x = v[0];
                            o.y = v[1];
                            *reinterpret_cast<V2 *>(jp + i) = o;
                        }
                    } else {
#pragma unroll
                        for (int x = 0; x < L::VW; ++x)
                            if (i + x < m) jp[i + x] = v[x];
                    }
                }
            }
        }
    }
}

template <typename T, int N, int P, int R, bool WITH_D, int W = 1> int launch_one(const ExtArgs<T> &a, hipStream_t stream) {
    hipLaunchKernelGGL((ext_evaluate_kernel<T, N, P, R, WITH_D, W>), dim3((unsigned)a.nprob), dim3(64 * W),
                       (size_t)group_xch_bytes<W>(), stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// one row of the table of compiled shapes (vp_inst_ext*.hip)
template <typename T> struct ExtEntry {
    int N, P, R;   // P == 0: the kernel without derivative columns (set_params / residuals only)
    int (*launch)(const ExtArgs<T> &, hipStream_t);
    int W = 1;     // waves per problem: the kernel holds 64 R W rows
};
template <typename T> std::vector<ExtEntry<T>> &ext_table() {
    static std::vector<ExtEntry<T>> t;
    return t;
}
template <typename T> struct ExtRegistrar {
    explicit ExtRegistrar(const ExtEntry<T> &e) { ext_table<T>().push_back(e); }
};

// the streamed kernels of vp_blk_ext.hpp (any m): one row per (N, P) shape; P == 0: the pass without derivative columns
template <typename T> struct ExtStreamEntry {
    int N, P;
    int (*launch)(const ExtArgs<T> &, hipStream_t);
};
template <typename T> std::vector<ExtStreamEntry<T>> &ext_stream_table() {
    static std::vector<ExtStreamEntry<T>> t;
    return t;
}
template <typename T> struct ExtStreamRegistrar {
    explicit ExtStreamRegistrar(const ExtStreamEntry<T> &e) { ext_stream_table<T>().push_back(e); }
};
template <typename T> const ExtStreamEntry<T> *find_ext_stream(int n, int np, bool with_d) {
    const ExtStreamEntry<T> *best = nullptr;
    for (const ExtStreamEntry<T> &e : ext_stream_table<T>()) {
        if (e.N != n) continue;
        if (with_d ? (e.P == 0 || e.P < np) : e.P != 0) continue;
        if (!best || e.P < best->P) best = &e;
    }
    return best;
}

// the resident kernel that covers (n, np pairs, m) with / without derivative columns, or null
template <typename T> const ExtEntry<T> *find_ext(int n, int np, int m, bool with_d) {
    const ExtEntry<T> *best = nullptr;
    for (const ExtEntry<T> &e : ext_table<T>()) {
        if (e.N != n || 64 * e.R * e.W < m) continue;
        if (with_d ? (e.P == 0 || e.P < np) : e.P != 0) continue;
        // the smallest capacity that holds the problem, on the fewest waves, with the fewest spare derivative columns
        const int cap = e.R * e.W, bcap = best ? best->R * best->W : 0;
        if (!best || cap < bcap || (cap == bcap && (e.W < best->W || (e.W == best->W && e.P < best->P)))) best = &e;
    }
    return best;
}

// evaluate entry of the external kernel set: the resident kernel when the shape is in the table, else the generic one
template <typename T> int launch_evaluate(const LaunchParams &p, int (*fallback)(const LaunchParams &)) {
    const int n = p.model->n_basis, q = p.model->n_params;
    const bool with_d = p.J_out != nullptr && p.ext_np > 0;
    const bool usable = p.ext_rows == p.m && p.m >= n && (!with_d || p.ext_dphi) && p.ext_phi;
    const ExtEntry<T> *e = usable ? find_ext<T>(n, p.ext_np, p.m, with_d) : nullptr;
    // no resident kernel holds the shape at this length: the streamed kernel (vp_blk_ext.hpp) if the shape has one
    const ExtStreamEntry<T> *es = (usable && !e) ? find_ext_stream<T>(n, p.ext_np, with_d) : nullptr;
    if (!e && !es) return fallback(p);
    if (p.J_out && !with_d && q > 0) {
        // a model without a single derivative column (every parameter unused): the kernel without derivative columns has no
        // Jacobian store, and J = 0 -- as eval_partial_deriv leaves it (src/model/mod.rs:473-512)
        if (hipMemsetAsync(p.J_out, 0, (size_t)p.B * (size_t)q * (size_t)p.S * (size_t)p.m * sizeof(T), p.stream) != hipSuccess)
            return VP_ERR_HIP;
    }
    ExtArgs<T> a;
    a.phi = (const T *)p.ext_phi;
    a.dphi = (const T *)p.ext_dphi;
    a.w = (const T *)p.w;
    a.yw = (const T *)p.yw;
    a.r_out = (T *)p.r_out;
    a.J_out = (T *)p.J_out;
    a.C_out = (T *)p.C_out;
    a.cost_out = p.cost_out;
    a.status = p.status;
    for (int i = 0; i < VP_MAX_PAIRS; ++i) {
        a.pb[i] = i < p.ext_np ? p.ext_pb[i] : 0;
        a.pp[i] = i < p.ext_np ? p.ext_pp[i] : -1;
    }
    a.np = p.ext_np;
    a.q = q;
    a.m = p.m;
    a.S = p.S;
    a.nprob = p.B * p.S;
    a.w_stride = p.w_stride;
    a.eps = (T)p.eps;
    a.vec = host_aligned<T>(p.m, {p.ext_phi, p.ext_dphi, p.w, p.yw, p.r_out, p.J_out}) ? 1 : 0;
    if (a.nprob <= 0) return VP_ERR_OK;
    if (e) return e->launch(a, p.stream);
    const int rc = es->launch(a, p.stream);
    return rc == VP_ERR_UNSUPPORTED ? fallback(p) : rc; // (more blocks than the LDS carry record holds)
}

} // namespace ext
} // namespace vp

#define VP_EXT_CAT_(a, b) a##b
#define VP_EXT_CAT(a, b) VP_EXT_CAT_(a, b)
// one shape with derivative columns (P >= 1) ...
#define VP_REGISTER_EXT(T, NN, PP, RR)                                                                                 \
    static ::vp::ext::ExtRegistrar<T> VP_EXT_CAT(vp_ext_reg_, __COUNTER__)(                                           \
        ::vp::ext::ExtEntry<T>{NN, PP, RR, &::vp::ext::launch_one<T, NN, PP, RR, true>});
// ... and the kernel without them
#define VP_REGISTER_EXT0(T, NN, RR)                                                                                    \
    static ::vp::ext::ExtRegistrar<T> VP_EXT_CAT(vp_ext_reg_, __COUNTER__)(                                           \
        ::vp::ext::ExtEntry<T>{NN, 0, RR, &::vp::ext::launch_one<T, NN, 1, RR, false>});
// the same pair for problems spread over WW waves (64 RR WW rows)
#define VP_REGISTER_EXT_W(T, NN, PP, RR, WW)                                                                           \
    static ::vp::ext::ExtRegistrar<T> VP_EXT_CAT(vp_ext_reg_, __COUNTER__)(                                           \
        ::vp::ext::ExtEntry<T>{NN, PP, RR, &::vp::ext::launch_one<T, NN, PP, RR, true, WW>, WW});
#define VP_REGISTER_EXT0_W(T, NN, RR, WW)                                                                              \
    static ::vp::ext::ExtRegistrar<T> VP_EXT_CAT(vp_ext_reg_, __COUNTER__)(                                           \
        ::vp::ext::ExtEntry<T>{NN, 0, RR, &::vp::ext::launch_one<T, NN, 1, RR, false, WW>, WW});
