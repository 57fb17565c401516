// vp_blk_extfit.hpp -- the batched reverse-communication LM fit of caller-evaluated models (vp_extfit.hpp) at ANY length.
//
// == LevMarSolver::fit (src/solvers/levmar/mod.rs:238-254) over any SeparableNonlinearModel (src/model/mod.rs:239-363) of any
// output_len() (:263).  ext_fit_eval_kernel holds the m rows of [Phi | y | dPhi] of one problem in the registers of one to four
// wavefronts (4 096 rows at most); here m is a run-time number: the caller's columns are streamed in blocks of 64 RB rows
// through the TSQR carry of vp_block.hpp (stacked_qr over ALL n + 1 + p columns: one forward pass, no recorded carries), and
// the evaluation + Jacobian factor are taken from the compressed problem exactly as blk_fit_kernel takes them -- T's leading
// n x n block and (T_y)[0:n] give c and e, row n of T_y the residual norm, rows >= n of T_D the Kaufman columns in
// Q-coordinates, MINPACK's pivoted qrfac on those two-rows-per-lane columns (jac_qrfac) the factor.  The results land in the
// candidate slot of the problem's LM record (ExtFitLayout C_*), which is all ext_fit_lm_kernel reads: the LM launch, the
// protocols (eager / VP_FIT_DERIVATIVES_ON_ACCEPT) and the host entry points are those of vp_extfit.hpp unchanged.
// The residual norm of a point does not depend on the protocol (with or without derivative columns in the step): see `fold`.
#pragma once
#include "vp_blk_ext.hpp"
#include "vp_extfit.hpp"

namespace vp {
namespace blk {

// waves per SIMD: the two resident blocks of the double buffer are 2 NC RB values; beyond 112 register words they leave a
// second wave no room (eight fp64 columns of four rows spilled 74-158 VGPRs at two waves per SIMD)
// rows per lane and block.  Shapes of up to 14 register words per row (seven fp64 columns): ONE resident block of 8 rows per
// lane -- the second wave of the SIMD loads while this one folds -- instead of two of 4 (a block's reduction rounds and
// reflector scalars cost the same whatever its height); wider shapes: the double buffer of ext_block_rows
#ifndef VP_EXTFIT_STREAM_SINGLE
#define VP_EXTFIT_STREAM_SINGLE 1
#endif
template <typename T, int NC> constexpr bool ext_fit_stream_single() {
    return VP_EXTFIT_STREAM_SINGLE && NC * (int)(sizeof(T) / 4) > 8 && NC * (int)(sizeof(T) / 4) <= 14;
}
template <typename T, int NC> constexpr int ext_fit_stream_rows() { return ext_fit_stream_single<T, NC>() ? 8 : ext_block_rows<T, NC>(); }
template <typename T, int NC, int RB> constexpr int ext_fit_stream_waves() {
    if (ext_fit_stream_single<T, NC>()) return 2;
    return (2 * NC * RB * (int)(sizeof(T) / 4) <= 112 && (NC <= 10 || sizeof(T) == 4)) ? 2 : 1; // (eleven and more fp64 columns: the carry and the q x q factor)
}

template <typename T, int N, int P, int Q, int RB>
__global__ void __launch_bounds__(64, (ext_fit_stream_waves<T, N + 1 + P, RB>())) ext_fit_stream_eval_kernel(const ext::ExtFitArgs<T> a) {
    constexpr int NC = N + 1 + P;
    constexpr int ROWS = 64 * RB;
    using G = Grp<1>;
    using F = ext::ExtFitLayout<Q>;
    G grp = G::make(nullptr);
    const int lane = grp.gl;
    const int64_t b = ext::extfit_problem_of<T>(a, blockIdx.x);
    if (b < 0) return;
    const int m = a.m;
    const bool vec = a.vec != 0;
    T *st = reinterpret_cast<T *>(a.state);
    int32_t *si = ext::extfit_ints<T, Q>(a.state, a.B);

    int want = ext::EXTFIT_WANT_BASIS | ext::EXTFIT_WANT_DERIVS;
    if (!a.init) {
        if (uni(si[F::TERM * a.B + b]) != 0) return; // finished in an earlier step
        want = uni(si[F::WANT * a.B + b]);
    }
    const bool with_d = (want & ext::EXTFIT_WANT_DERIVS) != 0 && a.dphi != nullptr; // (uniform)
    const T *ph = a.phi + b * (int64_t)N * m;
    const T *yp = a.yw + b * (int64_t)m;
    const T *dp = with_d ? a.dphi + b * (int64_t)a.np * m : nullptr;
    const T *wp = a.w ? a.w + b * a.w_stride : nullptr;
    const int np = a.np;

    auto load_block = [&](const int off, T (&Cb)[NC][RB]) __attribute__((always_inline)) {
        const int mrem = m - off;
#pragma unroll
        for (int j = 0; j < N; ++j) load_rows<T, RB, 1>(ph + (int64_t)a.perm[j] * m + off, mrem, lane, vec, Cb[j]);
        load_rows<T, RB, 1>(yp + off, mrem, lane, vec, Cb[N]);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (dp && p < np) { // (uniform)
                load_rows<T, RB, 1>(dp + (int64_t)p * m + off, mrem, lane, vec, Cb[N + 1 + p]);
            } else {
#pragma unroll
                for (int r = 0; r < RB; ++r) Cb[N + 1 + p][r] = T(0);
            }
        }
        if (wp) { // `&self.weights * ...` (src/util/weights.rs:82-99); y_w was weighted when the handle was made
            T wt[RB];
            load_rows<T, RB, 1>(wp + off, mrem, lane, vec, wt);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (j == N) continue;
#pragma unroll
                for (int r = 0; r < RB; ++r) Cb[j][r] *= wt[r];
            }
        }
    };

    // ---- compress: every block folded into the (n + 1 + p)-column carry ----
    T K[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) K[j][0] = K[j][1] = T(0);
    const int nb = (m + ROWS - 1) / ROWS;
    // the trailing columns [y | dPhi_1 .. dPhi_P] are final after the n wave-wide reflectors of a block: each LANE folds its own
    // rows of them into a private (p + 1)^2 triangle (lane_trail_update, vp_block.hpp: no reduction, no broadcast) and the 64
    // triangles are merged once per evaluation -- n reduction rounds per block instead of n + 1 + p.  A step without derivative
    // columns runs the same code on zero columns: the data column is the first of the trailing ones, its reflectors do not
    // see the others, so the residual norm of a point is the same number under both protocols.
    constexpr int PT = P + 1;
    constexpr bool kLaneTrail = PT <= 5;
    T Tl[kLaneTrail ? PT : 1][kLaneTrail ? PT : 1];
    if constexpr (kLaneTrail) {
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) Tl[i][j] = T(0);
    }
    auto fold = [&](T (&Cb)[NC][RB]) __attribute__((always_inline)) {
        if constexpr (kLaneTrail) {
            stacked_qr<T, NC, N, RB, G>(K, Cb, grp);
            lane_trail_update<T, NC, N, PT, RB>(Tl, Cb);
        } else {
            if (with_d) stacked_qr<T, NC, NC, RB, G>(K, Cb, grp);
            else stacked_qr<T, NC, N + 1, RB, G>(K, Cb, grp);
        }
    };
    if constexpr (ext_fit_stream_single<T, NC>()) {
        for (int ib = 0; ib < nb; ++ib) {
            T Ca[NC][RB];
            load_block(ib * ROWS, Ca);
            fold(Ca);
        }
    } else {
        T Ca[NC][RB], Cc[NC][RB];
        load_block(0, Ca);
        for (int ib = 0; ib < nb; ib += 2) {
            if (ib + 1 < nb) load_block((ib + 1) * ROWS, Cc);
            fold(Ca);
            if (ib + 1 < nb) {
                if (ib + 2 < nb) load_block((ib + 2) * ROWS, Ca);
                fold(Cc);
            }
        }
    }

    if constexpr (kLaneTrail) lane_trail_merge<T, NC, N, PT, G>(Tl, K, grp);

    // ---- set_params on the compressed problem: src/solvers/levmar/mod.rs:42-73 ----
    T Rm[N][N], qty[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) Rm[i][j] = (j >= i) ? readlane(K[j][i % 2], i / 2) : T(0);
        qty[i] = readlane(K[N][i % 2], i / 2);
    }
    T c[N], e[N];
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, a.eps, c, e, truncated);
    T sq = T(0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const T v = (2 * lane + r >= N) ? K[N][r] : T(0); // (only row n is non-zero)
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
        st[F::C_FN * a.B + b] = usqrt(fn2);
#pragma unroll
        for (int k = 0; k < N; ++k) st[(F::C_C + a.perm[k]) * a.B + b] = c[k];
        si[F::C_OK * a.B + b] = ok ? 1 : 0;
        si[F::C_HASJ * a.B + b] = (with_d && ok) ? 1 : 0;
    }
    if (!with_d || !ok) return;

    // ---- jacobian at the same point (:101-201): Kaufman columns in Q-coordinates from the carry rows >= n, pivoted qrfac ----
    residual_qcoords<T, 2, N>(K[N], e, grp);
    T Zs[Q][2];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        Zs[k][0] = Zs[k][1] = T(0);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (p < np && a.pp[p] == k) { // (uniform)
                const T cj = -dyn_get<N>(c, a.pb[p]);
#pragma unroll
                for (int r = 0; r < 2; ++r) Zs[k][r] = tfma(cj, K[N + 1 + p][r], Zs[k][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (2 * lane + r < N) Zs[k][r] = T(0); // (the P_perp)
    }
    T Rj[Q][Q], acn[Q], qtf[Q];
    int ipv[Q];
    jac_qrfac<T, 2, Q, N>(Zs, K[N], Rj, acn, ipv, qtf, grp);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            st[(F::C_ACN + k) * a.B + b] = acn[k];
            st[(F::C_QTF + k) * a.B + b] = qtf[k];
            si[(F::C_IPVT + k) * a.B + b] = ipv[k];
#pragma unroll
            for (int j = 0; j < Q; ++j) st[(F::C_RJ + k * Q + j) * a.B + b] = Rj[k][j];
        }
    }
}

template <typename T, int N, int P, int Q> int launch_fit_stream_eval(const ext::ExtFitArgs<T> &a, hipStream_t stream) {
    constexpr int RB = ext_fit_stream_rows<T, N + 1 + P>();
    hipLaunchKernelGGL((ext_fit_stream_eval_kernel<T, N, P, Q, RB>), dim3((unsigned)a.grid_problems), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace blk
} // namespace vp

// one streamed shape of the batched external fit: capacity 2^26 rows (R = 2^20, W = 1 in the table: it loses against every
// resident kernel that covers m and serves every length none does)
#define VP_REGISTER_EXTFIT_STREAM(T, NN, PP, QQ)                                                                       \
    static ::vp::ext::ExtFitRegistrar<T> VP_EXT_CAT(vp_extfit_stream_reg_, __COUNTER__)(                              \
        ::vp::ext::ExtFitEntry<T>{NN, PP, QQ, 1 << 20, 1, &::vp::blk::launch_fit_stream_eval<T, NN, PP, QQ>});
