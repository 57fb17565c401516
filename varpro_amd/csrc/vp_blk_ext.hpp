// vp_blk_ext.hpp -- caller-evaluated models at ANY length: the evaluation of vp_ext.hpp with the rows streamed in blocks.
//
// The reference takes any `output_len()` (src/model/mod.rs:263) and any `SeparableNonlinearModel` (:239-363).  The
// register-resident kernels of vp_ext.hpp hold all m rows of Phi, y and dPhi on chip (one wave to 1 024 rows fp64, eight
// waves to 8 192); six fp64 columns of 10 000 rows are a CU's whole register file.  Here m is a run-time number, as in
// vp_block.hpp -- the same two passes as blk_evaluate_kernel, with the column BUILD replaced by column LOADS:
//
//   pass 1 (forward)   block by block: load the rows of [W Phi | y_w | W dPhi_1 .. W dPhi_P], fold them into the
//                      (N + 1 + P)-column carry by the N reflectors of Phi (stacked_qr<NREF = N>): R, Q^T y, ||r||^2 -> c,
//                      cost, status (src/solvers/levmar/mod.rs:42-73); the carry rows every block started from are kept in
//                      LDS (N x NC values per block);
//   pass 2 (backward)  block by block in reverse: reload the rows, recompute the block's reflectors from the recorded carry
//                      (identical arithmetic), take the block's rows of Q^T y and Q^T dPhi_p, carry [r~ | D~_1 .. D~_P] back
//                      through the reflectors (stacked_apply_q) and store r and  J_k = -sum_{pairs p of k} c_{j(p)} Q D~_p
//                      (:91-95, 101-201; assembled at the store, q and the pair table are run-time as in vp_ext.hpp).
//
// Round 6: a problem whose factor is well conditioned (diagonal ratio <= 1e4, no truncation) takes the DIRECT route instead --
// pass 1 lane-private (every lane folds its own rows, one merge per problem: R, (Q^T [y | D])_{<N}, ||r||), pass 2 row-local
// (r = y - Phi c, P_perp D = D - Phi R^-1 (Q^T D)_{<N}) from the last block backwards: no reduction round inside either loop,
// the kernel then runs at the memory system's pace on its 15 column transfers (5.7 TB/s; 1.02 -> 0.86 ms per 4 096 problems of
// 10 000 rows).  Everything else: the two Householder passes described above.
//
// Exact Householder arithmetic on both passes.  The caller's columns cross HBM TWICE (forward and backward), r and J once:
// 2 (n + p + 1) + 1 + q column transfers against the n + p + 2 + q of the resident kernels -- the price of not holding m rows
// on chip.  Blocks are double-buffered in registers (the loads of block i + 1 are in flight while block i is folded).
#pragma once
#ifndef VP_BLKEXT_NT
#define VP_BLKEXT_NT 1 /* non-temporal stores of r and J: they leave the forward pass's blocks in the caches for the backward pass (round 6: 1-6 % on two boxes at m = 10 000, B = 4 096; tools/ext_probe.py) */
#endif
#include "vp_block.hpp"
#include "vp_ext.hpp"

namespace vp {
namespace blk {

// rows per lane and block of the streamed external kernels: two blocks are resident (double buffer)
template <typename T, int NC> constexpr int ext_block_rows() {
    constexpr int words = NC * (int)(sizeof(T) / 4);
    return words <= 8 ? 8 : (words <= 18 ? 4 : 2);
}
// the evaluation kernel on shapes of 9 .. 12 register words per row (five / six fp64 columns): ONE resident block of 8 rows per
// lane -- the SIMD's second wave loads while this one folds -- instead of a double buffer of two of 4 (end of round 5: the
// batched fit's streamed step went from 0.49 to 0.61 of HBM that way, vp_blk_extfit.hpp)
#ifndef VP_EXT_STREAM_SINGLE
#define VP_EXT_STREAM_SINGLE 1
#endif
template <typename T, int NC> constexpr bool ext_stream_single() {
    return VP_EXT_STREAM_SINGLE && NC * (int)(sizeof(T) / 4) > 8 && NC * (int)(sizeof(T) / 4) <= 12;
}
template <typename T, int NC> constexpr int ext_stream_rows() { return ext_stream_single<T, NC>() ? 8 : ext_block_rows<T, NC>(); }

// Lane-private TSQR of a block (round 6): every lane folds its own RB rows of ALL NC columns into a private NREF x NC upper
// trapezoid Tl by NREF Householder reflectors on the stacked [Tl; rows] -- per-lane arithmetic only, no reduction, no
// broadcast.  A tall-skinny QR does not care how the rows are grouped: the 64 trapezoids are merged ONCE per problem by a
// wave-wide QR of their 64 NREF rows (lane_tsqr_merge).  The R factor, the first NREF entries of Q^T(.) of every column and
// the NREF-th diagonal (the norm of what the first NREF - 1 columns leave of column NREF - 1: ||r|| with y at NREF - 1) are
// what the DIRECT form of the evaluation needs -- the block's reflectors themselves are not.
template <typename T, int NC, int NREF, int RB>
__device__ __forceinline__ void lane_tsqr_update(T (&Tl)[NREF][NC], T (&Cb)[NC][RB]) {
    static_for<0, NREF>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        T d[NC - k];
#pragma unroll
        for (int j = k; j < NC; ++j) {
            T acc = Tl[k][k] * Tl[k][j];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc = tfma(Cb[k][r], Cb[j][r], acc);
            d[j - k] = acc;
        }
        const T alpha = Tl[k][k], nrm2 = d[0];
        const bool live = nrm2 > num<T>::norm2_min && is_finite(nrm2);
        const T y = live ? frsqrt(nrm2) : T(0);
        const T s0 = nrm2 * y;
        const T sigma = tfma(tfma(-s0, s0, nrm2), T(0.5) * y, s0);
        const T beta = live ? -tcopysign(sigma, alpha) : ((nrm2 <= num<T>::norm2_min) ? alpha : nrm2);
        const T u = live ? alpha - beta : T(0);
        const T gk = live ? -y * frcp(tabs(alpha) + sigma) : T(0);
        Tl[k][k] = beta;
#pragma unroll
        for (int j = k + 1; j < NC; ++j) {
            const T f = gk * tfma(-beta, Tl[k][j], d[j - k]);
#pragma unroll
            for (int r = 0; r < RB; ++r) Cb[j][r] = tfma(f, Cb[k][r], Cb[j][r]);
            Tl[k][j] = tfma(f, u, Tl[k][j]);
        }
    });
}
// merge the 64 private trapezoids into the carry K (row i of the merged factor in lane i/2, register i%2; rows >= NREF zero)
template <typename T, int NC, int NREF, class G>
__device__ __forceinline__ void lane_tsqr_merge(const T (&Tl)[NREF][NC], T (&K)[NC][2], G &grp) {
    constexpr int RZ = (NREF + 1) / 2 * 2;
    T Z[NC][RZ];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < RZ; ++i) Z[j][i] = (i < NREF && i <= j) ? Tl[i < NREF ? i : 0][j] : T(0);
#pragma unroll
    for (int j = 0; j < NC; ++j) K[j][0] = K[j][1] = T(0);
    stacked_qr<T, NC, NREF, RZ, G>(K, Z, grp);
}

#ifndef VP_EXT_STREAM_DIRECT
#define VP_EXT_STREAM_DIRECT 1 // (A/B switch: 0 = two Householder passes for every problem, round 5)
#endif
#ifndef VP_EXT_STREAM_DIRECT_COND
#define VP_EXT_STREAM_DIRECT_COND 1e4
#endif
template <typename T, int N, int P, int RB>
__global__ void __launch_bounds__(64, 2) ext_stream_evaluate_kernel(const ext::ExtArgs<T> a) {
    constexpr int NC = N + 1 + P, NW = 1 + P;
    constexpr int ROWS = 64 * RB;
    using G = Grp<1>;
    G grp = G::make(nullptr);
    const int lane = grp.gl;
    const int64_t prob = blockIdx.x; // problem * S + rhs
    if (prob >= a.nprob) return;
    const int64_t b = prob / a.S;
    const int s = (int)(prob - b * a.S);
    const int m = a.m;
    const bool vec = a.vec != 0;
    const T *ph = a.phi + b * (int64_t)N * m;
    const T *yp = a.yw + prob * (int64_t)m;
    const T *dp = (P > 0 && a.dphi) ? a.dphi + b * (int64_t)a.np * m : nullptr;
    const T *wp = a.w ? a.w + b * a.w_stride : nullptr;
    const int np = a.np;
    const bool want_j = P > 0 && a.J_out != nullptr && dp != nullptr && np > 0;
    const bool want_rj = a.r_out != nullptr || want_j;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *snap = reinterpret_cast<T *>(smem_raw); // [block][N][NC]: the carry rows block i started from

    // the block of ROWS rows at `off`: every column of this lane's RB rows (zero past the end), weighted
    auto load_block = [&](const int off, T (&Cb)[NC][RB]) __attribute__((always_inline)) {
        const int mrem = m - off;
#pragma unroll
        for (int j = 0; j < N; ++j) load_rows<T, RB, 1>(ph + (int64_t)j * m + off, mrem, lane, vec, Cb[j]);
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

    const int nb_all = (m + ROWS - 1) / ROWS;
#if VP_EXT_STREAM_DIRECT
    // ================= the DIRECT route (round 6): lane-private pass 1, row-local pass 2, no reduction inside either loop ======
    // The Householder route below is bound by its dependent reduction rounds per block (N forward, 2 N backward: 12 us per block
    // step, 3 TB/s of input), not by HBM.  With R and the first N rows of Q^T [y | D] known,
    //     r = y_w - Phi_w c,      P_perp D_p = D_p - Phi_w (R^-1 (Q^T D_p)_{<N})
    // are ROW-LOCAL expressions of the caller's columns: pass 2 needs neither the blocks' reflectors nor their order, and pass 1
    // only has to deliver R, (Q^T [y | D])_{<N} and ||r|| -- every lane folds its own rows into a private trapezoid
    // (lane_tsqr_update), one wave-wide merge per problem.  The price is cond(Phi_w): y - Phi c carries eps |Phi| |c| where the
    // orthogonal form carries eps ||y||.  Taken where the merged factor's diagonal says cond <~ 1e4 (error <= 1e-12 relative,
    // two digits inside north_star's 1e-10), the solve is not truncated and everything is finite; any other problem falls
    // through to the exact Householder route (a second pass 1: rare).
    {
        constexpr int NREF = N + 1; // Phi and y: the (N, N) entry of the merged factor is ||r||
        T Kd[NC][2];
        {
            T Tl[NREF][NC];
#pragma unroll
            for (int i = 0; i < NREF; ++i)
#pragma unroll
                for (int j = 0; j < NC; ++j) Tl[i][j] = T(0);
            for (int ib = 0; ib < nb_all; ++ib) {
                T Ca[NC][RB];
                load_block(ib * ROWS, Ca);
                lane_tsqr_update<T, NC, NREF, RB>(Tl, Ca);
            }
            lane_tsqr_merge<T, NC, NREF, G>(Tl, Kd, grp);
        }
        T Rd[N][N], qd[N];
        T dmx = T(0), dmn = num<T>::huge;
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < N; ++j) Rd[i][j] = (j >= i) ? readlane(Kd[j][i % 2], i / 2) : T(0);
            qd[i] = readlane(Kd[N][i % 2], i / 2);
            dmx = tmax(dmx, tabs(Rd[i][i]));
            dmn = tmin(dmn, tabs(Rd[i][i]));
        }
        const T rn = readlane(Kd[N][N % 2], N / 2); // the data column's diagonal: ||r||
        T cd[N], ed[N];
        bool trunc_d;
        solve_coeffs<T, N>(Rd, qd, a.eps, cd, ed, trunc_d);
        const T fn2d = rn * rn;
        bool okd = is_finite(fn2d) && is_finite(dmx) && dmn > T(0) && dmx <= T(VP_EXT_STREAM_DIRECT_COND) * dmn && !trunc_d;
#pragma unroll
        for (int k = 0; k < N; ++k) okd = okd && is_finite(cd[k]);
        if (uni(okd)) {
            if (lane == 0) {
                if (a.status) a.status[prob] = VP_ST_OK;
                if (a.cost_out) a.cost_out[prob] = 0.5 * (double)fn2d;
            }
            if (a.C_out && lane < N) a.C_out[prob * N + lane] = dyn_get<N>(cd, lane);
            if (!want_rj) return;
            T Gm[N][P > 0 ? P : 1]; // R^-1 (Q^T D_p)_{<N}, column p
#pragma unroll
            for (int p = 0; p < P; ++p) {
                T rhs[N];
#pragma unroll
                for (int i = 0; i < N; ++i) rhs[i] = readlane(Kd[N + 1 + p][i % 2], i / 2);
#pragma unroll
                for (int i = N - 1; i >= 0; --i) {
                    T acc = rhs[i];
#pragma unroll
                    for (int j = i + 1; j < N; ++j) acc = tfma(-Rd[i][j], Gm[j][p], acc);
                    Gm[i][p] = acc / Rd[i][i];
                }
            }
            for (int ib = nb_all - 1; ib >= 0; --ib) { // (any order is right: the LAST block first -- it is the one still in the caches)
                const int off = ib * ROWS;
                T Ca[NC][RB];
                load_block(off, Ca);
                if (a.r_out) {
                    T v[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        T acc = Ca[N][r];
#pragma unroll
                        for (int j = 0; j < N; ++j) acc = tfma(-cd[j], Ca[j][r], acc);
                        v[r] = acc;
                    }
                    store_rows<T, RB, 1, VP_BLKEXT_NT != 0>(a.r_out + prob * (int64_t)m + off, m - off, lane, vec, v);
                }
                if (want_j) {
#pragma unroll
                    for (int p = 0; p < P; ++p)
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            T acc = Ca[N + 1 + p][r];
#pragma unroll
                            for (int j = 0; j < N; ++j) acc = tfma(-Gm[j][p], Ca[j][r], acc);
                            Ca[N + 1 + p][r] = acc; // P_perp D_p, this block's rows
                        }
                    for (int k = 0; k < a.q; ++k) { // J[b][k][s][m]
                        T cj[P > 0 ? P : 1];
#pragma unroll
                        for (int p = 0; p < P; ++p) cj[p] = (p < np && a.pp[p] == k) ? -dyn_get<N>(cd, a.pb[p]) : T(0);
                        T v[RB];
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            T acc = T(0);
#pragma unroll
                            for (int p = 0; p < P; ++p) acc = tfma(cj[p], Ca[N + 1 + p][r], acc);
                            v[r] = acc;
                        }
                        T *jp = a.J_out + ((b * a.q + k) * (int64_t)a.S + s) * (int64_t)m + off;
                        store_rows<T, RB, 1, VP_BLKEXT_NT != 0>(jp, m - off, lane, vec, v);
                    }
                }
            }
            return;
        }
    }
#endif
    // ================= pass 1: forward, N reflectors per block =================
    T K[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) K[j][0] = K[j][1] = T(0);
    T sq = T(0);
    const int nb = (m + ROWS - 1) / ROWS;
    auto fold = [&](T (&Cb)[NC][RB], const int ib) __attribute__((always_inline)) {
        if (want_rj) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (lane == i / 2) {
#pragma unroll
                    for (int j = 0; j < NC; ++j) snap[((size_t)ib * N + i) * NC + j] = K[j][i % 2];
                }
        }
        stacked_qr<T, NC, N, RB, G>(K, Cb, grp);
#pragma unroll
        for (int r = 0; r < RB; ++r) sq = tfma(Cb[N][r], Cb[N][r], sq);
    };
    if constexpr (ext_stream_single<T, NC>()) {
        for (int ib = 0; ib < nb; ++ib) {
            T Ca[NC][RB];
            load_block(ib * ROWS, Ca);
            fold(Ca, ib);
        }
    } else {
        T Ca[NC][RB], Cc[NC][RB];
        load_block(0, Ca);
        for (int ib = 0; ib < nb; ib += 2) {
            if (ib + 1 < nb) load_block((ib + 1) * ROWS, Cc);
            fold(Ca, ib);
            if (ib + 1 < nb) {
                if (ib + 2 < nb) load_block((ib + 2) * ROWS, Ca);
                fold(Cc, ib + 1);
            }
        }
    }
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
    if (!want_rj) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the recorded carries are in LDS

    // ================= pass 2: backward, r and J block by block =================
    T Wc[NW][2]; // carry part of [r~ | D~_1 .. D~_P]: rows < N hold e (residual) and 0 (P_perp drops the range(Q) part)
#pragma unroll
    for (int z = 0; z < NW; ++z) Wc[z][0] = Wc[z][1] = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) Wc[0][i % 2] = (lane == i / 2) ? e[i] : Wc[0][i % 2];
    auto unfold = [&](T (&Cb)[NC][RB], const int ib) __attribute__((always_inline)) {
        const int off = ib * ROWS;
        T Kb[NC][2];
#pragma unroll
        for (int j = 0; j < NC; ++j) Kb[j][0] = Kb[j][1] = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (lane == i / 2) {
#pragma unroll
                for (int j = 0; j < NC; ++j) Kb[j][i % 2] = snap[((size_t)ib * N + i) * NC + j];
            }
        T u[N], g[N];
        stacked_qr_keep<T, NC, N, RB, G>(Kb, Cb, u, g, grp);
        T Wb[NW][RB];
#pragma unroll
        for (int z = 0; z < NW; ++z)
#pragma unroll
            for (int r = 0; r < RB; ++r) Wb[z][r] = Cb[N + z][r];
        stacked_apply_q<T, NC, N, NW, RB, G>(Cb, u, g, Wc, Wb, grp);
        if (a.r_out) store_rows<T, RB, 1, VP_BLKEXT_NT != 0>(a.r_out + prob * (int64_t)m + off, m - off, lane, vec, Wb[0]);
        if (want_j) {
            for (int k = 0; k < a.q; ++k) { // J[b][k][s][m]
                T cj[P > 0 ? P : 1];
#pragma unroll
                for (int p = 0; p < P; ++p) cj[p] = (p < np && a.pp[p] == k) ? -dyn_get<N>(c, a.pb[p]) : T(0);
                T v[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    T acc = T(0);
#pragma unroll
                    for (int p = 0; p < P; ++p) acc = tfma(cj[p], Wb[1 + p][r], acc);
                    v[r] = acc;
                }
                T *jp = a.J_out + ((b * a.q + k) * (int64_t)a.S + s) * (int64_t)m + off;
                store_rows<T, RB, 1, VP_BLKEXT_NT != 0>(jp, m - off, lane, vec, v);
            }
        }
    };
    if constexpr (ext_stream_single<T, NC>()) {
        for (int ib = nb - 1; ib >= 0; --ib) {
            T Ca[NC][RB];
            load_block(ib * ROWS, Ca);
            unfold(Ca, ib);
        }
    } else {
        T Ca[NC][RB], Cc[NC][RB];
        load_block((nb - 1) * ROWS, Ca);
        for (int ib = nb - 1; ib >= 0; ib -= 2) {
            if (ib - 1 >= 0) load_block((ib - 1) * ROWS, Cc);
            unfold(Ca, ib);
            if (ib - 1 >= 0) {
                if (ib - 2 >= 0) load_block((ib - 2) * ROWS, Ca);
                unfold(Cc, ib - 1);
            }
        }
    }
}

// LDS of the recorded carries; beyond the cap the generic kernels (m > ~70 000 rows of a six-column fp64 shape)
template <typename T, int N, int P, int RB> constexpr size_t ext_stream_snap_bytes(int64_t m) {
    return (size_t)((m + 64 * RB - 1) / (64 * RB)) * N * (N + 1 + P) * sizeof(T);
}

template <typename T, int N, int P> int launch_ext_stream(const ext::ExtArgs<T> &a, hipStream_t stream) {
    constexpr int RB = ext_stream_rows<T, N + 1 + P>();
    const bool want_rj = a.r_out != nullptr || a.J_out != nullptr;
    const size_t snap = want_rj ? ext_stream_snap_bytes<T, N, P, RB>(a.m) : 0;
    if (snap > kEvalSnapMax) return VP_ERR_UNSUPPORTED; // (the caller falls back to the generic kernels)
    // (round 6, tools/ext_stream_cap_probe.py: capping the resident waves through the LDS footprint so that the backward pass's
    // re-reads come out of the 256 MB Infinity Cache LOSES -- 1.01 ms at 2 048 resident waves, 1.20 at 1 024, 1.81 at 512, 3.07
    // at 256 per 4 096 problems of 10 000 rows: a wave is bound by its own chain of reduction rounds per block, not by HBM)
    // (round 6, tools/ext_stream_cap_probe.py: capping the resident waves through the LDS footprint so that pass 2 re-reads from
    // the 256 MB Infinity Cache LOSES on both routes -- a wave holds one block (24 KB) in flight, fewer waves are fewer bytes in
    // flight: direct route 0.86 ms at 2 048 resident waves, 0.94 at 1 536, 1.02 at 1 024, 1.52 at 512)
    hipLaunchKernelGGL((ext_stream_evaluate_kernel<T, N, P, RB>), dim3((unsigned)a.nprob), dim3(64), snap, stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace blk
} // namespace vp

// one streamed shape: n basis functions, up to PP derivative columns (PP = 0: the pass without them)
#define VP_REGISTER_EXT_STREAM(T, NN, PP)                                                                              \
    static ::vp::ext::ExtStreamRegistrar<T> VP_EXT_CAT(vp_ext_reg_, __COUNTER__)(                                     \
        ::vp::ext::ExtStreamEntry<T>{NN, PP, &::vp::blk::launch_ext_stream<T, NN, PP>});
