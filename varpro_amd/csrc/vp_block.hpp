// vp_block.hpp -- LENGTH-AGNOSTIC single-RHS fit: the rows of a problem are streamed in blocks through a TSQR-style update.
//
// The reference takes any `output_len()` (src/model/mod.rs:263).  The register-resident kernels (vp_fit.hpp, vp_fit2.hpp)
// keep all m rows of every column in VGPRs and therefore exist per (model, rows-per-lane, waves) SET; one row past the
// largest set a problem used to fall to the generic kernels at a 60-100x step.  Here m is a run-time number:
//
//   One wavefront owns one problem.  Its columns  A = [W Phi | y_w | W dPhi_1 .. W dPhi_P]  (NC = N + 1 + P) are never
//   resident: per evaluation the wave walks the rows in blocks of 64*RB, builds the block's columns in registers
//   (build_columns on an offset row source -- the same code, incl. the uniform-grid exp recurrence, re-seeded per block) and
//   folds it into an NC x NC upper-triangular CARRY  T  by Householder reflectors on the stacked matrix [T; block]
//   (stacked_qr: reflector k touches row k of T and the block rows only, because T is triangular).  That is a
//   sequential TSQR: T^T T = A^T A with the conditioning of A, not of A^T A -- nothing is squared.
//
//   After the last block  A = Q_A T  with orthonormal Q_A, so the compressed problem (T_Phi, T_y, T_D) has the same
//   coefficients c, the same ||r|| and -- the Kaufman Jacobian being  J = -P_perp D c  -- J = Q_A J_T, r = Q_A r_T: the
//   LM loop, which only sees quantities invariant under an orthogonal change of basis of the residual space (vp_fit.hpp),
//   runs UNCHANGED on the NC-row problem held in the carry: solve_coeffs on T's leading N x N block
//   (src/solvers/levmar/mod.rs:51-59), MINPACK qrfac on the (P + 1) x Q compressed Jacobian (jac_qrfac at two rows per
//   lane), lm_after_eval / lm_next_step (vp_lm_core.hpp) -- LevenbergMarquardt::minimize (:247) as everywhere else.
//
// The carry sits in the first register PAIR of every column (lane l holds carry rows 2l, 2l+1, i.e. row i of T lives in
// lane i/2): NC * 2 VGPR pairs, whatever m is.  y (and the grid / weights) are re-read per evaluation, 16 bytes per lane:
// T*m bytes per evaluation from L2 / HBM, the price of not holding m rows on chip.
#pragma once
#ifndef VP_BLKEVAL_NT
#define VP_BLKEVAL_NT 1 /* non-temporal stores of r and J in blk_evaluate_kernel (round 6: 0.953 -> 0.918 ms per 8 192 evaluations of 10 000 rows, tools/blk_eval_probe.py) */
#endif
#include "vp_lm_core.hpp"

namespace vp {
namespace blk {

// Householder reflectors k = 0 .. NREF-1 on the stacked matrix [K; Cb]:
//   K  [NC][2]   the carry: upper triangular in its first NC rows (row i in lane i/2, register i%2), zero below
//   Cb [NC][RB]  the block's rows
// On return K holds the updated triangle (rows < NREF), Cb the rows of Q^T(block) that later reflectors of THIS call still
// acted on -- for columns >= NREF they are the block's contribution to the part of the column orthogonal to the first NREF
// columns (final: later blocks never touch them again); the reflector vectors themselves are not kept.
template <typename T, int NC, int NREF, int RB, class G>
__device__ __forceinline__ void stacked_qr(T (&K)[NC][2], T (&Cb)[NC][RB], G &grp) {
    const int lane = grp.gl;
    static_for<0, NREF>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int NREM = NC - k;
        constexpr int own = k / 2, reg = k % 2; // carry row k lives in lane `own`, register `reg`
        // block part of the raw dot products a_k^T a_j (j = k .. NC-1) and the carry row's entries T[k][j]
        T d[NREM], top[NREM];
#pragma unroll
        for (int j = k; j < NC; ++j) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc = tfma(Cb[k][r], Cb[j][r], acc);
            d[j - k] = acc;
            top[j - k] = K[j][reg];
        }
        group_allreduce(grp, d);
        group_bcast<NREM>(grp, top, own);
        // (T is upper triangular: column k has no carry entries below row k, so the carry contributes top_k * top_j)
#pragma unroll
        for (int j = 0; j < NREM; ++j) d[j] = tfma(top[0], top[j], d[j]);
        const T alpha = top[0], nrm2 = d[0];
        const bool live = nrm2 > num<T>::norm2_min && is_finite(nrm2);
        const T y = live ? frsqrt(nrm2) : T(0);
        const T s0 = nrm2 * y;
        const T sigma = tfma(tfma(-s0, s0, nrm2), T(0.5) * y, s0);
        const T beta = live ? -tcopysign(sigma, alpha) : ((nrm2 <= num<T>::norm2_min) ? alpha : nrm2);
        const T u = live ? alpha - beta : T(0);
        const T gk = live ? -y * frcp(tabs(alpha) + sigma) : T(0);
        K[k][reg] = (lane == own) ? beta : K[k][reg];
#pragma unroll
        for (int j = k + 1; j < NC; ++j) {
            const T f = gk * tfma(-beta, top[j - k], d[j - k]); // g * v^T a_j,  v = [u; block part of a_k]
#pragma unroll
            for (int r = 0; r < RB; ++r) Cb[j][r] = tfma(f, Cb[k][r], Cb[j][r]);
            const T tj = tfma(f, u, top[j - k]); // row k of the updated column: the new T[k][j]
            K[j][reg] = (lane == own) ? tj : K[j][reg];
        }
    });
}

// ---- lane-private compression of the TRAILING columns --------------------------------------------------------------------
// After the N wave-wide reflectors of a block the rows of the trailing columns [y | D_1 .. D_P] (PT = P + 1) are final: no
// later block touches them.  Their R factor is all the LM loop needs of them, and a tall-skinny QR does not care how the
// rows are grouped (TSQR): every LANE folds its own RB rows into a PRIVATE PT x PT triangle by Householder reflectors on
// the stacked [Tl; rows] -- per-lane arithmetic only, no reduction, no broadcast -- and the 64 triangles are merged ONCE per
// evaluation by an ordinary wave-wide QR of the 64*PT rows.  A block then costs N reduction rounds instead of N + 1 + P.
template <typename T, int NC, int N, int PT, int RB>
__device__ __forceinline__ void lane_trail_update(T (&Tl)[PT][PT], T (&Cb)[NC][RB]) {
    static_for<0, PT>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        T d[PT - k];
#pragma unroll
        for (int j = k; j < PT; ++j) {
            T acc = Tl[k][k] * Tl[k][j];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc = tfma(Cb[N + k][r], Cb[N + j][r], acc);
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
        for (int j = k + 1; j < PT; ++j) {
            const T f = gk * tfma(-beta, Tl[k][j], d[j - k]);
#pragma unroll
            for (int r = 0; r < RB; ++r) Cb[N + j][r] = tfma(f, Cb[N + k][r], Cb[N + j][r]);
            Tl[k][j] = tfma(f, u, Tl[k][j]);
        }
    });
}
// merge the 64 private triangles and write the PT x PT result into the carry rows N .. N + PT - 1
template <typename T, int NC, int N, int PT, class G>
__device__ __forceinline__ void lane_trail_merge(const T (&Tl)[PT][PT], T (&K)[NC][2], G &grp) {
    constexpr int RZ = (PT + 1) / 2 * 2;
    T Z[PT][RZ];
#pragma unroll
    for (int j = 0; j < PT; ++j)
#pragma unroll
        for (int i = 0; i < RZ; ++i) Z[j][i] = (i <= j && i < PT) ? Tl[i < PT ? i : 0][j] : T(0);
    T gq[PT], Rz[PT][PT], qd[PT];
    house_qr<T, RZ, PT, PT, 0, false, G>(Z, gq, Rz, qd, grp);
    const int lane = grp.gl;
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = i; j < PT; ++j) K[N + j][(N + i) % 2] = (lane == (N + i) / 2) ? Rz[i][j] : K[N + j][(N + i) % 2];
}

// rows per lane and block: the NC x (RB + 2) register columns + the wave-uniform LM state must fit two waves per SIMD
// The wave's LDS ring of row blocks: 2 slots x {grid, data[, weights]} x 64*RB rows in natural row order.  Block i's rows
// are staged by ASYNCHRONOUS global -> LDS DMA (global_load_lds_dwordx4: no VGPRs; lane l of instruction k delivers the
// 16-byte group k*64 + l) issued ONE BLOCK AHEAD of their use: without it every block started with a dependent global round
// trip for its grid values that two resident waves per SIMD cannot hide (measured at m = 10^5: 2.9x).  vmcnt is hand-counted
// as in vp_mrhs.hpp: between begin() and the last stage() of a pass the wave must issue nothing else that counts in vmcnt
// except what the caller declares (`extra` stores per block).  Arrays that do not allow whole 16-byte groups (m not a
// multiple of 16/sizeof(T), unaligned bases) are staged element-wise through registers, without prefetch.
// NOGRID (unit weights, uniform grid): only the data is staged -- the caller computes the grid values (RowSource TCALC).
template <typename T, int RB, bool WEIGHTED, bool NOGRID = false> struct RowRing {
    static_assert(!(NOGRID && WEIGHTED), "the data-only ring serves unit-weight problems");
    static constexpr int ROWS = 64 * RB;
    static constexpr int NARR = WEIGHTED ? 3 : (NOGRID ? 1 : 2);
    static constexpr int EL = 16 / (int)sizeof(T); // elements per lane and DMA instruction
    static constexpr int KA = RB / EL;             // DMA instructions per array and block
    static constexpr int KD = NARR * KA;           // ... per block
    static_assert(RB % EL == 0 && KD <= 40, "block rows must be whole DMA instructions; vmcnt is a 6-bit counter");
    T *ring;
    unsigned ring_lds;
    const T *tp, *yp, *wp;
    int m, lane, it;
    bool dma;
    __device__ __forceinline__ void init(T *ring_, const T *tp_, const T *yp_, const T *wp_, int m_, int lane_) {
        ring = ring_;
        ring_lds = (unsigned)(uintptr_t)(VP_LDS unsigned char *)ring_;
        tp = tp_;
        yp = yp_;
        wp = wp_;
        m = m_;
        lane = lane_;
        it = 0;
        dma = (m % EL) == 0 && (((NOGRID ? 0 : reinterpret_cast<uintptr_t>(tp)) | reinterpret_cast<uintptr_t>(yp) |
                                 (WEIGHTED ? reinterpret_cast<uintptr_t>(wp) : 0)) & 15) == 0;
    }
    __device__ __forceinline__ void issue(const int off, const int slot) {
        const int mrem = m - off;
#pragma unroll
        for (int arr = 0; arr < NARR; ++arr) {
            // (the base must sit in an SGPR pair: made explicitly wave-uniform, or under register pressure the "s" operand
            // below is handed over in VGPRs and the instruction does not assemble)
            const uint64_t sv = reinterpret_cast<uint64_t>((NOGRID ? yp : (arr == 0 ? tp : (arr == 1 ? yp : wp))) + off);
            const uint64_t src = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv >> 32)) << 32) |
                                 (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(sv & 0xffffffffu));
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                const int row = (k * 64 + lane) * EL; // groups past the end re-read the block's first group (zeroed afterwards)
                const unsigned roff = (unsigned)(row < mrem ? row : 0) * (unsigned)sizeof(T);
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)((slot * NARR + arr) * ROWS * (int)sizeof(T)) + (unsigned)k * 1024u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(roff), "s"(src), "s"(dst)
                             : "memory");
            }
        }
    }
    // start of a pass over the blocks whose first block begins at row `off0`
    __device__ __forceinline__ void begin(const int off0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        it = 0;
        if (dma) issue(off0, 0);
    }
    // the block at row `off` alone, no prefetch: for passes that issue stores (loads and stores share vmcnt and do not
    // complete in order with respect to each other, so a prefetch in flight cannot be counted past them)
    __device__ __forceinline__ void stage_sync(const int off, T *&s_t, T *&s_y, T *&s_w) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        it = 0;
        if (dma) issue(off, 0);
        stage(off, -1, s_t, s_y, s_w);
    }
    // Make the block at row `off` readable and request the one at `next_off` (< 0: none).
    __device__ __forceinline__ void stage(const int off, const int next_off, T *&s_t, T *&s_y, T *&s_w) {
        const int slot = it & 1;
        ++it;
        const int mrem = m - off;
        s_t = ring + (size_t)slot * NARR * ROWS;
        s_y = NOGRID ? s_t : s_t + ROWS;
        s_w = s_y + ROWS;
        if (dma) {
            if (next_off >= 0) {
                issue(next_off, slot ^ 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KD) : "memory"); // this block has landed, the next one is in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (mrem < ROWS) { // the last block: groups past the end were clamped for the DMA -> zero data, zero weight
#pragma unroll
                for (int k = 0; k < KA; ++k) {
                    const int row = (k * 64 + lane) * EL;
                    if (row >= mrem) {
#pragma unroll
                        for (int x = 0; x < EL; ++x) {
                            s_y[row + x] = T(0);
                            if constexpr (WEIGHTED) s_w[row + x] = T(0);
                        }
                    }
                }
            }
        } else {
            T tmp[RB];
            if constexpr (!NOGRID) {
                load_rows<T, RB, 1>(tp + off, mrem, lane, false, tmp);
                store_rows<T, RB, 1>(s_t, ROWS, lane, true, tmp);
            }
            load_rows<T, RB, 1>(yp + off, mrem, lane, false, tmp);
            store_rows<T, RB, 1>(s_y, ROWS, lane, true, tmp);
            if constexpr (WEIGHTED) {
                load_rows<T, RB, 1>(wp + off, mrem, lane, false, tmp);
                store_rows<T, RB, 1>(s_w, ROWS, lane, true, tmp);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
};

#ifndef VP_BLK_MULTI_MIN_BLOCKS
#define VP_BLK_MULTI_MIN_BLOCKS 2 /* blocks per wave from which four waves per problem pay */
#endif
#ifndef VP_BLK_WAVES
#define VP_BLK_WAVES 2
#endif
#ifndef VP_BLK_LANE_ALL
#define VP_BLK_LANE_ALL 0 /* blk_fit_kernel: ALL columns folded lane-privately (no reduction inside the block loop).
   Measured in round 5 (tools/stream_ab_probe.py): 28 % faster per evaluation at m = 10 000 -- and 25 % MORE evaluations per
   fit (178 747 vs 143 560; a plain Householder sweep on the CPU takes 146 161): the residual norm that comes out of 64 merged lane triangles is ~10x
   noisier than the one the wave-wide reflectors leave, actred drowns in it and ftol fires late.  Same minima, same
   success classes, no gain in fits/s, worse parity of the trajectories: off. */
#endif
// (run-time-descriptor models evaluate every basis kind per element and their column build needs about as many registers
// again as the block itself: their kernels run ONE wave per SIMD -- 512 VGPRs -- like the resident ones, model_waves_for)
// (so do static models of more than ten columns: five exponentials in fp64 -- 12 columns of 4 rows + the carry + the lane-private
// trailing triangle spilled 176-570 VGPRs at two waves per SIMD)
#ifndef VP_BLK_PREDICT
#define VP_BLK_PREDICT 0 /* fold y - Phi(alpha) c_prev instead of y (c_prev = the coefficients of the last evaluation; see
   blk_fit_problem).  Built in round 6 to take the noise out of VP_BLK_LANE_ALL's residual norm, and it does: 16 384 fits of
   10 000 rows take 171 613 evaluations lane-private, 142 617 lane-private + predicted, 142 988 as shipped.  But the lane-private
   fold is no longer the faster one -- the wave-wide kernel moved to 20-row blocks on a computed grid since round 5 (8.04 ms);
   lane-private needs 512 VGPRs + 37 spilled there and takes 10.6 ms, 9.35 at 12 / 16 rows -- and on the shipped fold the
   prediction costs 1 % (N more fma per row) for 0.5 % fewer evaluations, with 3 more failed fits in 32 768 at m = 3 000 (a
   prediction from a far-away trial point is a LARGER data column: cancellation).  Both off (tools/stream_ab_probe.py). */
#endif
template <class M> constexpr int blk_waves() { return (M::kStatic && M::N + 1 + M::P <= 10) ? VP_BLK_WAVES : 1; }
template <typename T, int NC, bool STATIC = true> constexpr int block_rows() {
#ifdef VP_BLK_RB
    return VP_BLK_RB;
#else
    constexpr int words = NC * (int)(sizeof(T) / 4);
    if constexpr (STATIC) return words <= 14 ? 8 : ((words <= 28 || sizeof(T) == 4) ? 4 : 2);
    else return words <= 14 ? 8 : ((words <= 24 || sizeof(T) == 4) ? 4 : 2); // (one wave per SIMD: blk_waves)
#endif
}

// W waves per problem (W = 1, or 4 for launches smaller than the device, where the launch is its LONGEST fit's chain of
// evaluations -- 100+ evaluations x blocks x ~4 us -- and not its work): wave w streams the blocks w, w + W, ... into its own
// carry and private trailing triangles; the W carries are then stacked through LDS and every wave reduces the stack by the
// same stacked_qr (TSQR merges carries as easily as blocks), so all waves hold the bit-identical compressed problem and
// run the LM bookkeeping redundantly -- no LM state is ever exchanged (as in the resident multi-wave groups, Grp<W>).
// Rows per lane and block of the fit kernel on LONG problems (block_rows_long) and the waves per SIMD a block of RB rows
// per lane leaves room for (blk_fit_waves).  A block costs N dependent reduction rounds + the trailing update whatever its
// height -- about half of a 512-row block's instructions (round 5, 16 384 double-exponential fits of 10 000 rows: RB = 4 at
// three waves per SIMD 16.8 ms, RB = 8 at two 11.3, RB = 12 at one 10.3, RB = 16 at one 8.9: more resident waves hide
// nothing, the kernel is bound by the instructions of its reductions) -- so problems of at least four such blocks take the
// tallest block (<= 16 rows per lane) whose columns fit ONE wave per SIMD (512 VGPRs: 208 words of block columns) and
// whose row ring leaves room for four waves per CU (36 KiB per wave).
#ifndef VP_BLK_RB_LONG
#define VP_BLK_RB_LONG 16
#endif
#ifndef VP_BLK_RB_LONG_TC
#define VP_BLK_RB_LONG_TC 32
#endif
#ifndef VP_BLK_TC_WORDS
#define VP_BLK_TC_WORDS 264
#endif
// TC: unit weights on a uniform grid -- the grid values are computed, the ring holds the data alone (half the LDS per row)
template <typename T, class M, bool WEIGHTED, bool TC = false> constexpr int block_rows_long() {
    constexpr int RB = block_rows<T, M::N + 1 + M::P, M::kStatic>();
    constexpr int words = (M::N + 1 + M::P) * (int)(sizeof(T) / 4);
    constexpr int ring_per_row = 2 * (WEIGHTED ? 3 : (TC ? 1 : 2)) * 64 * (int)sizeof(T);
    int best = RB;
    // (more than ten columns: the 4-row block already takes one wave per SIMD -- blk_waves -- and taller ones spill 64-150 VGPRs)
    if (M::kStatic && M::N + 1 + M::P <= 10 && !(TC && (WEIGHTED || sizeof(T) != 8)))
        for (int rb = RB + 4; rb <= (TC ? VP_BLK_RB_LONG_TC : VP_BLK_RB_LONG); rb += 4)
            if (words * rb <= (TC ? VP_BLK_TC_WORDS : 208) && ring_per_row * rb <= 36 * 1024) best = rb;
    return best;
}
template <typename T, class M, int RB> constexpr int blk_fit_waves() {
    return RB > block_rows<T, M::N + 1 + M::P, M::kStatic>() ? 1 : blk_waves<M>();
}

// The streamed fit of ONE problem by the W waves of a workgroup: the body of blk_fit_kernel.  RESCUE (round 6): the re-fit of a
// problem whose Jacobian factor came out non-finite (vp_fit.hpp jac_not_finite) with every derivative column built as 2^-ks
// times its value and the coefficient entering the Kaufman columns as c 2^ks -- run by blk_fit_kernel<..., RESCUE> over the
// handle's list (launch_fit_rescue), for the models of fit_rescue_v; SELF: the caller re-fits, the problem is not pushed.
// Returns true when the problem was flagged (and the handle has re-fits switched on).
template <typename T, class M, int RB, bool WEIGHTED, int W, bool TC, bool RESCUE, bool SELF>
__device__ __forceinline__ bool blk_fit_problem(const FitArgs<T, M> &a, const int64_t b, T *ring_mem, T *s_merge, LmVars<T, M::N, M::Q> *s_lm,
                                                T (*s_cb)[M::N]) {
    constexpr int N = M::N, P = M::P, Q = M::Q, NC = N + 1 + P;
    constexpr int ROWS = 64 * RB;
    constexpr int NTRI = NC * (NC + 1) / 2;
    static_assert(W * NC <= 128, "the stacked carries must fit two rows per lane");
    static_assert(!RESCUE || fit_rescue_v<T, M>, "the scaled re-fit needs diagonal pairs");
    using G = Grp<1>; // (reductions are per wave: each wave owns its blocks and, after the merge, a full copy of the problem)
    G grp = G::make(nullptr);
    const int lane = grp.gl;
    const int wave = (W > 1) ? (int)(threadIdx.x >> 6) : 0;
    const int m = a.m;
    const T *tp = a.t + b * a.t_stride;
    const T *wp = WEIGHTED ? a.w + b * a.w_stride : nullptr;
    const T *yp = a.yw + b * (int64_t)m;

    LmVars<T, N, Q> S;
    LmOpts<T> opt;
    opt.ftol = a.ftol;
    opt.xtol = a.xtol;
    opt.gtol = a.gtol;
    opt.stepbound = a.stepbound;
    opt.patience = a.patience;
    opt.scale_diag = a.scale_diag;
    {
        T a0[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) a0[k] = a.alpha[b * Q + k];
        lm_init<T, N, Q>(S, a0);
    }
    bool flagged = false; // handed to the re-fit launch (vp_fit.hpp, jac_not_finite)
    T cbest[N];
#pragma unroll
    for (int k = 0; k < N; ++k) cbest[k] = T(0);
    int trow = 0;
    // (VP_BLK_PREDICT) the data column is folded as  y' = y_w - Phi_w(alpha) c_prev : the same projection residual, the same
    // R factor, coefficients c = c_prev + dc -- but every carry entry of the data column is O(|y'|) ~ O(||r||) near the minimum
    // instead of O(||y||), so the rounding of the sequential updates (eps * entry per fold) no longer shows in ||r||^2
    T cprev[N];
#pragma unroll
    for (int k = 0; k < N; ++k) cprev[k] = T(0);
    [[maybe_unused]] bool shifted = false;

    static_assert(!TC || (!WEIGHTED && sizeof(T) == 8), "computed grid: unit weights, fp64 (the recurrence's own condition)");
    using Ring = RowRing<T, RB, WEIGHTED, TC>;
    Ring ring;
    ring.init(ring_mem + (size_t)wave * 2 * Ring::NARR * ROWS, tp, yp, wp, m, lane);
    int parity = 0;
    // distance between a lane's consecutive row pairs on a uniform grid (RowSource::set_uniform, from the WHOLE grid)
    const T dpair = (m >= 3) ? (tp[m - 1] - tp[0]) / T(m - 1) * T(128) : T(0);
    // (TC) the lattice the handle's grid check held the grid to: t_i = t_0 + i dt
    const T tc_t0 = TC ? tp[0] : T(0), tc_dt = dpair * T(1.0 / 128.0);

    while (S.term == 0) {
        // ================= compress the problem at xt: stream the rows, fold every block into the carry =================
        T K[NC][2];
#pragma unroll
        for (int j = 0; j < NC; ++j) K[j][0] = K[j][1] = T(0);
        // kLaneAll (round 5): EVERY column is folded lane-privately -- each lane keeps an NC x NC triangle of its own rows
        // (TSQR over the 64 lanes' row sets) and a block costs NO wave-wide reduction at all; the 64 triangles are merged once
        // per evaluation (NC rounds on 64 NC rows).  A block was N dependent reduction rounds + the private trailing update
        // (~620 instructions per 512 rows, three round trips through the packed reduction); it is ~460 without a single
        // cross-lane dependency, and the reflector scalars -- private to a lane -- are useful work on all 64 lanes.
        constexpr bool kLaneAll = VP_BLK_LANE_ALL && NC <= 7;
        constexpr bool kPark = true; // the LM record waits in LDS while the rows stream (see below)
        constexpr int PT = kLaneAll ? NC : P + 1;  // columns folded lane-privately ...
        constexpr int NF = kLaneAll ? 0 : N;       // ... after NF wave-wide reflectors per block
        constexpr bool kLaneTrail = PT <= 7; // (the private triangle: PT (PT + 1) / 2 values per lane)
        T Tl[kLaneTrail ? PT : 1][kLaneTrail ? PT : 1];
        if constexpr (kLaneTrail) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) Tl[i][j] = T(0);
        }
        // the wave-uniform LM record (~45 values: twice as many VGPRs on gfx950, which has no scalar fp64 registers) is PARKED
        // in LDS while the rows stream -- the block loop is where the registers are short (round 5: 16-55 spilled VGPRs -> 0-5
        // on the double-exponential sets, the run-time-descriptor sets from 24-44 to 0-12); only the trial point stays
        T xt_now[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) xt_now[k] = S.xt[k];
        int ks[N]; // (RESCUE) binary exponents the derivative columns are scaled down by: the largest exponent of exp(-t/tau_j)
                   // over the grid, from its two ends
#pragma unroll
        for (int j = 0; j < N; ++j) ks[j] = 0;
        if constexpr (RESCUE) {
            const T t_first = tp[0], t_last = tp[m - 1];
#pragma unroll
            for (int j = 0; j < N - 1; ++j) {
                const T rt = T(1) / xt_now[j];
                const T e2 = tmax(-t_first * rt, -t_last * rt) * T(1.4426950408889634);
                ks[j] = uni((e2 > T(64) && e2 < T(1100)) ? (int)e2 : 0);
            }
        }
        if constexpr (kPark) {
            if (lane == 0) {
                s_lm[wave] = S;
#pragma unroll
                for (int k = 0; k < N; ++k) s_cb[wave][k] = cbest[k];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (wave * ROWS < m) ring.begin(wave * ROWS);
        for (int off = wave * ROWS; off < m; off += W * ROWS) {
            T *s_t, *s_y, *s_w;
            ring.stage(off, off + W * ROWS < m ? off + W * ROWS : -1, s_t, s_y, s_w);
            // (TC: a whole block takes the source without row masks, the last, partial one the masked source)
            auto fold_block = [&](auto padm_c) __attribute__((always_inline)) {
                constexpr int PADM = decltype(padm_c)::value;
                using Src = RowSource<T, RB, true, WEIGHTED ? 1 : 0, 1, 1, true, PADM, TC, TC>;
                Src src;
                src.t = s_t;
                src.w = WEIGHTED ? s_w : nullptr;
                src.m = m - off;
                src.lane = lane;
                src.vec = true;
                src.uniform = Src::kRecur && (TC || a.grid_uniform != 0) && m >= 3;
                src.delta = dpair;
                if constexpr (TC) {
                    src.tl[0] = tfma(T(off + 2 * lane), tc_dt, tc_t0);
                    src.tl[1] = tfma(T(off + 2 * lane + 1), tc_dt, tc_t0);
                }
                T Cb[NC][RB];
                load_rows_lds<T, RB, 1>(s_y, lane, Cb[N]);
                build_columns<T, M, RB, NC, Src, M::N + 1, false, true, true, -1, RESCUE>(a.mdl, xt_now, src, Cb, ks);
                if constexpr (VP_BLK_PREDICT != 0) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        T acc = Cb[N][r];
#pragma unroll
                        for (int j = 0; j < N; ++j) acc = tfma(-cprev[j], Cb[j][r], acc);
                        Cb[N][r] = acc;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kLaneTrail) {
                    if constexpr (NF > 0) stacked_qr<T, NC, NF, RB, G>(K, Cb, grp);
                    lane_trail_update<T, NC, NF, PT, RB>(Tl, Cb);
                } else {
                    stacked_qr<T, NC, NC, RB, G>(K, Cb, grp);
                }
            };
            if constexpr (TC) {
                if (m - off >= ROWS) fold_block(std::integral_constant<int, 1>{});
                else fold_block(std::integral_constant<int, 0>{});
            } else {
                fold_block(std::integral_constant<int, 0>{});
            }
            asm volatile("" ::: "memory");
        }
        if constexpr (kLaneTrail) lane_trail_merge<T, NC, NF, PT, G>(Tl, K, grp);
        if constexpr (kPark) { // un-park
            asm volatile("" ::: "memory");
            S = s_lm[wave];
#pragma unroll
            for (int k = 0; k < Q; ++k) S.ipvt[k] = uni(S.ipvt[k]);
            S.first = uni(S.first);
            S.first_tr = uni(S.first_tr);
            S.first_update = uni(S.first_update);
            S.nfev = uni(S.nfev);
            S.term = uni(S.term);
            S.status = uni(S.status);
            S.accepted = uni(S.accepted);
#pragma unroll
            for (int k = 0; k < N; ++k) cbest[k] = s_cb[wave][k];
        }
        if constexpr (W > 1) {
            // ---- merge the W carries: stack them through LDS, every wave reduces the same stack ----
            T *mg = s_merge + (size_t)parity * W * NTRI;
            parity ^= 1;
#pragma unroll
            for (int i = 0; i < NC; ++i)
                if (lane == i / 2) {
#pragma unroll
                    for (int j = i; j < NC; ++j) mg[wave * NTRI + i * NC - i * (i - 1) / 2 + (j - i)] = K[j][i % 2];
                }
            __syncthreads();
            T Cm[NC][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int rho = 2 * lane + r; // row of the stack [K_0; K_1; ...; K_{W-1}]
                const int w = rho / NC, i = rho - w * NC;
                const bool in = rho < W * NC;
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    Cm[j][r] = (in && i <= j) ? mg[w * NTRI + i * NC - i * (i - 1) / 2 + (j - i)] : T(0);
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) K[j][0] = K[j][1] = T(0);
            stacked_qr<T, NC, NC, 2, G>(K, Cm, grp);
        }
        // ================= the compressed problem: T's leading N x N block, (T_y)[0:N], the rows >= N =================
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
        if constexpr (VP_BLK_PREDICT != 0) {
            // a truncated solve returns the minimum-norm coefficients of WHAT IT WAS GIVEN: redo the evaluation on y itself
            // (the reference's minimum-norm c, src/solvers/levmar/mod.rs:51-59) -- rank-deficient trial points are rare
            if (uni(truncated) && shifted) {
#pragma unroll
                for (int k = 0; k < N; ++k) cprev[k] = T(0);
                shifted = false;
                continue;
            }
#pragma unroll
            for (int k = 0; k < N; ++k) c[k] += cprev[k];
        }
        // ||r||^2 = ||e||^2 + sum over the carry rows >= N of (T_y)^2   (only row N is non-zero)
        T sq = T(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const T v = (2 * lane + r >= N) ? K[N][r] : T(0);
            sq = tfma(v, v, sq);
        }
        T fn2 = group_sum(grp, sq);
#pragma unroll
        for (int k = 0; k < N; ++k) fn2 = tfma(e[k], e[k], fn2);
        bool ok = is_finite(fn2);
#pragma unroll
        for (int k = 0; k < N; ++k) ok = ok && is_finite(c[k]) && is_finite(Rm[k][k]);
        ok = uni(ok);
        if constexpr (VP_BLK_PREDICT != 0) {
            shifted = ok && !uni(truncated);
#pragma unroll
            for (int k = 0; k < N; ++k) cprev[k] = shifted ? c[k] : T(0);
        }

        const T fnorm1 = usqrt(fn2);
        const bool need = lm_after_eval<T, N, Q, true>(S, opt, fnorm1, ok, (long)m);
        if (S.accepted) {
#pragma unroll
            for (int k = 0; k < N; ++k) cbest[k] = c[k];
        }
        if (a.trace && trow < a.trace_rows && lane == 0 && wave == 0) {
            double *tr = a.trace + ((size_t)b * a.trace_rows + trow) * (Q + 4);
#pragma unroll
            for (int k = 0; k < Q; ++k) tr[k] = (double)S.xt[k];
            tr[Q] = (double)fnorm1;
            tr[Q + 1] = 0.0 / 0.0;
            tr[Q + 2] = (double)S.delta;
            tr[Q + 3] = (double)S.par;
        }
        ++trow;
        if (S.term != 0) break;
        if (need) {
            // Jacobian of the compressed problem in Q-coordinates: z_k = -sum_{pairs p of k} c_{basis(p)} (T_D)_p, rows >= N;
            // residual column: rows < N <- e (zero at full rank), rows >= N keep T_y.  Then MINPACK's pivoted QR.
            residual_qcoords<T, 2, N>(K[N], e, grp);
            T Zs[Q][2];
            if constexpr (M::kDiagonalPairs) {
                T Zd[1][2];
                T cj[N]; // (RESCUE: c_k 2^ks against the derivative column's 2^-ks)
#pragma unroll
                for (int k = 0; k < N; ++k) cj[k] = RESCUE ? tldexp(c[k], ks[k]) : c[k];
                jacobian_qcoords<T, M, 2, NC, G, N + 1>(a.mdl, K, cj, Zd, grp); // in place: z_k = K[N + 1 + k]
#pragma unroll
                for (int k = 0; k < Q; ++k) Zs[k][0] = K[N + 1 + k][0], Zs[k][1] = K[N + 1 + k][1];
            } else {
                jacobian_qcoords<T, M, 2, NC, G, N + 1>(a.mdl, K, c, Zs, grp);
            }
            jac_qrfac<T, 2, Q, N>(Zs, K[N], S.Rj, S.acnorm, S.ipvt, S.qtf, grp);
        }
        // (a refreshed factor with non-finite column norms: flag and re-fit, vp_fit.hpp jac_not_finite)
        if (lm_next_step<T, N, Q, true>(S, opt, need)) flagged = !RESCUE && a.rescue != nullptr;
    }

    if (lane == 0 && wave == 0) {
        vp_report rep;
        rep.termination = S.term;
        rep.n_evals = S.nfev;
        rep.objective = (double)S.objective;
        a.report[b] = rep;
        if (a.cost_out) a.cost_out[b] = (double)S.objective;
        if (a.status) a.status[b] = S.status;
        if (flagged) { // alpha[b] keeps the initial guess for the re-fit
            if (!SELF) rescue_push(a.rescue, a.rescue_slot, b);
        } else {
#pragma unroll
            for (int k = 0; k < Q; ++k) a.alpha[b * Q + k] = S.x[k];
        }
        if (a.C_out) {
#pragma unroll
            for (int k = 0; k < N; ++k) a.C_out[b * N + a.mdl.out_index(k)] = cbest[k];
        }
    }
    return flagged;
}

// RESCUE = true: the launch that re-fits the problems the streamed kernels FLAGGED (jac_not_finite) -- workgroup i takes problem
// list[2 + i], i < list[slot] -- with scaled derivative columns (blk_fit_problem<..., RESCUE>).  A separate, list-driven launch
// and not an exit path of the fit kernel: the out-of-line call at the kernel's end cost the streamed fit 2.2 % (8.12 -> 8.30 ms
// per 16 384 fits of 10 000 rows, stack frame + reserved registers) where two small launches cost 0.15 %; the slot kernel of
// the resident sets, whose waves are persistent, keeps its exit path (vp_fit2.hpp).
template <typename T, class M, int RB, bool WEIGHTED, int W = 1, bool TC = false, bool RESCUE = false>
__global__ void __launch_bounds__(64 * W, (blk_fit_waves<T, M, RB>())) blk_fit_kernel(const FitArgs<T, M> a) {
    constexpr int N = M::N, Q = M::Q, NC = N + 1 + M::P;
    constexpr int ROWS = 64 * RB;
    constexpr int NTRI = NC * (NC + 1) / 2;
    using Ring = RowRing<T, RB, WEIGHTED, TC>;
    __shared__ __attribute__((aligned(16))) T ring_mem[W * 2 * Ring::NARR * ROWS];
    __shared__ T s_merge[W > 1 ? 2 * W * NTRI : 1]; // the waves' carries, double-buffered by the evaluation's parity
    __shared__ LmVars<T, N, Q> s_lm[W];             // the parked LM records
    __shared__ T s_cb[W][N];
    int64_t b = blockIdx.x;
    if constexpr (RESCUE) {
        if (b >= (int64_t)uni(a.rescue[a.rescue_slot])) return;
        b = (int64_t)uni(a.rescue[2 + b]);
    } else {
        if (b >= a.B) return;
    }
    (void)blk_fit_problem<T, M, RB, WEIGHTED, W, TC, RESCUE, false>(a, b, ring_mem, s_merge, s_lm, s_cb);
}

// ---- trait-level evaluation at any m: set_params (+ residuals + Jacobian), src/solvers/levmar/mod.rs:42-73, 91-95, 101-201 ----
// Pass 1 streams the blocks forward with the N reflectors of Phi only (stacked_qr<NREF = N>): R, Q^T y and Q^T D in the
// carry's first N rows, ||r||^2 from the rows of the data column the reflectors leave behind -- c, cost, status.  With
// r / J wanted it also records, per block, the carry rows it STARTED from (N x NC values in LDS).  Pass 2 walks the blocks
// BACKWARDS: block i is rebuilt, its reflectors are recomputed from the recorded carry (identical arithmetic, identical
// reflectors), which also yields the block's rows of Q^T y and Q^T D; r~ = [e; (Q^T y)_>=N], J~_k = -[0; sum_p c_j(p) (Q^T D_p)_>=N]
// are then carried back through the block's reflectors in reverse order -- Q = Q_1 ... Q_nb applied exactly as the resident
// kernels' apply_q does, block by block -- and the block's rows of r and J are stored.  Exact Householder arithmetic on both
// passes: no normal equations, no R^-1 Phi products.

// stacked_qr with NREF = N that keeps what the back-application needs: u_k (carry entry of v_k), g_k; Cb[k] keeps the block
// part of v_k, Cb[j >= N] the block's rows of Q^T (.)
template <typename T, int NC, int N, int RB, class G>
__device__ __forceinline__ void stacked_qr_keep(T (&K)[NC][2], T (&Cb)[NC][RB], T (&uo)[N], T (&go)[N], G &grp) {
    const int lane = grp.gl;
    static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int NREM = NC - k;
        constexpr int own = k / 2, reg = k % 2;
        T d[NREM], top[NREM];
#pragma unroll
        for (int j = k; j < NC; ++j) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc = tfma(Cb[k][r], Cb[j][r], acc);
            d[j - k] = acc;
            top[j - k] = K[j][reg];
        }
        group_allreduce(grp, d);
        group_bcast<NREM>(grp, top, own);
#pragma unroll
        for (int j = 0; j < NREM; ++j) d[j] = tfma(top[0], top[j], d[j]);
        const T alpha = top[0], nrm2 = d[0];
        const bool live = nrm2 > num<T>::norm2_min && is_finite(nrm2);
        const T y = live ? frsqrt(nrm2) : T(0);
        const T s0 = nrm2 * y;
        const T sigma = tfma(tfma(-s0, s0, nrm2), T(0.5) * y, s0);
        const T beta = live ? -tcopysign(sigma, alpha) : ((nrm2 <= num<T>::norm2_min) ? alpha : nrm2);
        const T u = live ? alpha - beta : T(0);
        const T gk = live ? -y * frcp(tabs(alpha) + sigma) : T(0);
        uo[k] = u;
        go[k] = gk;
        K[k][reg] = (lane == own) ? beta : K[k][reg];
#pragma unroll
        for (int j = k + 1; j < NC; ++j) {
            const T f = gk * tfma(-beta, top[j - k], d[j - k]);
#pragma unroll
            for (int r = 0; r < RB; ++r) Cb[j][r] = tfma(f, Cb[k][r], Cb[j][r]);
            const T tj = tfma(f, u, top[j - k]);
            K[j][reg] = (lane == own) ? tj : K[j][reg];
        }
    });
}

// [Wc; Wb] <- Q_block [Wc; Wb] for NW columns: the block's reflectors k = N-1 .. 0, v_k = [u_k at carry row k; Cb[k]]
template <typename T, int NC, int N, int NW, int RB, class G>
__device__ __forceinline__ void stacked_apply_q(const T (&Cb)[NC][RB], const T (&u)[N], const T (&g)[N], T (&Wc)[NW][2],
                                                T (&Wb)[NW][RB], G &grp) {
    const int lane = grp.gl;
    static_for<0, N>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = N - 1 - decltype(kc)::value;
        constexpr int own = k / 2, reg = k % 2;
        T w[NW], top[NW];
#pragma unroll
        for (int z = 0; z < NW; ++z) {
            T acc = T(0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc = tfma(Cb[k][r], Wb[z][r], acc);
            w[z] = acc;
            top[z] = Wc[z][reg];
        }
        group_allreduce(grp, w);
        group_bcast<NW>(grp, top, own);
#pragma unroll
        for (int z = 0; z < NW; ++z) {
            const T f = g[k] * tfma(u[k], top[z], w[z]); // g v^T [Wc; Wb]
#pragma unroll
            for (int r = 0; r < RB; ++r) Wb[z][r] = tfma(f, Cb[k][r], Wb[z][r]);
            const T tz = tfma(f, u[k], top[z]);
            Wc[z][reg] = (lane == own) ? tz : Wc[z][reg];
        }
    });
}

// LDS the evaluate kernel needs for the per-block carry records of a problem of m rows (0 when r / J are not wanted)
template <typename T, class M, int RB> constexpr size_t eval_snap_bytes(int64_t m) {
    return (size_t)((m + 64 * RB - 1) / (64 * RB)) * M::N * (M::N + 1 + M::P) * sizeof(T);
}
constexpr size_t kEvalSnapMax = 40 * 1024; // beyond: the generic kernels (vp_generic.hpp)

template <typename T, class M, int RB, bool WEIGHTED>
__global__ void __launch_bounds__(64, (blk_waves<M>())) blk_evaluate_kernel(const EvalArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q, NC = N + 1 + P, NW = 1 + Q;
    constexpr int ROWS = 64 * RB;
    using G = Grp<1>;
    using LB = Layout<RB, 1>;
    G grp = G::make(nullptr);
    const int lane = grp.gl;
    const int64_t prob = blockIdx.x; // problem * S + rhs
    if (prob >= a.nprob) return;
    const int64_t b = prob / a.S;
    const int s = (int)(prob - b * a.S);
    const int m = a.m;
    const T *tp = a.t + b * a.t_stride;
    const T *wp = WEIGHTED ? a.w + b * a.w_stride : nullptr;
    const T *yp = a.yw + prob * (int64_t)m;
    const bool want_rj = a.r_out != nullptr || a.J_out != nullptr;
    T alpha[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];

    using Ring = RowRing<T, RB, WEIGHTED>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *ring_mem = reinterpret_cast<T *>(smem_raw);
    T *snap = ring_mem + 2 * Ring::NARR * ROWS; // [block][N][NC]: the carry rows block i started from
    Ring ring;
    ring.init(ring_mem, tp, yp, wp, m, lane);
    const T dpair = (m >= 3) ? (tp[m - 1] - tp[0]) / T(m - 1) * T(128) : T(0);
    using Src = RowSource<T, RB, true, WEIGHTED ? 1 : 0, 1, 1, true, 0>;
    auto make_src = [&](T *s_t, T *s_w, const int off) __attribute__((always_inline)) {
        Src src;
        src.t = s_t;
        src.w = WEIGHTED ? s_w : nullptr;
        src.m = m - off;
        src.lane = lane;
        src.vec = true;
        src.uniform = Src::kRecur && a.grid_uniform != 0 && m >= 3;
        src.delta = dpair;
        return src;
    };

    // ================= pass 1: forward, N reflectors per block =================
    T K[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) K[j][0] = K[j][1] = T(0);
    T sq = T(0); // this lane's share of sum_{rows >= N} (Q^T y)^2
    ring.begin(0);
    for (int off = 0, ib = 0; off < m; off += ROWS, ++ib) {
        T *s_t, *s_y, *s_w;
        ring.stage(off, off + ROWS < m ? off + ROWS : -1, s_t, s_y, s_w);
        if (want_rj) { // record the carry rows < N this block starts from (row i: lane i/2, register i%2)
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (lane == i / 2) {
#pragma unroll
                    for (int j = 0; j < NC; ++j) snap[((size_t)ib * N + i) * NC + j] = K[j][i % 2];
                }
        }
        const Src src = make_src(s_t, s_w, off);
        T Cb[NC][RB];
        load_rows_lds<T, RB, 1>(s_y, lane, Cb[N]);
        build_columns<T, M, RB, NC, Src>(a.mdl, alpha, src, Cb);
        __builtin_amdgcn_sched_barrier(0);
        stacked_qr<T, NC, N, RB, G>(K, Cb, grp);
#pragma unroll
        for (int r = 0; r < RB; ++r) sq = tfma(Cb[N][r], Cb[N][r], sq);
        asm volatile("" ::: "memory");
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

    // ================= pass 2: backward, r and J block by block =================
    const bool vec = ring.dma; // (whole 16-byte groups: m even, aligned bases) -> 2-element stores where the outputs allow it
    const bool vec_r = vec && a.r_out && ((reinterpret_cast<uintptr_t>(a.r_out + prob * (int64_t)m) & (2 * sizeof(T) - 1)) == 0);
    const bool vec_j = vec && a.J_out && ((reinterpret_cast<uintptr_t>(a.J_out) & (2 * sizeof(T) - 1)) == 0);
    T Wc[NW][2]; // carry part of [r~ | J~_1 .. J~_Q]: rows < N hold e (residual) and 0 (Kaufman columns)
#pragma unroll
    for (int z = 0; z < NW; ++z) Wc[z][0] = Wc[z][1] = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) Wc[0][i % 2] = (lane == i / 2) ? e[i] : Wc[0][i % 2];
    const int nb = (m + ROWS - 1) / ROWS;
    for (int ib = nb - 1; ib >= 0; --ib) {
        const int off = ib * ROWS;
        T *s_t, *s_y, *s_w;
        ring.stage_sync(off, s_t, s_y, s_w);
        T Kb[NC][2];
#pragma unroll
        for (int j = 0; j < NC; ++j) Kb[j][0] = Kb[j][1] = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (lane == i / 2) {
#pragma unroll
                for (int j = 0; j < NC; ++j) Kb[j][i % 2] = snap[((size_t)ib * N + i) * NC + j];
            }
        const Src src = make_src(s_t, s_w, off);
        T Cb[NC][RB];
        load_rows_lds<T, RB, 1>(s_y, lane, Cb[N]);
        build_columns<T, M, RB, NC, Src>(a.mdl, alpha, src, Cb);
        __builtin_amdgcn_sched_barrier(0);
        T u[N], g[N];
        stacked_qr_keep<T, NC, N, RB, G>(Kb, Cb, u, g, grp);
        // the block's rows of r~ and J~_k (Q-coordinates)
        T Wb[NW][RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) Wb[0][r] = Cb[N][r];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
#pragma unroll
            for (int r = 0; r < RB; ++r) Wb[1 + k][r] = T(0);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (a.mdl.pair_param(p) == k) {
                    const T cj = -dyn_get<N>(c, a.mdl.pair_basis(p));
#pragma unroll
                    for (int r = 0; r < RB; ++r) Wb[1 + k][r] = tfma(cj, Cb[N + 1 + p][r], Wb[1 + k][r]);
                }
            }
        }
        stacked_apply_q<T, NC, N, NW, RB, G>(Cb, u, g, Wc, Wb, grp);
        if (a.r_out) store_rows<T, RB, 1, VP_BLKEVAL_NT != 0>(a.r_out + prob * (int64_t)m + off, m - off, lane, vec_r, Wb[0]);
        if (a.J_out) {
#pragma unroll
            for (int k = 0; k < Q; ++k) { // J[b][k][s][m]
                T *jp = a.J_out + ((b * Q + k) * (int64_t)a.S + s) * (int64_t)m + off;
                store_rows<T, RB, 1, VP_BLKEVAL_NT != 0>(jp, m - off, lane, vec_j && ((m & 1) == 0), Wb[1 + k]);
            }
        }
        asm volatile("" ::: "memory");
    }
    (void)sizeof(LB);
}

template <typename T, class M> int launch_evaluate(const LaunchParams &p, int (*fallback)(const LaunchParams &)) {
    constexpr int RB = block_rows<T, M::N + 1 + M::P, M::kStatic>();
    using Ring = RowRing<T, RB, false>;
    const bool want_rj = p.r_out || p.J_out;
    const size_t snap = want_rj ? eval_snap_bytes<T, M, RB>(p.m) : 0;
    if (snap > kEvalSnapMax || p.m < M::N) return fallback(p); // (m < N: the padded handles of vp_batch_create stay generic)
    EvalArgs<T, M> a;
    if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
    a.t = (const T *)p.t;
    a.w = (const T *)p.w;
    a.yw = (const T *)p.yw;
    a.alpha = (const T *)p.alpha;
    a.r_out = (T *)p.r_out;
    a.J_out = (T *)p.J_out;
    a.C_out = (T *)p.C_out;
    a.cost_out = p.cost_out;
    a.status = p.status;
    a.m = p.m;
    a.S = p.S;
    a.nprob = p.B * p.S;
    a.t_stride = p.t_stride;
    a.w_stride = p.w_stride;
    a.eps = (T)p.eps;
    a.grid_uniform = p.grid_uniform;
    if (a.nprob <= 0) return VP_ERR_OK;
    const size_t lds = (size_t)2 * (p.w ? 3 : 2) * 64 * RB * sizeof(T) + snap;
    (void)sizeof(Ring);
    if (p.w) hipLaunchKernelGGL((blk_evaluate_kernel<T, M, RB, true>), dim3((unsigned)a.nprob), dim3(64), lds, p.stream, a);
    else hipLaunchKernelGGL((blk_evaluate_kernel<T, M, RB, false>), dim3((unsigned)a.nprob), dim3(64), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

template <typename T, class M> int launch_fit(const LaunchParams &p) {
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
    constexpr int RB = block_rows<T, M::N + 1 + M::P, M::kStatic>();
    // A launch that does not fill the device several times over ends when its LONGEST fit does (evaluation counts are
    // heavy-tailed: mean ~8, max 100+): four waves per problem then shorten every chain ~3.5x at no cost in throughput that
    // matters there.  Results do not depend on the choice beyond rounding (a different but equally valid TSQR tree).
    constexpr int WM = 4;
    const bool multi = p.B <= (int64_t)16 * (p.num_cus > 0 ? p.num_cus : 256) && p.m >= VP_BLK_MULTI_MIN_BLOCKS * WM * 64 * RB && WM * (M::N + 1 + M::P) <= 128;
    // (long problems: taller blocks, one wave per SIMD -- block_rows_long)
    constexpr int RLU = block_rows_long<T, M, false>(), RLW = block_rows_long<T, M, true>(), RLT = block_rows_long<T, M, false, true>();
#define VP_BLK_FIT(RB_, WEIGHTED_, W_)                                                                                  \
    hipLaunchKernelGGL((blk_fit_kernel<T, M, RB_, WEIGHTED_, W_>), dim3((unsigned)a.B), dim3(64 * (W_)), 0, p.stream, a)
#define VP_BLK_FIT_TC(RB_, W_)                                                                                          \
    hipLaunchKernelGGL((blk_fit_kernel<T, M, RB_, false, W_, true>), dim3((unsigned)a.B), dim3(64 * (W_)), 0, p.stream, a)
    // unit weights on one shared uniform grid: the grid values are computed and the ring holds the data alone
    const bool tc = RLT > RLU && !p.w && p.grid_uniform != 0 && p.m >= 3;
    if (multi) {
        if (p.w) {
            if (RLW > RB && p.m >= (int64_t)4 * WM * 64 * RLW) VP_BLK_FIT(RLW, true, WM);
            else VP_BLK_FIT(RB, true, WM);
        } else {
            if (tc && p.m >= (int64_t)4 * WM * 64 * RLT) {
                if constexpr (RLT > RLU) VP_BLK_FIT_TC(RLT, WM);
            } else if (RLU > RB && p.m >= (int64_t)4 * WM * 64 * RLU) VP_BLK_FIT(RLU, false, WM);
            else VP_BLK_FIT(RB, false, WM);
        }
    } else {
        if (p.w) {
            if (RLW > RB && p.m >= (int64_t)4 * 64 * RLW) VP_BLK_FIT(RLW, true, 1);
            else VP_BLK_FIT(RB, true, 1);
        } else {
            if (tc && p.m >= (int64_t)4 * 64 * RLT) {
                if constexpr (RLT > RLU) VP_BLK_FIT_TC(RLT, 1);
            } else if (RLU > RB && p.m >= (int64_t)4 * 64 * RLU) VP_BLK_FIT(RLU, false, 1);
            else VP_BLK_FIT(RB, false, 1);
        }
    }
#undef VP_BLK_FIT
#undef VP_BLK_FIT_TC
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// the re-fit launch of the streamed kernels' flagged problems: one wave per problem at the base block height (any m, grid values
// from memory), up to kFitRescueGrid of them; weights / per-problem grids, models outside fit_rescue_v and whatever is left go
// to the generic kernel (vp_api.hip rescue_refit)
template <typename T, class M> int launch_fit_rescue(const LaunchParams &p) {
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
        constexpr int RB = block_rows<T, M::N + 1 + M::P, M::kStatic>();
        // (a flagged fit is a chain of ~70 evaluations nothing overlaps: four waves per problem shorten it ~3.5x where the
        // problem has the blocks for them -- 4.0 -> 1.2 ms behind a batch of 16 384 problems of 10 000 rows)
        constexpr int WM = 4;
        if constexpr (WM * (M::N + 1 + M::P) <= 128) {
            if (p.m >= VP_BLK_MULTI_MIN_BLOCKS * WM * 64 * RB) {
                hipLaunchKernelGGL((blk_fit_kernel<T, M, RB, false, WM, false, true>), dim3(kFitRescueGrid), dim3(64 * WM), 0, p.stream, a);
                return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
            }
        }
        hipLaunchKernelGGL((blk_fit_kernel<T, M, RB, false, 1, false, true>), dim3(kFitRescueGrid), dim3(64), 0, p.stream, a);
        return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
    }
}

} // namespace blk
} // namespace vp
