// vp_kernels.hpp -- the __global__ kernels of the hot path (one wavefront == one problem).
//
//   evaluate_kernel : set_params (+ residuals + Jacobian + coefficients + cost) in one launch
//                     == src/solvers/levmar/mod.rs:42-73, 91-95, 101-201 for a batch
//   basis_kernel    : stand-alone Phi / dPhi evaluation (pure streaming writes; the kernel whose
//                     HBM-roofline fraction BASELINE.json asks for) == src/model/mod.rs:308,359-362
//   fit_kernel      : device-resident Levenberg-Marquardt over the VarPro functional (vp_fit.hpp)
#pragma once
#include <initializer_list>

#include "vp_core.hpp"

namespace vp {

// Minimum waves per SIMD the register allocator must leave room for (2 <=> 256 VGPRs per lane, 1 <=> 512),
// chosen from the register footprint of the NC resident columns (R rows x sizeof(T)/4 VGPRs each).
#ifndef VP_TWO_WAVE_VGPRS
#define VP_TWO_WAVE_VGPRS 200
#endif
template <typename T, int R, int NC> constexpr int waves_for() {
    return (NC * R * (int)(sizeof(T) / 4) <= VP_TWO_WAVE_VGPRS) ? 2 : 1;
}
// run-time-descriptor models (kinds and dependency table read at run time: every basis kind is evaluated per element, and
// the general Jacobian keeps q extra columns): their kernels at 12 and more rows per lane run ONE wave per SIMD (512
// VGPRs) -- at two they spilled 300-600 VGPRs
#ifndef VP_RT_ONE_WAVE
#define VP_RT_ONE_WAVE 1
#endif
template <typename T, class M, int R, int NC> constexpr int model_waves_for() {
    return (VP_RT_ONE_WAVE && !M::kStatic && R >= 12) ? 1 : waves_for<T, R, NC>();
}
// launch bound (2nd argument = waves per SIMD) for a kernel whose workgroup is one group of W waves
template <typename T, int R, int NC, int W> constexpr int group_waves_per_eu() {
    return W >= 4 ? ((waves_for<T, R, NC>() * W) / 4 > 0 ? (waves_for<T, R, NC>() * W) / 4 : 1) : waves_for<T, R, NC>();
}

// type-erased launch parameters (host side); every pointer is a device pointer
struct LaunchParams {
    const vp_model_desc *model;
    const void *t;      // [m] or [B][m]
    const void *w;      // NULL, [m] or [B][m]
    const void *yw;     // [B][S][m] weighted data
    const void *alpha;  // [B][q]
    void *alpha_out;    // fit: [B][q]
    void *r_out;        // [B][S][m] or NULL
    void *J_out;        // [B][q][S][m] or NULL
    void *C_out;        // [B][S][n] or NULL
    double *cost_out;   // [B*S] partial costs (per problem and RHS) or NULL
    int32_t *status;    // [B*S] or NULL
    void *Phi_out;      // basis: [B][n or n_alpha][m]
    void *dPhi_out;     // basis: [B][p][m]
    vp_report *report;  // fit: [B]
    const vp_lm_opts *opts;
    double *trace;      // fit diagnostics: [B][trace_rows][q+4] or NULL
    int trace_rows;
    int fit_group;      // fit: kernel selection (VP_FIT_KERNEL_*): 0 = automatic, 1 = one problem per wave, 2 = slots
    int *queue;         // fit (slot kernel): device int, the problem queue head
    double *gram_dbg;   // Gram fit kernel: evaluate-only diagnostics output (vp_debug_gram_evaluate) or null
    int num_cus;        // compute units of the device
    void *gen_ws;       // generic fallback kernels (vp_generic.hpp): workspace, gen_blocks slots
    int gen_phase;      // generic global fit in phases (right-hand sides sharded over ranks): see GenArgs::phase
    void *gen_lm_state; // [B] LM state between the phases
    double *gen_acc;    // [B][2 + q*q + q] sums between `sums` and `step` (all-reduced by the caller's collective)
    int32_t *gen_nactive;
    int gen_blocks;
    const void *mrhs_ws; // MRHS path: pointer to the handle's MrhsWs
    const void *mrhs_fws; // MRHS LM step + factorisation: the workspace the factorisation writes when it is not mrhs_ws (null: mrhs_ws)
    const void *mrhs_io; // MRHS whole-fit graph: device address of the pinned MrhsIo record (the caller's arrays of this call) or null
    int32_t *mrhs_hflag; // MRHS finish: pinned host words [active count, max evaluations] (device address) or null
    int mrhs_mode;      // MRHS stream: 0 = reduced quantities (fit), 1 = trait-level outputs
    int mrhs_init;      // MRHS LM step: 1 = initialise the state
    int mrhs_gx;        // MRHS LM step: partial-sum slots per problem (0 = the streaming kernel's grid)
    int64_t mrhs_S_global; // MRHS LM step: right-hand sides of the WHOLE problem when S is sharded (0 = S)
    int basis_flags;
    int m;
    int S;
    int64_t B;
    int64_t t_stride; // 0 (shared) or m
    int64_t w_stride; // 0 (shared) or m
    double eps;
    int grid_uniform;   // every grid of the handle passed grid_check_kernel: kernels may use the exp recurrence
    // caller-evaluated models (vp_batch_create_external): the columns come from these arrays instead of the descriptor
    const void *ext_phi;     // [B][n][ext_rows] UNWEIGHTED basis matrices, or null (descriptor model)
    const void *ext_dphi;    // [B][ext_np][ext_rows] UNWEIGHTED derivative columns in pair-table order, or null
    const int32_t *ext_pb;   // [ext_np] pair -> basis   (host pointers: copied into the kernel arguments)
    const int32_t *ext_pp;   // [ext_np] pair -> parameter
    int ext;                 // 1: external model (the fields above are meaningful)
    int ext_np;              // dependency pairs of the external model
    int ext_rows;            // rows per column in the caller's arrays (== m unless the handle padded m < n)
    // flag-and-refit (round 6): problems whose Jacobian is not representable column by column are NOT redone inside the fit
    // kernels; they append their index to this list, keep alpha0 in place and are re-fitted by a second, tiny launch of the
    // generic kernel with power-of-two column scaling (rescue_push below, gen_fit_kernel; vp_api.hip: rescue_refit)
    int32_t *rescue;         // [2 + B]: two ping-pong counters, then the problem indices; null: nothing is flagged
    int rescue_slot;         // the counter this fit appends to (0 / 1); the re-fit launch zeroes the other one
    int *rescue_used;        // (host) set to 1 by a launcher whose kernel may append to the list (the slot kernels of models
                             // with the scaled re-fit built in re-fit what they flag themselves: nothing to launch afterwards)
    const int32_t *gen_list; // generic fit kernel: fit the problems gen_list[2 + i], i < gen_list[gen_list_slot] (null: all B)
    int gen_list_slot;
    int gen_list_first;      // ... starting at entry gen_list_first (the ones before it were re-fitted by the set's own kernel)
    int gen_scale_cols;      // generic kernels: power-of-two scaling of huge basis columns (and their derivative columns)
    hipStream_t stream;
};

// workgroups of the fast re-fit launch (fit_kernel<..., RESCUE>, vp_fit.hpp): flagged problems beyond that go to the generic kernel
constexpr int kFitRescueGrid = 256;

// One fit whose Jacobian came out non-finite after an evaluation that was itself fine -- the reference forms D_k c BEFORE it
// projects (src/solvers/levmar/mod.rs:156-171), the register kernels sweep the unscaled derivative columns, so a decay time
// stepping through zero (exp(+t/0.035) = 1e153 with c = 1e-152) overflows here and not there -- hands itself over: called by
// ONE lane of the problem; the caller then skips its store of the final parameters, so alpha[b] still holds the initial guess
// the re-fit starts from.
__device__ __forceinline__ void rescue_push(int32_t *base, const int slot, const int64_t b) {
    const int i = atomicAdd(&base[slot], 1);
    base[2 + i] = (int32_t)b;
}

// ---- row-distributed loads / stores ------------------------------------------------------------
// branch-free: rows >= m read element 0 and are zeroed by a select
template <typename T, int R, int W = 1>
__device__ __forceinline__ void load_rows(const T *__restrict__ base, const int m, const int lane, const bool vec_ok,
                                          T (&out)[R]) {
    using L = Layout<R, W>;
    if constexpr (L::VW == 2) {
        if (vec_ok) { // m even and base 2*sizeof(T)-aligned: one 2-element access per register pair
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int i = L::row_of(r, lane);
                const bool in = i < m;
                using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
                const V2 v = *reinterpret_cast<const V2 *>(base + (in ? i : 0));
                out[r] = in ? v.x : T(0);
                out[r + 1] = in ? v.y : T(0);
            }
            return;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = L::row_of(r, lane);
        const bool in = i < m;
        const T v = base[in ? i : 0];
        out[r] = in ? v : T(0);
    }
}

// NT: non-temporal stores (`nt`: streamed past the caches) -- for the OUTPUT arrays of vp_basis (Phi, dPhi) and vp_evaluate (r, J):
// gigabytes per launch that nothing on the device reads back.  Round 6, 25 launches each on one box, alternating builds
// (tools/nt_store_probe.py): vp_basis (2.1 GB of stores, nothing else) median 0.388 -> 0.342 ms (0.69 -> 0.79 of 8 TB/s; the fastest
// launch 0.32 -> 0.30-0.31) -- ordinary stores leave the previous launch's lines dirty in L2 / MALL and the next launch waits for
// their write-back; vp_evaluate (r + J) 0.390 -> 0.384; residuals only: even.  The caller-evaluated-model kernels (vp_ext.hpp), which
// READ 2.7 GB next to the 1.6 GB they write, gain nothing (0.84-0.90 vs 0.90-0.92 ms) and keep ordinary stores
#ifndef VP_NT_STORES
#define VP_NT_STORES 1
#endif
template <typename T, int R, int W = 1, bool NT = false>
__device__ __forceinline__ void store_rows(T *__restrict__ base, const int m, const int lane, const bool vec_ok,
                                           const T (&in)[R]) {
    using L = Layout<R, W>;
    if constexpr (L::VW == 2) {
        if (vec_ok) {
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int i = L::row_of(r, lane);
                if (i < m) {
                    if constexpr (NT && VP_NT_STORES != 0) {
                        typedef T v2_t __attribute__((ext_vector_type(2)));
                        const v2_t v = {in[r], in[r + 1]};
                        __builtin_nontemporal_store(v, reinterpret_cast<v2_t *>(base + i));
                    } else {
                        using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
                        V2 v;
                        v.x = in[r];
                        v.y = in[r + 1];
                        *reinterpret_cast<V2 *>(base + i) = v;
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = L::row_of(r, lane);
        if (i < m) {
            if constexpr (NT && VP_NT_STORES != 0) __builtin_nontemporal_store(in[r], base + i);
            else base[i] = in[r];
        }
    }
}
__device__ __forceinline__ void store_out2(double *p, const double x, const double y) {
    if constexpr (VP_NT_STORES != 0) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t v = {x, y};
        __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(p));
    } else {
        *reinterpret_cast<double2 *>(p) = make_double2(x, y);
    }
}
// the output arrays of the trait-level calls
template <typename T, int R, int W = 1>
__device__ __forceinline__ void store_rows_out(T *__restrict__ base, const int m, const int lane, const bool vec_ok, const T (&in)[R]) {
    store_rows<T, R, W, true>(base, m, lane, vec_ok, in);
}

// zero-padded, 16-byte aligned LDS column (64*R*W rows): ONE lane address + immediate offsets, no bounds logic
template <typename T, int R, int W = 1>
__device__ __forceinline__ void load_rows_lds(const T *__restrict__ base, const int lane, T (&out)[R]) {
    using L = Layout<R, W>;
    static_assert(L::VW == 2, "padded LDS columns are read in row pairs");
    using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
    const V2 *p = reinterpret_cast<const V2 *>(base) + lane;
#pragma unroll
    for (int r = 0; r < R; r += 2) {
        const V2 v = p[(r / 2) * 64 * W];
        out[r] = v.x;
        out[r + 1] = v.y;
    }
}

template <typename T> __device__ __forceinline__ bool vec_aligned(const void *p, int m) {
    return ((m & 1) == 0) && ((reinterpret_cast<uintptr_t>(p) & (2 * sizeof(T) - 1)) == 0);
}

template <typename T, int R>
__device__ __forceinline__ RowSource<T, R> make_row_source(const T *t, const T *w, int m, int lane) {
    RowSource<T, R> s;
    s.t = t;
    s.w = w;
    s.m = m;
    s.lane = lane;
    s.vec = vec_aligned<T>(t, m) && (w == nullptr || vec_aligned<T>(w, m));
    return s;
}

template <typename T, class M> struct EvalArgs {
    M mdl;
    const T *t;
    const T *w;
    const T *yw;
    const T *alpha;
    T *r_out;
    T *J_out;
    T *C_out;
    double *cost_out;
    int32_t *status;
    int m;
    int S;
    int64_t nprob; // B*S
    int64_t t_stride, w_stride;
    T eps;
    int grid_uniform;
};

// MODE 1: coefficients / cost / status + residuals (without a residual pointer: set_params alone); 2: + Jacobian
// ALIGNED: m even and every array 16-byte aligned (checked on the host) -> 2-element accesses only
// WEIGHTED: weights present (decided on the host)
// UNIFORM (MODE 2 only): the handle's grid check found a uniform grid -- the exponentials by recurrence, and ONLY that path
// compiled.  MODE 2 is register-tight (six 32-register columns at 16 rows per lane): with both the per-row-exponential and
// the recurrence path in the kernel it spilled 65 VGPRs and was slower than per-row exponentials alone (0.59 vs 0.50 ms per
// 65 536 problems); with the recurrence alone 38 and 0.45-0.48 ms (600 of ~2 000 instructions per problem less).
// SPLIT form of MODE 2 (uniform grid, constant column last, one derivative column per parameter): the columns are never all
// in registers at once.  Phase 1: [exp_1 .. exp_NE | y] with the constant column implicit (evaluate_core_const_first) -> c,
// cost, r written.  Phase 2: the derivative columns are REBUILT (the exponentials were overwritten by the reflectors; on a
// uniform grid a rebuild is 2 exponentials per lane and column), carried through Q^T, scaled to the Kaufman columns and
// carried back through Q.  Peak: 2 NE register columns instead of 2 NE + 2 -- VP_EVAL2_SPLIT_WAVES waves per SIMD.
#ifndef VP_EVAL2_SPLIT
#define VP_EVAL2_SPLIT 1
#endif
#ifndef VP_EVAL2_TCALC
#define VP_EVAL2_TCALC 1
#endif
#ifndef VP_EVAL2_PHASE1_WAVES
#define VP_EVAL2_PHASE1_WAVES 2
#endif
#ifndef VP_EVAL2_SPLIT_WAVES
#define VP_EVAL2_SPLIT_WAVES 2
#endif
// FULL: m == 64 R W and unit weights (host dispatch): every row is valid, so the row scale is the literal 1 and there is no
// validity mask anywhere -- with masks the 16 scale values of a lane (0.0 / 1.0 selects) are common subexpressions of every
// loop that asks the row source and end up as a seventh register column.
template <typename T, class M, int R, int W, int MODE, bool UNIFORM, bool FULL> constexpr bool eval2_split() {
    return VP_EVAL2_SPLIT && FULL && UNIFORM && W == 1 && R == 16 && M::kStatic && M::kConstLast && M::kDiagonalPairs &&
           sizeof(T) == 8;
}
// The GENERAL split form (any grid, weights, any length; static models with one derivative column per parameter, r and J
// out): phase 1 = evaluate_core on [Phi | y]; phase 2 = one derivative column at a time, rebuilt, carried through Q^T, scaled,
// carried back through Q, stored.  N + 2 register columns at the peak instead of N + 1 + P.  Used where the one-sweep
// kernel's N + 1 + P columns take 168 and more register pairs per lane (double exponential + offset from 28 rows per lane
// on, triple from 24), i.e. where it spills 80-190 VGPRs at one wave per SIMD: double exponential, m = 2048, B = 32 768:
// 1.00 -> 0.70 ms, weighted 1.24 -> 0.89.  At 16 rows per lane it has no spills either but loses to the one-sweep kernel's
// ~40 (0.27 -> 0.31 ms: twice the reduction rounds and a rebuild per column); double exponential at 24: even.
#ifndef VP_EVAL2_SPLITG
#define VP_EVAL2_SPLITG 1
#endif
template <typename T, class M, int R, int W, int MODE, bool FULL> constexpr bool eval2_split_general() {
    return VP_EVAL2_SPLITG && !FULL && MODE == 2 && W == 1 && (M::N + 1 + M::P) * R >= 168 && M::kStatic && M::kDiagonalPairs &&
           sizeof(T) == 8;
}
template <typename T, class M, int R, int W, int MODE, bool UNIFORM, bool FULL> constexpr int eval_waves() {
    if (eval2_split_general<T, M, R, W, MODE, FULL>()) return model_waves_for<T, M, R, M::N + 2>();
    return eval2_split<T, M, R, W, MODE, UNIFORM, FULL>()
               ? (MODE == 2 ? VP_EVAL2_SPLIT_WAVES : VP_EVAL2_PHASE1_WAVES)
               : model_waves_for<T, M, R, M::N + 1 + M::P + ((MODE == 2 && !M::kDiagonalPairs) ? 1 + M::Q : 0)>();
}
template <typename T, class M, int R, int W, int MODE, bool ALIGNED, bool WEIGHTED, bool UNIFORM = false, bool FULL = false>
__global__ void __launch_bounds__(64 * W, (eval_waves<T, M, R, W, MODE, UNIFORM, FULL>()))
    evaluate_kernel(const EvalArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q, NC = N + 1 + P;
    __shared__ __attribute__((aligned(16))) unsigned char s_xch[group_xch_bytes<W>() > 0 ? group_xch_bytes<W>() : 16];
    using G = Grp<W>;
    G grp = G::make(s_xch);
    const int lane = grp.gl; // group lane: row ownership
    const int64_t prob = blockIdx.x; // problem * S + rhs
    if (prob >= a.nprob) return;
    const int64_t b = prob / a.S;
    const int s = (int)(prob - b * a.S);
    constexpr bool SPLIT = eval2_split<T, M, R, W, MODE, UNIFORM, FULL>();
    static_assert(!FULL || (SPLIT && !WEIGHTED && ALIGNED), "FULL exists for the split kernel only");
    const int m = SPLIT ? 64 * R * W : a.m;
    T alpha[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];

    // MODE 2 (residual + Jacobian output) is register-tight: per-row exponentials unless the grid is known to be uniform
    static_assert(!UNIFORM || MODE == 2 || FULL, "UNIFORM specialises MODE 2 (and every mode of the split kernel)");
    using Src = typename std::conditional<SPLIT, RowSource<T, R, true, 0, 1, W, true, 1, true, VP_EVAL2_TCALC != 0>,
                                          RowSource<T, R, false, WEIGHTED ? 1 : 0, ALIGNED ? 1 : 0, W, (MODE != 2) || UNIFORM, 0, UNIFORM>>::type;
    Src src;
    src.t = a.t + b * a.t_stride;
    src.w = WEIGHTED ? a.w + b * a.w_stride : nullptr;
    src.m = m;
    src.lane = lane;
    src.vec = ALIGNED;
    src.set_uniform(a.grid_uniform != 0);
    const T *yp = a.yw + prob * (int64_t)m;
    constexpr bool yvec = ALIGNED;
    if constexpr (SPLIT) {
        constexpr int NE = N - 1, NCX = NE + 1;
        using L = Layout<R, G::W>;
        T C[NCX][R];
        load_rows<T, R, W>(yp, m, lane, yvec, C[NE]);
        // every row valid, unit scale: ||s||^2 = 64 R W, s_0 = 1 -- the reflector of the scale column is a constant
        ConstReflector<T> h0;
        {
            const T sigma = tsqrt(T(64 * R * W));
            h0.live = true;
            h0.beta = -sigma;
            h0.u = T(1) + sigma;
            h0.g = T(1) / (h0.beta * h0.u);
        }
        EvalUniform<T, N> u;
        evaluate_core_const_first<T, M, R, NCX, Src, G, false, true>(a.mdl, alpha, src, a.eps, grp, h0, C, u);
        if (lane == 0) {
            if (a.status) a.status[prob] = u.ok ? VP_ST_OK : VP_ST_NONFINITE;
            if (a.cost_out) a.cost_out[prob] = 0.5 * (double)u.fn2;
        }
        if (a.C_out && lane < N) a.C_out[prob * N + lane] = dyn_get<N>(u.c, lane);
        T g1[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) g1[j] = u.g[1 + j];
        // (phase 1 alone is the set_params of a full-length problem: 3 register columns.  A set_params call is the MODE 1
        // kernel without a residual pointer: one kernel per set instead of two, the branch is uniform over the launch)
        if constexpr (MODE == 0) return;
        if (MODE == 1 && !a.r_out) return;
        // r = Q r~ = H_0 H_1 .. H_NE r~
        residual_qcoords<T, R, N>(C[NE], u.e, grp);
        apply_q_cols<T, R, NE, NCX, NE, NCX>(C, g1, grp);
        {
            T Y1[1][R];
#pragma unroll
            for (int r = 0; r < R; ++r) Y1[0][r] = C[NE][r];
            apply_const_reflector<T, R, 1, Src, G>(Y1, h0, src, grp);
            if (a.r_out) store_rows_out<T, R, W>(a.r_out + prob * (int64_t)m, m, lane, yvec, Y1[0]);
        }
        if constexpr (MODE == 1) return;
        asm volatile("" ::: "memory"); // (the grid loads of phase 2 must not be hoisted into phase 1)
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: D_p rebuilt, Z_k = -c_k P_perp D_k = -c_k Q [0; (Q^T D_k)(rows >= N)]
        if constexpr (NE >= 3) {
        // one derivative column at a time: NE + 1 register columns at the peak
        static_for<0, Q>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            T D[1][R];
            build_columns<T, M, R, 1, Src, -k, true, false, true, k>(a.mdl, alpha, src, D);
            apply_const_reflector<T, R, 1, Src, G>(D, h0, src, grp);
            apply_qt<T, R, NE, NCX, 1, G>(C, g1, D, grp);
            const T ck = -u.c[k];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool top = (r < L::VW) && (L::row_of(r, lane) < N);
                D[0][r] = top ? T(0) : ck * D[0][r];
            }
            apply_q<T, R, NE, NCX, 1, G>(C, g1, D, grp);
            apply_const_reflector<T, R, 1, Src, G>(D, h0, src, grp);
            if (a.J_out) store_rows_out<T, R, W>(a.J_out + ((b * Q + k) * (int64_t)a.S + s) * (int64_t)m, m, lane, ALIGNED, D[0]);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
        } else {
        T D[P][R];
        build_columns<T, M, R, P, Src, 0, true, false, true>(a.mdl, alpha, src, D);
        apply_const_reflector<T, R, P, Src, G>(D, h0, src, grp);
        apply_qt<T, R, NE, NCX, P, G>(C, g1, D, grp);
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const T ck = -u.c[k];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool top = (r < L::VW) && (L::row_of(r, lane) < N);
                D[k][r] = top ? T(0) : ck * D[k][r];
            }
        }
        apply_q<T, R, NE, NCX, P, G>(C, g1, D, grp);
        apply_const_reflector<T, R, P, Src, G>(D, h0, src, grp);
        if (a.J_out) {
#pragma unroll
            for (int k = 0; k < Q; ++k) // J[b][k][s][m]
                store_rows_out<T, R, W>(a.J_out + ((b * Q + k) * (int64_t)a.S + s) * (int64_t)m, m, lane, ALIGNED, D[k]);
        }
        }
        return;
    }
    if constexpr (eval2_split_general<T, M, R, W, MODE, FULL>()) {
        using L = Layout<R, G::W>;
        T C[N + 1][R];
        load_rows<T, R, W>(yp, m, lane, yvec, C[N]);
        EvalUniform<T, N> u;
        evaluate_core<T, M, R, N + 1, Src, G, true>(a.mdl, alpha, src, a.eps, grp, C, u);
        if (lane == 0) {
            if (a.status) a.status[prob] = u.ok ? VP_ST_OK : VP_ST_NONFINITE;
            if (a.cost_out) a.cost_out[prob] = 0.5 * (double)u.fn2;
        }
        if (a.C_out && lane < N) a.C_out[prob * N + lane] = dyn_get<N>(u.c, lane);
        residual_qcoords<T, R, N>(C[N], u.e, grp);
        apply_q_cols<T, R, N, N + 1, N, N + 1>(C, u.g, grp);
        if (a.r_out) store_rows_out<T, R, W>(a.r_out + prob * (int64_t)m, m, lane, yvec, C[N]);
        if (!a.J_out) return;
        static_for<0, Q>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            asm volatile("" ::: "memory"); // (the grid / weight loads of this column stay here)
            __builtin_amdgcn_sched_barrier(0);
            // the row count through an empty asm: the validity masks / 0-1 scales of this column are recomputed here instead
            // of being kept (spilled) from phase 1 as common subexpressions
            Src src2 = src;
            int m2 = m;
            asm volatile("" : "+s"(m2));
            src2.m = m2;
            T D[1][R];
            build_columns<T, M, R, 1, Src, -k, false, false, true, k>(a.mdl, alpha, src2, D);
            apply_qt<T, R, N, N + 1, 1, G>(C, u.g, D, grp);
            const T ck = -u.c[k];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool top = (r < L::VW) && (L::row_of(r, lane) < N);
                D[0][r] = top ? T(0) : ck * D[0][r];
            }
            apply_q<T, R, N, N + 1, 1, G>(C, u.g, D, grp);
            store_rows_out<T, R, W>(a.J_out + ((b * Q + k) * (int64_t)a.S + s) * (int64_t)m, m2, lane, ALIGNED, D[0]);
        });
        return;
    }
    T C[NC][R];
    load_rows<T, R, W>(yp, m, lane, yvec, C[N]);

    EvalUniform<T, N> u;
    evaluate_core<T, M, R, NC, Src, G>(a.mdl, alpha, src, a.eps, grp, C, u);

    if (lane == 0) {
        if (a.status) a.status[prob] = u.ok ? VP_ST_OK : VP_ST_NONFINITE;
        if (a.cost_out) a.cost_out[prob] = 0.5 * (double)u.fn2;
    }
    if (a.C_out && lane < N) a.C_out[prob * N + lane] = dyn_get<N>(u.c, lane);

    if constexpr (MODE >= 1) {
        if (MODE == 1 && !a.r_out) return; // set_params alone (launch_evaluate has no MODE 0 kernels any more)
        residual_qcoords<T, R, N>(C[N], u.e, grp);
        T *rp = a.r_out ? a.r_out + prob * (int64_t)m : nullptr;
        if constexpr (MODE == 1) {
            // r = Q r~ : back-sweep on the data column only, in place
            apply_q_cols<T, R, N, NC, N, N + 1>(C, u.g, grp);
            if (rp) store_rows_out<T, R, W>(rp, m, lane, yvec, C[N]);
        } else if constexpr (M::kDiagonalPairs) {
            // J~_k = -c_k (Q^T D_k) in place, then ONE back-sweep over [r~ | J~_1 .. J~_q] in place
            T Zs[1][R];
            jacobian_qcoords<T, M, R, NC>(a.mdl, C, u.c, Zs, grp);
            apply_q_cols<T, R, N, NC, N, NC>(C, u.g, grp);
            if (rp) store_rows_out<T, R, W>(rp, m, lane, yvec, C[N]);
            if (a.J_out) {
#pragma unroll
                for (int k = 0; k < Q; ++k) { // J[b][k][s][m]
                    T *jp = a.J_out + ((b * Q + k) * (int64_t)a.S + s) * (int64_t)m;
                    store_rows_out<T, R, W>(jp, m, lane, ALIGNED, C[N + 1 + k]);
                }
            }
        } else {
            T Z[1 + Q][R];
            {
                T Zs[Q][R];
                jacobian_qcoords<T, M, R, NC>(a.mdl, C, u.c, Zs, grp);
#pragma unroll
                for (int r = 0; r < R; ++r) Z[0][r] = C[N][r];
#pragma unroll
                for (int k = 0; k < Q; ++k)
#pragma unroll
                    for (int r = 0; r < R; ++r) Z[1 + k][r] = Zs[k][r];
            }
            apply_q<T, R, N, NC, 1 + Q>(C, u.g, Z, grp);
            if (rp) store_rows_out<T, R, W>(rp, m, lane, yvec, Z[0]);
            if (a.J_out) {
#pragma unroll
                for (int k = 0; k < Q; ++k) { // J[b][k][s][m]
                    T *jp = a.J_out + ((b * Q + k) * (int64_t)a.S + s) * (int64_t)m;
                    store_rows_out<T, R, W>(jp, m, lane, ALIGNED, Z[1 + k]);
                }
            }
        }
    }
}

template <typename T, class M> struct BasisArgs {
    M mdl;
    const T *t;
    const T *alpha;
    T *Phi_out;
    T *dPhi_out;
    int m;
    int skip_invariant;
    int n_phi_cols; // columns per problem in Phi_out
    int64_t B;
    int64_t t_stride;
};

// Stand-alone Phi/dPhi: reads q scalars (+ the shared grid from L2), writes (n + p) * m scalars per
// problem with 16-byte-per-lane fully coalesced stores: HBM-write-bound by construction.
// W == 1: EIGHT problems per workgroup, one per wave: 0.377 vs 0.403 ms (4 per workgroup) vs 0.429 (16) on the same box
// (tools/basis_var_probe.py).  What was measured around it (tools/store_pattern.hip, pure stores of the same 2.1 GB):
// one 16-byte store per thread in a linear sweep 0.313 ms (6.85 TB/s), hipMemset 0.33, one wave per problem 0.36-0.38,
// persistent waves 0.41-0.46; an element-wise form of this kernel (thread = one row pair of one column) follows the
// linear sweep but its extra index / reciprocal arithmetic per 32 bytes makes the clocks sag (0.36-0.50 ms); stores
// issued row pair by row pair at 41 VGPRs: no change; non-temporal stores: 5-15 % slower.
#ifndef VP_BASIS_WPB
#define VP_BASIS_WPB 8
#endif
#ifndef VP_BASIS_ROWPAIR
#define VP_BASIS_ROWPAIR 1 /* multi-exponential fp64: the thread-per-row-pair kernel below */
#endif
// problems (waves) per workgroup of basis_kernel: run-time-descriptor models at 12 and more rows per lane need the 512
// VGPRs of one wave per SIMD (4 waves per workgroup), see model_waves_for
template <class M, int R, int W> constexpr int basis_ppb() {
    return (W == 1) ? ((!M::kStatic && R >= 12 && VP_RT_ONE_WAVE) ? 4 : VP_BASIS_WPB) : 1;
}
template <typename T, class M, int R, int W, bool ALIGNED>
__global__ void __launch_bounds__((64 * W * basis_ppb<M, R, W>())) basis_kernel(const BasisArgs<T, M> a) {
    constexpr int N = M::N, P = M::P, Q = M::Q, NC = N + 1 + P;
    constexpr int PPB = basis_ppb<M, R, W>();
    const int lane = (W == 1) ? (int)(threadIdx.x & 63u) : (int)threadIdx.x; // group lane
    const int64_t b = (int64_t)blockIdx.x * PPB + ((W == 1) ? (int)(threadIdx.x >> 6) : 0);
    if (b >= a.B) return;
    const int m = a.m;
    T alpha[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) alpha[k] = a.alpha[b * Q + k];
    // (per-row exponentials: the stand-alone Phi kernel is the model-evaluation API and keeps its 2-ulp accuracy; the
    // uniform-grid recurrence would cut its VALU work 3x but only buys ~5 % here -- the kernel is store-bound)
    using Src = RowSource<T, R, false, 0, ALIGNED ? 1 : 0, W>;
    Src src;
    src.t = a.t + b * a.t_stride;
    src.w = nullptr;
    src.m = m;
    src.lane = lane;
    src.vec = ALIGNED;
    T C[NC][R];
    build_columns<T, M, R, NC, Src>(a.mdl, alpha, src, C);
    if (a.Phi_out) {
        int col = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (a.skip_invariant && a.mdl.kind(j) == VP_BASIS_CONST) continue;
            T *p = a.Phi_out + (b * a.n_phi_cols + col) * (int64_t)m;
            store_rows_out<T, R, W>(p, m, lane, ALIGNED, C[j]);
            ++col;
        }
    }
    if (a.dPhi_out) {
#pragma unroll
        for (int pidx = 0; pidx < P; ++pidx) {
            T *p = a.dPhi_out + (b * P + pidx) * (int64_t)m;
            store_rows_out<T, R, W>(p, m, lane, ALIGNED, C[N + 1 + pidx]);
        }
    }
}

// Multi-exponential models in fp64 with aligned arrays (the case the HBM target is priced on): a THREAD owns one row pair
// of one problem -- it evaluates the NE exponentials and their derivatives there, stores 2 NE x 16 bytes and retires; a
// 256-thread workgroup covers 512 rows.  Workgroups are dispatched in address order, so every output column is written by
// short-lived waves sweeping forward -- the closest this output layout gets to a memset's store pattern
// (tools/store_pattern.hip: linear sweep 6.85 TB/s, one wave per problem 5.7-6.1): 0.352 vs 0.376 ms for 2.15 GB.  Same
// arithmetic per element as build_columns; tau's reciprocal is recomputed per thread (8 of ~125 instructions).
template <class M> __global__ void __launch_bounds__(256) basis_rowpair_kernel(const BasisArgs<double, M> a, const int blocks_per_problem) {
    constexpr int NE = M::kConstLast ? M::N - 1 : M::N, Q = M::Q, P = M::P;
    const int64_t b = blockIdx.x / (unsigned)blocks_per_problem;
    const int piece = (int)(blockIdx.x - (unsigned)b * (unsigned)blocks_per_problem);
    const int i = 2 * (piece * 256 + (int)threadIdx.x);
    const int m = a.m;
    if (i >= m) return;
    const double2 tv = *reinterpret_cast<const double2 *>(a.t + b * a.t_stride + i);
    double *phi = a.Phi_out ? a.Phi_out + b * (int64_t)a.n_phi_cols * m + i : nullptr;
    double *dphi = a.dPhi_out ? a.dPhi_out + b * (int64_t)P * m + i : nullptr;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const double tau = a.alpha[b * Q + k];
        const double rt = frcp(tau), rt2 = rt * rt;
        double2 f, d;
        f.x = texp(-div_refined(tv.x, tau, rt));
        f.y = texp(-div_refined(tv.y, tau, rt));
        d.x = (f.x * tv.x) * rt2;
        d.y = (f.y * tv.y) * rt2;
        if (phi) store_out2(phi + (int64_t)k * m, f.x, f.y);
        if (dphi) store_out2(dphi + (int64_t)k * m, d.x, d.y);
    }
    if (M::kConstLast && !a.skip_invariant && phi) store_out2(phi + (int64_t)NE * m, 1.0, 1.0);
}

// The same with a thread owning one row pair of ONE column: when Phi is written without its constant column, Phi [B][NE][m]
// and dPhi [B][NE][m] have the same shape and thread e writes 16 bytes at the same flat offset of both -- workgroups in
// dispatch order sweep each array linearly from end to end (two memset-like streams instead of 2 NE interleaved ones).
#ifndef VP_BASIS_FLAT
#define VP_BASIS_FLAT 1
#endif

template <class M> __global__ void __launch_bounds__(256) basis_flat_kernel(const BasisArgs<double, M> a, const int blocks_per_col) {
    constexpr int NE = M::kConstLast ? M::N - 1 : M::N, Q = M::Q;
    const unsigned col = blockIdx.x / (unsigned)blocks_per_col; // b * NE + k
    const int piece = (int)(blockIdx.x - col * (unsigned)blocks_per_col);
    const int i = 2 * (piece * 256 + (int)threadIdx.x);
    const int m = a.m;
    if (i >= m) return;
    const unsigned b = col / (unsigned)NE;
    const int k = (int)(col - b * (unsigned)NE);
    const double2 tv = *reinterpret_cast<const double2 *>(a.t + (int64_t)b * a.t_stride + i);
    const double tau = a.alpha[(int64_t)b * Q + k];
    const double rt = frcp(tau), rt2 = rt * rt;
    double2 f, d;
    f.x = texp(-div_refined(tv.x, tau, rt));
    f.y = texp(-div_refined(tv.y, tau, rt));
    d.x = (f.x * tv.x) * rt2;
    d.y = (f.y * tv.y) * rt2;
    const int64_t off = (int64_t)col * m + i;
    if (a.Phi_out) store_out2(a.Phi_out + off, f.x, f.y);
    if (a.dPhi_out) store_out2(a.dPhi_out + off, d.x, d.y);
}

// ---- host-side launch templates ------------------------------------------------------------------
// every per-problem / per-column slice starts at a multiple of m elements from its base: with m even and
// 16-byte aligned bases all 2-element accesses are aligned
template <typename T> inline bool host_aligned(int m, std::initializer_list<const void *> ptrs) {
    if (m & 1) return false;
    for (const void *q : ptrs)
        if (q && (reinterpret_cast<uintptr_t>(q) & (2 * sizeof(T) - 1)) != 0) return false;
    return true;
}

template <class M> inline bool bind_model(const vp_model_desc &d, M &out) {
    if constexpr (M::kStatic) {
        (void)d;
        (void)out;
        return true;
    } else {
        return make_rt_model(d, out);
    }
}

template <typename T, class M, int R, int W = 1> int launch_evaluate(const LaunchParams &p) {
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
    dim3 grid((unsigned)a.nprob), block(64 * W);
    // run-time-descriptor models only get the element-wise (any alignment) variants: half the instantiations of a
    // path that is not throughput-critical
    const bool aligned = M::kStatic && host_aligned<T>(p.m, {p.t, p.w, p.yw, p.r_out, p.J_out});
    const int mode = p.J_out ? 2 : (p.r_out ? 1 : 0);
    if constexpr (M::kStatic && sizeof(T) == 8 && (R > 2)) {
        if (mode == 2 && aligned && p.grid_uniform != 0 && p.m >= 3) { // (RowSource::set_uniform's own conditions)
            if constexpr (eval2_split<T, M, R, W, 2, true, true>()) {
                if (!p.w && p.m == 64 * R * W) { // a full-length, unweighted problem: the split kernel
                    hipLaunchKernelGGL((evaluate_kernel<T, M, R, W, 2, true, false, true, true>), grid, block, 0, p.stream, a);
                    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
                }
            }
            if (p.w) hipLaunchKernelGGL((evaluate_kernel<T, M, R, W, 2, true, true, true>), grid, block, 0, p.stream, a);
            else hipLaunchKernelGGL((evaluate_kernel<T, M, R, W, 2, true, false, true>), grid, block, 0, p.stream, a);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
    }
    if constexpr (eval2_split<T, M, R, W, 1, true, true>()) {
        // set_params (+ residuals) of a full-length, unweighted problem on a uniform grid: phase 1 of the split kernel
        if (mode < 2 && aligned && p.grid_uniform != 0 && !p.w && p.m == 64 * R * W) {
            hipLaunchKernelGGL((evaluate_kernel<T, M, R, W, 1, true, false, true, true>), grid, block, 0, p.stream, a);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
    }
    const int variant = (mode == 0 ? 1 : mode) * 4 + (aligned ? 2 : 0) + (p.w ? 1 : 0);
#define VP_EV(MODE_, AL_, W_)                                                                                          \
    case (MODE_) * 4 + (AL_) * 2 + (W_):                                                                               \
        if constexpr (M::kStatic || (AL_) == 0)                                                                        \
            hipLaunchKernelGGL((evaluate_kernel<T, M, R, W, MODE_, (AL_) != 0, (W_) != 0>), grid, block, 0, p.stream, \
                               a);                                                                                     \
        break;
    switch (variant) {
        VP_EV(1, 0, 0) VP_EV(1, 0, 1) VP_EV(1, 1, 0) VP_EV(1, 1, 1)
        VP_EV(2, 0, 0) VP_EV(2, 0, 1) VP_EV(2, 1, 0) VP_EV(2, 1, 1)
    }
#undef VP_EV
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

template <typename T, class M, int R, int W = 1> int launch_basis(const LaunchParams &p) {
    BasisArgs<T, M> a;
    if (!bind_model(*p.model, a.mdl)) return VP_ERR_UNSUPPORTED;
    a.t = (const T *)p.t;
    a.alpha = (const T *)p.alpha;
    a.Phi_out = (T *)p.Phi_out;
    a.dPhi_out = (T *)p.dPhi_out;
    a.m = p.m;
    a.skip_invariant = (p.basis_flags & VP_BASIS_SKIP_INVARIANT) ? 1 : 0;
    int ncols = 0;
    for (int j = 0; j < p.model->n_basis; ++j)
        if (!(a.skip_invariant && p.model->kind[j] == VP_BASIS_CONST)) ++ncols;
    a.n_phi_cols = ncols;
    a.B = p.B;
    a.t_stride = p.t_stride;
    if (a.B <= 0) return VP_ERR_OK;
#if VP_BASIS_ROWPAIR
    if constexpr (M::kStatic && M::kDiagonalPairs && sizeof(T) == 8) {
        const int bpp = (p.m / 2 + 255) / 256;
#if VP_BASIS_FLAT
        {
            constexpr int NE = M::kConstLast ? M::N - 1 : M::N;
            if (host_aligned<T>(p.m, {p.t, p.Phi_out, p.dPhi_out}) && a.n_phi_cols == NE && M::P == NE &&
                a.B * NE * bpp < ((int64_t)1 << 31)) {
                hipLaunchKernelGGL((basis_flat_kernel<M>), dim3((unsigned)(a.B * NE * bpp)), dim3(256), 0, p.stream, a, bpp);
                return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
            }
        }
#endif
        if (host_aligned<T>(p.m, {p.t, p.Phi_out, p.dPhi_out}) && a.B * bpp < ((int64_t)1 << 31)) {
            hipLaunchKernelGGL((basis_rowpair_kernel<M>), dim3((unsigned)(a.B * bpp)), dim3(256), 0, p.stream, a, bpp);
            return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
        }
    }
#endif
    constexpr int PPB = basis_ppb<M, R, W>();
    const dim3 grid((unsigned)((a.B + PPB - 1) / PPB)), block(64 * W * PPB);
    if (host_aligned<T>(p.m, {p.t, p.Phi_out, p.dPhi_out}))
        hipLaunchKernelGGL((basis_kernel<T, M, R, W, true>), grid, block, 0, p.stream, a);
    else hipLaunchKernelGGL((basis_kernel<T, M, R, W, false>), grid, block, 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
