// vp_extfit.hpp -- a BATCHED Levenberg-Marquardt fit of CALLER-EVALUATED models by reverse communication.
//
// == LevMarSolver::fit -> levenberg_marquardt::LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:238-254, call
// site :247) over ANY `SeparableNonlinearModel` (src/model/mod.rs:239-363), for a batch.  The reference's driver calls
// the trait: set_params(x_trial) -> model.eval() (:43-45), residuals() (:91-95) and, at accepted points only,
// jacobian() -> model.eval_partial_deriv(k) (:141).  A model the device cannot evaluate crosses the C ABI as those VALUES;
// everything else of the loop stays on the device:
//
//   vp_fit_begin            one LM record per problem (the LmVars of vp_lm_core.hpp, structure-of-arrays) from alpha0
//   vp_fit_step_with_basis  TWO launches per step, with Phi (and dPhi) at the current trial points:
//     ext_fit_eval_kernel   one WAVEFRONT per active problem, columns in registers exactly as ext_evaluate_kernel (vp_ext.hpp):
//                             Phi_w = W Phi, Householder QR applied to [y_w | W dPhi], truncated solve, ||r||        :42-73
//                             with derivative columns at hand: Kaufman columns in Q-coordinates, MINPACK qrfac of the
//                             m x q Jacobian with Q_J^T r alongside (jac_qrfac)                                   :101-201
//                           -> per problem { ||r||, ok, c, [R_J, Q_J^T r, column norms, pivots] }: ~q^2 + 3q + n + 2 scalars.
//                           HBM-bound: it streams m (n + 1 + p) scalars per active problem in and nothing else out.
//     ext_fit_lm_kernel     one LANE per problem (the wave-uniform bookkeeping of 64 problems at once instead of 64 times in a
//                           row on 64 redundant lanes: lmpar alone is thousands of dependent fp64 instructions at q = 4):
//                             trust-region update, accept / reject, termination tests              (lm_after_eval)
//                             gradient test, diag, lmpar, predicted reduction, next trial point    (lm_next_step)
//                           -> alpha_trial [B][q], what every problem wants next [B], the best point so far
//   vp_fit_end              parameters, coefficients and MinimizationReport of every problem
//
// The Jacobian J [B][q][m] and the residuals never leave the chip (the trait-level route -- vp_evaluate_with_basis and one
// host LM driver per problem -- writes and copies them at every accepted point: 1 GB per step at the headline shape).
// The Jacobian factor of a trial point is formed BEFORE the LM kernel decides whether the point is accepted (the columns
// are in registers then, not later): it lands in a candidate slot of the record and replaces the current factor on
// acceptance; on a rejected step it is dropped, as the reference never asks for it.
//
// Two protocols (vp_fit_begin flags):
//   eager (default)              every step carries Phi AND dPhi at the trial point; a step is one LM iteration
//   VP_FIT_DERIVATIVES_ON_ACCEPT the driver's own order: a trial point is evaluated with Phi alone; a problem that
//                                accepts it asks for the derivative columns AT THAT POINT (want = basis | derivatives, same
//                                alpha_trial) and forms its Jacobian in the next step -- eval_partial_deriv is then called
//                                exactly as often as the reference calls it, at the price of a second pass over Phi
// Evaluation counts (MinimizationReport::number_of_evaluations) count residual evaluations, as the reference's do; the
// Jacobian pass of the second protocol is not one.
#pragma once
#include <vector>

#include "vp_ext.hpp"
#include "vp_lm_core.hpp"

namespace vp {
namespace ext {

#ifndef VP_EXTFIT_TWO_WAVE_VGPRS
#define VP_EXTFIT_TWO_WAVE_VGPRS 200
#endif
enum { EXTFIT_WANT_BASIS = 1, EXTFIT_WANT_DERIVS = 2 }; // == VP_WANT_* (include/varpro_hip.h)
enum { EXTFIT_PH_EVAL = 0, EXTFIT_PH_JAC = 1 };

// The per-problem state, structure-of-arrays: field f of problem b is base[f * B + b] (the LM kernel's lane b reads and
// writes it coalesced; the evaluation kernel's wave b touches a handful of scalars).  Q parameters, up to VP_MAX_BASIS
// coefficients.
template <int Q> struct ExtFitLayout {
    // fields of the problem's scalar type
    static constexpr int X = 0, XT = X + Q, DIAG = XT + Q, QTF = DIAG + Q, ACN = QTF + Q, RJ = ACN + Q, SC = RJ + Q * Q;
    // SC + {0 fnorm, 1 delta, 2 par, 3 xnorm, 4 gnorm, 5 pnorm, 6 prered, 7 dirder, 8 objective}
    static constexpr int CBEST = SC + 9, C_FN = CBEST + VP_MAX_BASIS, C_C = C_FN + 1, C_RJ = C_C + VP_MAX_BASIS;
    static constexpr int C_ACN = C_RJ + Q * Q, C_QTF = C_ACN + Q, NT = C_QTF + Q;
    // 32-bit fields
    static constexpr int IPVT = 0, FIRST = IPVT + Q, FIRST_TR = FIRST + 1, FIRST_UP = FIRST_TR + 1, NFEV = FIRST_UP + 1;
    static constexpr int TERM = NFEV + 1, STATUS = TERM + 1, WANT = STATUS + 1, PHASE = WANT + 1, C_OK = PHASE + 1;
    static constexpr int C_HASJ = C_OK + 1, C_IPVT = C_HASJ + 1, NI = C_IPVT + Q;
    // bytes per problem (+ 8 per handle for the alignment of the 32-bit block: added by the host)
    static constexpr size_t bytes(size_t tsize) { return (size_t)NT * tsize + (size_t)NI * 4; }
};
// The same offsets for a RUN-TIME parameter count: the generic step kernel (vp_gen_extfit.hpp: any (n, pairs, q) the header
// admits, any number of right-hand sides) is compiled once and writes the candidate slot of whichever ExtFitLayout<Q> the LM
// kernel of the handle's q reads.
struct ExtFitOffsets {
    int C_FN, C_C, C_RJ, C_ACN, C_QTF, NT;
    int TERM, WANT, C_OK, C_HASJ, C_IPVT;
};
__host__ __device__ constexpr ExtFitOffsets extfit_offsets(const int Q) {
    ExtFitOffsets o{};
    const int SC = 5 * Q + Q * Q, CBEST = SC + 9;
    o.C_FN = CBEST + VP_MAX_BASIS;
    o.C_C = o.C_FN + 1;
    o.C_RJ = o.C_C + VP_MAX_BASIS;
    o.C_ACN = o.C_RJ + Q * Q;
    o.C_QTF = o.C_ACN + Q;
    o.NT = o.C_QTF + Q;
    const int NFEV = Q + 3;
    o.TERM = NFEV + 1;
    o.WANT = o.TERM + 2;
    o.C_OK = o.WANT + 2;
    o.C_HASJ = o.C_OK + 1;
    o.C_IPVT = o.C_HASJ + 1;
    return o;
}
template <int Q> constexpr bool extfit_offsets_match() {
    using F = ExtFitLayout<Q>;
    constexpr ExtFitOffsets o = extfit_offsets(Q);
    return o.C_FN == F::C_FN && o.C_C == F::C_C && o.C_RJ == F::C_RJ && o.C_ACN == F::C_ACN && o.C_QTF == F::C_QTF && o.NT == F::NT &&
           o.TERM == F::TERM && o.WANT == F::WANT && o.C_OK == F::C_OK && o.C_HASJ == F::C_HASJ && o.C_IPVT == F::C_IPVT;
}
static_assert(extfit_offsets_match<1>() && extfit_offsets_match<2>() && extfit_offsets_match<3>() && extfit_offsets_match<4>() &&
                  extfit_offsets_match<5>() && extfit_offsets_match<6>() && extfit_offsets_match<7>() && extfit_offsets_match<8>(),
              "extfit_offsets must mirror ExtFitLayout");

// the 32-bit fields start behind the NT * B scalars, 8-byte aligned
template <typename T> __host__ __device__ inline int32_t *extfit_ints_rt(void *state, int64_t B, const int NT) {
    size_t off = (size_t)NT * sizeof(T) * (size_t)B;
    off = (off + 7) & ~(size_t)7;
    return reinterpret_cast<int32_t *>(reinterpret_cast<char *>(state) + off);
}
template <typename T, int Q> __host__ __device__ inline int32_t *extfit_ints(void *state, int64_t B) {
    return extfit_ints_rt<T>(state, B, ExtFitLayout<Q>::NT);
}

template <typename T> struct ExtFitArgs {
    const T *phi;  // [B][N][m]   UNWEIGHTED, at alpha_trial of the previous step (alpha0 for the first)
    const T *dphi; // [B][np][m]  UNWEIGHTED, pair-table order (null: no problem may want derivatives)
    const T *w;
    const T *yw;   // [B][m]
    void *state;   // ExtFitLayout arrays
    const T *alpha0; // [B][q]: read when init != 0
    T *alpha_best;   // [B][q] best point so far      (the handle's parameter array)
    T *C_best;       // [B][n] its coefficients       (the handle's coefficient array)
    double *cost;    // [B]
    int32_t *status; // [B]
    vp_report *report; // [B] termination == 0 while the problem is running
    T *alpha_trial;  // [B][q] out: where Phi (and dPhi) are wanted next; the final parameters once a problem is done
    int32_t *want;   // [B] out: EXTFIT_WANT_* bits, 0 = done
    int32_t *nactive; // [2] device counters: [step & 1] receives this step's number of still-active problems (one atomic per
                      // wavefront of the LM kernel), [(step + 1) & 1] is zeroed for the next step
    int step;
    int32_t pb[VP_MAX_PAIRS], pp[VP_MAX_PAIRS]; // pb: SWEEP column of the pair's basis function (see perm)
    int32_t perm[VP_MAX_BASIS]; // sweep column j is basis function perm[j]: the INVARIANT functions (no derivative pair) first.
                                // Their reflectors are then the same in every evaluation of a fit and their rounding cancels in
                                // actred = 1 - (||r_trial|| / ||r||)^2, which the ftol test reads at the 30-eps level (the headline
                                // problems as an external model, constant column last: +0.69 evaluations per fit against the
                                // reference algorithm on the CPU, within 3 on 90.8 %; invariant columns first: -0.08, 98.2 % -- what vp_fit's implicit
                                // constant-first sweep has; tools/extfit_headline_probe.py)
    int np;
    int n; // basis functions (the LM kernel is compiled per Q only)
    int m;
    int64_t B;
    int64_t w_stride;
    T eps;
    LmOpts<T> o;
    int init; // first step after vp_fit_begin: records are created from alpha0
    int lazy; // VP_FIT_DERIVATIVES_ON_ACCEPT
    int vec;
    // several right-hand sides (generic step kernel only): the problem's S data columns share alpha; the residual is the
    // stacked one (m S entries, src/solvers/levmar/mod.rs:91-95, 172-186) and the coefficients are [B][S][n]: the evaluation
    // writes those of the trial point to C_trial, the LM kernel copies them to C_best when the point is accepted
    int S;            // 1 on every specialised path
    const T *C_trial; // [B][S][n] (S > 1) or null
    // the ACTIVE SET, compacted (round 6): the LM kernel of step k writes the indices of the problems still running to
    // active_out (their number is nactive[k & 1]); the evaluation launch of step k + 1 covers only those -- workgroup i takes
    // problem active_in[i], i < *active_count; its grid is the host's last known count (an upper bound: counts never grow).
    // The first step of a fit has no list (active_in == null: workgroup b <-> problem b).
    const int32_t *active_in;
    const int32_t *active_count;
    int32_t *active_out;
    int64_t grid_problems; // evaluation launch: workgroups (== B without a list)
};

// the problem of workgroup `wg` (-1: nothing to do)
template <typename T> __device__ __forceinline__ int64_t extfit_problem_of(const ExtFitArgs<T> &a, const int64_t wg) {
    if (!a.active_in) return wg < a.B ? wg : -1;
    const int cnt = uni(*a.active_count);
    return wg < cnt ? (int64_t)uni(a.active_in[wg]) : -1;
}

template <typename T, int R, int N, int P, int Q, int W> constexpr int extfit_waves() {
    // resident: N + 1 + P columns during the sweep, 1 + P + Q afterwards.  Two waves per SIMD (256 VGPRs) whenever the
    // columns leave ~60 registers for the rest: one wave computes while the other loads -- a wave cannot overlap its own
    // loads with its own sweep (the columns ARE the registers).  (2nd launch-bound argument: waves per SIMD of a workgroup
    // of W waves -- W >= 4 spreads over the 4 SIMDs of the CU.)
    constexpr int cols = (N + 1 + P) > (1 + P + Q) ? (N + 1 + P) : (1 + P + Q);
    constexpr int per_simd = (cols * R * (int)(sizeof(T) / 4) <= VP_EXTFIT_TWO_WAVE_VGPRS) ? 2 : 1;
    return W >= 4 ? (per_simd * W) / 4 : per_simd;
}

// ---- launch 1: the evaluation (and, with derivative columns at hand, the Jacobian factor) of every active problem ----
// W wavefronts per problem (one workgroup): rows dealt to 64 W lanes, R per lane -- m <= 64 R W.  W > 1 where the columns of
// one problem do not leave a single wave room for a second wave on its SIMD (vp_device.hpp Grp: reductions through LDS)
template <typename T, int N, int P, int Q, int R, int W>
__global__ void __launch_bounds__(64 * W, (extfit_waves<T, R, N, P, Q, W>())) ext_fit_eval_kernel(const ExtFitArgs<T> a) {
    constexpr int NC = N + 1 + P;
    static_assert(W == 1 || Q * (Q + 1) / 2 + Q <= VP_XV, "the Gram round of jac_qrfac must fit the group's exchange area");
    using G = Grp<W>;
    using L = Layout<R, W>;
    using F = ExtFitLayout<Q>;
    extern __shared__ __attribute__((aligned(16))) unsigned char extfit_smem[];
    G grp = G::make(W > 1 ? extfit_smem : nullptr);
    const int lane = grp.gl;
    const int64_t b = extfit_problem_of<T>(a, blockIdx.x);
    if (b < 0) return;
    const int m = a.m;
    const bool vec = a.vec != 0;
    T *st = reinterpret_cast<T *>(a.state);
    int32_t *si = extfit_ints<T, Q>(a.state, a.B);

    int want = EXTFIT_WANT_BASIS | EXTFIT_WANT_DERIVS;
    if (!a.init) {
        if (uni(si[F::TERM * a.B + b]) != 0) return; // finished in an earlier step: nothing is read, nothing changes
        want = uni(si[F::WANT * a.B + b]);
    }
    const bool with_d = (want & EXTFIT_WANT_DERIVS) != 0 && a.dphi != nullptr; // (uniform)

    T C[NC][R];
    {
        const T *ph = a.phi + b * (int64_t)N * m;
#pragma unroll
        for (int j = 0; j < N; ++j) load_rows<T, R, W>(ph + (int64_t)a.perm[j] * m, m, lane, vec, C[j]);
        load_rows<T, R, W>(a.yw + b * (int64_t)m, m, lane, vec, C[N]);
        const T *dp = a.dphi + b * (int64_t)a.np * m;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (with_d && p < a.np) {
                load_rows<T, R, W>(dp + (int64_t)p * m, m, lane, vec, C[N + 1 + p]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) C[N + 1 + p][r] = T(0);
            }
        }
        if (a.w) { // `&self.weights * ...` (src/util/weights.rs:82-99)
            T wt[R];
            load_rows<T, R, W>(a.w + b * a.w_stride, m, lane, vec, wt);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (j == N) continue; // y_w was weighted when the handle was made
#pragma unroll
                for (int r = 0; r < R; ++r) C[j][r] *= wt[r];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- set_params at the trial point: src/solvers/levmar/mod.rs:42-73 ----
    T g[N], Rm[N][N], qty[N], c[N], e[N];
    house_qr<T, R, N, NC, 0, true, G>(C, g, Rm, qty, grp);
    bool truncated;
    solve_coeffs<T, N>(Rm, qty, a.eps, c, e, truncated);
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
        st[F::C_FN * a.B + b] = usqrt(fn2);
#pragma unroll
        for (int k = 0; k < N; ++k) st[(F::C_C + a.perm[k]) * a.B + b] = c[k];
        si[F::C_OK * a.B + b] = ok ? 1 : 0;
        si[F::C_HASJ * a.B + b] = (with_d && ok) ? 1 : 0;
    }
    if (!with_d || !ok) return;

    // ---- jacobian at the same point: Kaufman columns in Q-coordinates (rows < N: the P_perp), MINPACK qrfac with
    // Q_J^T r alongside (:101-201).  Whether the LM driver wants it is decided by the LM kernel. ----
    residual_qcoords<T, R, N>(C[N], e, grp);
    T Zs[Q][R];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
#pragma unroll
        for (int r = 0; r < R; ++r) Zs[k][r] = T(0);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (p < a.np && a.pp[p] == k) { // (uniform)
                const T cj = -dyn_get<N>(c, a.pb[p]);
#pragma unroll
                for (int r = 0; r < R; ++r) Zs[k][r] = tfma(cj, C[N + 1 + p][r], Zs[k][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < L::VW && r < R; ++r)
            if (L::row_of(r, lane) < N) Zs[k][r] = T(0);
    }
    T Rj[Q][Q], acn[Q], qtf[Q];
    int ipv[Q];
    jac_qrfac<T, R, Q, N>(Zs, C[N], Rj, acn, ipv, qtf, grp);
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

// ---- launch 2: the LM driver of every problem, one lane each ----
template <typename T, int Q> __global__ void __launch_bounds__(64) ext_fit_lm_kernel(const ExtFitArgs<T> a) {
    using F = ExtFitLayout<Q>;
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b == 0) a.nactive[(a.step + 1) & 1] = 0;
    if (b >= a.B) return;
    const int64_t B = a.B;
    T *st = reinterpret_cast<T *>(a.state);
    int32_t *si = extfit_ints<T, Q>(a.state, B);
    const int n = a.n;

    LmVars<T, 1, Q> s;
    int phase = EXTFIT_PH_EVAL;
    if (a.init) {
        lm_init<T, 1, Q>(s, a.alpha0 + b * Q);
#pragma unroll
        for (int k = 0; k < VP_MAX_BASIS; ++k) st[(F::CBEST + k) * B + b] = T(0);
        if (a.S > 1) { // (several right-hand sides: the coefficients live in C_best itself; a fit that never accepts a point reports zeros)
            const int64_t cn = (int64_t)a.S * n;
            for (int64_t i = 0; i < cn; ++i) a.C_best[b * cn + i] = T(0);
        }
    } else {
        if (si[F::TERM * B + b] != 0) {
            // finished in an earlier step: the caller's arrays of THIS step (they may be other buffers than the last step's)
            // still get what the header promises -- want = 0 and the final parameters
            a.want[b] = 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) a.alpha_trial[b * Q + k] = st[(F::X + k) * B + b];
            return;
        }
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            s.x[k] = st[(F::X + k) * B + b];
            s.xt[k] = st[(F::XT + k) * B + b];
            s.diag[k] = st[(F::DIAG + k) * B + b];
            s.qtf[k] = st[(F::QTF + k) * B + b];
            s.acnorm[k] = st[(F::ACN + k) * B + b];
            s.ipvt[k] = si[(F::IPVT + k) * B + b];
#pragma unroll
            for (int j = 0; j < Q; ++j) s.Rj[k][j] = st[(F::RJ + k * Q + j) * B + b];
        }
        s.fnorm = st[(F::SC + 0) * B + b];
        s.delta = st[(F::SC + 1) * B + b];
        s.par = st[(F::SC + 2) * B + b];
        s.xnorm = st[(F::SC + 3) * B + b];
        s.gnorm = st[(F::SC + 4) * B + b];
        s.pnorm = st[(F::SC + 5) * B + b];
        s.prered = st[(F::SC + 6) * B + b];
        s.dirder = st[(F::SC + 7) * B + b];
        s.objective = st[(F::SC + 8) * B + b];
        s.first = si[F::FIRST * B + b];
        s.first_tr = si[F::FIRST_TR * B + b];
        s.first_update = si[F::FIRST_UP * B + b];
        s.nfev = si[F::NFEV * B + b];
        s.term = 0;
        s.status = si[F::STATUS * B + b];
        s.accepted = 0;
        phase = si[F::PHASE * B + b];
    }
    const T fnorm1 = st[F::C_FN * B + b];
    const bool ok = si[F::C_OK * B + b] != 0;
    const bool has_jac = si[F::C_HASJ * B + b] != 0;

    bool need_jac;
    if (phase == EXTFIT_PH_JAC) {
        need_jac = true; // the columns were those of the point accepted in the previous step: no new evaluation
        if (!ok) {       // (the caller's columns at that point are not the ones it showed before: residuals() == None)
            s.term = VP_TERM_USER;
            s.status = VP_ST_NONFINITE;
        }
    } else {
        need_jac = lm_after_eval<T, 1, Q, false>(s, a.o, fnorm1, ok, (long)a.m * (long)(a.S > 1 ? a.S : 1));
        if (s.accepted) {
            if (a.S > 1) {
                const int64_t cn = (int64_t)a.S * n;
                for (int64_t i = 0; i < cn; ++i) a.C_best[b * cn + i] = a.C_trial[b * cn + i];
            } else {
                for (int k = 0; k < n; ++k) st[(F::CBEST + k) * B + b] = st[(F::C_C + k) * B + b];
            }
        }
    }
    bool deferred = false;
    const bool zero_jac = a.np == 0; // a model without derivative columns: eval_partial_deriv is zero for every parameter
    if (s.term == 0 && need_jac && !has_jac && !zero_jac) {
        if (phase == EXTFIT_PH_JAC) {
            // the derivative columns were asked for at this point in the previous step and did not come (dPhi == NULL):
            // the model's eval_partial_deriv failed -> jacobian() == None -> the driver ends with `User`
            // (src/solvers/levmar/mod.rs:101-104).  Deferring again would never end: nfev does not advance here.
            s.term = VP_TERM_USER;
            s.status = VP_ST_NONFINITE;
        } else {
            deferred = true; // accepted without derivative columns at hand (second protocol): ask for them at this very point
        }
    } else if (s.term == 0) {
        if (need_jac && zero_jac) {
            // J = 0: R_J = 0, Q_J^T r = 0, column norms 0 -> the scaled gradient is 0 <= gtol: `Orthogonal`, as in the reference
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                s.acnorm[k] = T(0);
                s.qtf[k] = T(0);
                s.ipvt[k] = k;
#pragma unroll
                for (int j = 0; j < Q; ++j) s.Rj[k][j] = T(0);
            }
        } else if (need_jac) { // the candidate factor becomes the factor of the current point
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                s.acnorm[k] = st[(F::C_ACN + k) * B + b];
                s.qtf[k] = st[(F::C_QTF + k) * B + b];
                s.ipvt[k] = si[(F::C_IPVT + k) * B + b];
#pragma unroll
                for (int j = 0; j < Q; ++j) s.Rj[k][j] = st[(F::C_RJ + k * Q + j) * B + b];
            }
        }
        lm_next_step<T, 1, Q, false>(s, a.o, need_jac);
    }

    int want_next;
    if (s.term != 0) {
        want_next = 0;
        phase = EXTFIT_PH_EVAL;
    } else if (deferred) {
        want_next = EXTFIT_WANT_BASIS | EXTFIT_WANT_DERIVS;
        phase = EXTFIT_PH_JAC;
    } else {
        want_next = a.lazy ? EXTFIT_WANT_BASIS : (EXTFIT_WANT_BASIS | EXTFIT_WANT_DERIVS);
        phase = EXTFIT_PH_EVAL;
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        st[(F::X + k) * B + b] = s.x[k];
        st[(F::XT + k) * B + b] = s.xt[k];
        st[(F::DIAG + k) * B + b] = s.diag[k];
        st[(F::QTF + k) * B + b] = s.qtf[k];
        st[(F::ACN + k) * B + b] = s.acnorm[k];
        si[(F::IPVT + k) * B + b] = s.ipvt[k];
#pragma unroll
        for (int j = 0; j < Q; ++j) st[(F::RJ + k * Q + j) * B + b] = s.Rj[k][j];
        // a finished problem reports its final parameters; a deferred one repeats the accepted point (x == xt)
        a.alpha_trial[b * Q + k] = (s.term != 0 || deferred) ? s.x[k] : s.xt[k];
        a.alpha_best[b * Q + k] = s.x[k];
    }
    st[(F::SC + 0) * B + b] = s.fnorm;
    st[(F::SC + 1) * B + b] = s.delta;
    st[(F::SC + 2) * B + b] = s.par;
    st[(F::SC + 3) * B + b] = s.xnorm;
    st[(F::SC + 4) * B + b] = s.gnorm;
    st[(F::SC + 5) * B + b] = s.pnorm;
    st[(F::SC + 6) * B + b] = s.prered;
    st[(F::SC + 7) * B + b] = s.dirder;
    st[(F::SC + 8) * B + b] = s.objective;
    si[F::FIRST * B + b] = s.first;
    si[F::FIRST_TR * B + b] = s.first_tr;
    si[F::FIRST_UP * B + b] = s.first_update;
    si[F::NFEV * B + b] = s.nfev;
    si[F::TERM * B + b] = s.term;
    si[F::STATUS * B + b] = s.status;
    si[F::WANT * B + b] = want_next;
    si[F::PHASE * B + b] = phase;
    a.want[b] = want_next;
    if (a.S <= 1)
        for (int k = 0; k < n; ++k) a.C_best[b * n + k] = st[(F::CBEST + k) * B + b];
    vp_report rep;
    rep.termination = s.term;
    rep.n_evals = s.nfev;
    rep.objective = (double)s.objective;
    a.report[b] = rep;
    a.cost[b] = (double)s.objective;
    a.status[b] = s.status;
    // the number of problems still running: one atomic per wavefront (lanes that returned early count as finished)
    // ... and their indices, compacted (the next step's evaluation launch covers only these; order = arrival of the wavefronts,
    // which nothing depends on)
    const unsigned long long running = __builtin_amdgcn_ballot_w64(s.term == 0);
    if (running != 0) {
        const int ln = (int)(threadIdx.x & 63u);
        int base = 0;
        if (ln == __builtin_ctzll(running)) base = atomicAdd(&a.nactive[a.step & 1], __builtin_popcountll(running));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(running));
        if (s.term == 0 && a.active_out) a.active_out[base + __builtin_popcountll(running & ((1ull << ln) - 1ull))] = (int32_t)b;
    }
}

template <typename T, int N, int P, int Q, int R, int W> int launch_fit_eval(const ExtFitArgs<T> &a, hipStream_t stream) {
    hipLaunchKernelGGL((ext_fit_eval_kernel<T, N, P, Q, R, W>), dim3((unsigned)a.grid_problems), dim3(64 * W), group_xch_bytes<W>(), stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}
template <typename T, int Q> int launch_fit_lm(const ExtFitArgs<T> &a, hipStream_t stream) {
    hipLaunchKernelGGL((ext_fit_lm_kernel<T, Q>), dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

// one row of the table of compiled evaluation shapes (vp_inst_extfit*.hip) ...
template <typename T> struct ExtFitEntry {
    int N, P, Q, R, W; // covers m <= 64 R W
    int (*launch)(const ExtFitArgs<T> &, hipStream_t);
};
// ... and of the LM kernels (one per parameter count)
template <typename T> struct ExtFitLmEntry {
    int Q;
    size_t rec_bytes; // state bytes per problem
    int (*launch)(const ExtFitArgs<T> &, hipStream_t);
};
template <typename T> std::vector<ExtFitEntry<T>> &extfit_table() {
    static std::vector<ExtFitEntry<T>> t;
    return t;
}
template <typename T> std::vector<ExtFitLmEntry<T>> &extfit_lm_table() {
    static std::vector<ExtFitLmEntry<T>> t;
    return t;
}
template <typename T> struct ExtFitRegistrar {
    explicit ExtFitRegistrar(const ExtFitEntry<T> &e) { extfit_table<T>().push_back(e); }
    explicit ExtFitRegistrar(const ExtFitLmEntry<T> &e) { extfit_lm_table<T>().push_back(e); }
};

// the evaluation kernel that covers (n, np pairs, q, m): exact n and q, the smallest capacity 64 R W, then the fewest spare
// pairs, then the FIRST registered of equal ones (the instantiation files list the preferred split R x W first)
template <typename T> const ExtFitEntry<T> *find_extfit(int n, int np, int q, int64_t m) {
    const ExtFitEntry<T> *best = nullptr;
    for (const ExtFitEntry<T> &e : extfit_table<T>()) {
        if (e.N != n || e.Q != q || e.P < np || 64 * (int64_t)e.R * e.W < m) continue;
        if (!best || e.R * e.W < best->R * best->W || (e.R * e.W == best->R * best->W && e.P < best->P)) best = &e;
    }
    return best;
}
template <typename T> const ExtFitLmEntry<T> *find_extfit_lm(int q) {
    for (const ExtFitLmEntry<T> &e : extfit_lm_table<T>())
        if (e.Q == q) return &e;
    return nullptr;
}

} // namespace ext
} // namespace vp

#define VP_REGISTER_EXTFIT_W(T, NN, PP, QQ, RR, WW)                                                                    \
    static ::vp::ext::ExtFitRegistrar<T> VP_EXT_CAT(vp_extfit_reg_, __COUNTER__)(                                     \
        ::vp::ext::ExtFitEntry<T>{NN, PP, QQ, RR, WW, &::vp::ext::launch_fit_eval<T, NN, PP, QQ, RR, WW>});
#define VP_REGISTER_EXTFIT(T, NN, PP, QQ, RR) VP_REGISTER_EXTFIT_W(T, NN, PP, QQ, RR, 1)
#define VP_REGISTER_EXTFIT_LM(T, QQ)                                                                                   \
    static ::vp::ext::ExtFitRegistrar<T> VP_EXT_CAT(vp_extfit_lm_reg_, __COUNTER__)(::vp::ext::ExtFitLmEntry<T>{       \
        QQ, ::vp::ext::ExtFitLayout<QQ>::bytes(sizeof(T)), &::vp::ext::launch_fit_lm<T, QQ>});
