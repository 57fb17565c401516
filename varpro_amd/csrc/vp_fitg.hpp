// vp_fitg.hpp -- Levenberg-Marquardt fit of fp32 problems on the fp64 GRAM matrix of [Phi | y | dPhi]
// (BASELINE.json configs[4]: five exponentials + offset, m = 4096, fp32 -- "wider Jacobian / Phi^T Phi path").
//
// == LevMarSolver::fit -> LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:238-254) for a batch of fp32
// problems, with the linear algebra of one evaluation (src/solvers/levmar/mod.rs:42-73, 101-201: thin decomposition of
// Phi_w, coefficients, projected residual, Kaufman Jacobian) restated on the normal equations IN DOUBLE.
//
// MOMENT form of the Gram matrix.  With e_k = w exp(-t/tau_k) (w = the row weight, src/util/weights.rs:82-99; 1 for
// Weights::Unit) and u_k = t e_k the weighted derivative column is d_k = u_k / tau_k^2
// (shared_test_code/src/lib.rs:109-114), so every inner product of X = [e_1..e_NE | w | y_w | d_1..d_NE] is one of
//     A0[i,k] = sum e_i e_k    A1[i,k] = sum e_i u_k (symmetric: t w^2 e_i e_k)    A2[i,k] = sum u_i u_k
//     B0[k] = sum e_k y        B1[k] = sum u_k y       YY = sum y^2
//     S0[k] = sum w e_k        S1[k] = sum w u_k       SY = sum w y       SW = sum w^2 (= m for unit weights)
// times powers of 1/tau_k^2 that are applied AFTER the pass: 3 NE(NE+1)/2 + 4 NE + 2 = 67 accumulators for NE = 5
// where the plain Gram of the 11 columns needs 77, and no per-row scaling of the derivative columns.
// ONE pass over the m rows accumulates them per lane (fp64 FMAs: the product of two fp32 values is exact in fp64), ONE
// packed wave reduction delivers them -- no column is ever resident, no per-reflector reduction round.  Then, per slot
// on one lane (gram_phase):
//     A = Phi^T Phi = L L^T,  z = L^-1 Phi^T y,  c = L^-T z,  ||r||^2 = y^T y - z^T z,
//     W = L^-1 Phi^T D,  D^T P_perp D = D^T D - W^T W,  D^T r = D^T y - W^T z,
//     J^T J = diag(c) (D^T P_perp D) diag(c),  J^T r = -c_k (D^T r)_k   (Kaufman, pair p = (basis p, parameter p)),
//     pivoted Cholesky of J^T J -> (R_J, acnorm, ipvt, qtf) exactly as the multiple-right-hand-side path (gram_to_qr).
//
// Why this is legitimate for fp32 and only for fp32: the Householder path in fp32 loses kappa(Phi)*eps32 (6e-8) of the
// coefficients -- at cfg4's kappa ~ 1e3..1e6 that is up to 6 %, and 15 % of the fits end non-finite; the Gram matrix in
// fp64 loses kappa^2*eps64.  For fp64 data the same trick would square the conditioning with nothing in reserve, so fp64
// handles never come here.  Rank-deficient Phi at a trial point (two decay times collide, a column degenerates into the
// constant): columns whose Cholesky pivot vanishes are dropped, the counterpart of the reference's truncated SVD.
//
// Execution (round 3): a workgroup of 8 waves owns a POOL of NS = 32 problem slots.  Waves 1..7 (stream waves) do nothing
// but moment passes: each claims the next slot that has a trial point, streams the rows of its problem ONCE (y_w -- and the
// grid / weights where they are not implied -- re-read from HBM / L2, 16 B per lane and stream, coalesced, the next chunk
// prefetched) and leaves the moments in LDS.  Wave 0 (the scalar wave) does nothing but the lane-serial part, lane s <->
// slot s, for every slot whose moments are ready: gram_phase (moments -> c, ||r||, J^T J, J^T r) and the LM bookkeeping
// (slot_scalar_phase<double>), all in fp64, and the refill of finished slots from the device-side queue.  Slot states live
// in LDS, claims are LDS compare-and-swaps, there is no barrier in the loop: the ~33 k issue cycles of one bookkeeping
// pass serve up to NS evaluations instead of one wave's own 2-3, and no stream wave ever waits for them.  Results do not
// depend on which wave served a slot, nor on what ran beside it (tested bit for bit under a permutation of the batch).
// Round-3 history at configs[4] (8192 fits): round 2's 4-wave groups 5.7 ms -> independent waves on the moment form 4.0 ms
// -> role-specialised waves 3.4 ms; measured without a gain: serving the oldest fit first, raised priority of the scalar
// wave, 4 / 5 / 6 waves per workgroup, 16 / 24 / 48 slots.
#pragma once
#include "vp_fit2.hpp"
#include "vp_lm_core.hpp"

#ifndef VP_FITG_IDLE_PARTNER
#define VP_FITG_IDLE_PARTNER 0
#endif
#ifndef VP_FITG_SCALAR_WAVES
#define VP_FITG_SCALAR_WAVES 1 // bookkeeping waves per workgroup (each owns NS / this many slots)
#endif
#ifndef VP_FITG_CHOL_LMPAR
#define VP_FITG_CHOL_LMPAR 1   // trust-region sub-problem on the Cholesky factor of J^T J + par D^2 (lmpar_chol) instead of qrsolv
#endif
#ifndef VP_FITG_RESOLUTION_GUARD
#define VP_FITG_RESOLUTION_GUARD 1 // a trial point whose ||r||^2 cancelled to <= 0 is rejected, not read as a zero residual
#endif
#ifndef VP_FITG_PIVOT_NOISE
#define VP_FITG_PIVOT_NOISE 1.0e-10
#endif
#ifndef VP_FITG_RESOLUTION
// (experiment, VP_FITG_RESOLUTION_GUARD 2 / 3: ||r||^2 floored at / rejected below this fraction of the error scale
// sum_i z_i^2 A_ii / d_i.  On configs[4] every value from 1e-12 to 5e-10 removes the fits that end on a wrong objective
// and shortens the launch, but on exact data the error scale is ~||y||^2 whatever the conditioning -- the last column's
// term is ||phi_l c_l||^2 -- so the fit stops at 1e-11 ||y||^2 instead of 1e-15: not adopted, the pivot threshold is.)
#define VP_FITG_RESOLUTION 1.6e-11
#endif
#ifndef VP_FITG_CLOSED
#define VP_FITG_CLOSED 1       // uniform grid + unit weights: the y-independent moments in closed form
#endif
#ifndef VP_FITG_HORNER
#define VP_FITG_HORNER 1       // ... and the y-dependent ones by Horner's rule in rho_k = e^{-dt / tau_k} (gram_pass)
#endif
#ifndef VP_FITG_Y_ONCE
#define VP_FITG_Y_ONCE 1       // ... with sum y^2, sum y taken once per fit instead of once per pass
#endif
#ifndef VP_FITG_RING
#define VP_FITG_RING 8         // ... and this many 256-row chunks of y in flight per stream wave
#endif
#ifndef VP_FITG_UNI_EXP_LANES
#define VP_FITG_UNI_EXP_LANES 1 // the wave-uniform ratios of the recurrence: one exponential over 2 NE lanes + broadcasts
#endif
#ifndef VP_FITG_EXP_HALVES
#define VP_FITG_EXP_HALVES 1   // closed form: the exponentials of four doubling steps as two per lane over both half-waves
#endif
#ifndef VP_FITG_EXP_BATCH
#define VP_FITG_EXP_BATCH 4    // steps of the closed form's doubling recurrence whose exponentials are evaluated together
#endif

namespace vp {

template <int NE> struct GramIdx {
    static constexpr int NT = NE * (NE + 1) / 2;
    __host__ __device__ static constexpr int tri(int i, int k) { // i <= k, row-major upper triangle
        return i * NE - i * (i - 1) / 2 + (k - i);
    }
    __host__ __device__ static constexpr int sym(int i, int k) { return i <= k ? tri(i, k) : tri(k, i); }
    __host__ __device__ static constexpr int A0(int i, int k) { return sym(i, k); }
    __host__ __device__ static constexpr int A1(int i, int k) { return NT + sym(i, k); }
    __host__ __device__ static constexpr int A2(int i, int k) { return 2 * NT + sym(i, k); }
    __host__ __device__ static constexpr int B0(int k) { return 3 * NT + k; }
    __host__ __device__ static constexpr int B1(int k) { return 3 * NT + NE + k; }
    static constexpr int YY = 3 * NT + 2 * NE;
    __host__ __device__ static constexpr int S0(int k) { return YY + 1 + k; }
    __host__ __device__ static constexpr int S1(int k) { return YY + 1 + NE + k; }
    static constexpr int SY = YY + 1 + 2 * NE;
    static constexpr int SW = SY + 1; // accumulated for weighted problems only
    static constexpr int NV = SW + 1; // LDS stride of one slot's moments
};

// Packed wave reduction of V values whose totals are STORED to LDS by the lanes that end up holding them (dst[v]) --
// the broadcast of wave_allreduce would need 2V SGPRs.  Same packing as wave_allreduce (vp_device.hpp).
template <int V, typename T> __device__ __forceinline__ void wave_reduce_store(T (&x)[V], VP_LDS T *dst) {
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
    // value v (mod 16) sits in the lanes with bit5 = v&1, bit4 = (v>>1)&1, bit0 = (v>>2)&1, bit1 = (v>>3)&1
    const int L = lane_id();
    const int v = ((L >> 5) & 1) | (((L >> 4) & 1) << 1) | ((L & 1) << 2) | (((L >> 1) & 1) << 3);
    if ((L & 0xC) == 0) {
#pragma unroll
        for (int i = 0; i < V4; ++i)
            if (16 * i + v < V) dst[16 * i + v] = y4[i];
    }
}

__device__ __forceinline__ double uni_d(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

struct FitgArgs {
    const float *t;   // [m] or [B][m]
    const float *w;   // null (unit weights), [m] or [B][m]
    const float *yw;  // [B][m] weighted data
    float *alpha;     // [B][q] in: guesses, out: parameters
    float *C_out;     // [B][n] or null
    double *cost_out;
    int32_t *status;
    vp_report *report;
    double *trace;
    double *dbg;      // null, or [B][1 + n + q + q*q]: evaluate the Gram formulation ONCE at alpha and write
                      // {1/2||r||^2, c, J^T r, J^T J} per problem instead of fitting (vp_debug_gram_evaluate)
    int *queue;
    int64_t B;
    int64_t t_stride, w_stride;
    int m;
    int trace_rows, scale_diag, patience;
    int gs_used;      // slots per wave taken by the static first assignment (<= GS)
    double eps, ftol, xtol, gtol, stepbound;
};

// Lane s: moments of slot s -> results of the evaluation in the slot's record (what the vector phase of fit2_kernel
// posts).  gram: [GS][GI::NV].  dbg != null: additionally write {1/2||r||^2, c, J^T r, J^T J} of the slot's problem.
template <int NE, int GS, bool WEIGHTED>
__device__ __forceinline__ void gram_phase(VP_LDS SlotRec<double, NE + 1, NE> *recs, VP_LDS const double *gram,
                                        VP_LDS const SlotConsts<double, float> *k, double *dbg, const bool act = true) {
    constexpr int N = NE + 1, Q = NE;
    using GI = GramIdx<NE>;
    const int lane = lane_id();
    if (!(act && lane < GS && recs[lane].prob >= 0)) return;
    VP_LDS SlotRec<double, N, Q> *rec = recs + lane;
    VP_LDS const double *g = gram + (size_t)lane * GI::NV;
    const double eps = k->eps;
    double it2[NE]; // 1 / tau_k^2: the derivative columns are u_k / tau_k^2
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const double rt = frcp(rec->xt[i]);
        it2[i] = rt * rt;
    }
    // ---- A = Phi^T Phi (basis order e_0..e_{NE-1}, const) = L L^T ----
    // A column whose pivot d_i (its squared distance from the span of the columns before it) is <= max(eps^2,
    // VP_FITG_PIVOT_NOISE A_ii) is DROPPED (c_i = 0, the projector is that of the remaining columns): the counterpart of the
    // reference's truncated SVD (singular values <= eps, src/solvers/levmar/mod.rs:52-54) at trial points where two decay
    // times collide or a column degenerates into the constant -- the step is then judged by its residual like any other
    // instead of ending the fit.  eps is the handle's svd_epsilon (absolute, like the reference's); the relative term is
    // NOT a user parameter but the resolution of the method: the moments are consistent with each other to ~1e-13 of A_ii
    // (exponentials by recurrence over a chunk, closed-form sums beside accumulated ones), so a pivot of 1e-13 A_ii is noise
    // and one of 1e-10 A_ii is known to three digits.  Round 4 raised the threshold from the noise floor (1e-13) to 1e-10:
    // at 1e-13 a column that was still "kept" with a pivot of 1e-12 A_ii made ||r||^2 = y^T y - z^T z garbage, and 0.33 % of
    // configs[4]'s fits ended on such a point with a reported objective off by 1e-2 .. 0.7 of the true cost there (0.14 % by
    // more than 0.1); at 1e-10: 1 fit of 8 192 by 0.07, none above -- and fewer failed fits (2.56 -> 2.51 %), every test of
    // the Gram suite (m = 200 .. 4096, general grids, weights, exact data) unchanged.  tools/cfg4_resolution_probe.py.
    constexpr double VP_GRAM_NOISE = VP_FITG_PIVOT_NOISE;
    double Lm[N][N], iL[N]; // L (strict lower part) and the reciprocals of its diagonal (0 for a dropped column)
    double amp[N];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double aij;
            if (i == NE) aij = (j == NE) ? (WEIGHTED ? g[GI::SW] : (double)k->m) : g[GI::S0(j)];
            else aij = g[GI::A0(j, i)];
            double acc = aij;
#pragma unroll
            for (int p = 0; p < j; ++p) acc = tfma(-Lm[i][p], Lm[j][p], acc);
            if (i == j) {
                ok = ok && is_finite(acc);
                const bool keep = acc > tmax(eps * eps, VP_GRAM_NOISE * aij);
                iL[i] = keep ? frsqrt(acc) : 0.0; // (Newton-refined v_rsq: 1-2 ulp, against kappa^2 eps64 of the method)
                amp[i] = keep ? aij * (iL[i] * iL[i]) : 0.0; // A_ii / d_i: by how much the rounding of the moments is amplified in d_i
            } else {
                Lm[i][j] = acc * iL[j]; // (a dropped column j has no sub-diagonal entries)
            }
        }
    }
    // ---- z = L^-1 Phi^T y, c = L^-T z, ||r||^2 ----
    double z[N], c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double acc = (i == NE) ? g[GI::SY] : g[GI::B0(i)];
#pragma unroll
        for (int p = 0; p < i; ++p) acc = tfma(-Lm[i][p], z[p], acc);
        z[i] = acc * iL[i];
    }
    double fn2 = g[GI::YY];
    double fn2_err = 0.0; // estimate of the rounding error of fn2: sum_i z_i^2 (A_ii / d_i), in units of the moments' relative error
#pragma unroll
    for (int i = 0; i < N; ++i) {
        fn2 = tfma(-z[i], z[i], fn2);
        fn2_err = tfma(z[i] * z[i], amp[i], fn2_err);
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double acc = z[i];
#pragma unroll
        for (int p = i + 1; p < N; ++p) acc = tfma(-Lm[p][i], c[p], acc);
        c[i] = acc * iL[i];
        ok = ok && is_finite(c[i]);
    }
    ok = ok && is_finite(fn2);
    const int fl_in = rec->flags;
    const bool first = (fl_in & 1) != 0;
    // ||r||^2 = y^T y - z^T z is a difference of two numbers of the size of ||y||^2: where it cancels to <= 0 -- a value no
    // residual has -- the trial point carries no information.  Read as 0 it was the best point ever seen and ended the fit
    // `ResidualsZero` with objective 0 (0.4 % of configs[4]'s fits, on data with 1e-3 of noise).  Such a point is REJECTED
    // like any step that fails badly (the trust region shrinks by MINPACK's factor 10), never accepted; the first
    // evaluation keeps its value.  (fn2_err / modes 2, 3: the experiment described at VP_FITG_RESOLUTION.)
#if VP_FITG_RESOLUTION_GUARD == 3   // experiment: reject below VP_FITG_RESOLUTION x the error scale
    const bool lost = ok && !first && !(fn2 > VP_FITG_RESOLUTION * fn2_err);
    const double fnorm1 = lost ? 1.0e150 : usqrt(tmax(fn2, 0.0));
#elif VP_FITG_RESOLUTION_GUARD == 2 // experiment: ||r||^2 floored at VP_FITG_RESOLUTION x the error scale
    const double fnorm1 = usqrt(tmax(tmax(fn2, VP_FITG_RESOLUTION * fn2_err), 0.0));
#elif VP_FITG_RESOLUTION_GUARD == 1 // a cancelled ||r||^2 <= 0 is a failed step
    const bool lost = ok && !first && !(fn2 > 0.0);
    const double fnorm1 = lost ? 1.0e150 : usqrt(tmax(fn2, 0.0));
#else
    const double fnorm1 = usqrt(tmax(fn2, 0.0));
#endif
    const double fnorm = rec->fnorm, prered = rec->prered;
    double actred = 0.0, ratio = 0.0;
    bool good = false;
    if (!first) {
        const double q1 = fnorm1 * frcp(fnorm);
        actred = (fnorm1 * 0.1 < fnorm) ? 1.0 - q1 * q1 : -1.0;
        ratio = (prered == 0.0) ? 0.0 : actred * frcp(prered);
        good = ratio >= 1.0e-4;
    }
    const bool need_jac = ok && (first || good);
    rec->fnorm1 = fnorm1;
    rec->actred = actred;
    rec->ratio = ratio;
#pragma unroll
    for (int i = 0; i < N; ++i) rec->cnew[i] = c[i];
    int fl = fl_in & 7;
    if (ok) fl |= 8;
    double *dbo = dbg ? dbg + (size_t)rec->prob * (1 + N + Q + Q * Q) : nullptr;
    if (dbo) {
        dbo[0] = ok ? 0.5 * tmax(fn2, 0.0) : 0.0 / 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) dbo[1 + i] = c[i];
    }
    if (need_jac) {
        // ---- W = L^-1 Phi^T D;  G2 = D^T P_perp D;  v = D^T r ----
        double Wm[N][Q];
#pragma unroll
        for (int kk = 0; kk < Q; ++kk)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double acc = ((i == NE) ? g[GI::S1(kk)] : g[GI::A1(i, kk)]) * it2[kk];
#pragma unroll
                for (int p = 0; p < i; ++p) acc = tfma(-Lm[i][p], Wm[p][kk], acc);
                Wm[i][kk] = acc * iL[i];
            }
        double Aj[Q][Q], bv[Q];
#pragma unroll
        for (int kk = 0; kk < Q; ++kk) {
            double vk = g[GI::B1(kk)] * it2[kk];
#pragma unroll
            for (int i = 0; i < N; ++i) vk = tfma(-Wm[i][kk], z[i], vk);
            bv[kk] = -c[kk] * vk;
#pragma unroll
            for (int l = kk; l < Q; ++l) {
                double gkl = g[GI::A2(kk, l)] * (it2[kk] * it2[l]);
#pragma unroll
                for (int i = 0; i < N; ++i) gkl = tfma(-Wm[i][kk], Wm[i][l], gkl);
                const double a = c[kk] * c[l] * gkl;
                Aj[kk][l] = a;
                Aj[l][kk] = a;
            }
        }
        if (dbo) {
#pragma unroll
            for (int kk = 0; kk < Q; ++kk) {
                dbo[1 + N + kk] = bv[kk];
#pragma unroll
                for (int l = 0; l < Q; ++l) dbo[1 + N + Q + kk * Q + l] = Aj[kk][l];
            }
        }
        double Rd[Q][Q], acd[Q], qd[Q];
        int ipv[Q];
        gram_to_qr<double, Q>(Aj, bv, Rd, acd, ipv, qd);
        fl |= 16;
#pragma unroll
        for (int kk = 0; kk < Q; ++kk) {
            rec->acnorm[kk] = acd[kk];
            rec->qtf[kk] = qd[kk];
            rec->ipvt[kk] = ipv[kk];
#pragma unroll
            for (int j = 0; j < Q; ++j) rec->Rj[kk][j] = Rd[kk][j];
        }
    } else if (dbo) {
        for (int i = 0; i < Q + Q * Q; ++i) dbo[1 + N + i] = 0.0 / 0.0;
    }
    if (good) fl |= 32;
    rec->flags = fl;
    if (dbg) rec->term = VP_TERM_USER; // evaluate-only: the slot is done (nothing else is written for it)
}

// One chunk = 256 consecutive rows: lane l owns rows 256 ch + 4 l .. + 3 (16 B per lane and stream, 1 KiB per wave
// instruction).  UNIFORM: the grid is t_0 + i dt to rounding (grid_check_kernel) -- t is never read, exp(-t/tau) of
// a lane's rows follows from one anchor per lane by a recurrence (ratio per row, ratio per chunk).
template <bool UNIFORM, bool WEIGHTED> struct GramChunk {
    float4 y, t, w;
};

template <bool UNIFORM, bool WEIGHTED>
__device__ __forceinline__ void gram_load_chunk(GramChunk<UNIFORM, WEIGHTED> &c, const float *yp, const float *tp, const float *wp,
                                                const int row0, const int m, const bool vec) {
    auto ld4 = [&](const float *p) __attribute__((always_inline)) -> float4 {
        float4 v;
        if (vec && row0 + 3 < m) {
            v = *reinterpret_cast<const float4 *>(p + row0);
        } else {
            v.x = (row0 < m) ? p[row0] : 0.0f;
            v.y = (row0 + 1 < m) ? p[row0 + 1] : 0.0f;
            v.z = (row0 + 2 < m) ? p[row0 + 2] : 0.0f;
            v.w = (row0 + 3 < m) ? p[row0 + 3] : 0.0f;
        }
        return v;
    };
    c.y = ld4(yp);
    if constexpr (!UNIFORM) c.t = ld4(tp);
    if constexpr (WEIGHTED) {
        if (wp) {
            c.w = ld4(wp);
        } else { // unit weights with a ragged last row group: the row mask plays the weights
            c.w.x = (row0 < m) ? 1.0f : 0.0f;
            c.w.y = (row0 + 1 < m) ? 1.0f : 0.0f;
            c.w.z = (row0 + 2 < m) ? 1.0f : 0.0f;
            c.w.w = (row0 + 3 < m) ? 1.0f : 0.0f;
        }
    }
}

// The y stream alone, 16 B per lane, through a BUFFER resource over the problem's m floats: a row group past the end reads as
// zeros by the hardware's range check -- no branch around the load, so the compiler counts the loads in flight (vmcnt)
// instead of waiting for all of them at the first use.  VEC: the rows are 16-byte aligned (one dwordx4), else four dwords.
template <bool VEC> __device__ __forceinline__ float4 gram_buf_load_y(const __amdgpu_buffer_rsrc_t rs, const int row0) {
    float4 v;
    // (the values go through named integers: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
    if constexpr (VEC) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, row0 * 4, 0, 0);
        const unsigned int q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        v.x = __uint_as_float(q0);
        v.y = __uint_as_float(q1);
        v.z = __uint_as_float(q2);
        v.w = __uint_as_float(q3);
    } else {
        const unsigned int q0 = __builtin_amdgcn_raw_buffer_load_b32(rs, row0 * 4, 0, 0);
        const unsigned int q1 = __builtin_amdgcn_raw_buffer_load_b32(rs, row0 * 4 + 4, 0, 0);
        const unsigned int q2 = __builtin_amdgcn_raw_buffer_load_b32(rs, row0 * 4 + 8, 0, 0);
        const unsigned int q3 = __builtin_amdgcn_raw_buffer_load_b32(rs, row0 * 4 + 12, 0, 0);
        v.x = __uint_as_float(q0);
        v.y = __uint_as_float(q1);
        v.z = __uint_as_float(q2);
        v.w = __uint_as_float(q3);
    }
    return v;
}

// The MOMENT PASS of one slot by one wavefront: streams the rows of problem `prob` once and leaves the NVR moments in
// gram_out (LDS).  rec->xt holds the trial parameters; grid2 = {t_0, dt} of the slot's grid (UNIFORM).
template <int NE, bool UNIFORM, bool WEIGHTED, bool VEC>
__device__ __forceinline__ void gram_pass_v(const FitgArgs &a, VP_LDS const SlotRec<double, NE + 1, NE> *rec, VP_LDS const double *grid2,
                                          VP_LDS double *gram_out, const int prob, const int lane, const int m, const int ch0,
                                          const int nchunk, const bool vec, const bool own_closed, VP_LDS double *ymom) {
    // chunks [ch0, nchunk) of 256 rows (a whole pass: ch0 = 0, nchunk = all; a PART of a split pass: its chunk range;
    // own_closed: this pass / part also delivers the moments that do not depend on y -- see below)
    using GI = GramIdx<NE>;
    constexpr int NVR = WEIGHTED ? GI::NV : GI::NV - 1;
    constexpr bool CLOSED = UNIFORM && !WEIGHTED && (VP_FITG_CLOSED != 0);
    constexpr bool HORNER = CLOSED && (VP_FITG_HORNER != 0);
    // The y stream of the Horner form: VP_FITG_RING chunks in flight per wave, the first of them requested BEFORE the
    // exponentials of the preamble.  With one chunk of prefetch a pass was its 16 memory latencies (a chunk is 0.1 us of
    // arithmetic; y comes from L2 / HBM): 9.8 us per pass where its instructions issue in 2.5.
    constexpr int RING = HORNER ? VP_FITG_RING : 1;
    const float *yp = a.yw + (int64_t)prob * m;
    float4 ring[RING];
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)yp, 0, m * 4, 0x00020000);
    if constexpr (HORNER) {
        // (chunks at or past nchunk -- a part's range ends before the rows do -- are loaded and never used)
#pragma unroll
        for (int j = 0; j < RING; ++j) ring[j] = gram_buf_load_y<VEC>(yrs, (ch0 + j) * 256 + 4 * lane);
    }
    double rt[NE];
#pragma unroll
    for (int kx = 0; kx < NE; ++kx) rt[kx] = frcp(rec->xt[kx]);
    double t0 = 0.0, dt = 0.0;
    double fa[NE], q1[NE], qc[NE];
    if constexpr (UNIFORM) {
        t0 = uni_d(grid2[0]);
        dt = uni_d(grid2[1]);
        // anchor at the lane's first row, ratio per row, ratio per chunk: 3 NE exponentials per evaluation -- the 2 NE ratios
        // are wave-uniform: ONE exponential with a different argument per lane (lane k: per row, lane NE + k: per chunk of
        // column k) and 2 NE broadcasts instead of 2 NE exponentials that every lane repeats
        const double tl = tfma((double)(ch0 * 256 + 4 * lane), dt, t0);
#if VP_FITG_UNI_EXP_LANES
        double ax[NE + 1], ex[NE + 1];
#pragma unroll
        for (int kx = 0; kx < NE; ++kx) ax[kx] = -tl * rt[kx];
        {
            const int lk = lane < NE ? lane : (lane < 2 * NE ? lane - NE : 0);
            const double rtl = frcp(rec->xt[lk]);
            const double sc = lane < NE ? dt : 256.0 * dt;
            ax[NE] = -sc * rtl;
        }
        texp_n<NE + 1>(ax, ex);
#pragma unroll
        for (int kx = 0; kx < NE; ++kx) {
            fa[kx] = ex[kx];
            q1[kx] = readlane(ex[NE], kx);
            qc[kx] = readlane(ex[NE], NE + kx);
        }
#else
        double ax[3 * NE], ex[3 * NE];
#pragma unroll
        for (int kx = 0; kx < NE; ++kx) {
            ax[kx] = -tl * rt[kx];
            ax[NE + kx] = -dt * rt[kx];
            ax[2 * NE + kx] = -(256.0 * dt) * rt[kx];
        }
        texp_n<3 * NE>(ax, ex);
#pragma unroll
        for (int kx = 0; kx < NE; ++kx) {
            fa[kx] = ex[kx];
            q1[kx] = uni_d(ex[NE + kx]);
            qc[kx] = uni_d(ex[2 * NE + kx]);
        }
#endif
    }
    // ---- uniform grid, unit weights: 55 of the 67 moments do not depend on y and have a closed form ----
    // A0_ik = sum_r e^{-s t_r}, A1_ik = sum_r t_r e^{-s t_r}, A2_ik = sum_r t_r^2 e^{-s t_r} with s = 1/tau_i + 1/tau_k, and
    // S0_k, S1_k the same with s = 1/tau_k: on the lattice t_r = t_0 + r dt these are e^{-s t_0} times polynomials in
    // (t_0, dt) of H_j = sum_{r<m} r^j rho^r, rho = e^{-s dt}, j = 0, 1, 2.  The H_j follow from a DOUBLING recurrence
    // over the bits of m (H(2n) = H(n) + rho^n * [H(n) shifted by n]; every term positive: no cancellation; rho^n from
    // one exponential of n * (-s dt) per step, not by repeated squaring, which would double the relative error of rho^n
    // at every step).  One lane per value of s (NE (NE + 1) / 2 + NE = 20 lanes for five exponentials), ~13 exponentials
    // each: the row loop below then carries 12 accumulators instead of 67 (B0, B1, YY, SY) -- 28 instead of 86
    // instructions per row.
    // sum y^2 and sum y do not depend on the parameters: accumulated by the FIRST pass of a fit (never a split one) and kept
    // in ymom[2] for its later passes
    const bool y_once = HORNER && (VP_FITG_Y_ONCE != 0) && ymom != nullptr;
    const bool need_y = !y_once || uni((rec->flags & 1) != 0);
    constexpr int NPAIR = NE * (NE + 1) / 2;
    double G0 = 0.0, G1 = 0.0, G2 = 0.0;
    if constexpr (CLOSED) {
        // The lanes of the upper half-wave mirror the lower one (ls = lane & 31): the rho^n of FOUR steps of the recurrence
        // are two exponentials per lane -- lane ls takes steps 1 and 3 of the four, lane ls + 32 steps 2 and 4, two half-wave
        // swaps hand them over -- instead of four.  (The recurrences themselves matter on lanes < NPAIR + NE only.)
        const int ls = VP_FITG_EXP_HALVES ? (lane & 31) : lane;
        if (own_closed && ls < NPAIR + NE) {
            // lane -> (i, k): pairs in the row-major upper-triangle order of GramIdx::tri, then the NE single columns
            double sv = 0.0;
            {
                int idx = 0;
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int k2 = i; k2 < NE; ++k2) {
                        sv = (ls == idx) ? rt[i] + rt[k2] : sv;
                        ++idx;
                    }
#pragma unroll
                for (int k2 = 0; k2 < NE; ++k2) sv = (ls == NPAIR + k2) ? rt[k2] : sv;
            }
            const double x1 = -sv * dt; // log rho
            double H0 = 0.0, H1 = 0.0, H2 = 0.0;
            int n = 0;
            auto step = [&](const int bit, const double Pn) __attribute__((always_inline)) {
                const double dn = (double)n; // Pn = rho^n
                H2 = tfma(Pn, tfma(dn * dn, H0, tfma(2.0 * dn, H1, H2)), H2);
                H1 = tfma(Pn, tfma(dn, H0, H1), H1);
                H0 = tfma(Pn, H0, H0);
                n *= 2;
                if ((m >> bit) & 1) { // append the term r = n
                    const double Pa = texp(x1 * (double)n), da = (double)n;
                    H0 += Pa;
                    H1 = tfma(da, Pa, H1);
                    H2 = tfma(da * da, Pa, H2);
                    n += 1;
                }
            };
            // (n at the step of `bit` is the binary prefix m >> (bit + 1): the rho^n of several steps are independent of each
            // other and evaluated together -- interleaved polynomial chains instead of one after the other)
#if VP_FITG_EXP_HALVES
            const bool upper = lane >= 32;
            for (int b0 = 31 - __builtin_clz((unsigned)m); b0 >= 0; b0 -= 4) { // (m > 0; uniform over the wave)
                auto n_of = [&](const int bit) __attribute__((always_inline)) { return bit >= 0 ? (m >> (bit + 1)) : 0; };
                double ax[2], pn[2];
                ax[0] = x1 * (double)(upper ? n_of(b0 - 1) : n_of(b0));
                ax[1] = x1 * (double)(upper ? n_of(b0 - 3) : n_of(b0 - 2));
                texp_n<2>(ax, pn);
                const auto s01 = lane_swap<0>(pn[0], pn[0]); // .a: the lower half's value on every lane, .b: the upper half's
                const auto s23 = lane_swap<0>(pn[1], pn[1]);
                step(b0, s01.a);
                if (b0 >= 1) step(b0 - 1, s01.b);
                if (b0 >= 2) step(b0 - 2, s23.a);
                if (b0 >= 3) step(b0 - 3, s23.b);
            }
#else
            for (int b0 = 31 - __builtin_clz((unsigned)m); b0 >= 0; b0 -= VP_FITG_EXP_BATCH) { // (m > 0; uniform over the wave)
                double ax[VP_FITG_EXP_BATCH], pn[VP_FITG_EXP_BATCH];
#pragma unroll
                for (int j = 0; j < VP_FITG_EXP_BATCH; ++j) ax[j] = x1 * (double)((b0 - j) >= 0 ? (m >> (b0 - j + 1)) : 0);
                texp_n<VP_FITG_EXP_BATCH>(ax, pn);
#pragma unroll
                for (int j = 0; j < VP_FITG_EXP_BATCH; ++j)
                    if (b0 - j >= 0) step(b0 - j, pn[j]);
            }
#endif
            const double e0 = texp(-sv * t0);
            G0 = e0 * H0;
            G1 = e0 * tfma(dt, H1, t0 * H0);
            G2 = e0 * tfma(dt * dt, H2, tfma(2.0 * t0 * dt, H1, t0 * t0 * H0));
        }
    }
    const float *tp = a.t + (int64_t)prob * a.t_stride;
    const float *wp = a.w ? a.w + (int64_t)prob * a.w_stride : nullptr;
    double acc[NVR];
#pragma unroll
    for (int i = 0; i < NVR; ++i) acc[i] = 0.0;
    if constexpr (HORNER) {
        // B0_k = sum_r e_k(t_r) y_r and B1_k = sum_r e_k(t_r) (t_r y_r) on the lattice are POLYNOMIALS in rho_k = e^{-dt/tau_k}:
        // a lane's four rows of the chunk by Horner (3 FMAs per moment and exponential), times the lane's anchor
        // e_k(t_row0) -- 9 instructions per exponential and row group where carrying e_k and u_k = t e_k row by row
        // took 16.  (Rows past the end were loaded as zeros.)
        // (two instances of the loop: a predicated block inside it would still issue its instructions when sum y^2 and
        // sum y are not wanted)
        auto horner_rows = [&](auto with_y) __attribute__((always_inline)) {
        constexpr bool WITH_Y = decltype(with_y)::value;
#pragma nounroll
        for (int cb = ch0; cb < nchunk; cb += RING) {
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                const int ch = cb + j;
                if (ch >= nchunk) break; // (uniform)
                const float4 cur = ring[j];
                const int row0 = ch * 256 + 4 * lane;
                ring[j] = gram_buf_load_y<VEC>(yrs, row0 + RING * 256); // (past the part's range or the rows: unused / zeros)
                const double yd[4] = {(double)cur.x, (double)cur.y, (double)cur.z, (double)cur.w};
                double ty[4];
                const double tb = tfma((double)row0, dt, t0);
#pragma unroll
                for (int e = 0; e < 4; ++e) ty[e] = (e == 0 ? tb : tfma((double)e, dt, tb)) * yd[e];
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) {
                    const double r = q1[kx];
                    const double i0 = tfma(r, tfma(r, tfma(r, yd[3], yd[2]), yd[1]), yd[0]);
                    const double i1 = tfma(r, tfma(r, tfma(r, ty[3], ty[2]), ty[1]), ty[0]);
                    acc[GI::B0(kx)] = tfma(fa[kx], i0, acc[GI::B0(kx)]);
                    acc[GI::B1(kx)] = tfma(fa[kx], i1, acc[GI::B1(kx)]);
                    fa[kx] *= qc[kx];
                }
                if constexpr (WITH_Y) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[GI::YY] = tfma(yd[e], yd[e], acc[GI::YY]);
                        acc[GI::SY] += yd[e];
                    }
                }
            }
        }
        };
        if (need_y) horner_rows(std::true_type{});
        else horner_rows(std::false_type{});
    }
    GramChunk<UNIFORM, WEIGHTED> nxt;
    if constexpr (!HORNER) gram_load_chunk(nxt, yp, tp, wp, ch0 * 256 + 4 * lane, m, vec);
#pragma nounroll
    for (int ch = ch0; ch < (HORNER ? ch0 : nchunk); ++ch) {
        const GramChunk<UNIFORM, WEIGHTED> cur = nxt;
        const int row0 = ch * 256 + 4 * lane;
        if (ch + 1 < nchunk) gram_load_chunk(nxt, yp, tp, wp, row0 + 256, m, vec);
        const float yv[4] = {cur.y.x, cur.y.y, cur.y.z, cur.y.w};
        float tv[4] = {0.f, 0.f, 0.f, 0.f}, wv4[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (!UNIFORM) {
            tv[0] = cur.t.x, tv[1] = cur.t.y, tv[2] = cur.t.z, tv[3] = cur.t.w;
        }
        if constexpr (WEIGHTED) {
            wv4[0] = cur.w.x, wv4[1] = cur.w.y, wv4[2] = cur.w.z, wv4[3] = cur.w.w;
        }
        // unit weights: m % 4 == 0 (host dispatch), so a lane's row group is valid or padding as a whole
        const bool gvalid = row0 < m;
        double f[NE];
        if constexpr (UNIFORM) {
#pragma unroll
            for (int kx = 0; kx < NE; ++kx) f[kx] = (WEIGHTED || gvalid) ? fa[kx] : 0.0;
        }
        const double tb = UNIFORM ? tfma((double)row0, dt, t0) : 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double td = UNIFORM ? tfma((double)e, dt, tb) : (double)tv[e];
            const double yd = (double)yv[e];
            double eh[NE], uh[NE];
            if constexpr (UNIFORM) {
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) {
                    eh[kx] = f[kx];
                    if (e < 3) f[kx] *= q1[kx];
                }
            } else { // general grid: one fp64 exponential per element
                double ax[NE];
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) ax[kx] = -td * rt[kx];
                texp_n<NE>(ax, eh);
                if constexpr (!WEIGHTED) {
#pragma unroll
                    for (int kx = 0; kx < NE; ++kx) eh[kx] = gvalid ? eh[kx] : 0.0;
                }
            }
            double wd = 1.0;
            if constexpr (WEIGHTED) {
                wd = (double)wv4[e];
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) eh[kx] *= wd; // (padding rows: w = 0)
            }
#pragma unroll
            for (int kx = 0; kx < NE; ++kx) uh[kx] = td * eh[kx];
            if constexpr (!CLOSED) {
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int k2 = i; k2 < NE; ++k2) {
                        acc[GI::A0(i, k2)] = tfma(eh[i], eh[k2], acc[GI::A0(i, k2)]);
                        acc[GI::A1(i, k2)] = tfma(eh[i], uh[k2], acc[GI::A1(i, k2)]);
                        acc[GI::A2(i, k2)] = tfma(uh[i], uh[k2], acc[GI::A2(i, k2)]);
                    }
            }
#pragma unroll
            for (int kx = 0; kx < NE; ++kx) {
                acc[GI::B0(kx)] = tfma(eh[kx], yd, acc[GI::B0(kx)]);
                acc[GI::B1(kx)] = tfma(uh[kx], yd, acc[GI::B1(kx)]);
            }
            acc[GI::YY] = tfma(yd, yd, acc[GI::YY]);
            if constexpr (WEIGHTED) {
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) {
                    acc[GI::S0(kx)] = tfma(wd, eh[kx], acc[GI::S0(kx)]);
                    acc[GI::S1(kx)] = tfma(wd, uh[kx], acc[GI::S1(kx)]);
                }
                acc[GI::SY] = tfma(wd, yd, acc[GI::SY]);
                acc[GI::SW] = tfma(wd, wd, acc[GI::SW]);
            } else {
                if constexpr (!CLOSED) {
#pragma unroll
                    for (int kx = 0; kx < NE; ++kx) {
                        acc[GI::S0(kx)] += eh[kx];
                        acc[GI::S1(kx)] += uh[kx];
                    }
                }
                acc[GI::SY] += yd;
            }
        }
        if constexpr (UNIFORM) {
#pragma unroll
            for (int kx = 0; kx < NE; ++kx) fa[kx] *= qc[kx];
        }
    }
    if constexpr (CLOSED) {
        // the 12 y-dependent moments: B0, B1, YY are contiguous in the layout, SY travels as a 12th value (it lands on
        // S0(0), is moved to its place, and S0(0) is then written with the rest of the closed-form moments)
        constexpr int NY = 2 * NE + 2;
        double ya[NY];
#pragma unroll
        for (int i = 0; i < 2 * NE + 1; ++i) ya[i] = acc[GI::B0(0) + i];
        ya[NY - 1] = acc[GI::SY];
        static_assert(GI::B1(0) == GI::B0(0) + NE && GI::YY == GI::B0(0) + 2 * NE && GI::S0(0) == GI::YY + 1, "layout");
        if (need_y) {
            wave_reduce_store<NY>(ya, gram_out + GI::B0(0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
                const double sy = gram_out[GI::S0(0)];
                gram_out[GI::SY] = sy;
                if (y_once) {
                    ymom[0] = gram_out[GI::YY];
                    ymom[1] = sy;
                }
            }
        } else {
            double yb[2 * NE];
#pragma unroll
            for (int i = 0; i < 2 * NE; ++i) yb[i] = ya[i];
            wave_reduce_store<2 * NE>(yb, gram_out + GI::B0(0));
            if (lane == 0) { // (the parts of a split pass are summed: one of them carries the values)
                gram_out[GI::YY] = own_closed ? ymom[0] : 0.0;
                gram_out[GI::SY] = own_closed ? ymom[1] : 0.0;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (a part that does not own the closed-form moments contributes zeros: the parts are summed)
        if (lane < NPAIR) {
            gram_out[GI::NT * 0 + lane] = G0;
            gram_out[GI::NT * 1 + lane] = G1;
            gram_out[GI::NT * 2 + lane] = G2;
        } else if (lane < NPAIR + NE) {
            gram_out[GI::S0(lane - NPAIR)] = G0;
            gram_out[GI::S1(lane - NPAIR)] = G1;
        }
    } else {
        wave_reduce_store<NVR>(acc, gram_out);
    }
}

// (dispatch on the alignment of the rows ONCE per pass: each instance has its loads in straight-line code)
template <int NE, bool UNIFORM, bool WEIGHTED>
__device__ __forceinline__ void gram_pass(const FitgArgs &a, VP_LDS const SlotRec<double, NE + 1, NE> *rec, VP_LDS const double *grid2,
                                          VP_LDS double *gram_out, const int prob, const int lane, const int m, const int ch0,
                                          const int nchunk, const bool vec, const bool own_closed = true,
                                          VP_LDS double *ymom = nullptr) {
    constexpr bool HORNER = UNIFORM && !WEIGHTED && (VP_FITG_CLOSED != 0) && (VP_FITG_HORNER != 0);
    if constexpr (HORNER) {
        if (vec) gram_pass_v<NE, UNIFORM, WEIGHTED, true>(a, rec, grid2, gram_out, prob, lane, m, ch0, nchunk, vec, own_closed, ymom);
        else gram_pass_v<NE, UNIFORM, WEIGHTED, false>(a, rec, grid2, gram_out, prob, lane, m, ch0, nchunk, vec, own_closed, ymom);
    } else {
        gram_pass_v<NE, UNIFORM, WEIGHTED, false>(a, rec, grid2, gram_out, prob, lane, m, ch0, nchunk, vec, own_closed, ymom);
    }
}

// ---- role-specialised waves ------------------------------------------------------------------------------------------
// If every wave alternated moment passes and bookkeeping (measured: 4.0 ms per 8192 fits) the lane-serial LM bookkeeping --
// ~33 k cycles of issue per pass however few lanes are active (tools/gram_clocks.py: gram_phase 7 k + slot_scalar_phase
// 26 k) -- would be paid once per 2-3 evaluations and stretch every fit's round.  Here a workgroup of 8 waves owns a POOL
// of NS slots: wave 0 (the scalar wave) does nothing but the bookkeeping, lane s <-> slot s, for every slot whose moments
// are ready -- one pass serves up to NS evaluations -- and waves 1..7 (the stream waves) do nothing but moment passes, each
// claiming the next slot that has a trial point.  Slots move through 1 (needs a pass) -> 2 (being streamed) -> 3 (moments
// ready; bookkeeping) -> 1 | refill | 0 (empty); the state words live in LDS, claims are LDS compare-and-swaps, there is no
// barrier in the loop.  A fit's result is independent of who processed it and when.
// (re)fill the slot of THIS lane: only the LM record -- no data is staged
template <int N, int Q>
__device__ __forceinline__ void slotg_fill_lane(VP_LDS SlotRec<double, N, Q> *rec, VP_LDS const SlotConsts<double, float> *k,
                                                const int prob) {
    const float *a0 = k->alpha + (int64_t)prob * Q;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const double v = (double)a0[i];
        rec->xt[i] = v;
        rec->x[i] = v;
        rec->diag[i] = 1.0;
        rec->qtf[i] = 0.0;
        rec->acnorm[i] = 0.0;
        rec->ipvt[i] = i;
#pragma unroll
        for (int j = 0; j < Q; ++j) rec->Rj[i][j] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        rec->cbest[i] = 0.0;
        rec->cnew[i] = 0.0;
    }
    rec->fnorm = rec->delta = rec->par = rec->xnorm = rec->gnorm = rec->pnorm = rec->prered = rec->dirder = 0.0;
    rec->objective = 0.0 / 0.0;
    rec->fnorm1 = rec->actred = rec->ratio = 0.0;
    rec->qty0 = 0.0;
    rec->flags = 1 | 2 | 4;
    rec->nfev = 0;
    rec->term = VP_TERM_NOT_RUN;
    rec->status = VP_ST_NOT_EVALUATED;
    rec->prob = prob;
    rec->trow = 0;
#if VP_FITG_TIMELINE
    if (k->status) k->status[prob] = (int32_t)(wall_clock64() & 0x7fffffffull);
#endif
}

#ifndef VP_FITG_NS
#define VP_FITG_NS 32
#endif
#ifndef VP_FITG2_NWAVES
#define VP_FITG2_NWAVES 8
#endif
#ifndef VP_FITG_PARTS
#define VP_FITG_PARTS 4        // row ranges of a split moment pass
#endif
#ifndef VP_FITG_SPLIT_AFTER
#define VP_FITG_SPLIT_AFTER 24 // a fit's passes are split from this evaluation on
#endif
#ifndef VP_FITG_TIMELINE
#define VP_FITG_TIMELINE 0     // debug builds: start / finish clock of every fit in status / cost (100 MHz ticks)
#endif
#ifndef VP_FITG_OLD_FIRST
#define VP_FITG_OLD_FIRST 1    // fits past VP_FITG_SPLIT_AFTER evaluations: claimed first, bookkeeping by the wave that streamed them
#endif
#ifndef VP_FITG_TAIL_LIVE
#define VP_FITG_TAIL_LIVE 7    // live fits per workgroup from which a stream wave keeps the slot for the bookkeeping
#endif
constexpr int VP_FITG2_WAVES = VP_FITG2_NWAVES;

template <class M, int NS, bool UNIFORM, bool WEIGHTED>
__global__ void __launch_bounds__(64 * VP_FITG2_WAVES, VP_FITG2_WAVES == 8 ? 2 : 1) fitg2_kernel(const FitgArgs a) {
    constexpr int N = M::N, Q = M::Q, NE = M::N - 1;
    static_assert(M::kStatic && M::kConstLast && M::kDiagonalPairs && M::Q == NE, "exponentials + offset");
    static_assert(NS <= 64, "one lane of the scalar wave per slot");
    using GI = GramIdx<NE>;
    using Rec = SlotRec<double, N, Q>;
    using KC = SlotConsts<double, float>;
    __shared__ __attribute__((aligned(16))) double s_gram[NS][GI::NV];
    __shared__ __attribute__((aligned(16))) Rec s_recs[NS];
    __shared__ __attribute__((aligned(16))) KC s_kc;
    __shared__ double s_grid[NS][2];
    __shared__ double s_ymom[NS][2]; // sum y^2, sum y of the slot's problem (gram_pass)
    __shared__ int s_state[NS]; // 0 empty | 1 needs a moment pass | 2 being streamed | 3 moments ready / bookkeeping | 4 split pass, parts unclaimed
    __shared__ int s_live;      // slots that hold a fit
    // split passes (the tail of a launch): the parts' moments, the next unclaimed part, the parts finished
    __shared__ __attribute__((aligned(16))) double s_pgram[NS][VP_FITG_PARTS][GI::NV];
    __shared__ int s_next[NS], s_done[NS];
    const int lane = lane_id();
    const int wv = (int)(threadIdx.x >> 6);
    const int m = a.m;
    VP_LDS Rec *recs = (VP_LDS Rec *)&s_recs[0];
    VP_LDS double *gram = (VP_LDS double *)&s_gram[0][0];
    VP_LDS const KC *kc = (VP_LDS const KC *)&s_kc;
    if (threadIdx.x == 0) {
        KC *k = &s_kc;
        k->ftol = a.ftol;
        k->xtol = a.xtol;
        k->gtol = a.gtol;
        k->stepbound = a.stepbound;
        k->alpha = a.alpha;
        k->C_out = a.C_out;
        k->cost_out = a.cost_out;
        k->status = a.status;
        k->report = a.report;
        k->trace = VP_FITG_TIMELINE ? nullptr : a.trace;
        k->yw = a.yw;
        k->queue = a.queue;
        k->rescue = nullptr; // (the Gram kernel's failures are conditioning, not representability: no re-fit list)
        k->rescue_slot = 0;
        k->B = a.B;
        k->trace_rows = a.trace_rows;
        k->scale_diag = a.scale_diag;
        k->max_fev = a.patience * (Q + 1);
        k->m = a.m;
        k->eps = a.eps;
        s_live = 0;
    }
    __syncthreads();
    auto grid_of = [&](const int s, const int prob) __attribute__((always_inline)) {
        if constexpr (UNIFORM) {
            const float *tp = a.t + (int64_t)prob * a.t_stride;
            const double t0 = (double)tp[0];
            s_grid[s][0] = t0;
            s_grid[s][1] = ((double)tp[m - 1] - t0) / (double)(m - 1);
        }
    };
    // static first assignment: workgroup g takes problems g * ns_used .. + ns_used - 1 (lane s of wave 0 fills slot s)
    if (wv == 0 && lane < NS) {
        const int64_t prob = (int64_t)blockIdx.x * a.gs_used + lane;
        const bool have = lane < a.gs_used && prob < a.B;
        if (have) {
            slotg_fill_lane<N, Q>(recs + lane, kc, (int)prob);
            grid_of(lane, (int)prob);
            __hip_atomic_fetch_add(&s_live, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            s_recs[lane].prob = -1;
            s_recs[lane].term = VP_TERM_NOT_RUN;
        }
        s_state[lane] = have ? 1 : 0;
        s_next[lane] = VP_FITG_PARTS; // (nothing to claim)
        s_done[lane] = 0;
    }
    __syncthreads();
    const int nchunk = (m + 255) / 256;
    const bool vec = (m & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.yw) | reinterpret_cast<uintptr_t>(a.t) |
                                       reinterpret_cast<uintptr_t>(a.w)) & 15) == 0;
    auto lds_release = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    constexpr int NSC = VP_FITG_SCALAR_WAVES, GSW = NS / NSC; // scalar waves, slots per scalar wave
    static_assert(NS % NSC == 0 && NSC < VP_FITG2_WAVES, "the pool divides evenly over the scalar waves");
    // after the bookkeeping of slot `slot` (executed by ONE lane, `on`): a finished fit's slot takes the next problem of the
    // device-side queue or becomes empty; then the slot is handed back to the stream waves
    auto slot_advance = [&](const int slot, const bool on) __attribute__((always_inline)) {
        if (on) {
            int next_state = 1;
            if (s_recs[slot].term != 0) { // the fit of this slot is finished (results written): next problem, or empty
#if VP_FITG_TIMELINE
                if (a.cost_out) a.cost_out[s_recs[slot].prob] = (double)(wall_clock64() & 0x7fffffffull); // (tools/cfg4_timeline.py)
#endif
                const int next = __hip_atomic_fetch_add(a.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int64_t)next < a.B) {
                    slotg_fill_lane<N, Q>(recs + slot, kc, next);
                    grid_of(slot, next);
                } else {
                    s_recs[slot].prob = -1;
                    s_recs[slot].term = VP_TERM_NOT_RUN;
                    next_state = 0;
                    __hip_atomic_fetch_add(&s_live, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            lds_release(); // the record is complete before the slot is handed to a stream wave
            __hip_atomic_store(&s_state[slot], next_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // The moments of slot s are in LDS.  While the workgroup holds many fits the slot goes to the scalar wave (state 3), whose
    // one bookkeeping pass serves every slot that is ready; a slot that becomes ready just after that pass began waits for the
    // whole of it, though, and at the END of a launch -- a handful of long fits per workgroup, out of phase with each other --
    // that wait was half of every round (31 us per round of the longest fit against 3 + 14 us of pass + bookkeeping).  With at
    // most VP_FITG_TAIL_LIVE fits left there is a stream wave per fit: the wave that delivered the moments keeps the slot
    // (state 2) and runs the same two functions itself on its lane 0 -- the same arithmetic on the same record, so the result
    // is the same whoever served the slot.
#if VP_FITG_TIMELINE
    unsigned long long tl_own_clocks = 0;
#endif
    auto moments_ready = [&](const int s) __attribute__((always_inline)) {
        const bool own = VP_FITG_TAIL_LIVE > 0 && !a.dbg &&
                         (uni(__hip_atomic_load(&s_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= VP_FITG_TAIL_LIVE ||
                          (VP_FITG_OLD_FIRST && uni(s_recs[s].nfev) >= VP_FITG_SPLIT_AFTER));
        if (!own) {
            if (lane == 0) __hip_atomic_store(&s_state[s], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
#if VP_FITG_TIMELINE
        const unsigned long long tl_b = wall_clock64();
#endif
        gram_phase<NE, GSW, WEIGHTED>(recs + s, gram + (size_t)s * GI::NV, kc, nullptr, lane == 0);
        lds_release();
        slot_scalar_phase_inl<double, N, Q, GSW, float, VP_FITG_CHOL_LMPAR != 0>(recs + s, kc, lane == 0);
        lds_release();
        slot_advance(s, lane == 0);
#if VP_FITG_TIMELINE
        tl_own_clocks += wall_clock64() - tl_b;
#endif
    };
    if (wv < NSC) {
        // ======================= a scalar wave: lane s <-> slot base + s =======================
        // (the bookkeeping costs its ~33 k cycles of issue per pass however few lanes are active: with the moment passes
        // three times shorter -- closed-form moments -- one scalar wave per pool was what every fit's round waited for)
        const int base = wv * GSW;
        VP_LDS Rec *wrecs = recs + base;
#if VP_FITG_TIMELINE
        const unsigned long long tl_t0 = wall_clock64();
        unsigned long long tl_busy = 0, tl_trips = 0, tl_served = 0;
#endif
        for (;;) {
            const int st = (lane < GSW) ? __hip_atomic_load(&s_state[base + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
            asm volatile("" ::: "memory");
            const bool act = st == 3;
            if (!uni(act)) {
                if (uni(__hip_atomic_load(&s_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= 0) break;
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
#if VP_FITG_TIMELINE
            const unsigned long long tl_a = wall_clock64();
            tl_trips += 1;
            tl_served += __builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
#endif
            gram_phase<NE, GSW, WEIGHTED>(wrecs, gram + (size_t)base * GI::NV, kc, a.dbg, act);
            lds_release();
            if (!a.dbg) {
                slot_scalar_phase_inl<double, N, Q, GSW, float, VP_FITG_CHOL_LMPAR != 0>(wrecs, kc, act);
                lds_release();
            }
            slot_advance(base + lane, act);
#if VP_FITG_TIMELINE
            tl_busy += wall_clock64() - tl_a;
#endif
        }
#if VP_FITG_TIMELINE
        if (a.trace && lane == 0) {
            double *o = a.trace + ((size_t)blockIdx.x * VP_FITG2_WAVES + wv) * 8;
            o[0] = 0.0; o[1] = (double)(wall_clock64() - tl_t0); o[2] = (double)tl_busy; o[3] = (double)tl_trips; o[4] = (double)tl_served; o[5] = 0.0;
        }
#endif
    } else {
#if VP_FITG_IDLE_PARTNER
        // the wave that shares its SIMD with a scalar wave (waves of a workgroup are dealt round-robin over the 4 SIMDs) stays
        // idle: the lane-serial bookkeeping is the latency of every fit's round and runs at full issue rate alone
        if (wv >= 4 && wv - 4 < NSC) return;
#endif
        // ======================= stream waves: claim a slot with a trial point, stream its rows =======================
        int start = (wv - NSC) * (NS / (VP_FITG2_WAVES - NSC)); // spread the first claims over the pool
#if VP_FITG_TIMELINE
        const unsigned long long tl_t0 = wall_clock64();
        unsigned long long tl_busy = 0, tl_pass = 0, tl_parts = 0, tl_a = 0;
#define VP_TL_END()                                                                                                     \
    if (a.trace && lane == 0) {                                                                                         \
        double *o = a.trace + ((size_t)blockIdx.x * VP_FITG2_WAVES + wv) * 8;                                           \
        o[0] = 1.0; o[1] = (double)(wall_clock64() - tl_t0); o[2] = (double)tl_busy; o[3] = (double)tl_pass;            \
        o[4] = (double)tl_parts; o[5] = (double)tl_own_clocks;                                                                 \
    }
#else
#define VP_TL_END()
#endif
        for (;;) {
            // lanes look at one slot each; the first slot at or after `start` (cyclically) that has a trial point is claimed
            // (serving the OLDEST fit first instead was measured: no gain -- a fit's round is the pass + the bookkeeping, not
            // the queueing)
            int sl = lane + start;
            sl = sl >= NS ? sl - NS : sl;
            const int st = (lane < NS) ? __hip_atomic_load(&s_state[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
            unsigned long long ready = __builtin_amdgcn_ballot_w64(lane < NS && (st == 1 || st == 4));
            if (VP_FITG_OLD_FIRST && ready != 0ull) {
                // the fits that have outlived VP_FITG_SPLIT_AFTER evaluations decide when the launch ends: they are served first
                const bool old_fit = lane < NS && (st == 4 || (st == 1 && s_recs[sl].nfev >= VP_FITG_SPLIT_AFTER));
                const unsigned long long older = __builtin_amdgcn_ballot_w64(old_fit);
                if (older != 0ull) ready = older;
            }
            if (ready == 0ull) {
                if (uni(__hip_atomic_load(&s_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= 0) break;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
#if VP_FITG_TIMELINE
            tl_a = wall_clock64();
#endif
            const int off = (int)__builtin_ctzll(ready);
            int s = off + start;
            s = s >= NS ? s - NS : s;
            const int sst = __builtin_amdgcn_readlane(st, off);
            // A fit that has outlived VP_FITG_SPLIT_AFTER evaluations is (with the evaluation counts of this workload: mean
            // 18) one of the few its workgroup still holds while most stream waves idle: its pass is SPLIT into
            // VP_FITG_PARTS row ranges that as many waves stream at once, the last one to finish adds the parts' moments
            // in a fixed order.  Whether a pass is split depends on the fit's own evaluation count only, never on timing:
            // results stay independent of who processed a slot and when.
            int part = -1; // -1: a whole pass
            if (sst == 1) {
                int won = 0;
                if (lane == 0) {
                    int expect = 1;
                    won = __hip_atomic_compare_exchange_strong(&s_state[s], &expect, 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
                }
                if (!uni(won)) continue; // another stream wave was faster
                asm volatile("" ::: "memory");
                if (VP_FITG_PARTS > 1 && !a.dbg && uni(s_recs[s].nfev) >= VP_FITG_SPLIT_AFTER) {
                    part = 0;
                    if (lane == 0) {
                        s_done[s] = 0;
                        s_next[s] = 1;
                    }
                    lds_release();
                    if (lane == 0) __hip_atomic_store(&s_state[s], 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else { // a split pass with unclaimed parts
                int pp = 0;
                if (lane == 0) pp = __hip_atomic_fetch_add(&s_next[s], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pp = uni(pp);
                if (pp >= VP_FITG_PARTS) { // all taken (the state word was about to say so)
                    start = s + 1 >= NS ? 0 : s + 1;
                    continue;
                }
                if (pp == VP_FITG_PARTS - 1 && lane == 0) {
                    int expect = 4; // (only while it still is THIS pass that is being split)
                    (void)__hip_atomic_compare_exchange_strong(&s_state[s], &expect, 2, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                part = pp;
                asm volatile("" ::: "memory");
            }
            const int prob = uni(s_recs[s].prob);
            int ready_slot = -1;
            if (part < 0) {
                gram_pass<NE, UNIFORM, WEIGHTED>(a, recs + s, (VP_LDS const double *)&s_grid[s][0], gram + (size_t)s * GI::NV, prob, lane, m,
                                                 0, nchunk, vec, true, (VP_LDS double *)&s_ymom[s][0]);
                lds_release(); // the moments are in LDS before the slot is handed on
                ready_slot = s;
            } else {
                const int c0 = (int)((long)part * nchunk / VP_FITG_PARTS), c1 = (int)((long)(part + 1) * nchunk / VP_FITG_PARTS);
                gram_pass<NE, UNIFORM, WEIGHTED>(a, recs + s, (VP_LDS const double *)&s_grid[s][0],
                                                 (VP_LDS double *)&s_pgram[s][part][0], prob, lane, m, c0, c1, vec, part == 0,
                                                 (VP_LDS double *)&s_ymom[s][0]);
                lds_release();
                int d = 0;
                if (lane == 0) d = __hip_atomic_fetch_add(&s_done[s], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (uni(d) == VP_FITG_PARTS - 1) { // the last part: total in the order of the parts
                    asm volatile("" ::: "memory");
                    for (int v = lane; v < GI::NV; v += 64) {
                        double t = s_pgram[s][0][v];
#pragma unroll
                        for (int q2 = 1; q2 < VP_FITG_PARTS; ++q2) t += s_pgram[s][q2][v];
                        s_gram[s][v] = t;
                    }
                    lds_release();
                    ready_slot = s;
                }
            }
            if (ready_slot >= 0) moments_ready(ready_slot); // (one call site: the bookkeeping may be inlined here)
            start = s + 1 >= NS ? 0 : s + 1;
#if VP_FITG_TIMELINE
            tl_busy += wall_clock64() - tl_a;
            if (part < 0) tl_pass += 1; else tl_parts += 1;
#endif
        }
        VP_TL_END();
    }
}

// fp32 handle, single right-hand side, NE exponentials + offset: the Gram kernel for every combination of grid
// (shared / per problem, uniform or not) and weights (none / shared / per problem)
template <class M> int launch_fitg(const LaunchParams &p, double *dbg = nullptr) {
    if (p.S != 1 || !p.queue) return VP_ERR_UNSUPPORTED;
    FitgArgs a;
    a.t = (const float *)p.t;
    a.w = (const float *)p.w;
    a.yw = (const float *)p.yw;
    a.alpha = (float *)p.alpha_out;
    a.C_out = (float *)p.C_out;
    a.cost_out = p.cost_out;
    a.status = p.status;
    a.report = p.report;
    a.trace = p.trace;
    a.dbg = dbg;
    a.queue = p.queue;
    a.B = p.B;
    a.t_stride = p.t_stride;
    a.w_stride = p.w_stride;
    a.m = p.m;
    a.trace_rows = p.trace_rows;
    a.scale_diag = p.opts->scale_diag;
    a.patience = p.opts->patience;
    a.eps = p.eps;
    a.ftol = p.opts->ftol;
    a.xtol = p.opts->xtol;
    a.gtol = p.opts->gtol;
    a.stepbound = p.opts->stepbound;
    if (a.B <= 0) return VP_ERR_OK;
    const bool uniform = p.grid_uniform != 0 && p.m >= 3;
    const bool weighted = p.w != nullptr || (p.m & 3) != 0; // a ragged last row group is masked through the weights
    // role-specialised waves: one 8-wave workgroup per CU (1 scalar wave + 7 stream waves) with a pool of NS slots; the
    // static first assignment spreads the batch over all workgroups
    constexpr int NS = VP_FITG_NS;
    const int64_t cap_groups = (int64_t)p.num_cus;
    int64_t ns_used = (a.B + cap_groups - 1) / cap_groups;
    if (ns_used > NS) ns_used = NS;
    if (ns_used < 1) ns_used = 1;
    a.gs_used = (int)ns_used;
    int64_t blocks = (a.B + ns_used - 1) / ns_used;
    if (blocks > cap_groups) blocks = cap_groups;
    if (hipMemsetD32Async((hipDeviceptr_t)p.queue, (int)(blocks * ns_used), 1, p.stream) != hipSuccess) return VP_ERR_HIP;
    const dim3 grid((unsigned)blocks), block(64 * VP_FITG2_WAVES);
    if (uniform && !weighted) hipLaunchKernelGGL((fitg2_kernel<M, NS, true, false>), grid, block, 0, p.stream, a);
    else if (uniform) hipLaunchKernelGGL((fitg2_kernel<M, NS, true, true>), grid, block, 0, p.stream, a);
    else if (!weighted) hipLaunchKernelGGL((fitg2_kernel<M, NS, false, false>), grid, block, 0, p.stream, a);
    else hipLaunchKernelGGL((fitg2_kernel<M, NS, false, true>), grid, block, 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
