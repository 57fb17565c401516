// vp_model.hpp -- device-side model policies: the closed basis-function descriptor language of
// include/varpro_hip.h evaluated into register-resident columns.
//
// Reference semantics being reproduced: SeparableNonlinearModel::{set_params, eval,
// eval_partial_deriv} (src/model/mod.rs:266-267,308,359-362) for the models of
// shared_test_code/src/{lib.rs:101-135, models.rs:87-149, models.rs:310-372} and
// src/test_helpers/mod.rs:11-72, plus the row scaling by the weights (src/util/mod.rs:76-96).
#pragma once
#include "../../include/varpro_hip.h"
#include "vp_device.hpp"

#ifndef VP_BUILD_CHUNK
#define VP_BUILD_CHUNK 8
#endif

namespace vp {

// ---- static (compile-time) model: sum of NEXP exponential decays + optional constant offset ----
// basis j < NEXP: exp(-t/alpha_j); basis NEXP: 1.  Pair p == (basis p, param p).
template <int NEXP, bool OFFSET> struct MultiExpModel {
    static constexpr int N = NEXP + (OFFSET ? 1 : 0);
    static constexpr int Q = NEXP;
    static constexpr int P = NEXP;
    static constexpr bool kStatic = true;
    __host__ __device__ constexpr int kind(int j) const { return j < NEXP ? VP_BASIS_EXP_DECAY : VP_BASIS_CONST; }
    __host__ __device__ constexpr int param(int j, int a) const { return (j < NEXP && a == 0) ? j : -1; }
    __host__ __device__ constexpr int pair_basis(int p) const { return p; }
    __host__ __device__ constexpr int pair_arg(int) const { return 0; }
    __host__ __device__ constexpr int pair_param(int p) const { return p; }
    static constexpr bool kDiagonalPairs = true; // pair p <-> (basis p, param p), P == Q
    static constexpr bool kConstLast = OFFSET;   // the last basis is the constant: see evaluate_core_const_first
    __host__ __device__ constexpr int out_index(int j) const { return j; }
};

// ---- runtime model with compile-time sizes: any mix of kinds / shared parameters ---------------
template <int N_, int Q_, int P_> struct RtModel {
    static constexpr int N = N_, Q = Q_, P = P_;
    static constexpr bool kStatic = false;
    static constexpr bool kDiagonalPairs = false;
    static constexpr bool kConstLast = false;
    int32_t kind_[N_];
    int32_t par_[N_][VP_MAX_BASIS_PARAMS];
    int32_t pb_[P_], pa_[P_], pp_[P_];
    int32_t out_[N_]; // column j is basis function out_[j] of the caller's model (identity unless sweep_invariant_first reordered it)
    __host__ __device__ int out_index(int j) const { return out_[j]; }
    __host__ __device__ int kind(int j) const { return kind_[j]; }
    __host__ __device__ int param(int j, int a) const { return par_[j][a]; }
    __host__ __device__ int pair_basis(int p) const { return pb_[p]; }
    __host__ __device__ int pair_arg(int p) const { return pa_[p]; }
    __host__ __device__ int pair_param(int p) const { return pp_[p]; }
};

// fp64 exp for the basis columns: 2*m of these per evaluation dominate the vector work, so it is
// written out: k = rint(x log2 e); r = x - k ln2 (two FMAs, hi/lo split); degree-13 Taylor polynomial on
// |r| <= ln2/2 (truncation 6e-18 relative); ldexp.  19 instructions, < 1 ulp + 1 ulp of Horner rounding;
// no range-check selects: v_ldexp_f64 saturates to inf / flushes to 0 by itself, NaN propagates.
__device__ __forceinline__ double texp(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-k, 6.93147180559945286227e-01, x);
    r = __builtin_fma(-k, 2.31904681384629955842e-17, r);
    double p = 1.6059043836821613e-10;                 // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868100e-09);   // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441720e-08);   // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985893e-07);   // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985888e-06);   // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587302e-05);   // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841e-04);   // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889e-03);   // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332e-03);   // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}
__device__ __forceinline__ float texp(float x) { return __ocml_exp_f32(x); }

// NX exponentials at once: the Horner chains of the NX arguments interleave (ILP), and the 16 fp64 constants of the
// polynomial are materialised HERE (the empty asm makes each one opaque, i.e. not a loop-invariant the compiler may
// hoist): 32 SGPRs that would otherwise stay allocated -- and get spilled -- across the whole LM iteration.
template <int NX> __device__ __forceinline__ void texp_n(const double (&x)[NX], double (&out)[NX]) {
    double cs[16] = {1.4426950408889634074,  6.93147180559945286227e-01, 2.31904681384629955842e-17,
                     1.6059043836821613e-10, 2.0876756987868100e-09,     2.5052108385441720e-08,
                     2.7557319223985893e-07, 2.7557319223985888e-06,     2.4801587301587302e-05,
                     1.9841269841269841e-04, 1.3888888888888889e-03,     8.3333333333333332e-03,
                     4.1666666666666664e-02, 1.6666666666666666e-01,     0.5,
                     1.0};
#pragma unroll
    for (int i = 0; i < 14; ++i) asm volatile("" : "+s"(cs[i]));
    double k[NX], r[NX], p[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        k[i] = __builtin_rint(x[i] * cs[0]);
        r[i] = __builtin_fma(-k[i], cs[1], x[i]);
        r[i] = __builtin_fma(-k[i], cs[2], r[i]);
        p[i] = cs[3];
    }
#pragma unroll
    for (int c = 4; c < 15; ++c)
#pragma unroll
        for (int i = 0; i < NX; ++i) p[i] = __builtin_fma(p[i], r[i], cs[c]);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        p[i] = __builtin_fma(p[i], r[i], 1.0);
        p[i] = __builtin_fma(p[i], r[i], 1.0);
        out[i] = __builtin_ldexp(p[i], (int)k[i]);
    }
}
template <int NX> __device__ __forceinline__ void texp_n(const float (&x)[NX], float (&out)[NX]) {
#pragma unroll
    for (int i = 0; i < NX; ++i) out[i] = texp(x[i]);
}
// sin / cos only occur in run-time-descriptor models, whose column build is unrolled over rows x columns x kinds.
// The library routines stay out of line (one copy per kernel instead of R*N inlined Payne-Hanek expansions, which made MBs
// of code) and are only reached for huge arguments:
__device__ __noinline__ double tsin(double x) { return __ocml_sin_f64(x); }
__device__ __noinline__ float tsin(float x) { return __ocml_sin_f32(x); }
__device__ __noinline__ double tcos(double x) { return __ocml_cos_f64(x); }
__device__ __noinline__ float tcos(float x) { return __ocml_cos_f32(x); }
// sin AND cos of one fp64 argument, inline (round 4): k = rint(x 2/pi), three-term Cody-Waite reduction with FMAs
// (pi/2 = HI + MID + LO to ~160 bits: exact to rounding for |x| <= 2^20), the fdlibm kernel polynomials on |r| <= pi/4
// (< 1 ulp each), quadrant by k mod 4: ~45 instructions for both values.  Every element of an exp*cos / sin-phase
// column needed two out-of-line library calls before, and a CALL in the middle of the unrolled column build is what costs
// (the live columns around it): inlined, the O'Leary model fits 46 % faster (3.2 -> 4.7 M fits/s at m = 1024) -- and every
// run-time-descriptor kernel grows by N x R copies of it (library + 12 MB).  Inlined therefore only in the translation
// units that define VP_INLINE_SINCOS (the length-agnostic run-time-descriptor sets, where the blocks are short and the
// generic_fallback leg of bench.py lives); one out-of-line copy per kernel elsewhere.  |x| > 2^20 (never on a sane grid)
// takes the library routines.
#ifdef VP_INLINE_SINCOS
#define VP_SINCOS_ATTR __forceinline__
#else
#define VP_SINCOS_ATTR __noinline__
#endif
__device__ VP_SINCOS_ATTR void tsincos(double x, double &sn, double &cs) {
    if (__builtin_expect(!(__builtin_fabs(x) <= 1048576.0), 0)) { // (also NaN / inf)
        sn = tsin(x);
        cs = tcos(x);
        return;
    }
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    r = __builtin_fma(-k, -1.49738490485916983294e-33, r);
    const double z = r * r;
    // __kernel_sin: r + r^3 (S1 + z (S2 + ... ))
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double s0 = __builtin_fma(r * z, ps, r);
    // __kernel_cos: 1 - z/2 + z^2 (C1 + z (C2 + ...))
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double c0 = w1 + (((1.0 - w1) - hz) + z * z * pc);
    const int q = (int)k;
    const double a = (q & 1) ? c0 : s0, b = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void tsincos(float x, float &sn, float &cs) {
    sn = tsin(x);
    cs = tcos(x);
}

// ---- row sources: where the grid value t_i and the row scale of a lane's rows come from -----------
// scale_i = w_i for rows i < m (1 for unit weights) and 0 for padding rows i >= m, so that padding
// rows are exactly zero in every column and drop out of all later reductions.
// Values are fetched per register PAIR right where they are consumed (short live ranges: the grid is
// never held in 2R VGPRs across the whole column build).
//   PADDED : the arrays are 16-byte aligned and zero-padded to 64*R rows (the LDS copies of the fit
//            kernel): always 2-element accesses, no bounds clamp on the address
//   WMODE  : 0 = unit weights (compile time), 1 = weights present (compile time), 2 = decide by w != nullptr
//   VMODE  : 0 = scalar accesses only, 1 = 2-element aligned accesses (compile time), 2 = decide by `vec`
//   RECUR  : the kernel may use the uniform-grid exp recurrence of build_columns (it calls set_uniform);
//            false removes that code path at compile time (kernels that are HBM-bound or off the hot path)
//   PADM   : (PADDED, unit weights) which register pairs can hold padding rows, i.e. need the validity select on
//            the scale:  0 = any pair (general m);  1 = none (m == 64*R*W);  2 = only the LAST pair
//            (64*(R-2)*W < m < 64*R*W, e.g. m = 1000 in the 1024-row kernel)
//   UNI    : (with RECUR) the grid IS uniform -- the caller dispatched on the handle's grid check -- so only the recurrence
//            path of build_columns is compiled: a kernel that carries both paths is register-allocated for the wider one
//   TCALC  : (UNI, PADDED, unit weights) the grid values are COMPUTED, t = t(lane's first row pair) + k * delta,
//            instead of loaded: 16 loads that the scheduler issues together are 32 live registers; the lattice is within
//            4 ulp of the stored grid (grid_check_kernel)
template <typename T, int R, bool PADDED = false, int WMODE = 2, int VMODE = 2, int W = 1, bool RECUR = false,
          int PADM = 0, bool UNI = false, bool TCALC = false>
struct RowSource {
    const T *t;  // grid, indexed by row (LDS or global)
    const T *w;  // weights indexed by row, or nullptr for unit weights
    int m;       // rows >= m are padding
    int lane;    // GROUP lane (wave*64 + lane for multi-wave groups)
    bool vec;    // (!PADDED only) 2-element aligned accesses allowed
    // uniform-grid recurrence (see build_columns): the grid is t_0 + i*dt to rounding (checked once per handle by
    // grid_check_kernel) and `delta` = 64*W*VW*dt is the grid distance between a lane's consecutive row pairs
    bool uniform = false;
    T delta = T(0);
    T tl[2] = {T(0), T(0)}; // (TCALC) grid values of the lane's first row pair
    using L = Layout<R, W>;
    static constexpr int kGroupWaves = W;
    static constexpr bool kRecur = RECUR && sizeof(T) == 8 && (R > L::VW);
    static constexpr bool kAlwaysUniform = kRecur && UNI;
    // every row is valid and unweighted: the scale is the literal 1 and the recurrence needs no inf * 0 guard
    static constexpr bool kScaleOne = PADDED && WMODE == 0 && PADM == 1;
    __device__ __forceinline__ void set_uniform(bool flag) {
        if constexpr (kRecur) {
            uniform = flag && m >= 3;
            if (uniform) delta = (t[m - 1] - t[0]) / T(m - 1) * T(64 * W * L::VW);
            if constexpr (TCALC) {
                static_assert(PADDED && UNI && WMODE == 0 && L::VW == 2, "computed grid: unweighted, uniform");
                using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
                const V2 v = (reinterpret_cast<const V2 *>(t) + lane)[0];
                tl[0] = v.x;
                tl[1] = v.y;
            }
        } else {
            uniform = false;
        }
    }
    __device__ __forceinline__ bool weighted() const {
        if constexpr (WMODE == 0) return false;
        else if constexpr (WMODE == 1) return true;
        else return w != nullptr;
    }
    // BRANCH-FREE per row pair: out-of-range rows read a valid element and are zeroed by selects, so the
    // R row pairs of a column stay in one basic block.
    __device__ __forceinline__ void get(int r0, T (&tt)[2], T (&sc)[2]) const {
        // r0 even (or R == 1): registers r0, r0+1 hold rows i, i+1
        const int i = L::row_of(r0, lane);
        if constexpr (TCALC) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                tt[e] = (r0 == 0) ? tl[e] : tfma(T(r0 / 2), delta, tl[e]);
                sc[e] = T(1);
                if constexpr (PADM != 1) { // (a partially filled problem / block: rows >= m are padding)
                    const bool in = i + e < m;
                    tt[e] = in ? tt[e] : T(0);
                    sc[e] = in ? T(1) : T(0);
                }
            }
            return;
        }
        if constexpr (L::VW == 2) {
            using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
            if constexpr (PADDED) {
                // zero-padded LDS copies: the grid value of a padding row is 0 by construction, and so is its
                // weight; only the unit-weight scale of a partially filled problem needs the validity mask.
                // Addressed as (lane pointer)[constant]: ONE address register + immediate offsets (written as t + i the
                // R/2 addresses become R/2 hoisted loop invariants, i.e. spilled registers)
                const V2 v = (reinterpret_cast<const V2 *>(t) + lane)[(r0 / 2) * 64 * W];
                tt[0] = v.x;
                tt[1] = v.y;
                if constexpr (WMODE == 1) {
                    const V2 u = (reinterpret_cast<const V2 *>(w) + lane)[(r0 / 2) * 64 * W];
                    sc[0] = u.x;
                    sc[1] = u.y;
                    return;
                } else if constexpr (WMODE == 0) {
                    // r0 is a compile-time constant after unrolling: the select survives only where it can matter
                    const bool maybe_pad = (PADM == 0) || (PADM == 2 && r0 >= R - L::VW);
                    sc[0] = (!maybe_pad || i < m) ? T(1) : T(0);
                    sc[1] = (!maybe_pad || i + 1 < m) ? T(1) : T(0);
                    return;
                }
            }
            if (PADDED || VMODE == 1 || (VMODE == 2 && vec)) {
                const bool in0 = i < m, in1 = (i + 1) < m; // !PADDED: m even, in1 == in0
                const int ic = PADDED ? i : (in0 ? i : 0);
                const V2 v = *reinterpret_cast<const V2 *>(t + ic);
                tt[0] = in0 ? v.x : T(0);
                tt[1] = in1 ? v.y : T(0);
                if (weighted()) {
                    const V2 u = *reinterpret_cast<const V2 *>(w + ic);
                    sc[0] = in0 ? u.x : T(0);
                    sc[1] = in1 ? u.y : T(0);
                } else {
                    sc[0] = in0 ? T(1) : T(0);
                    sc[1] = in1 ? T(1) : T(0);
                }
                return;
            }
        }
#pragma unroll
        for (int e = 0; e < L::VW; ++e) {
            const bool in = (i + e) < m;
            const int ic = in ? (i + e) : 0;
            const T tv = t[ic];
            tt[e] = in ? tv : T(0);
            if (weighted()) {
                const T wv = w[ic];
                sc[e] = in ? wv : T(0);
            } else {
                sc[e] = in ? T(1) : T(0);
            }
        }
    }
};

// Build the (weighted) basis columns and derivative columns of one problem into the unified column
// array C:  C[j] = W phi_j  (j < N),  C[N] is left alone (data column),  C[N+1+p] = W dphi_pair_p.
//   DOFF: index of the first derivative column (N + 1 with a data column at N; N without one)
//   SKIP_CONST: constant basis columns are not written (the caller treats them implicitly)
//   WF / WD: write the basis columns / the derivative columns (the split evaluate kernel builds them in two phases; the
//            arithmetic of the phase that is not written is dead code)
//   JSEL:    >= 0: only basis JSEL is processed (with WF = false, WD = true and DOFF = -pair: ONE derivative column into C[0])
//   SHIFT:   (exponential kinds) the DERIVATIVE columns of basis j are built as 2^-ks[j] times their values, the factor
//            applied to the exponential before t / tau^2 multiplies it; the basis columns are untouched (rescue_jacobian,
//            vp_fit.hpp: a basis column within a few decades of overflow whose derivative column, or whose dot product with
//            it, is not representable)
//   PRE:     (static models, round 6) the wave-uniform scalars of the trial point come from the caller: pre[2 i] = 1 / alpha_i
//            (frcp) and pre[2 i + 1] = exp(-delta / alpha_i) (the recurrence ratio) -- the slot kernel computes them ONCE per
//            scalar phase, one lane per slot, instead of once per slot and evaluation on all 64 lanes (vp_fit2.hpp); the same
//            functions on the same inputs: bit-identical columns
template <typename T, class M, int R, int NC, class Src, int DOFF = M::N + 1, bool SKIP_CONST = false, bool WF = true,
          bool WD = true, int JSEL = -1, bool SHIFT = false, bool PRE = false>
__device__ __forceinline__ void build_columns(const M &mdl, const T (&alpha)[M::Q], const Src &src, T (&C)[NC][R],
                                              const int *ks = nullptr, const T *pre = nullptr) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
    constexpr int VW = Layout<R>::VW;
    static_assert(!WD || JSEL >= 0 || NC >= DOFF + P, "column array too small");
    // per-column invariants (wave-uniform)
    int kind[N], s0[N], s1[N];
    T p0[N], p1[N], rt[N], rt2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        kind[j] = mdl.kind(j);
        const int i0 = mdl.param(j, 0), i1 = mdl.param(j, 1);
        p0[j] = (i0 >= 0) ? dyn_get<Q>(alpha, i0) : T(0);
        p1[j] = (i1 >= 0) ? dyn_get<Q>(alpha, i1) : T(0);
        // derivative slots of this basis (pair index or -1)
        s0[j] = -1;
        s1[j] = -1;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (mdl.pair_basis(p) == j) {
                if (mdl.pair_arg(p) == 0) s0[j] = p;
                else s1[j] = p;
            }
        }
        // 1/tau by the Newton-refined v_rcp (1-2 ulp): the quotient t/tau is re-rounded by div_refined, and the
        // derivative scale 1/tau^2 = (1/tau)^2 carries ~2 ulp -- two IEEE division expansions (~13 VALU each, per
        // column and evaluation) less
        if constexpr (PRE) rt[j] = (kind[j] == VP_BASIS_EXP_DECAY) ? pre[2 * mdl.param(j, 0)] : T(0);
        else rt[j] = (kind[j] == VP_BASIS_EXP_DECAY) ? frcp(p0[j]) : T(0);
        rt2[j] = rt[j] * rt[j];
    }
    // UNIFORM-GRID RECURRENCE (fp64, R > 2): on a grid t_i = t_0 + i*dt a lane's row pairs are delta = 64*W*2*dt
    // apart, so exp(-t/tau) of pair k is the value of pair k-1 times the wave-uniform ratio exp(-delta/tau):
    // 2 full exponentials per lane and column instead of R, one multiply (+ a finite clamp that keeps the
    // zero-scaled padding rows from turning inf*0 into NaN) for the rest.  Error: <= (R/2)*1.5 ulp from the
    // chain plus the grid's own deviation from the lattice, bounded at handle creation (grid_check_kernel);
    // the derivative columns still use the actual t_i.
    constexpr bool kRecur = Src::kRecur;
    T fu[N][VW], qq[N];
    // run-time-descriptor models: the trigonometric kinds advance by a ROTATION on a uniform grid -- exp(-a t) (cos b t + i sin b t)
    // of pair k is the value of pair k-1 times the wave-uniform w = exp(-a delta) (cos b delta + i sin b delta): 2 multiplies
    // + 2 FMAs (+ the clamp of the exponential chain) per element instead of an exponential and a sine / cosine pair;
    // absolute error <= (R/2) * 2 ulp OF THE MODULUS per component (what a least-squares column is measured by)
    constexpr bool kTrig = kRecur && !M::kStatic;
    T fv[kTrig ? N : 1][VW], wr[kTrig ? N : 1], wi[kTrig ? N : 1];
    bool fast = false;
    // static models: the first row pair's exponentials and the ratios of ALL columns in one batched evaluation
    // (texp_n: interleaved Horner chains, polynomial constants live only here)
    constexpr bool kBatchExp = kRecur && M::kStatic;
    if constexpr (kRecur) {
        fast = src.uniform;
        if (fast) {
            if constexpr (kBatchExp && PRE) {
                static_assert(!PRE || M::kStatic, "precomputed trial-point scalars: static (multi-exponential) models");
                T tt0[2], sc0[2];
                src.get(0, tt0, sc0);
                T ax[N * VW], ex[N * VW];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const bool decay = kind[j] == VP_BASIS_EXP_DECAY, rate = kind[j] == VP_BASIS_EXP_RATE;
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        ax[j * VW + e] = decay ? -div_refined(tt0[e], p0[j], rt[j]) : (rate ? -p0[j] * tt0[e] : T(0));
                }
                texp_n(ax, ex);
#pragma unroll
                for (int j = 0; j < N; ++j) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) fu[j][e] = ex[j * VW + e];
                    qq[j] = (kind[j] == VP_BASIS_EXP_DECAY) ? pre[2 * mdl.param(j, 0) + 1] : T(1);
                }
            } else if constexpr (kBatchExp) {
                T tt0[2], sc0[2];
                src.get(0, tt0, sc0);
                T ax[N * (VW + 1)], ex[N * (VW + 1)];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const bool decay = kind[j] == VP_BASIS_EXP_DECAY, rate = kind[j] == VP_BASIS_EXP_RATE;
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        ax[j * (VW + 1) + e] = decay ? -div_refined(tt0[e], p0[j], rt[j]) : (rate ? -p0[j] * tt0[e] : T(0));
                    ax[j * (VW + 1) + VW] = decay ? -div_refined(src.delta, p0[j], rt[j]) : (rate ? -p0[j] * src.delta : T(0));
                }
                texp_n(ax, ex);
#pragma unroll
                for (int j = 0; j < N; ++j) {
#pragma unroll
                    for (int e = 0; e < VW; ++e) fu[j][e] = ex[j * (VW + 1) + e];
                    qq[j] = ex[j * (VW + 1) + VW];
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    qq[j] = T(1);
                    if (kind[j] == VP_BASIS_EXP_DECAY) qq[j] = texp(-div_refined(src.delta, p0[j], rt[j]));
                    else if (kind[j] == VP_BASIS_EXP_RATE) qq[j] = texp(-p0[j] * src.delta);
                    if constexpr (kTrig) {
                        wr[j] = T(1);
                        wi[j] = T(0);
                        if (kind[j] == VP_BASIS_EXP_COS || kind[j] == VP_BASIS_SIN_PHASE) {
                            const bool damped = kind[j] == VP_BASIS_EXP_COS;
                            T sn_, cs_;
                            tsincos((damped ? p1[j] : p0[j]) * src.delta, sn_, cs_);
                            const T q_ = damped ? texp(-p0[j] * src.delta) : T(1);
                            wr[j] = q_ * cs_;
                            wi[j] = q_ * sn_;
                        }
                    }
                }
            }
        }
    }
    // (re, im) <- (re, im) * (wr, wi); clamped like the exponential chain: a zero-scaled padding row must not see inf * 0
    auto rotate = [&](T &re, T &im, const T wr_, const T wi_) __attribute__((always_inline)) {
        const T a_ = re, b_ = im;
        re = tmax(tmin(tfma(a_, wr_, -(b_ * wi_)), num<T>::huge), -num<T>::huge);
        im = tmax(tmin(tfma(a_, wi_, b_ * wr_), num<T>::huge), -num<T>::huge);
    };
    auto shifted = [&](T v, int j) __attribute__((always_inline)) {
        if constexpr (SHIFT) return tldexp(v, -ks[j]);
        else return v;
    };
    // rows outermost: the grid value and row scale of a row pair are fetched (and masked) ONCE and feed all N
    // columns, whose independent transcendental pipelines interleave
    auto rows = [&](auto fast_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += VW) {
            // keep at most VP_BUILD_CHUNK rows in flight: without the fence the scheduler interleaves all R rows
            // (R x N x ~4 fp64 temporaries) and spills
            if constexpr (R > VP_BUILD_CHUNK)
                if (r0 % VP_BUILD_CHUNK == 0 && r0 != 0) __builtin_amdgcn_sched_barrier(0);
            T tt[2], sc[2];
            src.get(r0, tt, sc);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if constexpr (JSEL >= 0) {
                    if (j != JSEL) continue;
                }
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const int r = r0 + e;
                    const T t = tt[e], scl = sc[e];
                    T f, d0 = T(0), d1 = T(0);
                    if (kind[j] == VP_BASIS_CONST) {
                        if constexpr (SKIP_CONST) continue;
                        f = scl;
                    } else if (kind[j] == VP_BASIS_EXP_DECAY) {
                        // exp(-t/tau);  d/dtau = exp(-t/tau) * t / tau^2   (shared_test_code/src/lib.rs:101-114)
                        if constexpr (FAST) {
                            if constexpr (kBatchExp) {
                                if (r0 != 0) fu[j][e] = Src::kScaleOne ? fu[j][e] * qq[j] : tmin(fu[j][e] * qq[j], num<T>::huge);
                            } else {
                                fu[j][e] = (r0 == 0) ? texp(-div_refined(t, p0[j], rt[j]))
                                                     : tmin(fu[j][e] * qq[j], num<T>::huge);
                            }
                            f = fu[j][e] * scl;
                        } else {
                            f = texp(-div_refined(t, p0[j], rt[j])) * scl;
                        }
                        d0 = (shifted(f, j) * t) * rt2[j];
                    } else if (kind[j] == VP_BASIS_EXP_RATE) {
                        if constexpr (FAST) {
                            if constexpr (kBatchExp) {
                                if (r0 != 0) fu[j][e] = Src::kScaleOne ? fu[j][e] * qq[j] : tmin(fu[j][e] * qq[j], num<T>::huge);
                            } else {
                                fu[j][e] = (r0 == 0) ? texp(-p0[j] * t) : tmin(fu[j][e] * qq[j], num<T>::huge);
                            }
                            f = fu[j][e] * scl;
                        } else {
                            f = texp(-p0[j] * t) * scl;
                        }
                        d0 = -t * shifted(f, j);
                    } else if (kind[j] == VP_BASIS_EXP_COS) {
                        // exp(-a t) cos(b t)   (shared_test_code/src/models.rs:313-314, 349-372)
                        if constexpr (FAST && kTrig) {
                            if (r0 == 0) {
                                const T ex = texp(-p0[j] * t);
                                T sn_, cs_;
                                tsincos(p1[j] * t, sn_, cs_);
                                fu[j][e] = ex * cs_;
                                fv[j][e] = ex * sn_;
                            } else {
                                rotate(fu[j][e], fv[j][e], wr[j], wi[j]);
                            }
                            f = fu[j][e] * scl;
                            d0 = f * (-t);
                            d1 = -t * (fv[j][e] * scl);
                        } else {
                            const T ex = texp(-p0[j] * t) * scl;
                            T sn_, cs_;
                            tsincos(p1[j] * t, sn_, cs_);
                            f = ex * cs_;
                            d0 = f * (-t);
                            d1 = -t * ex * sn_;
                        }
                    } else { // VP_BASIS_SIN_PHASE   (src/test_helpers/mod.rs:28-52)
                        T sn_, cs_;
                        if constexpr (FAST && kTrig) {
                            if (r0 == 0) {
                                tsincos(p0[j] * t + p1[j], sn_, cs_);
                                fu[j][e] = cs_;
                                fv[j][e] = sn_;
                            } else {
                                rotate(fu[j][e], fv[j][e], wr[j], wi[j]);
                            }
                            cs_ = fu[j][e];
                            sn_ = fv[j][e];
                        } else {
                            tsincos(p0[j] * t + p1[j], sn_, cs_);
                        }
                        const T cs = cs_ * scl;
                        f = sn_ * scl;
                        d0 = t * cs;
                        d1 = cs;
                    }
                    if constexpr (WF) C[j][r] = f;
                    if constexpr (WD) {
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            if (p == s0[j]) C[DOFF + p][r] = d0;
                            if (p == s1[j]) C[DOFF + p][r] = d1;
                        }
                    }
                }
            }
        }
    };
    if constexpr (kRecur) {
        if constexpr (Src::kAlwaysUniform) rows(std::true_type{}); // (the caller dispatched on the handle's grid check)
        else if (uni(fast)) rows(std::true_type{});
        else rows(std::false_type{});
    } else {
        rows(std::false_type{});
    }
}

// Build an RtModel from the public descriptor (host side).  Returns false if sizes do not match.
template <int N_, int Q_, int P_> inline bool make_rt_model(const vp_model_desc &d, RtModel<N_, Q_, P_> &out) {
    if (d.n_basis != N_ || d.n_params != Q_) return false;
    int p = 0;
    for (int j = 0; j < N_; ++j) {
        out.kind_[j] = d.kind[j];
        out.out_[j] = j;
        for (int a = 0; a < VP_MAX_BASIS_PARAMS; ++a) {
            out.par_[j][a] = d.param[j][a];
            if (d.param[j][a] >= 0) {
                if (p >= P_) return false;
                out.pb_[p] = j;
                out.pa_[p] = a;
                out.pp_[p] = d.param[j][a];
                ++p;
            }
        }
    }
    return p == P_;
}

// FIT kernels of run-time-descriptor models: the invariant (constant) columns FIRST in the sweep, whatever their place in the
// model.  Their reflectors are then the same in every evaluation of a fit and their rounding cancels in
// actred = 1 - (||r_trial|| / ||r||)^2, which the ftol test reads at the 30-eps level -- what the static models' implicit
// constant-first sweep (evaluate_core_const_first) and the external fit's ExtFitArgs::perm do.  Only the coefficients leave a
// fit kernel per basis function: stored through out_index().  (host side; static models: nothing to do)
template <class M> inline void sweep_invariant_first(M &mdl) {
    if constexpr (!M::kStatic) {
        constexpr int N = M::N, P = M::P;
        M src = mdl;
        int pos[N];
        int k = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < N; ++j)
                if ((src.kind_[j] == VP_BASIS_CONST) == (pass == 0)) {
                    mdl.kind_[k] = src.kind_[j];
                    mdl.out_[k] = src.out_[j];
                    for (int a = 0; a < VP_MAX_BASIS_PARAMS; ++a) mdl.par_[k][a] = src.par_[j][a];
                    pos[j] = k++;
                }
        for (int p = 0; p < P; ++p) mdl.pb_[p] = pos[src.pb_[p]];
    } else {
        (void)mdl;
    }
}

} // namespace vp
