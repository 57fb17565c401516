// vp_generic.hpp -- generic fallback kernels: ANY model descriptor include/varpro_hip.h admits (n <= VP_MAX_BASIS,
// q <= VP_MAX_PARAMS, p <= VP_MAX_PAIRS, any mix of basis kinds and shared parameters) at ANY m, single right-hand
// side.  The reference accepts any SeparableNonlinearModel (src/model/mod.rs:239-363); the specialised kernels
// (vp_kernels.hpp, vp_fit.hpp, vp_fit2.hpp) keep all columns of a problem in registers and therefore exist only for
// compile-time shapes and sizes (vp_registry.hpp).  Everything else lands here: slower, but HIP -- never a CPU path.
//
// One 256-thread workgroup per problem, persistent over the batch; the problem's columns
//     [ W Phi (n) | y_w (1) | W dPhi_pair (p) | J (q) ]
// live in a global-memory workspace slot owned by the workgroup (L2-resident for the sizes this path sees), rows
// strided over the threads.  Same algorithm as the register kernels -- Householder QR of Phi_w applied to y and the
// derivative columns, truncated-SVD solve of the n x n factor (absolute epsilon, src/solvers/levmar/mod.rs:52-54),
// Kaufman Jacobian J_k = -P_perp (sum_p c_j(p) W dPhi_p) (:101-201), MINPACK qrfac with column pivoting on the
// explicit J and the crate's LM driver (:238-254) -- with run-time loop bounds.  Dot products of a reflector with all
// remaining columns are taken in ONE pass and reduced through LDS ("multi-dot").  The wave-uniform LM bookkeeping
// (lmpar / qrsolv / termination rules) is the SAME code as everywhere else (vp_lm_core.hpp), instantiated for
// q = 1..8 and dispatched on the run-time q by thread 0.
#pragma once
#include <cstring>

#include <mutex>

#include "vp_lm_core.hpp"

namespace vp {
namespace gen {

constexpr int TB = 256;                                                  // threads per workgroup
constexpr int MAXC = VP_MAX_BASIS + 1 + VP_MAX_PAIRS + VP_MAX_PARAMS;    // workspace columns
constexpr int MAXV = VP_MAX_BASIS + 1 + VP_MAX_PAIRS;                    // values per multi-dot

template <typename T> struct GenArgs {
    vp_model_desc mdl;
    int P;                                   // dependency pairs
    int pb[VP_MAX_PAIRS], pa[VP_MAX_PAIRS], pp[VP_MAX_PAIRS]; // pair -> basis, argument slot, parameter
    const T *t, *w, *yw;
    int S;              // right-hand sides per problem (evaluate only; fits are single-RHS here)
    const T *alpha;     // evaluate / basis / best_fit: [B][q]
    T *alpha_io;        // fit: in guess, out result
    T *r_out, *J_out, *C_out;
    const T *C_in;      // best_fit
    double *cost_out;
    int32_t *status;
    vp_report *report;
    T *Phi_out, *dPhi_out;
    int skip_invariant, n_phi_cols;
    T *ws;              // [gridDim.x][MAXC-ish][m] workspace
    int64_t ws_cols;    // columns per slot = n + 1 + P + q
    int m;
    int64_t B;
    int64_t t_stride, w_stride;
    T eps;
    LmOpts<T> lm;
    double *trace;
    int trace_rows;
    // global fit with the right-hand sides sharded over ranks (gen_mrhs_fit_kernel in phases, the caller's collective
    // between `sums` and `step`): 0 = the whole fit in one launch; 1 = init; 2 = sums at the trial point -> acc;
    // 3 = LM step on the (all-reduced) acc; 4 = results at the final point
    int phase;
    void *lm_state;     // [B] LmAny<T>
    double *acc;        // [B][gen_nacc(q)]: sum ||r||^2 | J^T J | J^T r | columns whose evaluation failed
    int32_t *nactive;   // [1]
    int64_t S_global;   // right-hand sides of the WHOLE problem (0: S)
    // caller-evaluated model (vp_batch_create_external; the reference's trait boundary, src/model/mod.rs:239-363): the
    // columns are READ -- Phi [B][n][ext_rows], dPhi [B][P][ext_rows] (UNWEIGHTED, pair-table order) -- not evaluated
    int ext;
    const T *ext_phi, *ext_dphi;
    int ext_rows;       // rows per column in the caller's arrays (< m only when the handle padded m < n: zero rows)
    // fit: the problems to fit -- null: all B; else list[2 + i], i < list[list_slot] (the flag-and-refit list the register and
    // streamed fit kernels append to, vp_kernels.hpp rescue_push); the launch zeroes the OTHER counter for the next fit
    const int32_t *list;
    int list_slot;
    int list_first;     // first list entry this launch fits
    // power-of-two scaling of huge columns (evaluate below): 1 for every fit launched here
    int scale_cols;
    // 1: the workgroup's column workspace lives in its dynamic LDS (it fits: ws_cols * m scalars <= kGenLdsMax) instead of the
    // global-memory slot -- every pass over the columns then costs an LDS round trip instead of an L2 / HBM one (round 6)
    int ws_lds;
};
// dynamic LDS a generic kernel may take for its columns: the CU's 160 KiB less the static records (GenShared, the LM state)
#ifndef VP_GEN_LDS_KB
#define VP_GEN_LDS_KB 150 // (A/B switch: 0 = columns always in the global-memory workspace, round 5's layout)
#endif
constexpr size_t kGenLdsMax = (size_t)VP_GEN_LDS_KB * 1024;
template <typename T> __device__ __forceinline__ T *gen_workspace(const GenArgs<T> &a, unsigned char *dyn_lds) {
    return a.ws_lds ? reinterpret_cast<T *>(dyn_lds) : a.ws + (int64_t)blockIdx.x * a.ws_cols * a.m;
}
__host__ __device__ constexpr int gen_nacc(int q) { return 2 + q * q + q; }

template <typename T> __device__ __forceinline__ T g_exp(T x);
template <> __device__ __forceinline__ double g_exp(double x) { return texp(x); }
template <> __device__ __forceinline__ float g_exp(float x) { return texp(x); }

// value and the (up to two) argument derivatives of one basis function at t -- the formulas of vp_model.hpp
template <typename T>
__device__ __forceinline__ void basis_eval(int kind, T t, T p0, T p1, T &f, T &d0, T &d1) {
    d0 = T(0);
    d1 = T(0);
    switch (kind) {
    case VP_BASIS_CONST: f = T(1); break;
    case VP_BASIS_EXP_DECAY: {
        const T rt = frcp(p0);
        f = g_exp(-div_refined(t, p0, rt));
        d0 = (f * t) * (rt * rt);
    } break;
    case VP_BASIS_EXP_RATE:
        f = g_exp(-p0 * t);
        d0 = -t * f;
        break;
    case VP_BASIS_EXP_COS: {
        const T ex = g_exp(-p0 * t);
        T sn_, cs_;
        tsincos(p1 * t, sn_, cs_);
        f = ex * cs_;
        d0 = f * (-t);
        d1 = -t * ex * sn_;
    } break;
    default: { // VP_BASIS_SIN_PHASE
        const T ph = p0 * t + p1;
        T cs;
        tsincos(ph, f, cs);
        d0 = t * cs;
        d1 = cs;
    } break;
    }
}

template <typename T> struct GenShared {
    T part[MAXV][TB / 64]; // multi-dot partials: one per wavefront (round 6: a wave-level reduction first -- the 51 KB of one partial
                           // per THREAD left no room for the problem's columns in LDS)
    unsigned long long mx[VP_MAX_BASIS]; // bit patterns of the column maxima (multi_reduce_max)
    T sw[VP_MAX_BASIS][VP_MAX_BASIS], sv[VP_MAX_BASIS][VP_MAX_BASIS]; // svd_solve's work arrays (thread 0; LDS, not scratch)
    T red[MAXV];        // reduced values
    T f[MAXV];          // per-column update factors of the current reflector
    T Rm[VP_MAX_BASIS][VP_MAX_BASIS];
    T g[VP_MAX_BASIS], qty[VP_MAX_BASIS], c[VP_MAX_BASIS], e[VP_MAX_BASIS];
    T alpha[VP_MAX_PARAMS];
    T cscale[VP_MAX_BASIS]; // 2^-e_j applied to basis column j and its derivative columns (1 unless scale_cols found it huge)
    T fn2;
    int ok;
    // Jacobian QR
    T rdiag[VP_MAX_PARAMS], wa[VP_MAX_PARAMS], acnorm[VP_MAX_PARAMS], qtf[VP_MAX_PARAMS];
    T Rj[VP_MAX_PARAMS][VP_MAX_PARAMS];
    int ipvt[VP_MAX_PARAMS], col[VP_MAX_PARAMS]; // col: workspace column of the logical (pivoted) Jacobian column
    int flag;
};

// nv dot-type sums at once: every thread holds vals[0..nv), afterwards sh.red[0..nv) holds the totals (all threads).  Per value a
// wave-level all-reduce (DPP / permlane, vp_device.hpp), then the TB / 64 wave totals added in a fixed order: deterministic.
template <typename T> __device__ __forceinline__ void multi_reduce(GenShared<T> &sh, const T *vals, int nv) {
    const int tid = (int)threadIdx.x, wave = tid >> 6, ln = tid & 63;
    for (int v = 0; v < nv; ++v) {
        const T s = wave_sum(vals[v]);
        if (ln == 0) sh.part[v][wave] = s;
    }
    __syncthreads();
    if (tid < nv) {
        T s = sh.part[tid][0];
        for (int w = 1; w < TB / 64; ++w) s += sh.part[tid][w];
        sh.red[tid] = s;
    }
    __syncthreads();
}

// One value at a time (round 6): run-time-indexed per-thread arrays (`vals[v]`, v < nv) live in SCRATCH -- every multiply-add of
// a dot pass was a scratch load + store, ~0.5 us of latency each.  The passes below keep ONE accumulator in a register per
// value and walk the rows once per value (the pivot column is re-read from LDS / L2, which is cheap): reduce_put hands the
// thread's partial of value v to the wave reduction, reduce_finish adds the wave totals (same order as multi_reduce).
template <typename T> __device__ __forceinline__ void reduce_put(GenShared<T> &sh, const int v, const T acc) {
    const T s = wave_sum(acc);
    if ((threadIdx.x & 63u) == 0) sh.part[v][threadIdx.x >> 6] = s;
}
template <typename T> __device__ __forceinline__ void reduce_finish(GenShared<T> &sh, const int nv) {
    const int tid = (int)threadIdx.x;
    __syncthreads();
    if (tid < nv) {
        T s = sh.part[tid][0];
        for (int w = 1; w < TB / 64; ++w) s += sh.part[tid][w];
        sh.red[tid] = s;
    }
    __syncthreads();
}

// the same for maxima of non-negative values: non-negative IEEE numbers order like their bit patterns, and a NaN's pattern
// lies above infinity's (a NaN wins) -- one LDS atomic per thread and value
template <typename T> __device__ __forceinline__ void max_put(GenShared<T> &sh, const int v, const T val) { // (sh.mx zeroed + barrier before)
    unsigned long long bits;
    if constexpr (sizeof(T) == 8) bits = (unsigned long long)__double_as_longlong((double)val);
    else bits = (unsigned long long)__float_as_uint((float)val);
    atomicMax(&sh.mx[v], bits);
}
template <typename T> __device__ __forceinline__ void max_finish(GenShared<T> &sh, const int nv) {
    const int tid = (int)threadIdx.x;
    __syncthreads();
    if (tid < nv) {
        if constexpr (sizeof(T) == 8) sh.red[tid] = (T)__longlong_as_double((long long)sh.mx[tid]);
        else sh.red[tid] = (T)__uint_as_float((unsigned)sh.mx[tid]);
    }
    __syncthreads();
}

// one-sided Jacobi SVD solve of the n x n upper-triangular Rm: minimum-norm c with the reference's absolute
// singular-value threshold, e = qty - Rm c (thread 0 only; run-time n)
template <typename T> __device__ void svd_solve(GenShared<T> &sh, int n, T eps) {
    // FAST PATH (round 6; the rule of the register kernels, vp_core.hpp solve_coeffs): sigma_min(R) >= 1 / ||R^-1||_F > eps
    // certifies that no singular value is truncated -> plain back-substitution, e = 0.  The Jacobi sweeps below ran for
    // EVERY evaluation before: tens of thousands of dependent single-lane instructions on scratch-resident arrays
    // (six exponentials + offset: 1.2 ms per evaluation, tools/gen_probe.py).
    {
        T(&Ri)[VP_MAX_BASIS][VP_MAX_BASIS] = sh.sw;
        bool zero_diag = false;
        for (int i = 0; i < n; ++i) zero_diag = zero_diag || (sh.Rm[i][i] == T(0));
        if (!zero_diag) {
            T inv_f2 = T(0);
            for (int j = 0; j < n; ++j)
                for (int i = j; i >= 0; --i) {
                    T acc = (i == j) ? T(1) : T(0);
                    for (int l = i + 1; l <= j; ++l) acc = tfma(-sh.Rm[i][l], Ri[l][j], acc);
                    Ri[i][j] = acc / sh.Rm[i][i];
                    inv_f2 = tfma(Ri[i][j], Ri[i][j], inv_f2);
                }
            if (inv_f2 * eps * eps < T(1)) {
                for (int i = n - 1; i >= 0; --i) {
                    T acc = sh.qty[i];
                    for (int j = i + 1; j < n; ++j) acc = tfma(-sh.Rm[i][j], sh.c[j], acc);
                    sh.c[i] = acc / sh.Rm[i][i];
                    sh.e[i] = T(0);
                }
                return;
            }
        }
    }
    T(&W)[VP_MAX_BASIS][VP_MAX_BASIS] = sh.sw;
    T(&V)[VP_MAX_BASIS][VP_MAX_BASIS] = sh.sv; // [col][row]
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            W[j][i] = (i <= j) ? sh.Rm[i][j] : T(0);
            V[j][i] = (i == j) ? T(1) : T(0);
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                T a = 0, b = 0, gg = 0;
                for (int i = 0; i < n; ++i) {
                    a = tfma(W[p][i], W[p][i], a);
                    b = tfma(W[q][i], W[q][i], b);
                    gg = tfma(W[p][i], W[q][i], gg);
                }
                if (gg != T(0) && tabs(gg) > num<T>::eps * T(0.25) * tsqrt(a * b)) {
                    rotated = true;
                    const T zeta = (b - a) / (T(2) * gg);
                    const T tt = tcopysign(T(1), zeta) / (tabs(zeta) + tsqrt(T(1) + zeta * zeta));
                    const T cs = T(1) / tsqrt(T(1) + tt * tt), sn = cs * tt;
                    for (int i = 0; i < n; ++i) {
                        T x = W[p][i], y = W[q][i];
                        W[p][i] = cs * x - sn * y;
                        W[q][i] = sn * x + cs * y;
                        x = V[p][i];
                        y = V[q][i];
                        V[p][i] = cs * x - sn * y;
                        V[q][i] = sn * x + cs * y;
                    }
                }
            }
        if (!rotated) break;
    }
    for (int i = 0; i < n; ++i) sh.c[i] = T(0);
    for (int j = 0; j < n; ++j) {
        T s2 = 0, d = 0;
        for (int i = 0; i < n; ++i) {
            s2 = tfma(W[j][i], W[j][i], s2);
            d = tfma(W[j][i], sh.qty[i], d);
        }
        const T sg = tsqrt(s2);
        const T coef = (sg > eps) ? d / s2 : T(0);
        for (int i = 0; i < n; ++i) sh.c[i] = tfma(coef, V[j][i], sh.c[i]);
    }
    for (int i = 0; i < n; ++i) {
        T acc = sh.qty[i];
        for (int j = i; j < n; ++j) acc = tfma(-sh.Rm[i][j], sh.c[j], acc);
        sh.e[i] = acc;
    }
}

// One evaluation at sh.alpha for problem b.  Workspace columns: [0,n) Phi -> Householder vectors, n: y -> Q^T y,
// [n+1, n+1+P): derivative columns -> Q^T(.).  With want_rj: column n <- r (original coordinates), columns
// [n+1+P, n+1+P+q) <- J_k.  Sets sh.c, sh.fn2, sh.ok.
template <typename T>
__device__ void evaluate(const GenArgs<T> &a, GenShared<T> &sh, T *ws, int64_t b, bool want_rj, int64_t prob = -1) {
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params, P = a.P;
    const int NCQ = n + 1 + P; // columns that take part in the sweep
    const T *tp = a.t + b * a.t_stride;
    const T *wp = a.w ? a.w + b * a.w_stride : nullptr;
    const T *yp = a.yw + (prob >= 0 ? prob : b) * (int64_t)m; // prob = b*S + s for multiple right-hand sides
    auto col = [&](int c) { return ws + (int64_t)c * m; };
    // ---- columns ----
    if (a.ext) {
        // == `&self.weights * self.model.eval()` (src/solvers/levmar/mod.rs:45-47) and `&self.weights *
        // model.eval_partial_deriv(k)` (:141) with the model calls replaced by the caller's arrays
        const int er = a.ext_rows;
        const T *ph = a.ext_phi + b * (int64_t)n * er;
        const T *dp = a.ext_dphi ? a.ext_dphi + b * (int64_t)P * er : nullptr;
        for (int i = tid; i < m; i += TB) {
            const bool in = i < er;
            const T sc = wp ? wp[i] : T(1);
            for (int j = 0; j < n; ++j) col(j)[i] = in ? ph[(int64_t)j * er + i] * sc : T(0);
            for (int p = 0; p < P; ++p) col(n + 1 + p)[i] = (in && dp) ? dp[(int64_t)p * er + i] * sc : T(0);
            col(n)[i] = yp[i];
        }
    } else
    for (int i = tid; i < m; i += TB) {
        const T t = tp[i], sc = wp ? wp[i] : T(1);
        for (int j = 0; j < n; ++j) {
            const int i0 = a.mdl.param[j][0], i1 = a.mdl.param[j][1];
            T f, d0, d1;
            basis_eval<T>(a.mdl.kind[j], t, i0 >= 0 ? sh.alpha[i0] : T(0), i1 >= 0 ? sh.alpha[i1] : T(0), f, d0, d1);
            col(j)[i] = f * sc;
            for (int p = 0; p < P; ++p)
                if (a.pb[p] == j) col(n + 1 + p)[i] = (a.pa[p] == 0 ? d0 : d1) * sc;
        }
        col(n)[i] = yp[i];
    }
    __syncthreads();
    // ---- huge columns: basis column j and its derivative columns times 2^-e_j (scale_cols) ----
    // The reference forms D_k c BEFORE it projects (src/solvers/levmar/mod.rs:156-171) and its SVD works on the matrix
    // scaled by its largest entry: a basis column of 1e153 with a coefficient of 1e-152 is an ordinary evaluation there,
    // while the raw dot products of this sweep overflow.  With e_j = the binary exponent of the largest entry of column j
    // and of its derivative columns: Phi' = Phi diag(2^-e), D'_p = D_p 2^-e_j(p) -> c' = diag(2^e) c, and r = y - Phi' c',
    // J_k = -P_perp sum_p c'_j(p) D'_p are UNCHANGED (power-of-two factors are exact); c = diag(2^-e) c' at the end.
    // Columns below 2^64 (fp32: 2^16) are left alone, i.e. every ordinary problem is bit for bit what it was.  (The
    // reference's absolute singular-value threshold then applies to the scaled factor: a stated deviation for problems that
    // are BOTH rank-deficient and hold a column beyond 2^64.)
    if (tid < n) sh.cscale[tid] = T(1);
    if (a.scale_cols) {
        if (tid < n) sh.mx[tid] = 0ull;
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            T mxj = T(0);
            for (int i = tid; i < m; i += TB) {
                mxj = tmax(mxj, tabs(col(j)[i]));
                for (int p = 0; p < P; ++p)
                    if (a.pb[p] == j) mxj = tmax(mxj, tabs(col(n + 1 + p)[i]));
            }
            max_put(sh, j, mxj);
        }
        max_finish(sh, n);
        if (tid < n) {
            const T mx = sh.red[tid];
            int e = 0;
            if (is_finite(mx) && mx > T(0)) (void)frexp(mx, &e);
            sh.cscale[tid] = (e > (sizeof(T) == 8 ? 64 : 16)) ? tldexp(T(1), -e) : T(1);
        }
        __syncthreads();
        bool any = false;
        for (int j = 0; j < n; ++j) any = any || sh.cscale[j] != T(1);
        if (any) { // (uniform)
            for (int i = tid; i < m; i += TB) {
                for (int j = 0; j < n; ++j) col(j)[i] *= sh.cscale[j];
                for (int p = 0; p < P; ++p) col(n + 1 + p)[i] *= sh.cscale[a.pb[p]];
            }
        }
        __syncthreads();
    }
    // ---- Householder sweep: H_k = I + g_k v_k v_k^T, v_k = a_k[k:] with v_k[k] = alpha - beta ----
    for (int k = 0; k < n; ++k) {
        const int nv = NCQ - k;
        const T *ak = col(k);
        for (int v = 0; v < nv; ++v) {
            const T *cv = col(k + v);
            T acc = T(0);
            for (int i = k + tid; i < m; i += TB) acc = tfma(ak[i], cv[i], acc);
            reduce_put(sh, v, acc);
        }
        reduce_finish(sh, nv);
        if (tid == 0) {
            const T alpha = ak[k], nrm2 = sh.red[0];
            const bool live = nrm2 > num<T>::norm2_min && is_finite(nrm2);
            const T sigma = live ? tsqrt(nrm2) : T(0);
            const T beta = live ? -tcopysign(sigma, alpha) : ((nrm2 <= num<T>::norm2_min) ? alpha : nrm2);
            const T u = live ? alpha - beta : T(0);
            const T gk = live ? T(1) / (beta * u) : T(0);
            sh.g[k] = gk;
            sh.Rm[k][k] = beta;
            sh.f[0] = u; // slot 0: the new pivot entry of v_k
            for (int v = 1; v < nv; ++v) {
                const T top = col(k + v)[k];
                const T f = gk * tfma(-beta, top, sh.red[v]);
                sh.f[v] = f;
                const T tj = tfma(f, u, top);
                if (k + v < n) sh.Rm[k][k + v] = tj;
                else if (k + v == n) sh.qty[k] = tj;
            }
        }
        __syncthreads();
        {
            const T u = sh.f[0];
            T *akw = col(k);
            if (tid == 0) akw[k] = u;
            __syncthreads();
            for (int i = k + tid; i < m; i += TB) {
                const T x = akw[i];
                for (int v = 1; v < nv; ++v) {
                    T *cj = col(k + v);
                    cj[i] = tfma(sh.f[v], x, cj[i]);
                }
            }
        }
        __syncthreads();
    }
    // ---- c, e, ||r||^2 ----
    if (tid == 0) {
        svd_solve(sh, n, a.eps);
        bool ok = true;
        for (int k = 0; k < n; ++k) ok = ok && is_finite(sh.c[k]) && is_finite(sh.Rm[k][k]);
        sh.ok = ok ? 1 : 0;
    }
    {
        T acc = T(0);
        const T *y = col(n);
        for (int i = n + tid; i < m; i += TB) acc = tfma(y[i], y[i], acc);
        reduce_put(sh, 0, acc);
        reduce_finish(sh, 1);
        if (tid == 0) {
            T fn2 = sh.red[0];
            for (int k = 0; k < n; ++k) fn2 = tfma(sh.e[k], sh.e[k], fn2);
            sh.fn2 = fn2;
            if (!is_finite(fn2)) sh.ok = 0;
        }
    }
    __syncthreads();
    if (!want_rj) {
        if (tid < n) sh.c[tid] *= sh.cscale[tid]; // c = diag(2^-e) c'
        __syncthreads();
        return;
    }
    // ---- r~ and J~ in Q-coordinates, then back with Q = H_0 ... H_{n-1} ----
    {
        T *y = col(n);
        if (tid < n) y[tid] = sh.e[tid];
        for (int k = 0; k < q; ++k) {
            T *zk = col(NCQ + k);
            for (int i = tid; i < m; i += TB) {
                T acc = T(0);
                if (i >= n)
                    for (int p = 0; p < P; ++p)
                        if (a.pp[p] == k) acc = tfma(-sh.c[a.pb[p]], col(n + 1 + p)[i], acc);
                zk[i] = acc;
            }
        }
    }
    __syncthreads();
    for (int k = n - 1; k >= 0; --k) {
        const int nv = 1 + q;
        const T *vk = col(k);
        for (int v = 0; v < nv; ++v) {
            const T *cv = (v == 0) ? col(n) : col(NCQ + v - 1);
            T acc = T(0);
            for (int i = k + tid; i < m; i += TB) acc = tfma(vk[i], cv[i], acc);
            reduce_put(sh, v, acc);
        }
        reduce_finish(sh, nv);
        const T gk = sh.g[k];
        for (int i = k + tid; i < m; i += TB) {
            const T x = vk[i];
            col(n)[i] = tfma(gk * sh.red[0], x, col(n)[i]);
            for (int v = 1; v < nv; ++v) {
                T *z = col(NCQ + v - 1);
                z[i] = tfma(gk * sh.red[v], x, z[i]);
            }
        }
        __syncthreads();
    }
    if (tid < n) sh.c[tid] *= sh.cscale[tid]; // c = diag(2^-e) c' (the Kaufman columns above used c')
    __syncthreads();
}

template <typename T> __global__ void __launch_bounds__(TB) gen_evaluate_kernel(const GenArgs<T> a) {
    __shared__ GenShared<T> sh;
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params, S = a.S;
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_dyn_lds[];
    T *ws = gen_workspace<T>(a, gen_dyn_lds);
    const bool want_rj = a.r_out || a.J_out;
    for (int64_t prob = blockIdx.x; prob < a.B * S; prob += gridDim.x) { // prob = b*S + s: every RHS on its own
        const int64_t b = prob / S;
        const int64_t s = prob - b * S;
        if (tid < q) sh.alpha[tid] = a.alpha[b * q + tid];
        __syncthreads();
        evaluate<T>(a, sh, ws, b, want_rj, prob);
        if (tid == 0) {
            if (a.status) a.status[prob] = sh.ok ? VP_ST_OK : VP_ST_NONFINITE;
            if (a.cost_out) a.cost_out[prob] = 0.5 * (double)sh.fn2;
        }
        if (a.C_out && tid < n) a.C_out[prob * n + tid] = sh.c[tid];
        if (a.r_out)
            for (int i = tid; i < m; i += TB) a.r_out[prob * (int64_t)m + i] = ws[(int64_t)n * m + i];
        if (a.J_out)
            for (int k = 0; k < q; ++k) // J[b][k][s][m]
                for (int i = tid; i < m; i += TB)
                    a.J_out[((b * q + k) * S + s) * (int64_t)m + i] = ws[(int64_t)(n + 1 + a.P + k) * m + i];
        __syncthreads();
    }
}

// MINPACK qrfac with column pivoting (lmder's factorisation, src/solvers/levmar/mod.rs:247) on the explicit Jacobian columns, cooperative; the
// residual column is carried along for qtf (lmder).  Columns are pivoted logically through sh.col[].
template <typename T> __device__ void jac_qrfac(const GenArgs<T> &a, GenShared<T> &sh, T *ws) {
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params;
    const int J0 = n + 1 + a.P;
    auto colp = [&](int c) { return ws + (int64_t)c * m; };
    T *rv = colp(n);
    {
        for (int k = 0; k < q; ++k) {
            const T *ck = colp(J0 + k);
            T acc = T(0);
            for (int i = tid; i < m; i += TB) acc = tfma(ck[i], ck[i], acc);
            reduce_put(sh, k, acc);
        }
        reduce_finish(sh, q);
        if (tid == 0)
            for (int k = 0; k < q; ++k) {
                sh.acnorm[k] = tsqrt(sh.red[k]);
                sh.rdiag[k] = sh.acnorm[k];
                sh.wa[k] = sh.acnorm[k];
                sh.ipvt[k] = k;
                sh.col[k] = J0 + k;
                for (int l = 0; l < q; ++l) sh.Rj[k][l] = T(0);
            }
        __syncthreads();
    }
    const int minmn = m < q ? m : q;
    for (int j = 0; j < minmn; ++j) {
        if (tid == 0) {
            int kmax = j;
            for (int k = j; k < q; ++k)
                if (sh.rdiag[k] > sh.rdiag[kmax]) kmax = k;
            if (kmax != j) {
                int ti = sh.col[j];
                sh.col[j] = sh.col[kmax];
                sh.col[kmax] = ti;
                for (int i = 0; i < j; ++i) { // finished rows of R follow their columns
                    const T tmp = sh.Rj[i][j];
                    sh.Rj[i][j] = sh.Rj[i][kmax];
                    sh.Rj[i][kmax] = tmp;
                }
                sh.rdiag[kmax] = sh.rdiag[j];
                sh.wa[kmax] = sh.wa[j];
                ti = sh.ipvt[j];
                sh.ipvt[j] = sh.ipvt[kmax];
                sh.ipvt[kmax] = ti;
            }
        }
        __syncthreads();
        T *aj = colp(sh.col[j]);
        // raw dots of the pivot column (rows >= j) with itself, the remaining columns and the residual
        const int nv = (q - j) + 1;
        for (int v = 0; v < nv; ++v) {
            const T *cv = (v == nv - 1) ? rv : colp(sh.col[j + v]);
            T acc = T(0);
            for (int i = j + tid; i < m; i += TB) acc = tfma(aj[i], cv[i], acc);
            reduce_put(sh, v, acc);
        }
        reduce_finish(sh, nv);
        if (tid == 0) {
            T ajnorm = tsqrt(sh.red[0]);
            if (ajnorm == T(0)) {
                sh.rdiag[j] = T(0);
                for (int k = j + 1; k < q; ++k) sh.Rj[j][k] = colp(sh.col[k])[j];
                sh.qtf[j] = rv[j];
                sh.flag = 0;
            } else {
                const T piv = aj[j];
                if (piv < T(0)) ajnorm = -ajnorm;
                const T vp = piv + ajnorm;         // unnormalised reflector: v' = a + ajnorm e_j, H = I + gj v' v'^T
                const T gj = T(-1) / (ajnorm * vp);
                sh.f[0] = vp;
                for (int k = j + 1; k < q; ++k) {
                    const T top = colp(sh.col[k])[j];
                    const T f = gj * tfma(ajnorm, top, sh.red[k - j]);
                    sh.f[k - j] = f;
                    const T akj = tfma(f, vp, top);
                    sh.Rj[j][k] = akj;
                    if (sh.rdiag[k] != T(0)) {
                        const T tq = akj / sh.rdiag[k];
                        const T d = T(1) - tq * tq;
                        sh.rdiag[k] *= tsqrt(d > T(0) ? d : T(0));
                    }
                }
                {
                    const T top = rv[j];
                    const T f = gj * tfma(ajnorm, top, sh.red[nv - 1]);
                    sh.f[nv - 1] = f;
                    sh.qtf[j] = tfma(f, vp, top);
                }
                sh.rdiag[j] = -ajnorm;
                sh.flag = 1;
            }
        }
        __syncthreads();
        if (sh.flag) {
            if (tid == 0) aj[j] = sh.f[0];
            __syncthreads();
            for (int i = j + tid; i < m; i += TB) {
                const T x = aj[i];
                for (int k = j + 1; k < q; ++k) {
                    T *ak = colp(sh.col[k]);
                    ak[i] = tfma(sh.f[k - j], x, ak[i]);
                }
                rv[i] = tfma(sh.f[nv - 1], x, rv[i]);
            }
            __syncthreads();
            // MINPACK's recompute rule for badly downdated norms
            for (int k = j + 1; k < q; ++k) {
                bool redo = false;
                if (sh.rdiag[k] != T(0)) {
                    const T r = sh.rdiag[k] / sh.wa[k];
                    redo = T(0.05) * (r * r) <= num<T>::eps;
                }
                if (redo) { // uniform: decided from shared values
                    T acc1 = T(0);
                    const T *ak = colp(sh.col[k]);
                    for (int i = j + 1 + tid; i < m; i += TB) acc1 = tfma(ak[i], ak[i], acc1);
                    reduce_put(sh, 0, acc1);
                    reduce_finish(sh, 1);
                    if (tid == 0) {
                        sh.rdiag[k] = tsqrt(sh.red[0]);
                        sh.wa[k] = sh.rdiag[k];
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0)
        for (int j = 0; j < q; ++j) sh.Rj[j][j] = sh.rdiag[j];
    __syncthreads();
}

// the scalar LM bookkeeping for a run-time q: the templated code of vp_lm_core.hpp, dispatched by thread 0
template <typename T, int Q> struct LmBox {
    LmVars<T, VP_MAX_BASIS, Q> s;
};
template <typename T> union LmAny {
    LmBox<T, 1> b1;
    LmBox<T, 2> b2;
    LmBox<T, 3> b3;
    LmBox<T, 4> b4;
    LmBox<T, 5> b5;
    LmBox<T, 6> b6;
    LmBox<T, 7> b7;
    LmBox<T, 8> b8;
};
#define VP_GEN_DISPATCH(q, ...)                                                                                        \
    switch (q) {                                                                                                       \
    case 1: { auto &S = lm.b1.s; constexpr int QQ = 1; (void)QQ; __VA_ARGS__; } break;                                        \
    case 2: { auto &S = lm.b2.s; constexpr int QQ = 2; (void)QQ; __VA_ARGS__; } break;                                        \
    case 3: { auto &S = lm.b3.s; constexpr int QQ = 3; (void)QQ; __VA_ARGS__; } break;                                        \
    case 4: { auto &S = lm.b4.s; constexpr int QQ = 4; (void)QQ; __VA_ARGS__; } break;                                        \
    case 5: { auto &S = lm.b5.s; constexpr int QQ = 5; (void)QQ; __VA_ARGS__; } break;                                        \
    case 6: { auto &S = lm.b6.s; constexpr int QQ = 6; (void)QQ; __VA_ARGS__; } break;                                        \
    case 7: { auto &S = lm.b7.s; constexpr int QQ = 7; (void)QQ; __VA_ARGS__; } break;                                        \
    default: { auto &S = lm.b8.s; constexpr int QQ = 8; (void)QQ; __VA_ARGS__; } break;                                       \
    }

template <typename T> __global__ void __launch_bounds__(TB) gen_fit_kernel(const GenArgs<T> a) {
    __shared__ GenShared<T> sh;
    __shared__ LmAny<T> lm;
    __shared__ int s_need_jac, s_term, s_trow;
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params;
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_dyn_lds[];
    T *ws = gen_workspace<T>(a, gen_dyn_lds);
    // (flag-and-refit launch: the problems of the list; the counter the NEXT fit appends to is zeroed here -- that fit starts
    // after this kernel on the handle's stream, and this launch never reads it)
    int64_t count = a.B;
    if (a.list) {
        count = a.list[a.list_slot];
        if (count > a.B) count = a.B;
        if (blockIdx.x == 0 && tid == 0) const_cast<int32_t *>(a.list)[a.list_slot ^ 1] = 0;
    }
    for (int64_t bi = blockIdx.x + (a.list ? a.list_first : 0); bi < count; bi += gridDim.x) {
        const int64_t b = a.list ? (int64_t)a.list[2 + bi] : bi;
        if (tid == 0) {
            T a0[VP_MAX_PARAMS];
            for (int k = 0; k < q; ++k) a0[k] = a.alpha_io[b * q + k];
            VP_GEN_DISPATCH(q, (lm_init<T, VP_MAX_BASIS, QQ>(S, a0)));
            for (int k = 0; k < q; ++k) sh.alpha[k] = a0[k];
            for (int k = 0; k < n; ++k) sh.c[k] = T(0);
            s_term = 0;
            s_trow = 0;
        }
        __syncthreads();
        T cbest[VP_MAX_BASIS]; // thread 0 only
        for (int k = 0; k < VP_MAX_BASIS; ++k) cbest[k] = T(0);
        if (q == 0) { // LevenbergMarquardt::minimize with no parameters: NoParameters (a failure)
            if (tid == 0) s_term = VP_TERM_NO_PARAMETERS;
            __syncthreads();
        }
        while (s_term == 0) {
            evaluate<T>(a, sh, ws, b, true);
            if (tid == 0) {
                const T fnorm1 = tsqrt(sh.fn2);
                bool need = false;
                VP_GEN_DISPATCH(q, {
                    // trace row: [x_trial, ||r||, ratio (not recorded here), delta, par] after the update
                    need = lm_after_eval<T, VP_MAX_BASIS, QQ, false>(S, a.lm, fnorm1, sh.ok != 0, (long)m);
                    if (S.accepted)
                        for (int k = 0; k < n; ++k) cbest[k] = sh.c[k];
                    if (a.trace && s_trow < a.trace_rows) {
                        double *tr = a.trace + ((size_t)b * a.trace_rows + s_trow) * (q + 4);
                        for (int k = 0; k < QQ; ++k) tr[k] = (double)S.xt[k];
                        tr[q] = (double)fnorm1;
                        tr[q + 1] = 0.0 / 0.0;
                        tr[q + 2] = (double)S.delta;
                        tr[q + 3] = (double)S.par;
                    }
                    ++s_trow;
                    s_term = S.term;
                });
                s_need_jac = need ? 1 : 0;
            }
            __syncthreads();
            if (s_term != 0) break;
            if (s_need_jac) jac_qrfac<T>(a, sh, ws);
            if (tid == 0) {
                VP_GEN_DISPATCH(q, {
                    if (s_need_jac) {
                        for (int k = 0; k < QQ; ++k) {
                            S.acnorm[k] = sh.acnorm[k];
                            S.qtf[k] = sh.qtf[k];
                            S.ipvt[k] = sh.ipvt[k];
                            for (int l = 0; l < QQ; ++l) S.Rj[k][l] = sh.Rj[k][l];
                        }
                    }
                    lm_next_step<T, VP_MAX_BASIS, QQ, false>(S, a.lm, s_need_jac != 0);
                    s_term = S.term;
                    for (int k = 0; k < QQ; ++k) sh.alpha[k] = S.xt[k];
                });
            }
            __syncthreads();
        }
        if (tid == 0) {
            vp_report rep;
            VP_GEN_DISPATCH(q, {
                rep.termination = S.term;
                rep.n_evals = S.nfev;
                rep.objective = (double)S.objective;
                for (int k = 0; k < QQ; ++k) a.alpha_io[b * q + k] = S.x[k];
                if (a.status) a.status[b] = S.status;
            });
            if (q == 0) {
                rep.termination = VP_TERM_NO_PARAMETERS;
                rep.n_evals = 0;
                rep.objective = 0.0 / 0.0;
                if (a.status) a.status[b] = VP_ST_NOT_EVALUATED;
            }
            a.report[b] = rep;
            if (a.cost_out) a.cost_out[b] = rep.objective;
            if (a.C_out)
                for (int k = 0; k < n; ++k) a.C_out[b * n + k] = cbest[k];
        }
        __syncthreads();
    }
}

// Global fit (one alpha, S right-hand sides; == LevMarSolver::fit on a SeparableProblem<MRHS>, src/solvers/levmar/
// mod.rs:172-186 for the Jacobian of the stacked residual) for ANY descriptor: one workgroup per problem walks the S
// columns per evaluation -- each gets its own projection (evaluate), its Kaufman columns stay in Q-coordinates (an
// orthogonal change of basis: J^T J and J^T r are the same) and only sum ||r||^2, J^T J, J^T r are carried, in double;
// the LM step factors J^T J by pivoted Cholesky (gram_to_qr, as the specialised MRHS path).  The fallback for models
// without MRHS kernel instantiations: one workgroup per problem, correctness over speed.
template <typename T> __global__ void __launch_bounds__(TB) gen_mrhs_fit_kernel(const GenArgs<T> a) {
    __shared__ GenShared<T> sh;
    __shared__ LmAny<T> lm;
    __shared__ int s_term, s_trow, s_okall;
    __shared__ double s_acc[2 + VP_MAX_PARAMS * VP_MAX_PARAMS + VP_MAX_PARAMS]; // sum ||r||^2 | J^T J (q x q) | J^T r | failed columns
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params, P = a.P, NS = a.S; // (S names the LM state inside VP_GEN_DISPATCH)
    const int NCQ = n + 1 + P;
    const int phase = a.phase;
    const long mres = (long)m * (long)(a.S_global > 0 ? a.S_global : NS);
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_dyn_lds[];
    T *ws = gen_workspace<T>(a, gen_dyn_lds);
    auto col = [&](int c) { return ws + (int64_t)c * m; };
    LmAny<T> *gstate = reinterpret_cast<LmAny<T> *>(a.lm_state);
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        if (tid == 0) {
            if (phase <= 1) {
                T a0[VP_MAX_PARAMS];
                for (int k = 0; k < q; ++k) a0[k] = a.alpha_io[b * q + k];
                VP_GEN_DISPATCH(q, (lm_init<T, VP_MAX_BASIS, QQ>(S, a0)));
                for (int k = 0; k < q; ++k) sh.alpha[k] = a0[k];
                s_term = (q == 0) ? VP_TERM_NO_PARAMETERS : 0;
                if (phase == 1) {
                    gstate[b] = lm;
                    if (q > 0) atomicAdd(a.nactive, 1);
                }
            } else {
                lm = gstate[b];
                s_term = 0;
                VP_GEN_DISPATCH(q, {
                    s_term = S.term;
                    for (int k = 0; k < QQ; ++k) sh.alpha[k] = (phase == 4) ? S.x[k] : S.xt[k];
                });
                if (q == 0) s_term = VP_TERM_NO_PARAMETERS;
            }
            s_trow = 0;
        }
        __syncthreads();
        if (phase == 1) continue;
        while (s_term == 0 && phase != 4) {
            if (phase != 3) { // ---- the sums of this rank's columns at the trial point ----
                if (tid == 0) {
                    for (int i = 0; i < gen_nacc(q); ++i) s_acc[i] = 0.0;
                    s_okall = 1;
                }
                __syncthreads();
                for (int s = 0; s < NS; ++s) {
                    evaluate<T>(a, sh, ws, b, false, b * NS + s);
                    // Kaufman columns in Q-coordinates: z_k = -sum_{pairs p of parameter k} c_{basis(p)} (Q^T D_p), rows >= n
                    for (int k = 0; k < q; ++k) {
                        T *zk = col(NCQ + k);
                        for (int i = n + tid; i < m; i += TB) {
                            T acc = T(0);
                            for (int p = 0; p < P; ++p)
                                if (a.pp[p] == k) acc = tfma(-sh.c[a.pb[p]], col(n + 1 + p)[i], acc);
                            zk[i] = acc;
                        }
                    }
                    __syncthreads();
                    for (int k = 0; k < q; ++k) {
                        const int nv = q - k + 1; // z_k . z_l (l >= k), z_k . r
                        const T *zk = col(NCQ + k);
                        for (int v = 0; v < nv; ++v) {
                            const T *cv = (v == nv - 1) ? col(n) : col(NCQ + k + v);
                            T acc = T(0);
                            for (int i = n + tid; i < m; i += TB) acc = tfma(zk[i], cv[i], acc);
                            reduce_put(sh, v, acc);
                        }
                        reduce_finish(sh, nv);
                        if (tid == 0) {
                            for (int l = k; l < q; ++l) s_acc[1 + k * q + l] += (double)sh.red[l - k];
                            s_acc[1 + q * q + k] += (double)sh.red[nv - 1];
                        }
                        __syncthreads();
                    }
                    if (tid == 0) {
                        s_acc[0] += (double)sh.fn2;
                        if (!sh.ok) s_okall = 0;
                    }
                    __syncthreads();
                }
                if (phase == 2) { // hand the sums to the caller's collective
                    if (tid == 0) {
                        s_acc[1 + q * q + q] = s_okall ? 0.0 : 1.0;
                        for (int i = 0; i < gen_nacc(q); ++i) a.acc[b * gen_nacc(q) + i] = s_acc[i];
                    }
                    __syncthreads();
                    break;
                }
            } else { // ---- phase 3: the totals over all ranks ----
                if (tid == 0) {
                    for (int i = 0; i < gen_nacc(q); ++i) s_acc[i] = a.acc[b * gen_nacc(q) + i];
                    s_okall = s_acc[1 + q * q + q] == 0.0;
                }
                __syncthreads();
            }
            if (tid == 0) {
                const T fnorm1 = tsqrt((T)s_acc[0]);
                VP_GEN_DISPATCH(q, {
                    if (phase == 3) s_trow = S.nfev; // (trace row of this evaluation)
                    const bool need = lm_after_eval<T, VP_MAX_BASIS, QQ, false>(S, a.lm, fnorm1, s_okall != 0 && is_finite(fnorm1), mres);
                    if (a.trace && s_trow < a.trace_rows) {
                        double *tr = a.trace + ((size_t)b * a.trace_rows + s_trow) * (q + 4);
                        for (int k = 0; k < QQ; ++k) tr[k] = (double)S.xt[k];
                        tr[q] = (double)fnorm1;
                        tr[q + 1] = 0.0 / 0.0;
                        tr[q + 2] = (double)S.delta;
                        tr[q + 3] = (double)S.par;
                    }
                    ++s_trow;
                    if (S.term == 0) {
                        if (need) {
                            double A[QQ][QQ], bv[QQ], Rd[QQ][QQ], acd[QQ], qd[QQ];
                            for (int k = 0; k < QQ; ++k) {
                                bv[k] = s_acc[1 + q * q + k];
                                for (int l = 0; l < QQ; ++l) A[k][l] = (l >= k) ? s_acc[1 + k * q + l] : s_acc[1 + l * q + k];
                            }
                            gram_to_qr<double, QQ>(A, bv, Rd, acd, S.ipvt, qd);
                            for (int k = 0; k < QQ; ++k) {
                                S.acnorm[k] = (T)acd[k];
                                S.qtf[k] = (T)qd[k];
                                for (int l = 0; l < QQ; ++l) S.Rj[k][l] = (T)Rd[k][l];
                            }
                        }
                        lm_next_step<T, VP_MAX_BASIS, QQ, false>(S, a.lm, need);
                    }
                    s_term = S.term;
                    for (int k = 0; k < QQ; ++k) sh.alpha[k] = S.xt[k];
                });
                if (phase == 3) {
                    gstate[b] = lm;
                    if (s_term != 0) atomicAdd(a.nactive, -1);
                }
            }
            __syncthreads();
            if (phase == 3) break;
        }
        if (phase == 2 || phase == 3) {
            if (phase == 2 && s_term != 0 && tid == 0) // finished earlier: contributes nothing (every rank agrees)
                for (int i = 0; i < gen_nacc(q); ++i) a.acc[b * gen_nacc(q) + i] = 0.0;
            __syncthreads();
            continue;
        }
        // ---- results: parameters + report, then coefficients / cost / status of every column at the final point ----
        if (tid == 0) {
            vp_report rep;
            rep.termination = VP_TERM_NO_PARAMETERS;
            rep.n_evals = 0;
            rep.objective = 0.0 / 0.0;
            VP_GEN_DISPATCH(q, {
                if (q > 0) {
                    rep.termination = S.term;
                    rep.n_evals = S.nfev;
                    rep.objective = (double)S.objective;
                    for (int k = 0; k < QQ; ++k) {
                        a.alpha_io[b * q + k] = S.x[k];
                        sh.alpha[k] = S.x[k];
                    }
                }
            });
            a.report[b] = rep;
        }
        __syncthreads();
        for (int s = 0; s < NS; ++s) {
            const int64_t prob = b * NS + s;
            evaluate<T>(a, sh, ws, b, false, prob);
            if (tid == 0) {
                if (a.status) a.status[prob] = sh.ok ? VP_ST_OK : VP_ST_NONFINITE;
                if (a.cost_out) a.cost_out[prob] = 0.5 * (double)sh.fn2;
            }
            if (a.C_out && tid < n) a.C_out[prob * n + tid] = sh.c[tid];
            __syncthreads();
        }
    }
}

// stand-alone Phi / dPhi (UNWEIGHTED) and best fit: element-wise, no workspace
template <typename T> __global__ void __launch_bounds__(TB) gen_basis_kernel(const GenArgs<T> a) {
    const int m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params, P = a.P;
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        const T *tp = a.t + b * a.t_stride;
        for (int i = (int)threadIdx.x; i < m; i += TB) {
            const T t = tp[i];
            int colo = 0;
            for (int j = 0; j < n; ++j) {
                const int i0 = a.mdl.param[j][0], i1 = a.mdl.param[j][1];
                T f, d0, d1;
                basis_eval<T>(a.mdl.kind[j], t, i0 >= 0 ? a.alpha[b * q + i0] : T(0), i1 >= 0 ? a.alpha[b * q + i1] : T(0), f,
                              d0, d1);
                if (a.Phi_out && !(a.skip_invariant && a.mdl.kind[j] == VP_BASIS_CONST)) {
                    a.Phi_out[(b * a.n_phi_cols + colo) * (int64_t)m + i] = f;
                    ++colo;
                }
                if (a.dPhi_out)
                    for (int p = 0; p < P; ++p)
                        if (a.pb[p] == j) a.dPhi_out[(b * P + p) * (int64_t)m + i] = (a.pa[p] == 0 ? d0 : d1);
            }
        }
    }
}

template <typename T> __global__ void __launch_bounds__(TB) gen_best_fit_kernel(const GenArgs<T> a, const int S) {
    // prob = b * S + s: right-hand side s of problem b shares the grid and the parameters of b (src/fit.rs:87-91)
    const int m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params;
    for (int64_t prob = blockIdx.x; prob < a.B * S; prob += gridDim.x) {
        const int64_t b = prob / S;
        const T *tp = a.t + b * a.t_stride;
        if (a.ext) { // the caller's (unweighted) Phi of the last vp_set_params_with_basis times the coefficients
            const int er = a.ext_rows;
            const T *ph = a.ext_phi + b * (int64_t)n * er;
            for (int i = (int)threadIdx.x; i < m; i += TB) {
                T acc = T(0);
                if (i < er)
                    for (int j = 0; j < n; ++j) acc = tfma(ph[(int64_t)j * er + i], a.C_in[prob * n + j], acc);
                a.r_out[prob * (int64_t)m + i] = acc;
            }
            continue;
        }
        for (int i = (int)threadIdx.x; i < m; i += TB) {
            const T t = tp[i];
            T acc = T(0);
            for (int j = 0; j < n; ++j) {
                const int i0 = a.mdl.param[j][0], i1 = a.mdl.param[j][1];
                T f, d0, d1;
                basis_eval<T>(a.mdl.kind[j], t, i0 >= 0 ? a.alpha[b * q + i0] : T(0), i1 >= 0 ? a.alpha[b * q + i1] : T(0), f,
                              d0, d1);
                acc = tfma(f, a.C_in[prob * n + j], acc);
            }
            a.r_out[prob * (int64_t)m + i] = acc;
        }
    }
}

// == FitStatistics::try_calculate (src/statistics/mod.rs:352-441) with run-time shapes: H = W [Phi, (dPhi/dalpha_k c)_k]
// = Q R by the same multi-dot Householder sweep (K = n + q <= 16 columns in the workspace), Cov = chi^2 R^-1 R^-T,
// sigma_i = sqrt(chi^2) ||R^-T j_i|| with the UNweighted rows j_i (the math of vp_stats.hpp).
constexpr int MAXK = VP_MAX_BASIS + VP_MAX_PARAMS;
template <typename T> struct GenStatsArgs {
    GenArgs<T> g;       // model, t, w, alpha, ws, m, B, strides
    const T *C;         // [B][n]
    const double *cost; // [B]
    const int32_t *status_in;
    T *cov_out;         // [B][K*K]
    double *chi2_out;   // [B]
    T *sigma_out;       // [B][m] or null
    int32_t *status_out;
};

template <typename T> __global__ void __launch_bounds__(TB) gen_stats_kernel(const GenStatsArgs<T> sa) {
    const GenArgs<T> &a = sa.g;
    __shared__ GenShared<T> sh;
    __shared__ T Rk[MAXK][MAXK], Ri[MAXK][MAXK], cc[VP_MAX_BASIS], al[VP_MAX_PARAMS];
    __shared__ int s_ok;
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = a.mdl.n_params, K = n + q;
    T *ws = a.ws + (int64_t)blockIdx.x * a.ws_cols * m;
    auto col = [&](int c) { return ws + (int64_t)c * m; };
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        if (tid < n) cc[tid] = sa.C[b * n + tid];
        if (tid < q) al[tid] = a.alpha[b * q + tid];
        __syncthreads();
        const T *tp = a.t + b * a.t_stride;
        const T *wp = a.w ? a.w + b * a.w_stride : nullptr;
        // J rows (optionally weighted) into columns 0..K-1
        auto build = [&](bool weighted) {
            for (int i = tid; i < m; i += TB) {
                const T t = a.ext ? T(0) : tp[i], sc = (weighted && wp) ? wp[i] : T(1);
                T jr[MAXK];
                for (int k = 0; k < q; ++k) jr[n + k] = T(0);
                if (a.ext) {
                    const int er = a.ext_rows;
                    const bool in = i < er;
                    const T *ph = a.ext_phi + b * (int64_t)n * er;
                    const T *dp = a.ext_dphi + b * (int64_t)a.P * er;
                    for (int j = 0; j < n; ++j) jr[j] = in ? ph[(int64_t)j * er + i] : T(0);
                    for (int p = 0; p < a.P; ++p)
                        jr[n + a.pp[p]] = tfma(cc[a.pb[p]], in ? dp[(int64_t)p * er + i] : T(0), jr[n + a.pp[p]]);
                } else
                for (int j = 0; j < n; ++j) {
                    const int i0 = a.mdl.param[j][0], i1 = a.mdl.param[j][1];
                    T f, d0, d1;
                    basis_eval<T>(a.mdl.kind[j], t, i0 >= 0 ? al[i0] : T(0), i1 >= 0 ? al[i1] : T(0), f, d0, d1);
                    jr[j] = f;
                    if (i0 >= 0) jr[n + i0] = tfma(cc[j], d0, jr[n + i0]);
                    if (i1 >= 0) jr[n + i1] = tfma(cc[j], d1, jr[n + i1]);
                }
                for (int c = 0; c < K; ++c) col(c)[i] = jr[c] * sc;
            }
            __syncthreads();
        };
        build(true);
        // Householder QR of the K columns (R only)
        for (int k = 0; k < K; ++k) {
            const int nv = K - k;
            const T *ak = col(k);
            for (int v = 0; v < nv; ++v) {
                const T *cv = col(k + v);
                T acc = T(0);
                for (int i = k + tid; i < m; i += TB) acc = tfma(ak[i], cv[i], acc);
                reduce_put(sh, v, acc);
            }
            reduce_finish(sh, nv);
            if (tid == 0) {
                const T alpha = ak[k], nrm2 = sh.red[0];
                const bool live = nrm2 > num<T>::norm2_min && is_finite(nrm2);
                const T sigma = live ? tsqrt(nrm2) : T(0);
                const T beta = live ? -tcopysign(sigma, alpha) : ((nrm2 <= num<T>::norm2_min) ? alpha : nrm2);
                const T u = live ? alpha - beta : T(0);
                const T gk = live ? T(1) / (beta * u) : T(0);
                Rk[k][k] = beta;
                sh.f[0] = u;
                for (int v = 1; v < nv; ++v) {
                    const T top = col(k + v)[k];
                    const T f = gk * tfma(-beta, top, sh.red[v]);
                    sh.f[v] = f;
                    Rk[k][k + v] = tfma(f, u, top);
                }
            }
            __syncthreads();
            {
                T *akw = col(k);
                if (tid == 0) akw[k] = sh.f[0];
                __syncthreads();
                for (int i = k + tid; i < m; i += TB) {
                    const T x = akw[i];
                    for (int v = 1; v < nv; ++v) {
                        T *cj = col(k + v);
                        cj[i] = tfma(sh.f[v], x, cj[i]);
                    }
                }
            }
            __syncthreads();
        }
        const int dof = m - K;
        if (tid == 0) {
            bool ok = dof > 0 && sa.status_in[b] == VP_ST_OK;
            for (int i = 0; i < K; ++i) ok = ok && Rk[i][i] != T(0) && is_finite(Rk[i][i]);
            const T chi2 = ok ? (T)(2.0 * sa.cost[b] / (double)dof) : T(0) / T(0);
            for (int i = 0; i < K; ++i)
                for (int j = 0; j < K; ++j) Ri[i][j] = T(0);
            if (ok)
                for (int j = 0; j < K; ++j)
                    for (int i = j; i >= 0; --i) {
                        T acc = (i == j) ? T(1) : T(0);
                        for (int l = i + 1; l <= j; ++l) acc = tfma(-Rk[i][l], Ri[l][j], acc);
                        Ri[i][j] = acc / Rk[i][i];
                    }
            const T nanv = T(0) / T(0);
            for (int bi = 0; bi < K; ++bi)
                for (int ai = 0; ai < K; ++ai) {
                    T val = T(0);
                    for (int l = (ai > bi ? ai : bi); l < K; ++l) val = tfma(Ri[ai][l], Ri[bi][l], val);
                    sa.cov_out[b * (K * K) + bi * K + ai] = ok ? val * chi2 : nanv;
                }
            sa.chi2_out[b] = (double)chi2;
            sa.status_out[b] = ok ? VP_ST_OK : 4 /* VP_ST_STATS_FAILED */;
            s_ok = ok ? 1 : 0;
            sh.fn2 = chi2;
        }
        __syncthreads();
        if (sa.sigma_out) {
            build(false);
            const T s0 = s_ok ? tsqrt(sh.fn2) : T(0) / T(0);
            for (int i = tid; i < m; i += TB) {
                T acc = T(0);
                for (int c = 0; c < K; ++c) {
                    T v = T(0);
                    for (int r = 0; r <= c; ++r) v = tfma(Ri[r][c], col(r)[i], v);
                    acc = tfma(v, v, acc);
                }
                sa.sigma_out[b * (int64_t)m + i] = s0 * tsqrt(acc);
            }
        }
        __syncthreads();
    }
}

// ---- host-side launchers (type-erased LaunchParams, vp_kernels.hpp) -------------------------------------------------
template <typename T> inline bool fill_args(const LaunchParams &p, GenArgs<T> &a) {
    std::memset(&a, 0, sizeof(a));
    a.mdl = *p.model;
    int np = 0;
    if (p.ext) { // caller-evaluated model: the pair table comes with the handle, there is no argument slot
        if (p.ext_np > VP_MAX_PAIRS || !p.ext_phi) return false;
        for (; np < p.ext_np; ++np) {
            a.pb[np] = p.ext_pb[np];
            a.pa[np] = 0;
            a.pp[np] = p.ext_pp[np];
        }
        a.ext = 1;
        a.ext_phi = (const T *)p.ext_phi;
        a.ext_dphi = (const T *)p.ext_dphi;
        a.ext_rows = p.ext_rows > 0 ? p.ext_rows : p.m;
    } else
    for (int j = 0; j < p.model->n_basis; ++j)
        for (int k = 0; k < VP_MAX_BASIS_PARAMS; ++k)
            if (p.model->param[j][k] >= 0) {
                if (np >= VP_MAX_PAIRS) return false;
                a.pb[np] = j;
                a.pa[np] = k;
                a.pp[np] = p.model->param[j][k];
                ++np;
            }
    a.P = np;
    a.t = (const T *)p.t;
    a.w = (const T *)p.w;
    a.yw = (const T *)p.yw;
    a.S = p.S > 0 ? p.S : 1;
    a.alpha = (const T *)p.alpha;
    a.alpha_io = (T *)p.alpha_out;
    a.r_out = (T *)p.r_out;
    a.J_out = (T *)p.J_out;
    a.C_out = (T *)p.C_out;
    a.cost_out = p.cost_out;
    a.status = p.status;
    a.report = p.report;
    a.Phi_out = (T *)p.Phi_out;
    a.dPhi_out = (T *)p.dPhi_out;
    a.ws = (T *)p.gen_ws;
    a.ws_cols = p.model->n_basis + 1 + np + p.model->n_params;
    a.m = p.m;
    a.B = p.B;
    a.t_stride = p.t_stride;
    a.w_stride = p.w_stride;
    a.eps = (T)p.eps;
    if (p.opts) {
        a.lm.ftol = (T)p.opts->ftol;
        a.lm.xtol = (T)p.opts->xtol;
        a.lm.gtol = (T)p.opts->gtol;
        a.lm.stepbound = (T)p.opts->stepbound;
        a.lm.patience = p.opts->patience;
        a.lm.scale_diag = p.opts->scale_diag;
    }
    a.trace = p.trace;
    a.trace_rows = p.trace_rows;
    a.list = p.gen_list;
    a.list_slot = p.gen_list_slot;
    a.list_first = p.gen_list_first;
    a.scale_cols = p.gen_scale_cols;
    return true;
}

// dynamic LDS of a launch: the columns of one problem when they fit (GenArgs::ws_lds), else 0
template <typename T, class K> inline size_t gen_lds_for(GenArgs<T> &a, K kernel) {
    const size_t need = (size_t)a.ws_cols * (size_t)a.m * sizeof(T);
    a.ws_lds = 0;
    if (need == 0 || need > kGenLdsMax) return 0;
    // (the attribute is set once per kernel: a small table of the kernels seen -- the pointer TYPE is the same for all of them,
    // a function-local static would be shared)
    static const void *seen[16];
    static bool seen_ok[16];
    static int nseen = 0;
    static std::mutex mtx; // (handles may be driven from several host threads)
    std::lock_guard<std::mutex> lock(mtx);
    bool attr_ok = false, found = false;
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == (const void *)kernel) {
            attr_ok = seen_ok[i];
            found = true;
        }
    if (!found) {
        attr_ok = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGenLdsMax) == hipSuccess;
        if (!attr_ok) (void)hipGetLastError();
        if (nseen < 16) {
            seen[nseen] = (const void *)kernel;
            seen_ok[nseen] = attr_ok;
            ++nseen;
        }
    }
    if (!attr_ok && need > 48 * 1024) return 0;
    a.ws_lds = 1;
    return need;
}
template <typename T> int launch_evaluate(const LaunchParams &p) {
    GenArgs<T> a;
    if (!fill_args(p, a) || !p.gen_ws) return VP_ERR_UNSUPPORTED;
    if (a.B <= 0) return VP_ERR_OK;
    const size_t lds = gen_lds_for<T>(a, &gen_evaluate_kernel<T>);
    hipLaunchKernelGGL((gen_evaluate_kernel<T>), dim3((unsigned)p.gen_blocks), dim3(TB), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}
template <typename T> int launch_fit(const LaunchParams &p) {
    GenArgs<T> a;
    if (!fill_args(p, a) || !p.gen_ws) return VP_ERR_UNSUPPORTED;
    if (a.B <= 0) return VP_ERR_OK;
    a.scale_cols = 1; // (this kernel never flags a problem for a re-fit: it IS the re-fit)
    const size_t lds = gen_lds_for<T>(a, &gen_fit_kernel<T>);
    hipLaunchKernelGGL((gen_fit_kernel<T>), dim3((unsigned)p.gen_blocks), dim3(TB), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}
template <typename T> size_t mrhs_lm_state_bytes() { return sizeof(LmAny<T>); }
template <typename T> int launch_mrhs_fit(const LaunchParams &p) {
    GenArgs<T> a;
    if (!fill_args(p, a) || !p.gen_ws) return VP_ERR_UNSUPPORTED;
    if (a.B <= 0) return VP_ERR_OK;
    a.phase = p.gen_phase;
    a.lm_state = p.gen_lm_state;
    a.acc = p.gen_acc;
    a.nactive = p.gen_nactive;
    a.S_global = p.mrhs_S_global;
    if (a.phase != 0 && (!a.lm_state || !a.acc || !a.nactive)) return VP_ERR_INVALID;
    const size_t lds = gen_lds_for<T>(a, &gen_mrhs_fit_kernel<T>);
    hipLaunchKernelGGL((gen_mrhs_fit_kernel<T>), dim3((unsigned)p.gen_blocks), dim3(TB), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}
template <typename T> int launch_basis(const LaunchParams &p) {
    GenArgs<T> a;
    if (!fill_args(p, a)) return VP_ERR_UNSUPPORTED;
    a.skip_invariant = (p.basis_flags & VP_BASIS_SKIP_INVARIANT) ? 1 : 0;
    int ncols = 0;
    for (int j = 0; j < p.model->n_basis; ++j)
        if (!(a.skip_invariant && p.model->kind[j] == VP_BASIS_CONST)) ++ncols;
    a.n_phi_cols = ncols;
    if (a.B <= 0) return VP_ERR_OK;
    const unsigned grid = (unsigned)(a.B < 4096 ? a.B : 4096);
    hipLaunchKernelGGL((gen_basis_kernel<T>), dim3(grid), dim3(TB), 0, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}
template <typename T> int launch_best_fit(const LaunchParams &p) {
    GenArgs<T> a;
    if (!fill_args(p, a)) return VP_ERR_UNSUPPORTED;
    a.C_in = (const T *)p.C_out;
    if (a.B <= 0) return VP_ERR_OK;
    const int64_t nprob = a.B * p.S;
    const unsigned grid = (unsigned)(nprob < 4096 ? nprob : 4096);
    hipLaunchKernelGGL((gen_best_fit_kernel<T>), dim3(grid), dim3(TB), 0, p.stream, a, (int)p.S);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

template <typename T> int launch_stats(const LaunchParams &p) {
    GenStatsArgs<T> sa;
    if (!fill_args(p, sa.g) || !p.gen_ws) return VP_ERR_UNSUPPORTED;
    sa.C = (const T *)p.C_out;
    sa.cost = p.cost_out;
    sa.status_in = p.status;
    sa.cov_out = (T *)p.Phi_out;
    sa.chi2_out = (double *)p.dPhi_out;
    sa.sigma_out = (T *)p.r_out;
    sa.status_out = (int32_t *)p.J_out;
    if (sa.g.B <= 0) return VP_ERR_OK;
    hipLaunchKernelGGL((gen_stats_kernel<T>), dim3((unsigned)p.gen_blocks), dim3(TB), 0, p.stream, sa);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace gen
} // namespace vp
