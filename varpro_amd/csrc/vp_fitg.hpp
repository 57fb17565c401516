// vp_fitg.hpp -- Levenberg-Marquardt fit of fp32 problems on the fp64 GRAM matrix of [Phi | y | dPhi]
// (BASELINE.json configs[4]: five exponentials + offset, m = 4096, fp32 -- "wider Jacobian / Phi^T Phi path").
//
// == LevMarSolver::fit -> LevenbergMarquardt::minimize (src/solvers/levmar/mod.rs:238-254) for a batch of fp32
// problems, with the linear algebra of one evaluation (src/solvers/levmar/mod.rs:42-73, 101-201: thin decomposition of
// Phi_w, coefficients, projected residual, Kaufman Jacobian) restated on the normal equations IN DOUBLE:
//
//     X = [e_1 .. e_NE | y | d_1 .. d_NE]   (fp32 data and grid, columns evaluated and multiplied in fp64),   const column implicit
//     ONE pass over the m rows accumulates the NX(NX+1)/2 inner products and NX column sums per lane (fp64 FMAs: the
//     product of two fp32 values is exact in fp64), ONE packed wave reduction delivers them -- no column is ever
//     resident, no per-reflector reduction round, no multi-wave group;
//     A = Phi^T Phi = L L^T,  z = L^-1 Phi^T y,  c = L^-T z,  ||r||^2 = y^T y - z^T z,
//     W = L^-1 Phi^T D,  D^T P_perp D = D^T D - W^T W,  D^T r = D^T y - W^T z,
//     J^T J = diag(c) (D^T P_perp D) diag(c),  J^T r = -c_k (D^T r)_k   (Kaufman, pair p = (basis p, parameter p)),
//     pivoted Cholesky of J^T J -> (R_J, acnorm, ipvt, qtf) exactly as the multiple-right-hand-side path (gram_to_qr).
//
// Why this is legitimate for fp32 and only for fp32: the Householder path in fp32 loses kappa(Phi)*eps32 (6e-8) of the
// coefficients -- at cfg4's kappa ~ 1e6 that is 6 %, and 15 % of the fits end non-finite; the Gram matrix in fp64 loses
// kappa^2*eps64 = 1e12*1e-16 = 1e-4.  For fp64 data the same trick would square the conditioning with nothing in
// reserve, so fp64 handles never come here.  Rank-deficient Phi at a trial point (two decay times collide, a column
// degenerates into the constant): columns whose Cholesky pivot vanishes are dropped, the counterpart of the reference's
// truncated SVD (gram_phase).
//
// Execution: the persistent-slot machinery of vp_fit2.hpp with W = 1 -- a wave owns GS slots, the VECTOR phase of a
// slot is the streaming Gram pass (y re-read from HBM/L2 per evaluation, coalesced 16 B per lane; the grid sits in LDS),
// then lane s turns slot s's Gram into the evaluation results (gram_phase) and runs the LM bookkeeping
// (slot_scalar_phase<double>), all in fp64; results are stored as fp32.
#pragma once
#include "vp_fit2.hpp"
#include "vp_lm_core.hpp"

namespace vp {

template <int NE> struct GramIdx {
    static constexpr int NX = 2 * NE + 1;        // e_0..e_{NE-1}, y (index NE), d_0..d_{NE-1} (index NE+1+k)
    static constexpr int NP = NX * (NX + 1) / 2; // products a <= b, row-major upper triangle
    static constexpr int NV = NP + NX;           // + column sums (products with the implicit constant column)
    __host__ __device__ static constexpr int prod(int a, int b) {
        return a <= b ? a * NX - a * (a - 1) / 2 + (b - a) : b * NX - b * (b - 1) / 2 + (a - b);
    }
    __host__ __device__ static constexpr int sum(int a) { return NP + a; }
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

template <class M> struct FitgArgs {
    const float *t;   // [m] shared grid
    const float *yw;  // [B][m]
    float *alpha;     // [B][q] in: guesses, out: parameters
    float *C_out;     // [B][n] or null
    double *cost_out;
    int32_t *status;
    vp_report *report;
    double *trace;
    int *queue;
    int64_t B;
    int m, mp;        // rows, rows padded to a multiple of 256
    int trace_rows, scale_diag, patience, grid_uniform;
    int gs_used;      // slots per wave taken by the static first assignment (<= GS)
    double eps, ftol, xtol, gtol, stepbound;
};

// (re)fill a slot: only the LM record -- no data is staged
template <int N, int Q>
__device__ __noinline__ void slotg_fill(VP_LDS SlotRec<double, N, Q> *rec, VP_LDS const SlotConsts<double, float> *k, const int prob) {
    if (lane_id() != 0) return;
    if (prob < 0) {
        rec->prob = -1;
        rec->term = VP_TERM_NOT_RUN;
        return;
    }
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
}

// Lane s: Gram of slot s -> results of the evaluation in the slot's record (what the vector phase of fit2_kernel posts).
//   gram: [GS][NV] Grams of the slots (partial 0 of the [W][GS][NV] area, already totalled over the waves)
template <int NE, int GS, int W>
__device__ __noinline__ void gram_phase(VP_LDS SlotRec<double, NE + 1, NE> *recs, VP_LDS double *gram,
                                        VP_LDS const SlotConsts<double, float> *k) {
    constexpr int N = NE + 1, Q = NE;
    using GI = GramIdx<NE>;
    const int lane = lane_id();
    if (!(lane < GS && recs[lane].prob >= 0)) return;
    VP_LDS SlotRec<double, N, Q> *rec = recs + lane;
    VP_LDS double *g = gram + (size_t)lane * GI::NV;
    // (the W per-wave partial Grams were totalled into partial 0 by the whole group before this call)
    const double eps = k->eps;
    // ---- A = Phi^T Phi (basis order e_0..e_{NE-1}, const) = L L^T ----
    // A column whose pivot d_i (its squared distance from the span of the columns before it) is <= max(eps^2,
    // 1e-13 A_ii) is DROPPED (c_i = 0, the projector is that of the remaining columns): the counterpart of the
    // reference's truncated SVD (singular values <= eps) at trial points where two decay times collide or a column
    // degenerates into the constant -- the step is then judged by its residual like any other instead of ending the fit.
    double Lm[N][N], iL[N]; // L (strict lower part) and the reciprocals of its diagonal (0 for a dropped column)
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double aij;
            if (i == NE) aij = (j == NE) ? (double)k->m : g[GI::sum(j)];
            else aij = g[GI::prod(j, i)];
            double acc = aij;
#pragma unroll
            for (int p = 0; p < j; ++p) acc = tfma(-Lm[i][p], Lm[j][p], acc);
            if (i == j) {
                ok = ok && is_finite(acc);
                const bool keep = acc > tmax(eps * eps, 1.0e-13 * aij);
                iL[i] = keep ? 1.0 / tsqrt(acc) : 0.0;
            } else {
                Lm[i][j] = acc * iL[j]; // (a dropped column j has no sub-diagonal entries)
            }
        }
    }
    // ---- z = L^-1 Phi^T y, c = L^-T z, ||r||^2 ----
    double z[N], c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double acc = (i == NE) ? g[GI::sum(NE)] : g[GI::prod(i, NE)];
#pragma unroll
        for (int p = 0; p < i; ++p) acc = tfma(-Lm[i][p], z[p], acc);
        z[i] = acc * iL[i];
    }
    double fn2 = g[GI::prod(NE, NE)];
#pragma unroll
    for (int i = 0; i < N; ++i) fn2 = tfma(-z[i], z[i], fn2);
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double acc = z[i];
#pragma unroll
        for (int p = i + 1; p < N; ++p) acc = tfma(-Lm[p][i], c[p], acc);
        c[i] = acc * iL[i];
        ok = ok && is_finite(c[i]);
    }
    ok = ok && is_finite(fn2);
    const double fnorm1 = tsqrt(tmax(fn2, 0.0));

    const int fl_in = rec->flags;
    const bool first = (fl_in & 1) != 0;
    const double fnorm = rec->fnorm, prered = rec->prered;
    double actred = 0.0, ratio = 0.0;
    bool good = false;
    if (!first) {
        const double q1 = fnorm1 / fnorm;
        actred = (fnorm1 * 0.1 < fnorm) ? 1.0 - q1 * q1 : -1.0;
        ratio = (prered == 0.0) ? 0.0 : actred / prered;
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
    if (need_jac) {
        // ---- W = L^-1 Phi^T D;  G2 = D^T P_perp D;  v = D^T r ----
        double Wm[N][Q];
#pragma unroll
        for (int kk = 0; kk < Q; ++kk)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double acc = (i == NE) ? g[GI::sum(NE + 1 + kk)] : g[GI::prod(i, NE + 1 + kk)];
#pragma unroll
                for (int p = 0; p < i; ++p) acc = tfma(-Lm[i][p], Wm[p][kk], acc);
                Wm[i][kk] = acc * iL[i];
            }
        double Aj[Q][Q], bv[Q];
#pragma unroll
        for (int kk = 0; kk < Q; ++kk) {
            double vk = g[GI::prod(NE, NE + 1 + kk)];
#pragma unroll
            for (int i = 0; i < N; ++i) vk = tfma(-Wm[i][kk], z[i], vk);
            bv[kk] = -c[kk] * vk;
#pragma unroll
            for (int l = kk; l < Q; ++l) {
                double gkl = g[GI::prod(NE + 1 + kk, NE + 1 + l)];
#pragma unroll
                for (int i = 0; i < N; ++i) gkl = tfma(-Wm[i][kk], Wm[i][l], gkl);
                const double a = c[kk] * c[l] * gkl;
                Aj[kk][l] = a;
                Aj[l][kk] = a;
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
    }
    if (good) fl |= 32;
    rec->flags = fl;
}

// One workgroup = one GROUP of W waves that owns GS slots: the waves stream interleaved 256-row chunks of a slot's rows
// (the latency of one evaluation is what bounds a launch once the queue is empty), wave 0 runs the lane-parallel phases.
template <class M, int GS, int W>
__global__ void __launch_bounds__(64 * W, (2 * W) / 4 > 0 ? (2 * W) / 4 : 1) fitg_kernel(const FitgArgs<M> a) {
    constexpr int N = M::N, Q = M::Q, NE = M::N - 1;
    static_assert(M::kStatic && M::kConstLast && M::kDiagonalPairs && M::Q == NE, "exponentials + offset");
    using GI = GramIdx<NE>;
    using Rec = SlotRec<double, N, Q>;
    using KC = SlotConsts<double, float>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *s_t = reinterpret_cast<float *>(smem_raw);                                   // [mp]
    double *gram = reinterpret_cast<double *>(smem_raw + (size_t)a.mp * sizeof(float)); // [W][GS][NV]
    Rec *recs = reinterpret_cast<Rec *>(gram + (size_t)W * GS * GI::NV);
    KC *kc = reinterpret_cast<KC *>(recs + GS);
    double *s_u = reinterpret_cast<double *>(kc + 1); // [W][4*NE] per-wave column constants of the running Gram pass
    int *s_pop = reinterpret_cast<int *>(s_u + (size_t)W * 4 * NE); // [GS] queue pops
    const int lane = lane_id();
    const int wv = (int)(threadIdx.x >> 6);
    const int gw = (int)blockIdx.x; // persistent group index
    const int m = a.m;
    for (int i = threadIdx.x; i < a.mp; i += blockDim.x) s_t[i] = (i < m) ? a.t[i] : 0.0f;
    if (threadIdx.x == 0) {
        kc->ftol = a.ftol;
        kc->xtol = a.xtol;
        kc->gtol = a.gtol;
        kc->stepbound = a.stepbound;
        kc->alpha = a.alpha;
        kc->C_out = a.C_out;
        kc->cost_out = a.cost_out;
        kc->status = a.status;
        kc->report = a.report;
        kc->trace = a.trace;
        kc->yw = a.yw;
        kc->queue = a.queue;
        kc->B = a.B;
        kc->trace_rows = a.trace_rows;
        kc->scale_diag = a.scale_diag;
        kc->max_fev = a.patience * (Q + 1);
        kc->m = a.m;
        kc->eps = a.eps;
    }
    __syncthreads();
    auto wave_sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const bool uniform = a.grid_uniform != 0 && m >= 3;
    const double t0 = (double)s_t[0];
    const double dt = uniform ? ((double)s_t[m - 1] - t0) / (double)(m - 1) : 0.0;
    const int nchunk = a.mp / 256;

    int nactive = 0;
#pragma nounroll
    for (int s = 0; s < GS; ++s) {
        const int64_t prob = (int64_t)gw * a.gs_used + s;
        const bool have = s < a.gs_used && prob < a.B;
        if (wv == 0) slotg_fill<N, Q>((VP_LDS Rec *)(recs + s), (VP_LDS const KC *)kc, have ? (int)prob : -1);
        nactive += have ? 1 : 0;
    }
    __syncthreads();

#ifdef VP_FITG_CLOCKS
    long long ck[4] = {0, 0, 0, 0};
    long long c0 = 0;
#define VP_CK(i)                                                                                                       \
    do {                                                                                                               \
        const long long c1 = __builtin_amdgcn_s_memtime();                                                             \
        ck[i] += c1 - c0;                                                                                              \
        c0 = c1;                                                                                                       \
    } while (0)
    c0 = __builtin_amdgcn_s_memtime();
#else
#define VP_CK(i)
#endif
    while (nactive > 0) {
        // ================= VECTOR phase: the Gram pass of every occupied slot =================
#pragma nounroll
        for (int s = 0; s < GS; ++s) {
            Rec *rec = recs + s;
            const int prob = uni(rec->prob);
            if (prob < 0) continue;
            // wave-uniform per-column constants live in LDS (su: 1/tau, 1/tau^2, ratio per row, ratio per chunk step), not
            // in registers: the 77 fp64 accumulators own the register file
            VP_LDS double *su = (VP_LDS double *)(s_u + (size_t)wv * 4 * NE);
            double fa[NE];
            {
                double rt[NE];
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) rt[kx] = 1.0 / rec->xt[kx];
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) {
                    // uniform grid: exp(-t/tau) of a lane's rows by recurrence (anchor at the first row of the wave's first
                    // chunk, ratio per row, ratio per W chunks)
                    fa[kx] = uniform ? texp(-(t0 + (double)(256 * wv + 4 * lane) * dt) * rt[kx]) : 0.0;
                    const double q1 = uniform ? texp(-dt * rt[kx]) : 0.0, qc = uniform ? texp(-((256.0 * W) * dt) * rt[kx]) : 0.0;
                    if (lane == 0) {
                        su[kx] = rt[kx];
                        su[NE + kx] = rt[kx] * rt[kx];
                        su[2 * NE + kx] = q1;
                        su[3 * NE + kx] = qc;
                    }
                }
            }
            wave_sync();
            const float *yp = a.yw + (int64_t)prob * m;
            const bool yvec = (reinterpret_cast<uintptr_t>(yp) & 15) == 0;
            double acc[GI::NV];
#pragma unroll
            for (int i = 0; i < GI::NV; ++i) acc[i] = 0.0;
#pragma nounroll
            for (int ch = wv; ch < nchunk; ch += W) {
                const int row0 = ch * 256 + 4 * lane;
                const float4 t4 = *reinterpret_cast<const float4 *>(s_t + row0);
                float4 y4;
                if (yvec && row0 + 3 < m) {
                    y4 = *reinterpret_cast<const float4 *>(yp + row0);
                } else {
                    y4.x = (row0 < m) ? yp[row0] : 0.0f;
                    y4.y = (row0 + 1 < m) ? yp[row0 + 1] : 0.0f;
                    y4.z = (row0 + 2 < m) ? yp[row0 + 2] : 0.0f;
                    y4.w = (row0 + 3 < m) ? yp[row0 + 3] : 0.0f;
                }
                const float tv[4] = {t4.x, t4.y, t4.z, t4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
                double f[NE];
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) f[kx] = fa[kx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool valid = row0 + e < m;
                    const double td = (double)tv[e];
                    double X[GI::NX];
                    if (uniform) {
#pragma unroll
                        for (int kx = 0; kx < NE; ++kx) {
                            X[kx] = f[kx];
                            f[kx] *= su[2 * NE + kx];
                        }
                    } else { // general grid: one fp32 exponential per element (the accuracy class of the data)
#pragma unroll
                        for (int kx = 0; kx < NE; ++kx) X[kx] = (double)texp(-(tv[e] * (float)su[kx]));
                    }
#pragma unroll
                    for (int kx = 0; kx < NE; ++kx) {
                        X[kx] = valid ? X[kx] : 0.0;
                        X[NE + 1 + kx] = X[kx] * (td * su[NE + kx]); // d/dtau exp(-t/tau) = exp(-t/tau) t / tau^2
                    }
                    X[NE] = valid ? (double)yv[e] : 0.0;
                    int idx = 0;
#pragma unroll
                    for (int p = 0; p < GI::NX; ++p)
#pragma unroll
                        for (int q = p; q < GI::NX; ++q) {
                            acc[idx] = tfma(X[p], X[q], acc[idx]);
                            ++idx;
                        }
#pragma unroll
                    for (int p = 0; p < GI::NX; ++p) acc[GI::NP + p] += X[p];
                }
#pragma unroll
                for (int kx = 0; kx < NE; ++kx) fa[kx] *= su[3 * NE + kx];
            }
            wave_reduce_store<GI::NV>(acc, (VP_LDS double *)(gram + ((size_t)wv * GS + s) * GI::NV));
        }
        __syncthreads();
        // total the W per-wave partial Grams in place (fixed order; all threads: one LDS round trip instead of 77 serial
        // ones on the lanes of wave 0)
        if constexpr (W > 1) {
            for (int i = threadIdx.x; i < GS * GI::NV; i += 64 * W) {
                double t = gram[i];
#pragma unroll
                for (int w = 1; w < W; ++w) t += gram[(size_t)w * GS * GI::NV + i];
                gram[i] = t;
            }
            __syncthreads();
        }
        VP_CK(0);
        // ===== wave 0, lane s: Gram -> evaluation results, then the LM bookkeeping of slot s; queue pops for finished slots =====
        if (wv == 0) {
            gram_phase<NE, GS, W>((VP_LDS Rec *)recs, (VP_LDS double *)gram, (VP_LDS const KC *)kc);
            wave_sync();
            VP_CK(1);
            slot_scalar_phase<double, N, Q, GS, float>((VP_LDS Rec *)recs, (VP_LDS const KC *)kc);
            wave_sync();
            VP_CK(2);
            if (lane == 0) {
                for (int s = 0; s < GS; ++s)
                    s_pop[s] = (recs[s].prob >= 0 && recs[s].term != 0)
                                   ? __hip_atomic_fetch_add(kc->queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : -1;
            }
        }
        __syncthreads();
        // ================= refill finished slots =================
        bool refilled = false;
#pragma nounroll
        for (int s = 0; s < GS; ++s) {
            if (uni(recs[s].prob) < 0 || uni(recs[s].term) == 0) continue;
            const int next = uni(s_pop[s]);
            const bool have = (int64_t)next < a.B;
            refilled = true;
            if (!have) nactive -= 1;
        }
        if (refilled) {
            __syncthreads(); // every wave has read the finished records
            if (wv == 0) {
#pragma nounroll
                for (int s = 0; s < GS; ++s) {
                    if (uni(recs[s].prob) < 0 || uni(recs[s].term) == 0) continue;
                    const int next = uni(s_pop[s]);
                    slotg_fill<N, Q>((VP_LDS Rec *)(recs + s), (VP_LDS const KC *)kc, (int64_t)next < a.B ? next : -1);
                }
            }
            __syncthreads();
        }
        VP_CK(3);
    }
#ifdef VP_FITG_CLOCKS
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) {
        double *tr = a.trace + (size_t)(a.trace_rows - 1) * (Q + 4);
        for (int i = 0; i < 4; ++i) tr[i] = (double)ck[i];
    }
#endif
}

// fp32 handle, unit weights, one shared grid, single right-hand side: the Gram kernel; everything else: `fallback`
template <class M> int launch_fitg(const LaunchParams &p, int (*fallback)(const LaunchParams &)) {
    constexpr int W = 4, GS = 8, NE = M::N - 1;
    using GI = GramIdx<NE>;
    const int mp = ((p.m + 255) / 256) * 256;
    const size_t lds = (size_t)mp * sizeof(float) + (size_t)GS * (W * GI::NV * sizeof(double) + sizeof(SlotRec<double, M::N, M::Q>)) +
                       sizeof(SlotConsts<double, float>) + (size_t)W * 4 * NE * sizeof(double) + (size_t)GS * sizeof(int) + 16;
    if (p.w || p.t_stride != 0 || !p.queue || p.fit_group == 1 || p.S != 1 || lds > 64 * 1024) return fallback(p);
    FitgArgs<M> a;
    a.t = (const float *)p.t;
    a.yw = (const float *)p.yw;
    a.alpha = (float *)p.alpha_out;
    a.C_out = (float *)p.C_out;
    a.cost_out = p.cost_out;
    a.status = p.status;
    a.report = p.report;
    a.trace = p.trace;
    a.queue = p.queue;
    a.B = p.B;
    a.m = p.m;
    a.mp = mp;
    a.trace_rows = p.trace_rows;
    a.scale_diag = p.opts->scale_diag;
    a.patience = p.opts->patience;
    a.grid_uniform = p.grid_uniform;
    a.eps = p.eps;
    a.ftol = p.opts->ftol;
    a.xtol = p.opts->xtol;
    a.gtol = p.opts->gtol;
    a.stepbound = p.opts->stepbound;
    if (a.B <= 0) return VP_ERR_OK;
    // persistent grid: 2 groups of W waves per CU; the static first assignment spreads the batch over all groups
    const int64_t cap_groups = (int64_t)p.num_cus * 2;
    int64_t gs_used = (a.B + cap_groups - 1) / cap_groups;
    if (gs_used > GS) gs_used = GS;
    if (gs_used < 1) gs_used = 1;
    a.gs_used = (int)gs_used;
    int64_t blocks = (a.B + gs_used - 1) / gs_used;
    if (blocks > cap_groups) blocks = cap_groups;
    if (hipMemsetD32Async((hipDeviceptr_t)p.queue, (int)(blocks * gs_used), 1, p.stream) != hipSuccess) return VP_ERR_HIP;
    hipLaunchKernelGGL((fitg_kernel<M, GS, W>), dim3((unsigned)blocks), dim3(64 * W), lds, p.stream, a);
    return hipGetLastError() == hipSuccess ? VP_ERR_OK : VP_ERR_HIP;
}

} // namespace vp
