// vp_gen_extfit.hpp -- the batched reverse-communication LM fit of caller-evaluated models (vp_extfit.hpp) for EVERY shape
// and any number of right-hand sides: the evaluation launch of a step on the generic kernels' machinery.
//
// == LevMarSolver::fit<Rhs> (src/solvers/levmar/mod.rs:238-254) over any SeparableNonlinearModel (src/model/mod.rs:239-363):
// the reference takes every model shape and both right-hand-side kinds (src/problem/builder.rs:194-225).  The specialised
// step kernels (ext_fit_eval_kernel: columns in registers; blk::ext_fit_stream_eval_kernel: rows streamed) exist for a table
// of compile-time (n, pairs, q) and for single right-hand sides; round 5 answered VP_ERR_UNSUPPORTED for everything else.
// This kernel is compiled ONCE per scalar type and takes what vp_batch_create_external admits -- n <= VP_MAX_BASIS,
// q <= VP_MAX_PARAMS, pairs <= VP_MAX_PAIRS, S >= 1 -- with run-time loop bounds: one 256-thread workgroup per active
// problem, the caller's columns copied (weighted) into a global-memory workspace slot, the Householder sweep / truncated
// solve of gen::evaluate, and into the candidate slot of the problem's LM record
//   S == 1   MINPACK's pivoted qrfac of the explicit Kaufman columns with Q_J^T r alongside (gen::jac_qrfac)     :101-201
//   S  > 1   the stacked residual's ||r||^2, J^T J, J^T r summed over the columns in double -- each column gets its own
//            projection, its Kaufman columns stay in Q-coordinates -- and the pivoted Cholesky of J^T J (gram_to_qr), as the
//            global fit of descriptor models does (gen_mrhs_fit_kernel, vp_mrhs.hpp)                             :172-186
// followed by the SAME lane-per-problem LM kernel (ext_fit_lm_kernel<T, Q>).  Slower than the specialised kernels (the
// columns travel through L2 / HBM once per reflector); never a CPU path.
#pragma once
#include "vp_extfit.hpp"
#include "vp_generic.hpp"

namespace vp {
namespace gen {

template <typename T> struct GenExtFitArgs {
    GenArgs<T> g;    // ext = 1: ext_phi / ext_dphi / pair table / w / yw / eps / ws
    void *state;     // the LM records (ExtFitLayout<q>)
    T *C_trial;      // [B][S][n] (S > 1) or null
    int init;        // first step: every problem is evaluated with derivative columns
    int q;
    const int32_t *active_in;    // the compacted active set of the previous step (ExtFitArgs), or null
    const int32_t *active_count;
};

template <typename T> __global__ void __launch_bounds__(TB) gen_extfit_eval_kernel(const GenExtFitArgs<T> xa) {
    __shared__ GenShared<T> sh;
    __shared__ double s_acc[2 + VP_MAX_PARAMS * VP_MAX_PARAMS + VP_MAX_PARAMS];
    __shared__ int s_okall;
    const GenArgs<T> &a = xa.g;
    const int tid = (int)threadIdx.x, m = a.m, n = a.mdl.n_basis, q = xa.q, P = a.P, NS = a.S;
    const int NCQ = n + 1 + P;
    const int64_t B = a.B;
    const ext::ExtFitOffsets F = ext::extfit_offsets(q);
    T *st = reinterpret_cast<T *>(xa.state);
    int32_t *si = ext::extfit_ints_rt<T>(xa.state, B, F.NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_dyn_lds[];
    T *ws = gen_workspace<T>(a, gen_dyn_lds);
    auto col = [&](int c) { return ws + (int64_t)c * m; };
    const int64_t count = xa.active_in ? (int64_t)*xa.active_count : B;
    for (int64_t bi = blockIdx.x; bi < count; bi += gridDim.x) {
        const int64_t b = xa.active_in ? (int64_t)xa.active_in[bi] : bi;
        int want = ext::EXTFIT_WANT_BASIS | ext::EXTFIT_WANT_DERIVS;
        if (!xa.init) {
            if (si[F.TERM * B + b] != 0) continue; // (uniform: finished in an earlier step)
            want = si[F.WANT * B + b];
        }
        const bool with_d = (want & ext::EXTFIT_WANT_DERIVS) != 0 && a.ext_dphi != nullptr && P > 0;
        if (NS == 1) {
            // (without derivative columns the copy loop of evaluate() zero-fills them: a.ext_dphi == nullptr is handled there;
            // a problem that does not want them simply ignores the Jacobian part)
            evaluate<T>(a, sh, ws, b, with_d, b);
            const bool ok = sh.ok != 0;
            if (with_d && ok) jac_qrfac<T>(a, sh, ws);
            if (tid == 0) {
                st[F.C_FN * B + b] = tsqrt(sh.fn2);
                for (int k = 0; k < n; ++k) st[(F.C_C + k) * B + b] = sh.c[k];
                si[F.C_OK * B + b] = ok ? 1 : 0;
                si[F.C_HASJ * B + b] = (with_d && ok) ? 1 : 0;
                if (with_d && ok)
                    for (int k = 0; k < q; ++k) {
                        st[(F.C_ACN + k) * B + b] = sh.acnorm[k];
                        st[(F.C_QTF + k) * B + b] = sh.qtf[k];
                        si[(F.C_IPVT + k) * B + b] = sh.ipvt[k];
                        for (int l = 0; l < q; ++l) st[(F.C_RJ + k * q + l) * B + b] = sh.Rj[k][l];
                    }
            }
            __syncthreads();
            continue;
        }
        // ---- several right-hand sides: sums over the columns (the stacked residual and its Jacobian) ----
        if (tid == 0) {
            for (int i = 0; i < gen_nacc(q); ++i) s_acc[i] = 0.0;
            s_okall = 1;
        }
        __syncthreads();
        for (int s = 0; s < NS; ++s) {
            evaluate<T>(a, sh, ws, b, false, b * NS + s);
            if (tid < n) xa.C_trial[(b * NS + s) * n + tid] = sh.c[tid];
            if (with_d) {
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
            }
            if (tid == 0) {
                s_acc[0] += (double)sh.fn2;
                if (!sh.ok) s_okall = 0;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const T fnorm1 = tsqrt((T)s_acc[0]);
            const bool ok = s_okall != 0 && is_finite(fnorm1);
            st[F.C_FN * B + b] = fnorm1;
            si[F.C_OK * B + b] = ok ? 1 : 0;
            si[F.C_HASJ * B + b] = (with_d && ok) ? 1 : 0;
            if (with_d && ok) {
                LmAny<T> *lmp = nullptr;
                (void)lmp;
                switch (q) {
#define VP_GX_CASE(QQ)                                                                                                 \
    case QQ: {                                                                                                         \
        double A[QQ][QQ], bv[QQ], Rd[QQ][QQ], acd[QQ], qd[QQ];                                                         \
        int ip[QQ];                                                                                                    \
        for (int k = 0; k < QQ; ++k) {                                                                                 \
            bv[k] = s_acc[1 + q * q + k];                                                                              \
            for (int l = 0; l < QQ; ++l) A[k][l] = (l >= k) ? s_acc[1 + k * q + l] : s_acc[1 + l * q + k];             \
        }                                                                                                              \
        gram_to_qr<double, QQ>(A, bv, Rd, acd, ip, qd);                                                                \
        for (int k = 0; k < QQ; ++k) {                                                                                 \
            st[(F.C_ACN + k) * B + b] = (T)acd[k];                                                                     \
            st[(F.C_QTF + k) * B + b] = (T)qd[k];                                                                      \
            si[(F.C_IPVT + k) * B + b] = ip[k];                                                                        \
            for (int l = 0; l < QQ; ++l) st[(F.C_RJ + k * q + l) * B + b] = (T)Rd[k][l];                               \
        }                                                                                                              \
    } break;
                    VP_GX_CASE(1)
                    VP_GX_CASE(2)
                    VP_GX_CASE(3)
                    VP_GX_CASE(4)
                    VP_GX_CASE(5)
                    VP_GX_CASE(6)
                    VP_GX_CASE(7)
                    VP_GX_CASE(8)
#undef VP_GX_CASE
                default: break;
                }
            }
        }
        __syncthreads();
    }
}

} // namespace gen
} // namespace vp
