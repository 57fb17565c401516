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
};

// ---- runtime model with compile-time sizes: any mix of kinds / shared parameters ---------------
template <int N_, int Q_, int P_> struct RtModel {
    static constexpr int N = N_, Q = Q_, P = P_;
    static constexpr bool kStatic = false;
    int32_t kind_[N_];
    int32_t par_[N_][VP_MAX_BASIS_PARAMS];
    int32_t pb_[P_], pa_[P_], pp_[P_];
    __host__ __device__ int kind(int j) const { return kind_[j]; }
    __host__ __device__ int param(int j, int a) const { return par_[j][a]; }
    __host__ __device__ int pair_basis(int p) const { return pb_[p]; }
    __host__ __device__ int pair_arg(int p) const { return pa_[p]; }
    __host__ __device__ int pair_param(int p) const { return pp_[p]; }
};

__device__ __forceinline__ double texp(double x) { return __ocml_exp_f64(x); }
__device__ __forceinline__ float texp(float x) { return __ocml_exp_f32(x); }
__device__ __forceinline__ double tsin(double x) { return __ocml_sin_f64(x); }
__device__ __forceinline__ float tsin(float x) { return __ocml_sin_f32(x); }
__device__ __forceinline__ double tcos(double x) { return __ocml_cos_f64(x); }
__device__ __forceinline__ float tcos(float x) { return __ocml_cos_f32(x); }

// Build the (weighted) basis columns A[N][R] and derivative columns D[P][R] of one problem.
//   t[r]      grid value of the lane's row r
//   scale[r]  w_i for rows i < m (1 for unit weights) and 0 for padding rows i >= m
// Padding rows come out exactly zero in every column, so they drop out of all later reductions.
template <typename T, class M, int R>
__device__ __forceinline__ void build_columns(const M &mdl, const T (&alpha)[M::Q], const T (&t)[R],
                                              const T (&scale)[R], T (&A)[M::N][R], T (&D)[M::P > 0 ? M::P : 1][R]) {
    constexpr int N = M::N, P = M::P, Q = M::Q;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int kind = mdl.kind(j);
        const int i0 = mdl.param(j, 0), i1 = mdl.param(j, 1);
        const T p0 = (i0 >= 0) ? dyn_get<Q>(alpha, i0) : T(0);
        const T p1 = (i1 >= 0) ? dyn_get<Q>(alpha, i1) : T(0);
        // derivative slots of this basis (pair index or -1)
        int s0 = -1, s1 = -1;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (mdl.pair_basis(p) == j) {
                if (mdl.pair_arg(p) == 0) s0 = p;
                else s1 = p;
            }
        }
        T d0[R], d1[R];
        if (kind == VP_BASIS_CONST) {
#pragma unroll
            for (int r = 0; r < R; ++r) A[j][r] = scale[r];
        } else if (kind == VP_BASIS_EXP_DECAY) {
            // exp(-t/tau);  d/dtau = exp(-t/tau) * t / tau^2     (shared_test_code/src/lib.rs:101-114)
            const T rt = T(1) / p0;
            const T rt2 = T(1) / (p0 * p0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T e = texp(-div_refined(t[r], p0, rt)) * scale[r];
                A[j][r] = e;
                d0[r] = (e * t[r]) * rt2;
            }
        } else if (kind == VP_BASIS_EXP_RATE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T e = texp(-p0 * t[r]) * scale[r];
                A[j][r] = e;
                d0[r] = -t[r] * e;
            }
        } else if (kind == VP_BASIS_EXP_COS) {
            // exp(-a t) cos(b t)   (shared_test_code/src/models.rs:313-314, 349-372)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T ex = texp(-p0 * t[r]) * scale[r];
                const T f = ex * tcos(p1 * t[r]);
                A[j][r] = f;
                d0[r] = f * (-t[r]);
                d1[r] = -t[r] * ex * tsin(p1 * t[r]);
            }
        } else { // VP_BASIS_SIN_PHASE   (src/test_helpers/mod.rs:28-52)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T ph = p0 * t[r] + p1;
                const T cs = tcos(ph) * scale[r];
                A[j][r] = tsin(ph) * scale[r];
                d0[r] = t[r] * cs;
                d1[r] = cs;
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (p == s0) {
#pragma unroll
                for (int r = 0; r < R; ++r) D[p][r] = d0[r];
            }
            if (p == s1) {
#pragma unroll
                for (int r = 0; r < R; ++r) D[p][r] = d1[r];
            }
        }
    }
}

// Build an RtModel from the public descriptor (host side).  Returns false if sizes do not match.
template <int N_, int Q_, int P_> inline bool make_rt_model(const vp_model_desc &d, RtModel<N_, Q_, P_> &out) {
    if (d.n_basis != N_ || d.n_params != Q_) return false;
    int p = 0;
    for (int j = 0; j < N_; ++j) {
        out.kind_[j] = d.kind[j];
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

} // namespace vp
