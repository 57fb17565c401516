/*
 * varpro_oracle.c -- CPU restatement of geo-ant/varpro 0.13.3's hot path (TEST INFRASTRUCTURE).
 * See varpro_oracle.h for scope, provenance and pinning status.  Plain C99 + optional OpenMP.
 * Build: oracle/Makefile (gcc -O2, no fast-math: fp64 parity matters).
 */
#include "varpro_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------- */
/* basis functions: the closed descriptor language of include/varpro_hip.h, evaluated with the  */
/* exact formulas of the reference's test/bench models                                          */
/* ------------------------------------------------------------------------------------------- */

/* value of basis `kind` at t.  EXP_DECAY: shared_test_code/src/lib.rs:101-106 `(-t / tau).exp()` */
static double basis_value(int kind, double t, double p0, double p1) {
    switch (kind) {
    case VP_BASIS_CONST: return 1.0; /* shared_test_code/src/lib.rs:123 */
    case VP_BASIS_EXP_DECAY: return exp(-t / p0);
    case VP_BASIS_EXP_RATE: return exp(-p0 * t);
    case VP_BASIS_EXP_COS: return exp(-p0 * t) * cos(p1 * t); /* shared_test_code/src/models.rs:313-314 */
    case VP_BASIS_SIN_PHASE: return sin(p0 * t + p1);         /* src/test_helpers/mod.rs:28-34 */
    default: return NAN;
    }
}

/* d basis / d (argument a).  EXP_DECAY: shared_test_code/src/lib.rs:109-114
 * `(-t / tau).exp() * t / (tau * tau)`; EXP_COS: shared_test_code/src/models.rs:349-372 */
static double basis_deriv(int kind, int a, double t, double p0, double p1) {
    switch (kind) {
    case VP_BASIS_EXP_DECAY: return exp(-t / p0) * t / (p0 * p0);
    case VP_BASIS_EXP_RATE: return -t * exp(-p0 * t);
    case VP_BASIS_EXP_COS:
        if (a == 0) return (exp(-p0 * t) * cos(p1 * t)) * (-1. * t);
        return -t * exp(-p0 * t) * sin(p1 * t);
    case VP_BASIS_SIN_PHASE:
        if (a == 0) return t * cos(p0 * t + p1);
        return cos(p0 * t + p1);
    default: return 0.0;
    }
}

static int kind_arity(int kind) {
    switch (kind) {
    case VP_BASIS_CONST: return 0;
    case VP_BASIS_EXP_DECAY:
    case VP_BASIS_EXP_RATE: return 1;
    case VP_BASIS_EXP_COS:
    case VP_BASIS_SIN_PHASE: return 2;
    default: return -1;
    }
}

/* src/model/mod.rs:441-471: column j = basis j evaluated on the whole grid */
void vpo_eval_phi(const vp_model_desc *model, int m, const double *t, const double *alpha, double *Phi) {
    for (int j = 0; j < model->n_basis; ++j) {
        int kind = model->kind[j];
        double p0 = model->param[j][0] >= 0 ? alpha[model->param[j][0]] : 0.0;
        double p1 = model->param[j][1] >= 0 ? alpha[model->param[j][1]] : 0.0;
        for (int i = 0; i < m; ++i) Phi[i + (size_t)j * m] = basis_value(kind, t[i], p0, p1);
    }
}

/* src/model/mod.rs:473-512: zero-filled m x n, only columns whose basis depends on alpha_k filled */
void vpo_eval_dphi(const vp_model_desc *model, int m, const double *t, const double *alpha, int k, double *Dk) {
    memset(Dk, 0, sizeof(double) * (size_t)m * model->n_basis);
    for (int j = 0; j < model->n_basis; ++j) {
        int kind = model->kind[j];
        int ar = kind_arity(kind);
        double p0 = model->param[j][0] >= 0 ? alpha[model->param[j][0]] : 0.0;
        double p1 = model->param[j][1] >= 0 ? alpha[model->param[j][1]] : 0.0;
        for (int a = 0; a < ar; ++a) {
            if (model->param[j][a] != k) continue;
            for (int i = 0; i < m; ++i) Dk[i + (size_t)j * m] += basis_deriv(kind, a, t[i], p0, p1);
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* thin SVD (restates nalgebra 0.33 `svd(true,true)` as used at src/solvers/levmar/mod.rs:51)   */
/* Householder QR + one-sided Jacobi SVD of the small triangular factor: backward stable.  Only  */
/* C, R and U U^T are comparable across SVD implementations (vector signs/ordering are not).    */
/* ------------------------------------------------------------------------------------------- */
/* one-sided Jacobi SVD of a small n x n matrix held in W (column-major, overwritten by U_r*Sigma) */
static void small_jacobi_svd(int n, long double *W, long double *V) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) V[i + j * n] = (i == j) ? 1.0L : 0.0L;
    for (int sweep = 0; sweep < 100; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                long double a = 0, b = 0, g = 0;
                for (int i = 0; i < n; ++i) {
                    a += W[i + p * n] * W[i + p * n];
                    b += W[i + q * n] * W[i + q * n];
                    g += W[i + p * n] * W[i + q * n];
                }
                if (g == 0.0L || fabsl(g) <= 1e-19L * sqrtl(a * b)) continue;
                rotated = 1;
                long double zeta = (b - a) / (2.0L * g);
                long double tt = (zeta >= 0 ? 1.0L : -1.0L) / (fabsl(zeta) + sqrtl(1.0L + zeta * zeta));
                long double c = 1.0L / sqrtl(1.0L + tt * tt), s = c * tt;
                for (int i = 0; i < n; ++i) {
                    long double x = W[i + p * n], y = W[i + q * n];
                    W[i + p * n] = c * x - s * y;
                    W[i + q * n] = s * x + c * y;
                    x = V[i + p * n];
                    y = V[i + q * n];
                    V[i + p * n] = c * x - s * y;
                    V[i + q * n] = s * x + c * y;
                }
            }
        }
        if (!rotated) break;
    }
}

/* Dot product in twice the working precision without x87 `long double` (which serialises on the x87 stack and does
 * not exist on other hosts): Ogita-Rump-Oishi Dot2 -- error-free product via fma(), error-free sum via TwoSum --
 * over four independent accumulator pairs so that the dependent add chains overlap. */
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target_clones("fma", "default"))) /* fma() inlined to one instruction where the host has it */
#endif
static double dot2(int m, const double *a, const double *b) {
    double s[4] = {0, 0, 0, 0}, c[4] = {0, 0, 0, 0};
    int i = 0;
    for (; i + 4 <= m; i += 4)
        for (int l = 0; l < 4; ++l) {
            const double x = a[i + l], y = b[i + l];
            const double pr = x * y, pe = fma(x, y, -pr);
            const double t = s[l] + pr, z = t - s[l];
            const double se = (s[l] - (t - z)) + (pr - z);
            s[l] = t;
            c[l] += pe + se;
        }
    for (; i < m; ++i) {
        const double x = a[i], y = b[i];
        const double pr = x * y, pe = fma(x, y, -pr);
        const double t = s[0] + pr, z = t - s[0];
        const double se = (s[0] - (t - z)) + (pr - z);
        s[0] = t;
        c[0] += pe + se;
    }
    /* combine the four partial sums, again error-free */
    double hi = s[0], lo = c[0];
    for (int l = 1; l < 4; ++l) {
        const double t = hi + s[l], z = t - hi;
        lo += ((hi - (t - z)) + (s[l] - z)) + c[l];
        hi = t;
    }
    return hi + lo;
}

/* Householder QR (m x n) -> explicit thin Q, then Jacobi SVD of the n x n R: A = (Q U_r) Sigma V^T.
 * Cost ~ 4 m n^2 flops, comparable to nalgebra's bidiagonalisation route (keeps the CPU baseline fair). */
static void thin_svd_ws(int m, int n, const double *A, double *U, double *sigma, double *V, double *W);

void vpo_thin_svd(int m, int n, const double *A, double *U, double *sigma, double *V) {
    double *W = (double *)malloc(sizeof(double) * (size_t)m * n);
    thin_svd_ws(m, n, A, U, sigma, V, W);
    free(W);
}

/* W: m x n workspace owned by the caller (the problem object: no allocation per evaluation) */
static void thin_svd_ws(int m, int n, const double *A, double *U, double *sigma, double *V, double *W) {
    if (m < n) {
        /* a WIDE matrix (fewer samples than basis functions): nalgebra's svd(true, true) returns U m x m, m singular
         * values, V^T m x n, and svd.solve the minimum-norm solution.  Realised here on [A; 0] (n x n): zero rows change
         * neither the non-zero singular values nor their right vectors, and the left vectors of the non-zero singular
         * values have zeros in the added rows.  Returned in the m >= n layout the callers use: n columns, those past
         * rank(A) with sigma = 0 (pinned by numpy's minimum-norm lstsq in tests/test_oracle_reference_vectors.py). */
        double Ap[VP_MAX_BASIS * VP_MAX_BASIS], Up[VP_MAX_BASIS * VP_MAX_BASIS], Wp[VP_MAX_BASIS * VP_MAX_BASIS];
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) Ap[i + j * n] = (i < m) ? A[i + (size_t)j * m] : 0.0;
        thin_svd_ws(n, n, Ap, Up, sigma, V, Wp);
        for (int j = 0; j < n; ++j) {
            if (j >= m) sigma[j] = 0.0; /* (rounding-level values of the n - m structural zeros) */
            for (int i = 0; i < m; ++i) U[i + (size_t)j * m] = (j < m) ? Up[i + j * n] : 0.0;
        }
        return;
    }
    double tau[VP_MAX_BASIS];
    long double Rm[VP_MAX_BASIS * VP_MAX_BASIS], Vr[VP_MAX_BASIS * VP_MAX_BASIS];
    memcpy(W, A, sizeof(double) * (size_t)m * n);
    for (int k = 0; k < n; ++k) {
        double *ak = W + (size_t)k * m;
        double xn2 = 0;
        for (int i = k + 1; i < m; ++i) xn2 += ak[i] * ak[i];
        double alpha = ak[k];
        if (xn2 == 0.0) {
            tau[k] = 0.0;
            continue;
        }
        double beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
        tau[k] = (beta - alpha) / beta;
        double scal = 1.0 / (alpha - beta);
        for (int i = k + 1; i < m; ++i) ak[i] *= scal;
        ak[k] = beta;
        for (int j = k + 1; j < n; ++j) {
            double *aj = W + (size_t)j * m;
            double dot = aj[k];
            for (int i = k + 1; i < m; ++i) dot += ak[i] * aj[i];
            dot *= tau[k];
            aj[k] -= dot;
            for (int i = k + 1; i < m; ++i) aj[i] -= dot * ak[i];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Rm[i + j * n] = (i <= j) ? (long double)W[i + (size_t)j * m] : 0.0L;
    small_jacobi_svd(n, Rm, Vr); /* Rm = U_r * Sigma */
    long double sg[VP_MAX_BASIS];
    int order[VP_MAX_BASIS];
    for (int j = 0; j < n; ++j) {
        long double s2 = 0;
        for (int i = 0; i < n; ++i) s2 += Rm[i + j * n] * Rm[i + j * n];
        sg[j] = sqrtl(s2);
        order[j] = j;
    }
    for (int a = 0; a < n - 1; ++a) /* descending, as nalgebra orders singular values */
        for (int b = a + 1; b < n; ++b)
            if (sg[order[b]] > sg[order[a]]) {
                int tmp = order[a];
                order[a] = order[b];
                order[b] = tmp;
            }
    /* U = Q * U_r : start from [U_r; 0] and apply H_0 ... H_{n-1} in reverse */
    for (int jj = 0; jj < n; ++jj) {
        int j = order[jj];
        sigma[jj] = (double)sg[j];
        double *u = U + (size_t)jj * m;
        for (int i = 0; i < n; ++i) u[i] = sg[j] > 0 ? (double)(Rm[i + j * n] / sg[j]) : 0.0;
        for (int i = n; i < m; ++i) u[i] = 0.0;
        for (int i = 0; i < n; ++i) V[i + jj * n] = (double)Vr[i + j * n];
        for (int k = n - 1; k >= 0; --k) {
            if (tau[k] == 0.0) continue;
            const double *ak = W + (size_t)k * m;
            double dot = u[k];
            for (int i = k + 1; i < m; ++i) dot += ak[i] * u[i];
            dot *= tau[k];
            u[k] -= dot;
            for (int i = k + 1; i < m; ++i) u[i] -= dot * ak[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* problem state                                                                                */
/* ------------------------------------------------------------------------------------------- */
static int all_finite(const double *x, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!isfinite(x[i])) return 0;
    return 1;
}

vpo_problem *vpo_problem_create(const vp_model_desc *model, int m, int S, const double *t, const double *Y,
                                const double *w, double svd_epsilon, int *build_err) {
    int err = VP_BUILD_OK;
    /* src/problem/builder.rs:280-302 */
    if (!Y) err = VP_BUILD_Y_DATA_MISSING;
    else if (m <= 0 || S <= 0) err = VP_BUILD_ZERO_LENGTH_VECTOR;
    if (build_err) *build_err = err;
    if (err) return NULL;
    vpo_problem *p = (vpo_problem *)calloc(1, sizeof(*p));
    int n = model->n_basis, q = model->n_params;
    p->model = *model;
    p->m = m;
    p->S = S;
    p->t = (double *)calloc(m, sizeof(double));
    if (t) memcpy(p->t, t, sizeof(double) * m); /* (NULL: a caller-evaluated model, vpo_problem_set_external_model) */
    p->w = NULL;
    if (w) {
        p->w = (double *)malloc(sizeof(double) * m);
        memcpy(p->w, w, sizeof(double) * m);
    }
    /* src/problem/builder.rs:307  Y_w = &weights * Y  (src/util/mod.rs:86-95: row i of every column * w_i) */
    p->Yw = (double *)malloc(sizeof(double) * (size_t)m * S);
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < m; ++i) p->Yw[i + (size_t)s * m] = (w ? w[i] : 1.0) * Y[i + (size_t)s * m];
    /* src/problem/builder.rs:246-251, 282: epsilon = |eps| or machine epsilon */
    p->eps = svd_epsilon < 0 ? DBL_EPSILON : fabs(svd_epsilon);
    p->alpha = (double *)calloc(q > 0 ? q : 1, sizeof(double));
    p->U = (double *)malloc(sizeof(double) * (size_t)m * n);
    p->sigma = (double *)malloc(sizeof(double) * n);
    p->V = (double *)malloc(sizeof(double) * n * n);
    p->C = (double *)malloc(sizeof(double) * (size_t)n * S);
    p->R = (double *)malloc(sizeof(double) * (size_t)m * S);
    /* workspace of set_params / jacobian / fit, owned by the problem: no allocation per evaluation */
    p->ws_phi = (double *)malloc(sizeof(double) * (size_t)m * n);
    p->ws_qr = (double *)malloc(sizeof(double) * (size_t)m * n);
    p->ws_dk = (double *)malloc(sizeof(double) * (size_t)m * n);
    p->ws_fvec = (double *)malloc(sizeof(double) * (size_t)m * S);
    p->ws_fwork = (double *)malloc(sizeof(double) * (size_t)m * S);
    p->ws_fjac = (double *)malloc(sizeof(double) * (size_t)m * S * (q > 0 ? q : 1));
    p->cached = 0;
    return p;
}

void vpo_problem_set_external_model(vpo_problem *p, void (*eval)(void *, const double *, double *),
                                    void (*dphi)(void *, const double *, int, double *), void *user) {
    p->ext_eval = eval;
    p->ext_dphi = dphi;
    p->ext_user = user;
    p->cached = 0;
}

/* the two model calls of the path: the descriptor formulas, or the user's trait impl (src/model/mod.rs:308, 359-362) */
static void model_eval(const vpo_problem *p, double *Phi) {
    if (p->ext_eval) p->ext_eval(p->ext_user, p->alpha, Phi);
    else vpo_eval_phi(&p->model, p->m, p->t, p->alpha, Phi);
}
static void model_dphi(const vpo_problem *p, int k, double *Dk) {
    if (p->ext_eval) p->ext_dphi(p->ext_user, p->alpha, k, Dk);
    else vpo_eval_dphi(&p->model, p->m, p->t, p->alpha, k, Dk);
}

/* New observations for an existing problem (same model, grid and weights): == building another SeparableProblem
 * without re-allocating -- the batched helpers keep ONE problem object per thread. */
void vpo_problem_reset(vpo_problem *p, const double *Y) {
    const int m = p->m, S = p->S;
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < m; ++i) p->Yw[i + (size_t)s * m] = (p->w ? p->w[i] : 1.0) * Y[i + (size_t)s * m];
    p->cached = 0;
    p->n_set_params = 0;
    p->n_jacobians = 0;
}

void vpo_problem_destroy(vpo_problem *p) {
    if (!p) return;
    free(p->t);
    free(p->w);
    free(p->Yw);
    free(p->alpha);
    free(p->U);
    free(p->sigma);
    free(p->V);
    free(p->C);
    free(p->R);
    free(p->ws_phi);
    free(p->ws_qr);
    free(p->ws_dk);
    free(p->ws_fvec);
    free(p->ws_fwork);
    free(p->ws_fjac);
    free(p);
}

/* src/solvers/levmar/mod.rs:42-73 */
void vpo_set_params(vpo_problem *p, const double *alpha) {
    const int m = p->m, S = p->S, n = p->model.n_basis, q = p->model.n_params;
    p->n_set_params++;
    memcpy(p->alpha, alpha, sizeof(double) * q); /* :43 model.set_params(params.clone()) */
    double *Phi_w = p->ws_phi;
    /* :47  Phi_w = &self.weights * Phi */
    model_eval(p, Phi_w);
    if (p->w)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i) Phi_w[i + (size_t)j * m] *= p->w[i];
    if (!all_finite(Phi_w, (size_t)m * n)) { /* nalgebra's SVD does not converge on non-finite input:
                                                treated as the error path, cached = None (:70-72) */
        p->cached = 0;
        return;
    }
    /* :51  Phi_w.clone().svd(true, true) */
    thin_svd_ws(m, n, Phi_w, p->U, p->sigma, p->V, p->ws_qr);
    /* :52-54  svd.solve(&Y_w, eps): x = V * diag(sigma_i > eps ? 1/sigma_i : 0) * U^T b (absolute threshold) */
    double utb[VP_MAX_BASIS];
    for (int s = 0; s < S; ++s) {
        const double *y = p->Yw + (size_t)s * m;
        for (int j = 0; j < n; ++j) {
            const double acc = dot2(m, p->U + (size_t)j * m, y);
            utb[j] = (p->sigma[j] > p->eps) ? acc / p->sigma[j] : 0.0;
        }
        for (int i = 0; i < n; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j) acc += p->V[i + j * n] * utb[j];
            p->C[i + (size_t)s * n] = acc;
        }
        /* :57-59  R = &Y_w - &Phi_w * C */
        double *r = p->R + (size_t)s * m;
        for (int i = 0; i < m; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j) acc += Phi_w[i + (size_t)j * m] * p->C[j + (size_t)s * n];
            r[i] = y[i] - acc;
        }
    }
    p->cached = all_finite(p->C, (size_t)n * S) && all_finite(p->R, (size_t)m * S);
}

/* src/solvers/levmar/mod.rs:91-95 + src/util/mod.rs:101-106 (column stacking == memory order) */
int vpo_residuals(const vpo_problem *p, double *r_out) {
    if (!p->cached) return 0;
    memcpy(r_out, p->R, sizeof(double) * (size_t)p->m * p->S);
    return 1;
}

/* src/solvers/levmar/mod.rs:101-201 -- both association orders, as written */
int vpo_jacobian(vpo_problem *p, double *J_out) {
    if (!p->cached) return 0;
    const int m = p->m, S = p->S, n = p->model.n_basis, q = p->model.n_params;
    p->n_jacobians++;
    double *Dk = p->ws_dk;
    const double *U = p->U;
    for (int k = 0; k < q; ++k) {
        /* :141  Dk = &self.weights * model.eval_partial_deriv(k) */
        model_dphi(p, k, Dk);
        if (p->w)
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < m; ++i) Dk[i + (size_t)j * m] *= p->w[i];
        double *Jk = J_out + (size_t)k * m * S; /* column k viewed as m x S (:147-153) */
        if (S <= q) {
            /* :156-171  T = Dk*C ; J_k = U (U^T T) - T */
            for (int s = 0; s < S; ++s) {
                double *T = Jk + (size_t)s * m;
                for (int i = 0; i < m; ++i) {
                    double acc = 0;
                    for (int j = 0; j < n; ++j) acc += Dk[i + (size_t)j * m] * p->C[j + (size_t)s * n];
                    T[i] = acc;
                }
                double utt[VP_MAX_BASIS];
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int i = 0; i < m; ++i) acc += U[i + (size_t)j * m] * T[i];
                    utt[j] = acc;
                }
                for (int i = 0; i < m; ++i) {
                    double acc = 0;
                    for (int j = 0; j < n; ++j) acc += U[i + (size_t)j * m] * utt[j];
                    T[i] = acc - T[i]; /* gemm(one, U, Ut_DkC, -one) */
                }
            }
        } else {
            /* :172-186  A = U (U^T Dk) - Dk ; J_k = A * C */
            double utd[VP_MAX_BASIS * VP_MAX_BASIS];
            for (int c = 0; c < n; ++c)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int i = 0; i < m; ++i) acc += U[i + (size_t)j * m] * Dk[i + (size_t)c * m];
                    utd[j + c * n] = acc;
                }
            for (int c = 0; c < n; ++c)
                for (int i = 0; i < m; ++i) {
                    double acc = 0;
                    for (int j = 0; j < n; ++j) acc += U[i + (size_t)j * m] * utd[j + c * n];
                    Dk[i + (size_t)c * m] = acc - Dk[i + (size_t)c * m];
                }
            for (int s = 0; s < S; ++s)
                for (int i = 0; i < m; ++i) {
                    double acc = 0;
                    for (int c = 0; c < n; ++c) acc += Dk[i + (size_t)c * m] * p->C[c + (size_t)s * n];
                    Jk[i + (size_t)s * m] = acc;
                }
        }
    }
    return 1;
}

/* src/fit.rs:55-59, 87-91: eval() (UNWEIGHTED) * coefficients */
int vpo_best_fit(const vpo_problem *p, double *fit_out) {
    if (!p->cached) return 0;
    const int m = p->m, S = p->S, n = p->model.n_basis;
    double *Phi = (double *)malloc(sizeof(double) * (size_t)m * n);
    model_eval(p, Phi);
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < m; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j) acc += Phi[i + (size_t)j * m] * p->C[j + (size_t)s * n];
            fit_out[i + (size_t)s * m] = acc;
        }
    free(Phi);
    return 1;
}

/* == FitStatistics::try_calculate (src/statistics/mod.rs:352-441), single RHS.
 * cov: (n+q) x (n+q) column-major, ordering [linear coefficients, nonlinear parameters];
 * conf_sigma[i] = sqrt(j_i^T cov j_i) (the reference's unscaled_confidence_sigma; the Student-t factor of
 * confidence_band_radius :271-304 is applied by the caller).  Returns 1 ok, 0 = Underdetermined /
 * MatrixInversion / no cached solution. */
int vpo_statistics(vpo_problem *p, double *cov, double *reduced_chi2, double *conf_sigma) {
    if (!p->cached || p->S != 1) return 0;
    const int m = p->m, n = p->model.n_basis, q = p->model.n_params, k = n + q;
    if (m <= k) return 0; /* Error::Underdetermined */
    double *J = (double *)malloc(sizeof(double) * (size_t)m * k); /* model_function_jacobian :481-511 */
    double *Dk = (double *)malloc(sizeof(double) * (size_t)m * n);
    model_eval(p, J);
    for (int a = 0; a < q; ++a) {
        model_dphi(p, a, Dk);
        for (int i = 0; i < m; ++i) {
            double acc = 0;
            for (int j = 0; j < n; ++j) acc += Dk[i + (size_t)j * m] * p->C[j];
            J[i + (size_t)(n + a) * m] = acc;
        }
    }
    const int dof = m - k;
    double rss = 0;
    for (int i = 0; i < m; ++i) rss += p->R[i] * p->R[i]; /* weighted residuals == cached R */
    const double chi2 = rss / (double)dof;
    /* H = W J ; (H^T H)^{-1} by Gauss-Jordan with partial pivoting in extended precision */
    long double A[2 * (VP_MAX_BASIS + VP_MAX_PARAMS)][2 * (VP_MAX_BASIS + VP_MAX_PARAMS)];
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
            long double acc = 0;
            for (int i = 0; i < m; ++i) {
                const long double w = p->w ? p->w[i] : 1.0;
                acc += (w * J[i + (size_t)a * m]) * (w * J[i + (size_t)b * m]);
            }
            A[a][b] = acc;
            A[a][k + b] = (a == b) ? 1.0L : 0.0L;
        }
    int ok = 1;
    for (int c = 0; c < k && ok; ++c) {
        int piv = c;
        for (int r = c + 1; r < k; ++r)
            if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0L || !isfinite((double)A[piv][c])) {
            ok = 0;
            break;
        }
        if (piv != c)
            for (int b = 0; b < 2 * k; ++b) {
                long double t = A[c][b];
                A[c][b] = A[piv][b];
                A[piv][b] = t;
            }
        const long double d = A[c][c];
        for (int b = 0; b < 2 * k; ++b) A[c][b] /= d;
        for (int r = 0; r < k; ++r)
            if (r != c) {
                const long double f = A[r][c];
                for (int b = 0; b < 2 * k; ++b) A[r][b] -= f * A[c][b];
            }
    }
    if (ok) {
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) cov[a + (size_t)b * k] = (double)(A[a][k + b] * chi2);
        for (int i = 0; i < m; ++i) {
            double acc = 0;
            for (int a = 0; a < k; ++a) {
                double t = 0;
                for (int b = 0; b < k; ++b) t += cov[a + (size_t)b * k] * J[i + (size_t)b * m];
                acc += J[i + (size_t)a * m] * t;
            }
            conf_sigma[i] = sqrt(acc);
        }
        *reduced_chi2 = chi2;
    }
    free(J);
    free(Dk);
    return ok;
}

/* ------------------------------------------------------------------------------------------- */
/* Levenberg-Marquardt: restatement of levenberg-marquardt 0.14 == MINPACK lmder/lmpar/qrfac/   */
/* qrsolv (More, Garbow, Hillstrom) with the crate's termination semantics.  */
/* Call site in the reference: src/solvers/levmar/mod.rs:247.                                   */
/* ------------------------------------------------------------------------------------------- */

double vpo_enorm(int n, const double *x) {
    /* MINPACK enorm: scaled accumulation in three ranges */
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        double xabs = fabs(x[i]);
        if (isnan(xabs)) return xabs;
        if (xabs >= agiant || xabs <= rdwarf) {
            if (xabs > rdwarf) {
                if (xabs > x1max) {
                    double d = x1max / xabs;
                    s1 = 1.0 + s1 * d * d;
                    x1max = xabs;
                } else {
                    double d = xabs / x1max;
                    s1 += d * d;
                }
            } else {
                if (xabs > x3max) {
                    double d = x3max / xabs;
                    s3 = 1.0 + s3 * d * d;
                    x3max = xabs;
                } else if (xabs != 0.0) {
                    double d = xabs / x3max;
                    s3 += d * d;
                }
            }
        } else {
            s2 += xabs * xabs;
        }
    }
    if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

/* MINPACK qrfac with column pivoting on a (mr x n, column-major, overwritten). */
static void qrfac(int mr, int n, double *a, int *ipvt, double *rdiag, double *acnorm, double *wa) {
    const double epsmch = DBL_EPSILON;
    for (int j = 0; j < n; ++j) {
        acnorm[j] = vpo_enorm(mr, a + (size_t)j * mr);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    int minmn = mr < n ? mr : n;
    for (int j = 0; j < minmn; ++j) {
        int kmax = j;
        for (int k = j; k < n; ++k)
            if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < mr; ++i) {
                double tmp = a[i + (size_t)j * mr];
                a[i + (size_t)j * mr] = a[i + (size_t)kmax * mr];
                a[i + (size_t)kmax * mr] = tmp;
            }
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            int k = ipvt[j];
            ipvt[j] = ipvt[kmax];
            ipvt[kmax] = k;
        }
        double *aj = a + (size_t)j * mr;
        double ajnorm = vpo_enorm(mr - j, aj + j);
        if (ajnorm == 0.0) {
            rdiag[j] = 0.0;
            continue;
        }
        if (aj[j] < 0.0) ajnorm = -ajnorm;
        for (int i = j; i < mr; ++i) aj[i] /= ajnorm;
        aj[j] += 1.0;
        for (int k = j + 1; k < n; ++k) {
            double *ak = a + (size_t)k * mr;
            double sum = 0;
            for (int i = j; i < mr; ++i) sum += aj[i] * ak[i];
            double temp = sum / aj[j];
            for (int i = j; i < mr; ++i) ak[i] -= temp * aj[i];
            if (rdiag[k] != 0.0) {
                double tq = ak[j] / rdiag[k];
                double d = 1.0 - tq * tq;
                rdiag[k] *= sqrt(d > 0.0 ? d : 0.0);
                double r = rdiag[k] / wa[k];
                if (0.05 * (r * r) <= epsmch) {
                    rdiag[k] = vpo_enorm(mr - j - 1, ak + j + 1);
                    wa[k] = rdiag[k];
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

/* MINPACK qrsolv.  r: n x n column-major, upper triangle = R (full diagonal), lower used as scratch */
static void qrsolv(int n, double *r, const int *ipvt, const double *diag, const double *qtb, double *x,
                   double *sdiag, double *wa) {
    for (int j = 0; j < n; ++j) {
        for (int i = j; i < n; ++i) r[i + j * n] = r[j + i * n];
        x[j] = r[j + j * n];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; ++j) {
        int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < n; ++k) {
                if (sdiag[k] == 0.0) continue;
                double c, s;
                if (fabs(r[k + k * n]) < fabs(sdiag[k])) {
                    double cotan = r[k + k * n] / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                    c = s * cotan;
                } else {
                    double tn = sdiag[k] / r[k + k * n];
                    c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    s = c * tn;
                }
                r[k + k * n] = c * r[k + k * n] + s * sdiag[k];
                double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; ++i) {
                    temp = c * r[i + k * n] + s * sdiag[i];
                    sdiag[i] = -s * r[i + k * n] + c * sdiag[i];
                    r[i + k * n] = temp;
                }
            }
        }
        sdiag[j] = r[j + j * n];
        r[j + j * n] = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        int j = nsing - k;
        double sum = 0;
        for (int i = j + 1; i < nsing; ++i) sum += r[i + j * n] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

/* MINPACK lmpar.  Returns par; x = step p (the new point is x_old - p); *dxnorm_out = ||diag .* p|| */
static double lmpar(int n, double *r, const int *ipvt, const double *diag, const double *qtb, double delta,
                    double par, double *x, double *sdiag, double *wa1, double *wa2, double *dxnorm_out) {
    const double p1 = 0.1, p001 = 0.001, dwarf = DBL_MIN;
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        wa1[j] = qtb[j];
        if (r[j + j * n] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        int j = nsing - k;
        wa1[j] /= r[j + j * n];
        double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[i + j * n] * temp;
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = vpo_enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= p1 * delta) {
        *dxnorm_out = dxnorm;
        return 0.0;
    }
    double parl = 0.0;
    if (nsing >= n) {
        for (int j = 0; j < n; ++j) {
            int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            double sum = 0;
            for (int i = 0; i < j; ++i) sum += r[i + j * n] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j + j * n];
        }
        double temp = vpo_enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; ++j) {
        double sum = 0;
        for (int i = 0; i <= j; ++i) sum += r[i + j * n] * qtb[i];
        int l = ipvt[j];
        wa1[j] = sum / diag[l];
    }
    double gnorm = vpo_enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = dwarf / (delta < p1 ? delta : p1);
    par = par > parl ? par : parl;
    par = par < paru ? par : paru;
    if (par == 0.0) par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (par == 0.0) par = dwarf > p001 * paru ? dwarf : p001 * paru;
        double temp = sqrt(par);
        for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
        qrsolv(n, r, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = vpo_enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= p1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < n; ++j) {
            int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
            for (int i = j + 1; i < n; ++i) wa1[i] -= r[i + j * n] * temp;
        }
        temp = vpo_enorm(n, wa1);
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = parl > par ? parl : par;
        if (fp < 0.0) paru = paru < par ? paru : par;
        par = parl > par + parc ? parl : par + parc;
    }
    *dxnorm_out = dxnorm;
    return par;
}

/* test hook: MINPACK lmpar as restated above on a caller's factor (r: n x n column-major upper triangle, left untouched) */
double vpo_lmpar(int n, const double *r, const int *ipvt, const double *diag, const double *qtb, double delta, double par,
                 double *x, double *dxnorm_out) {
    double *w = (double *)malloc(sizeof(double) * (size_t)(n * n + 3 * n));
    double *rw = w, *sdiag = w + n * n, *wa1 = sdiag + n, *wa2 = wa1 + n;
    memcpy(rw, r, sizeof(double) * (size_t)(n * n));
    const double out = lmpar(n, rw, ipvt, diag, qtb, delta, par, x, sdiag, wa1, wa2, dxnorm_out);
    free(w);
    return out;
}

void vpo_lm_opts_default(vp_lm_opts *o) {
    /* LevenbergMarquardt::new() without the minpack-compat feature (Cargo.toml:19 enables none) */
    o->ftol = 30.0 * DBL_EPSILON;
    o->xtol = 30.0 * DBL_EPSILON;
    o->gtol = 30.0 * DBL_EPSILON;
    o->stepbound = 100.0;
    o->patience = 100;
    o->scale_diag = 1;
}

static void trace_row(double *trace, int max_rows, int *row, int n, const double *xt, double fnorm1, double ratio,
                      double delta, double par) {
    if (!trace || *row >= max_rows) return;
    double *t = trace + (size_t)(*row) * (n + 4);
    for (int j = 0; j < n; ++j) t[j] = xt[j];
    t[n] = fnorm1;
    t[n + 1] = ratio;
    t[n + 2] = delta;
    t[n + 3] = par;
    ++*row;
}

void vpo_fit(vpo_problem *p, const vp_lm_opts *opts, vp_report *rep) { vpo_fit_trace(p, opts, rep, NULL, 0); }

/* == levenberg_marquardt::LevenbergMarquardt::minimize (call site src/solvers/levmar/mod.rs:247) over ANY
 * LeastSquaresProblem given as callbacks (vpo_lsq): MINPACK lmder with the crate's termination semantics.  The
 * problem is only touched through set_params / residuals / jacobian, exactly as the crate touches the trait
 * (src/solvers/levmar/mod.rs:22-202) -- so the same driver can run on the CPU restatement (vpo_fit_trace below) or,
 * in tests, on the C ABI of the device library (tests/c/test_trait_lm.c).  Records one row [x_trial(q),
 * ||r(x_trial)||, ratio, delta, par] per evaluation (row 0: the initial point with ratio = NaN).
 * fvec, fwork: mr doubles each; fjac: mr*n doubles (workspace owned by the caller). */
int vpo_lm_minimize(const vpo_lsq *P, const vp_lm_opts *opts, vp_report *rep, double *trace, int max_rows,
                    double *fvec, double *fwork, double *fjac) {
    int trow = 0;
    const int n = P->n;                    /* LM "n" = number of parameters q */
    const int mr = P->mr;                  /* LM "m" = number of residuals */
    const double epsmch = DBL_EPSILON;
    vp_report report;
    report.termination = VP_TERM_NO_PARAMETERS;
    report.n_evals = 1;
    report.objective = NAN;
    double x[VP_MAX_PARAMS], xt[VP_MAX_PARAMS], diag[VP_MAX_PARAMS], tmp[VP_MAX_PARAMS];
    double qtf[VP_MAX_PARAMS], step[VP_MAX_PARAMS], sdiag[VP_MAX_PARAMS], wa1[VP_MAX_PARAMS], wa2[VP_MAX_PARAMS];
    double rdiag[VP_MAX_PARAMS], acnorm[VP_MAX_PARAMS], wa[VP_MAX_PARAMS];
    double rmat[VP_MAX_PARAMS * VP_MAX_PARAMS];
    int ipvt[VP_MAX_PARAMS];
    double fnorm = 0, delta = 0, par = 0, xnorm = 0, gnorm = 0;
    int first_tr = 1, first_update = 1;
    const int max_fev = opts->patience * (n + 1);
    if (n == 0) goto done;
    P->params(P->user, x);
    if (!P->residuals(P->user, fvec)) {
        report.termination = VP_TERM_USER;
        goto done;
    }
    fnorm = vpo_enorm(mr, fvec);
    report.objective = 0.5 * fnorm * fnorm;
    trace_row(trace, max_rows, &trow, n, x, fnorm, NAN, 0.0, 0.0);
    if (mr == 0) {
        report.termination = VP_TERM_NO_RESIDUALS;
        goto done;
    }
    if (n > mr) {
        report.termination = VP_TERM_WRONG_DIMENSIONS;
        goto done;
    }
    if (!isfinite(fnorm)) {
        report.termination = VP_TERM_NUMERICAL;
        goto done;
    }
    if (fnorm <= DBL_MIN) {
        report.termination = VP_TERM_RESIDUALS_ZERO;
        goto done;
    }
    for (int j = 0; j < n; ++j) diag[j] = 1.0;

    for (;;) { /* outer loop: new Jacobian */
        if (!P->jacobian(P->user, fjac)) {
            report.termination = VP_TERM_USER;
            goto done;
        }
        qrfac(mr, n, fjac, ipvt, rdiag, acnorm, wa);
        /* first n entries of Q^T fvec (lmder) */
        memcpy(fwork, fvec, sizeof(double) * mr);
        for (int j = 0; j < n; ++j) {
            double *aj = fjac + (size_t)j * mr;
            if (aj[j] != 0.0) {
                double sum = 0;
                for (int i = j; i < mr; ++i) sum += aj[i] * fwork[i];
                double temp = -sum / aj[j];
                for (int i = j; i < mr; ++i) fwork[i] += aj[i] * temp;
            }
            qtf[j] = fwork[j];
        }
        /* upper_r = R with rdiag on the diagonal */
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) rmat[i + j * n] = (i < j) ? fjac[i + (size_t)j * mr] : (i == j ? rdiag[j] : 0.0);
        /* norm of the scaled gradient */
        gnorm = 0.0;
        int degenerate = 0;
        for (int j = 0; j < n; ++j) {
            int l = ipvt[j];
            if (acnorm[l] == 0.0) continue;
            double sum = 0;
            for (int i = 0; i <= j; ++i) sum += rmat[i + j * n] * qtf[i];
            double temp = fabs(sum / (acnorm[l] * fnorm));
            if (isnan(temp)) degenerate = 1;
            if (temp > gnorm) gnorm = temp;
        }
        if (degenerate) {
            report.termination = VP_TERM_NUMERICAL;
            goto done;
        }
        if (gnorm <= opts->gtol) {
            report.termination = VP_TERM_ORTHOGONAL;
            goto done;
        }
        if (first_update) {
            if (opts->scale_diag) {
                for (int j = 0; j < n; ++j) diag[j] = acnorm[j] == 0.0 ? 1.0 : acnorm[j];
                for (int j = 0; j < n; ++j) tmp[j] = diag[j] * x[j];
                xnorm = vpo_enorm(n, tmp);
            } else {
                xnorm = vpo_enorm(n, x);
            }
            if (!isfinite(xnorm)) {
                report.termination = VP_TERM_NUMERICAL;
                goto done;
            }
            delta = xnorm == 0.0 ? opts->stepbound : opts->stepbound * xnorm;
            first_update = 0;
        } else if (opts->scale_diag) {
            for (int j = 0; j < n; ++j) diag[j] = acnorm[j] > diag[j] ? acnorm[j] : diag[j];
        }

        for (;;) { /* inner loop: trust-region iterations */
            double pnorm;
            par = lmpar(n, rmat, ipvt, diag, qtf, delta, par, step, sdiag, wa1, wa2, &pnorm);
            if (!isfinite(pnorm)) {
                report.termination = VP_TERM_NUMERICAL;
                goto done;
            }
            /* predicted reduction and directional derivative: ||R P^T p|| */
            for (int i = 0; i < n; ++i) wa1[i] = 0.0;
            for (int j = 0; j < n; ++j) {
                double pj = step[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa1[i] += rmat[i + j * n] * pj;
            }
            double t1 = vpo_enorm(n, wa1) / fnorm;
            double temp1 = t1 * t1;
            double t2 = (sqrt(par) * pnorm) / fnorm;
            double temp2 = t2 * t2;
            if (!isfinite(temp1) || !isfinite(temp2)) {
                report.termination = VP_TERM_NUMERICAL;
                goto done;
            }
            double prered = temp1 + temp2 / 0.5;
            double dirder = -(temp1 + temp2);
            if (first_tr && pnorm < delta) delta = pnorm;
            first_tr = 0;
            for (int j = 0; j < n; ++j) xt[j] = x[j] - step[j];
            P->set_params(P->user, xt);
            report.n_evals += 1;
            if (!P->residuals(P->user, fwork)) {
                report.termination = VP_TERM_USER;
                goto done;
            }
            double fnorm1 = vpo_enorm(mr, fwork);
            double new_objective = 0.5 * fnorm1 * fnorm1;
            double actred = (fnorm1 * 0.1 < fnorm) ? 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm) : -1.0;
            double ratio = prered == 0.0 ? 0.0 : actred / prered;
            if (ratio <= 0.25) {
                double temp = !signbit(actred) ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
                if (fnorm1 * 0.1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * (delta < pnorm * 10.0 ? delta : pnorm * 10.0);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par *= 0.5;
            }
            int good = ratio >= 1.0e-4;
            trace_row(trace, max_rows, &trow, n, xt, fnorm1, ratio, delta, par);
            if (good) {
                memcpy(x, xt, sizeof(double) * n);
                if (opts->scale_diag) {
                    for (int j = 0; j < n; ++j) tmp[j] = diag[j] * x[j];
                    xnorm = vpo_enorm(n, tmp);
                } else {
                    xnorm = vpo_enorm(n, x);
                }
                if (!isfinite(xnorm)) {
                    report.termination = VP_TERM_NUMERICAL;
                    goto done;
                }
                fnorm = fnorm1;
                memcpy(fvec, fwork, sizeof(double) * mr);
                report.objective = new_objective;
            }
            int term = 0;
            int ftol_check = 0, xtol_check = 0;
            if (fnorm <= DBL_MIN) term = VP_TERM_RESIDUALS_ZERO;
            if (!term) {
                ftol_check = fabs(actred) <= opts->ftol && prered <= opts->ftol && ratio * 0.5 <= 1.0;
                xtol_check = delta <= opts->xtol * xnorm;
                if (ftol_check || xtol_check)
                    term = ftol_check && xtol_check ? VP_TERM_CONVERGED_BOTH
                                                    : (ftol_check ? VP_TERM_CONVERGED_FTOL : VP_TERM_CONVERGED_XTOL);
            }
            if (!term && report.n_evals >= max_fev) term = VP_TERM_LOST_PATIENCE;
            if (!term && fabs(actred) <= epsmch && prered <= epsmch && ratio * 0.5 <= 1.0) term = VP_TERM_NO_IMPROVEMENT;
            if (!term && delta <= epsmch * xnorm) term = VP_TERM_NO_IMPROVEMENT;
            if (!term && gnorm <= epsmch) term = VP_TERM_NO_IMPROVEMENT;
            if (term) {
                if (!good) P->set_params(P->user, x); /* reset_params_if(!update_considered_good) */
                report.termination = term;
                goto done;
            }
            if (good) break;
        }
    }
done:
    if (rep) *rep = report;
    return trow;
}

/* the CPU restatement as a LeastSquaresProblem */
static void lsq_set_params(void *u, const double *x) { vpo_set_params((vpo_problem *)u, x); }
static void lsq_params(void *u, double *x) {
    vpo_problem *p = (vpo_problem *)u;
    memcpy(x, p->alpha, sizeof(double) * p->model.n_params);
}
static int lsq_residuals(void *u, double *r) { return vpo_residuals((vpo_problem *)u, r); }
static int lsq_jacobian(void *u, double *J) { return vpo_jacobian((vpo_problem *)u, J); }

/* same as vpo_fit; additionally records the per-evaluation trace -- used by the per-iteration parity tests */
int vpo_fit_trace(vpo_problem *p, const vp_lm_opts *opts, vp_report *rep, double *trace, int max_rows) {
    vpo_lsq P;
    P.n = p->model.n_params;
    P.mr = p->m * p->S;
    P.user = p;
    P.set_params = lsq_set_params;
    P.params = lsq_params;
    P.residuals = lsq_residuals;
    P.jacobian = lsq_jacobian;
    return vpo_lm_minimize(&P, opts, rep, trace, max_rows, p->ws_fvec, p->ws_fwork, p->ws_fjac);
}

/* ------------------------------------------------------------------------------------------- */
/* batched helpers (tests + CPU baseline)                                                       */
/* ------------------------------------------------------------------------------------------- */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int vpo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

double vpo_fit_batch(const vp_model_desc *model, int m, int64_t B, const double *t, const double *Y,
                     const double *w, double svd_epsilon, const vp_lm_opts *opts, double *alpha_inout,
                     double *C_out, vp_report *rep, int n_threads) {
    const int n = model->n_basis, q = model->n_params;
    double total = 0.0;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads) reduction(max : total)
#endif
    {
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        int64_t lo = B * tid / nt, hi = B * (tid + 1) / nt;
        double mine = 0.0;
        int err;
        /* one problem object (and workspace arena) per thread, re-used for every fit of its share */
        vpo_problem *p = (lo < hi) ? vpo_problem_create(model, m, 1, t, Y + (size_t)lo * m, w, svd_epsilon, &err) : NULL;
        for (int64_t b = lo; b < hi && p; ++b) {
            vpo_problem_reset(p, Y + (size_t)b * m);
            vpo_set_params(p, alpha_inout + (size_t)b * q); /* build(): initial set_params, untimed (builder.rs:321) */
            vp_report r;
            double t0 = now_s();
            vpo_fit(p, opts, &r);
            mine += now_s() - t0;
            memcpy(alpha_inout + (size_t)b * q, p->alpha, sizeof(double) * q);
            if (C_out) {
                if (p->cached) memcpy(C_out + (size_t)b * n, p->C, sizeof(double) * n);
                else
                    for (int j = 0; j < n; ++j) C_out[(size_t)b * n + j] = NAN;
            }
            if (rep) rep[b] = r;
        }
        vpo_problem_destroy(p);
        total = mine;
    }
    return total;
}

void vpo_evaluate_batch(const vp_model_desc *model, int m, int64_t B, const double *t, const double *Y,
                        const double *w, double svd_epsilon, const double *alpha, double *r_out, double *J_out,
                        double *C_out, double *cost_out, int32_t *status, int n_threads) {
    const int n = model->n_basis, q = model->n_params;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (int64_t b = 0; b < B; ++b) {
        int err;
        vpo_problem *p = vpo_problem_create(model, m, 1, t, Y + (size_t)b * m, w, svd_epsilon, &err);
        if (!p) continue;
        vpo_set_params(p, alpha + (size_t)b * q);
        if (status) status[b] = p->cached ? VP_ST_OK : VP_ST_NONFINITE;
        if (p->cached) {
            if (r_out) vpo_residuals(p, r_out + (size_t)b * m);
            if (J_out) vpo_jacobian(p, J_out + (size_t)b * m * q);
            if (C_out) memcpy(C_out + (size_t)b * n, p->C, sizeof(double) * n);
            if (cost_out) {
                double nr = vpo_enorm(m, p->R);
                cost_out[b] = 0.5 * nr * nr;
            }
        } else {
            if (r_out)
                for (int i = 0; i < m; ++i) r_out[(size_t)b * m + i] = NAN;
            if (J_out)
                for (int i = 0; i < m * q; ++i) J_out[(size_t)b * m * q + i] = NAN;
            if (C_out)
                for (int j = 0; j < n; ++j) C_out[(size_t)b * n + j] = NAN;
            if (cost_out) cost_out[b] = NAN;
        }
        vpo_problem_destroy(p);
    }
}
