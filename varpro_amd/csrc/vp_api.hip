// vp_api.hip -- implementation of the C ABI declared in include/varpro_hip.h.
// Host-side state management + dispatch into the kernel registry.  No fallbacks: every compute
// entry point runs HIP kernels on a gfx950 device or returns an error.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/varpro_hip.h"
#include "../../include/varpro_hip_debug.h"
#include "vp_registry.hpp"
#include "vp_extfit_api.hpp"
#include "vp_mrhs.hpp"

using namespace vp;

// global fit: identical consecutive fits (by their longest problem's evaluation count) before the captured graph drops its
// spare iteration (mrhs_fit)
#ifndef VP_MRHS_EXACT_AFTER
#define VP_MRHS_EXACT_AFTER 8
#endif

namespace {

thread_local std::string g_err;
thread_local int g_detail = 0;

int fail(int code, const std::string &msg, int detail = 0) {
    g_err = msg;
    g_detail = detail;
    return code;
}

#define VP_HIP(expr)                                                                                                  \
    do {                                                                                                              \
        hipError_t e__ = (expr);                                                                                      \
        if (e__ != hipSuccess)                                                                                        \
            return fail(VP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                            \
    } while (0)

inline size_t tsize(int dtype) { return dtype == VP_F64 ? 8 : 4; }

// ---- small utility kernels (dtype-generic plumbing, not the hot path) ------------------------------
// (Y and Yw may be the same buffer -- vp_set_observations stages host data through Yw -- hence no __restrict__ on them:
// every element is read and written by the same thread)
template <typename T>
__global__ void weight_data_kernel(const T *Y, const T *__restrict__ w, T *Yw, int m, int64_t cols_per_problem,
                                   int64_t w_stride, int64_t total) {
    // Y_w = W * Y  (src/problem/builder.rs:307, src/util/mod.rs:86-95)
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = idx / m;
        const int i = (int)(idx - col * m);
        const int64_t b = col / cols_per_problem;
        Yw[idx] = w ? (T)(w[b * w_stride + i] * Y[idx]) : Y[idx];
    }
}

// per-problem reduction over the S right-hand sides: cost[b] = sum_s cost_bs ; status[b] = max_s status_bs
__global__ void reduce_rhs_kernel(const double *__restrict__ cost_bs, const int32_t *__restrict__ status_bs,
                                  double *__restrict__ cost_b, int32_t *__restrict__ status_b, int S, int64_t B) {
    const int64_t b = blockIdx.x;
    if (b >= B) return;
    double acc = 0.0;
    int st = 0;
    // eight loads in flight per thread and trip (one dependent load per trip made this 24 us for S = 16384)
    for (int s0 = threadIdx.x; s0 < S; s0 += 8 * blockDim.x) {
        double c[8];
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = s0 + k * (int)blockDim.x;
            const bool in = s < S;
            c[k] = in ? cost_bs[b * S + s] : 0.0;
            v[k] = in ? status_bs[b * S + s] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc += c[k];
            st = v[k] > st ? v[k] : st;
        }
    }
    __shared__ double sh[1024];
    __shared__ int shs[1024];
    sh[threadIdx.x] = acc;
    shs[threadIdx.x] = st;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sh[threadIdx.x] += sh[threadIdx.x + off];
            shs[threadIdx.x] = shs[threadIdx.x] > shs[threadIdx.x + off] ? shs[threadIdx.x] : shs[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (cost_b) cost_b[b] = sh[0];
        if (status_b) status_b[b] = shs[0];
    }
}

// {sum cost, #successful, #failed, sum n_evals} over the batch
__global__ void summary_kernel(const vp_report *__restrict__ rep, int64_t B, double *__restrict__ out4) {
    double c = 0, ok = 0, bad = 0, ev = 0;
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const vp_report r = rep[b];
        if (r.objective == r.objective) c += r.objective;
        if (r.termination > 0) ok += 1.0;
        else bad += 1.0;
        ev += (double)r.n_evals;
    }
    __shared__ double sh[4][256];
    sh[0][threadIdx.x] = c;
    sh[1][threadIdx.x] = ok;
    sh[2][threadIdx.x] = bad;
    sh[3][threadIdx.x] = ev;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; ++k) atomicAdd(&out4[k], sh[k][0]);
}

// Sum of the streaming kernel's per-workgroup partials in a fixed order: tot[b][i] = sum_g part[b][g][i].  Used
// when the right-hand sides are sharded over ranks: the totals are all-reduced (vp_set_rhs_allreduce) and the LM
// step then reads them as a single slot, so that every rank takes bit-identical decisions.
__global__ void mrhs_reduce_partials_kernel(const double *part, int gx, int nacc, int64_t B, double *tot) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= B * nacc) return;
    const int64_t b = idx / nacc;
    const int i = (int)(idx - b * nacc);
    double s = 0.0;
    for (int g = 0; g < gx; ++g) s += part[((size_t)b * nacc + i) * gx + g]; // [b][accumulator][workgroup]
    tot[idx] = s;
}

// Uniform-grid check, once per handle: grid g passes if every t_i lies within 4 ulp-of-the-offset of the lattice
// t_0 + i*dt, dt = (t_{m-1} - t_0)/(m-1):   |t_i - (t_0 + i dt)| <= 4 eps |t_i - t_0|.
// That is what a linspace-type grid anchored at its first sample satisfies, and it bounds the argument error of
// the recurrence exp(-(t_0 + i dt)/tau) to 4 eps (t_i - t_0)/|tau| -- the size of the reference's own rounding of
// the quotient t_i/tau.  Grids with a large offset (|t_0| >> m dt) or irregular sampling fail and keep the
// per-row exponential.  One block per grid; *flag is AND-ed.
template <typename T> __global__ void grid_check_kernel(const T *t, int m, int64_t ngrids, int *flag) {
    const int64_t g = blockIdx.x;
    if (g >= ngrids) return;
    const T *tg = t + g * (int64_t)m;
    const double t0 = (double)tg[0];
    const double dt = ((double)tg[m - 1] - t0) / (double)(m - 1);
    // to the rounding of the grid's own type: 4 ulp of the distance from t_0
    const double tol = 4.0 * (sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07);
    bool ok = (dt == dt) && (dt - dt == 0.0) && dt != 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double lat = __builtin_fma((double)i, dt, t0);
        const double dev = __builtin_fabs((double)tg[i] - lat);
        if (!(dev <= tol * __builtin_fabs((double)tg[i] - t0))) ok = false;
    }
    if (!ok) atomicAnd(flag, 0);
}

} // namespace

// ---- the handle ------------------------------------------------------------------------------------
struct vp_batch {
    vp_model_desc model;
    int dtype;
    int64_t m, S, B;
    int n, q, p;
    int flags;
    int device;
    double eps;
    hipStream_t stream;
    bool own_stream;
    const KernelEntry *kern;
    // device state (== SeparableProblem + CachedCalculations for the whole batch)
    void *d_t, *d_w, *d_yw;
    void *d_alpha;      // [B][q]
    void *d_C;          // [B][S][n]
    void *d_R;          // [B][S][m]  lazily allocated residual cache
    double *d_cost_bs;  // [B*S]
    int32_t *d_status_bs;
    double *d_cost;     // [B]   (aliases d_cost_bs when S == 1)
    int32_t *d_status;  // [B]
    vp_report *d_report; // [B]
    double *d_sum4;
    bool grid_uniform; // every grid is t_0 + i*dt to rounding (grid_check_kernel): kernels may use the exp recurrence
    bool have_params; // set_params/evaluate/fit has run
    bool r_valid;     // d_R matches d_alpha
    bool have_report;
    // timing
    bool timing;
    hipEvent_t ev0, ev1;
    float last_ms[3];
    // multiple-right-hand-side path (S > 1): factor/stream/LM-step kernels + their workspace
    bool have_mrhs;
    MrhsWs mrhs;
    // S-sharded global fits (vp_set_rhs_allreduce)
    vp_allreduce_fn rhs_allreduce;
    void *rhs_allreduce_user;
    int64_t rhs_global; // right-hand sides of the whole problem
    double *d_mrhs_tot; // [B][1 + n*n + p] totals of the reduced sums (all-reduced across ranks); generic kernels: [B][2 + q*q + q]
    void *d_gen_lm;     // generic kernels, sharded right-hand sides: [B] LM state between the phases
    int32_t *d_gen_nactive;
    // single-RHS fit kernel selection (vp_set_fit_kernel) and the slot kernel's problem queue
    int fit_kernel;
    int *d_queue;
    int num_cus;
    void *tmp_a, *tmp_b; // scratch allocations of vp_batch_create (freed by destroy if create fails half way)
    void *d_gen_ws;      // generic fallback kernels: gen_blocks workspace slots of (n + 1 + p + q) columns x m
    int gen_blocks;
    // MRHS fit: a captured HIP graph of VP_MRHS_GRAPH_ITERS {factor, stream, LM step} iterations (replayed per batch
    // of iterations: one graph launch instead of 3 x ITERS kernel launches), the options it was captured with
    int64_t m_user;  // != 0: the caller's row count m < n; the handle works on m = n rows, the extra ones with zero weight
    hipGraphExec_t mrhs_graph;
    hipGraphExec_t mrhs_graph_tail; // 12 further iterations + finish, for a fit that outlasts mrhs_graph
    // caller-evaluated model (vp_batch_create_external): shape + dependency-pair table, and where the columns of the
    // current parameters live (the caller's device arrays, or this handle's staged copies on host-pointer handles)
    bool external;
    int ext_np;
    int32_t ext_pb[VP_MAX_PAIRS], ext_pp[VP_MAX_PAIRS];
    const void *ext_phi, *ext_dphi;
    void *ext_phi_own, *ext_dphi_own;
    int mrhs_graph_len;     // LM iterations mrhs_graph holds
    int mrhs_graph_iters;   // ... and what the next capture should hold (evaluations of the previous fit + 1, >= 6)
    int mrhs_prev_nfev;     // largest evaluation count of the previous fit, and for how many fits in a row it has been the
    int mrhs_same_count;    // same: a stream of fits of one length runs a graph WITHOUT the spare iteration
    int32_t *h_nactive;     // pinned, device-mapped: the active count as the graph's last kernel leaves it
    int32_t *h_nactive_dev; // its device address
    MrhsIo *h_io;           // pinned, device-mapped: the caller's arrays of the current vp_fit (whole-fit graph, device-pointer handles)
    MrhsIo *h_io_dev;
    vp_lm_opts mrhs_graph_opts;
    hipStream_t cap_stream;
    bool mrhs_graph_failed;
    // batched reverse-communication LM fit of a caller-evaluated model (vp_fit_begin / vp_fit_step_with_basis / vp_fit_end)
    void *d_xf_state;       // [B] LM records of the step kernel
    void *d_xf_trial;       // [B][q] trial points of the last step
    int32_t *d_xf_want;     // [B] what every problem wants next
    void *d_xf_ctrial;      // [B][S][n] coefficients of the trial point (S > 1: generic step)
    int32_t *d_xf_active;   // [2][B] compacted indices of the still-active problems, written alternately by the LM kernel
    int64_t xf_known_active; // the last active count the host read (an upper bound of the current one)
    int32_t *d_xf_nactive;  // device counter of the last step
    int32_t *h_xf_nactive;  // pinned host copy
    bool xf_running;        // between vp_fit_begin and vp_fit_end
    bool xf_init;           // the next step is the first
    int xf_flags;
    int64_t xf_steps;
    vp_lm_opts xf_opts;
    // flag-and-refit of single-RHS fits (vp_fit.hpp jac_not_finite; rescue_refit below)
    int32_t *d_rescue;      // [2 + B]: two ping-pong counters + the flagged problems of the running fit
    void *d_rescue_ws;      // kRescueBlocks workspace slots of the generic fit kernel
    int rescue_slot;        // the counter the NEXT fit appends to
    bool rescue_off;        // vp_debug / VP_NO_RESCUE=1: fits keep the kernels' own `Numerical` (what rounds 1-4 returned)
};

namespace {

bool device_ptrs(const vp_batch *h) { return (h->flags & VP_FLAG_DEVICE_PTRS) != 0; }

// input staging: user pointer -> device pointer usable on h->stream
struct InBuf {
    const void *dptr = nullptr;
    void *tmp = nullptr;
    int init(vp_batch *h, const void *user, size_t bytes) {
        if (!user) {
            dptr = nullptr;
            return 0;
        }
        if (device_ptrs(h)) {
            dptr = user;
            return 0;
        }
        VP_HIP(hipMalloc(&tmp, bytes ? bytes : 1));
        VP_HIP(hipMemcpyAsync(tmp, user, bytes, hipMemcpyHostToDevice, h->stream));
        dptr = tmp;
        return 0;
    }
    ~InBuf() {
        if (tmp) {
            (void)hipFree(tmp); // hipFree synchronises
        }
    }
};

// output staging: kernels write to dptr; finish() lands the bytes in the user's buffer
struct OutBuf {
    void *dptr = nullptr;
    void *tmp = nullptr;
    void *user = nullptr;
    size_t bytes = 0;
    int init(vp_batch *h, void *user_, size_t bytes_) {
        user = user_;
        bytes = bytes_;
        if (!user) return 0;
        if (device_ptrs(h)) {
            dptr = user;
            return 0;
        }
        VP_HIP(hipMalloc(&tmp, bytes ? bytes : 1));
        dptr = tmp;
        return 0;
    }
    int finish(vp_batch *h) {
        if (tmp) {
            VP_HIP(hipMemcpyAsync(user, tmp, bytes, hipMemcpyDeviceToHost, h->stream));
            VP_HIP(hipStreamSynchronize(h->stream));
        }
        return 0;
    }
    ~OutBuf() {
        if (tmp) (void)hipFree(tmp);
    }
};

int copy_out(vp_batch *h, void *user, const void *dev, size_t bytes);
// ---- m < n (an underdetermined linear sub-problem; the reference's SVD solve accepts it, src/solvers/levmar/mod.rs:51-54)
// The handle then works on n rows: the caller's m rows plus n - m rows of zero weight (zero rows change neither the
// minimum-norm coefficients nor the residual nor any singular value > 0).  Everything with a row dimension that crosses the
// ABI is padded on the way in and stripped on the way out.
__global__ void strip_rows_kernel(const unsigned *__restrict__ src, unsigned *__restrict__ dst, int64_t blocks, int words_user,
                                  int words_pad) {
    const int64_t total = blocks * words_user;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t blk = i / words_user;
        dst[i] = src[blk * words_pad + (i - blk * words_user)];
    }
}

// the LM crate's answer when there are fewer residuals than nonlinear parameters: one evaluation at the initial point,
// then TerminationReason::WrongDimensions (vp_lm_core.hpp lm_after_eval, first evaluation)
__global__ void wrong_dimensions_report_kernel(const double *__restrict__ cost, const int32_t *__restrict__ status, int64_t B,
                                               vp_report *__restrict__ rep, double *__restrict__ trace, int trace_rows, int q,
                                               int dtype, const void *__restrict__ alpha) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b >= B) return;
    vp_report r;
    const bool ok = status[b] == 0;
    r.termination = ok ? VP_TERM_WRONG_DIMENSIONS : VP_TERM_USER;
    r.n_evals = 1;
    r.objective = ok ? cost[b] : 0.0 / 0.0;
    rep[b] = r;
    if (trace && trace_rows > 0) { // vp_fit_trace: row 0 = [alpha_0, ||r||, ratio = NaN, delta, par] of the one evaluation
        double *tr = trace + (size_t)b * trace_rows * (q + 4);
        for (int k = 0; k < q; ++k)
            tr[k] = dtype == VP_F64 ? ((const double *)alpha)[b * q + k] : (double)((const float *)alpha)[b * q + k];
        tr[q] = ok ? sqrt(2.0 * cost[b]) : 0.0 / 0.0;
        tr[q + 1] = 0.0 / 0.0;
        tr[q + 2] = 0.0;
        tr[q + 3] = 0.0;
    }
}

// `blocks` blocks of the handle's m rows in device memory -> the caller's array of `blocks` blocks of ITS m rows
int copy_out_rows(vp_batch *h, void *user, const void *dev, size_t blocks) {
    if (!user) return 0;
    const size_t ts = tsize(h->dtype);
    if (!h->m_user) return copy_out(h, user, dev, blocks * (size_t)h->m * ts);
    const size_t bytes = blocks * (size_t)h->m_user * ts;
    void *packed = device_ptrs(h) ? user : nullptr;
    if (!packed) VP_HIP(hipMalloc(&packed, bytes ? bytes : 1));
    const int wu = (int)(h->m_user * ts / 4), wp = (int)(h->m * ts / 4);
    const int64_t total = (int64_t)blocks * wu;
    hipLaunchKernelGGL(strip_rows_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, h->stream,
                       (const unsigned *)dev, (unsigned *)packed, (int64_t)blocks, wu, wp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !device_ptrs(h)) {
        e = hipMemcpyAsync(user, packed, bytes, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (!device_ptrs(h)) (void)hipFree(packed); // the temporary of host-pointer handles, on every path
    if (e != hipSuccess) return fail(VP_ERR_HIP, std::string("copy_out_rows: ") + hipGetErrorString(e));
    return 0;
}

// an output with a row dimension: `blocks` blocks of m rows.  Not padded: OutBuf as is.  Padded: the kernels write a
// private array of padded blocks, finish() strips it into the caller's
struct RowOut {
    OutBuf buf;
    void *pad = nullptr;
    void *user = nullptr;
    size_t blocks = 0;
    void *dptr = nullptr;
    int init(vp_batch *h, void *user_, size_t blocks_) {
        user = user_;
        blocks = blocks_;
        if (!h->m_user) {
            const int rc = buf.init(h, user_, blocks_ * (size_t)h->m * tsize(h->dtype));
            dptr = buf.dptr;
            return rc;
        }
        if (!user) return 0;
        VP_HIP(hipMalloc(&pad, blocks * (size_t)h->m * tsize(h->dtype)));
        dptr = pad;
        return 0;
    }
    int finish(vp_batch *h) {
        if (!h->m_user) return buf.finish(h);
        if (!user) return 0;
        return copy_out_rows(h, user, pad, blocks);
    }
    ~RowOut() {
        if (pad) (void)hipFree(pad);
    }
};

int copy_out(vp_batch *h, void *user, const void *dev, size_t bytes) {
    if (!user) return 0;
    if (device_ptrs(h)) {
        VP_HIP(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToDevice, h->stream));
    } else {
        VP_HIP(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, h->stream));
        VP_HIP(hipStreamSynchronize(h->stream));
    }
    return 0;
}

void fill_params(vp_batch *h, LaunchParams &p) {
    std::memset(&p, 0, sizeof(p));
    p.model = &h->model;
    p.t = h->d_t;
    p.w = h->d_w;
    p.yw = h->d_yw;
    p.alpha = h->d_alpha;
    p.m = (int)h->m;
    p.S = (int)h->S;
    p.B = h->B;
    p.t_stride = (h->flags & VP_FLAG_T_PER_PROBLEM) ? h->m : 0;
    p.w_stride = (h->flags & VP_FLAG_W_PER_PROBLEM) ? h->m : 0;
    p.eps = h->eps;
    p.grid_uniform = h->grid_uniform ? 1 : 0;
    p.stream = h->stream;
    p.queue = h->d_queue;
    p.num_cus = h->num_cus;
    p.gen_ws = h->d_gen_ws;
    p.gen_blocks = h->gen_blocks;
    p.fit_group = h->fit_kernel;
    p.ext = h->external ? 1 : 0;
    p.ext_np = h->ext_np;
    p.ext_pb = h->ext_pb;
    p.ext_pp = h->ext_pp;
    p.ext_phi = h->ext_phi;
    p.ext_dphi = h->ext_dphi;
    p.ext_rows = (int)(h->m_user ? h->m_user : h->m);
}

struct Timer {
    vp_batch *h;
    int which;
    Timer(vp_batch *h_, int w) : h(h_), which(w) {
        if (h->timing) (void)hipEventRecord(h->ev0, h->stream);
    }
    void stop() {
        if (h->timing) {
            (void)hipEventRecord(h->ev1, h->stream);
            (void)hipEventSynchronize(h->ev1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
            h->last_ms[which] = ms;
        }
    }
};

int reduce_rhs(vp_batch *h, hipStream_t stream_override = nullptr, bool use_override = false) {
    if (h->S == 1) return 0; // d_cost / d_status alias the per-(b,s) arrays
    hipLaunchKernelGGL(reduce_rhs_kernel, dim3((unsigned)h->B), dim3(h->S > 2048 ? 1024 : 256), 0, use_override ? stream_override : h->stream, h->d_cost_bs,
                       h->d_status_bs, h->d_cost, h->d_status, (int)h->S, h->B);
    VP_HIP(hipGetLastError());
    return 0;
}

// workspace of the generic kernels (vp_generic.hpp): one slot of (n + 1 + p + q) columns per persistent workgroup; at most
// 1024 workgroups and 4 GiB
int ensure_gen_ws(vp_batch *h) {
    if (h->d_gen_ws || !h->kern->uses_gen_ws) return 0;
    const size_t slot = (size_t)(h->n + 1 + h->p + h->q) * (size_t)h->m * tsize(h->dtype);
    int64_t blocks = std::min<int64_t>(h->B * h->S, 1024);
    while (blocks > 1 && (size_t)blocks * slot > ((size_t)4 << 30)) blocks /= 2;
    h->gen_blocks = (int)blocks;
    VP_HIP(hipMalloc(&h->d_gen_ws, (size_t)blocks * slot));
    return 0;
}

int ensure_R(vp_batch *h) {
    if (!h->d_R) VP_HIP(hipMalloc(&h->d_R, (size_t)h->B * h->S * h->m * tsize(h->dtype)));
    return 0;
}

// run the evaluate kernel at h->d_alpha; any output may be null
int run_evaluate(vp_batch *h, void *r_dev, void *J_dev, void *C_dev) {
    if (h->external && !h->ext_phi)
        return fail(VP_ERR_INVALID, "the columns of the handle's current parameters are not known (after vp_fit_end: call "
                                    "vp_set_params_with_basis with Phi / dPhi at the fitted parameters first)");
    if (h->external && !h->d_gen_ws &&
        !external_resident(h->dtype, h->n, h->ext_np, h->m, h->m_user ? h->m_user : h->m, J_dev != nullptr))
        if (int rc = ensure_gen_ws(h)) return rc;
    LaunchParams p;
    fill_params(h, p);
    p.r_out = r_dev;
    p.J_out = J_dev;
    p.C_out = C_dev;
    p.cost_out = h->d_cost_bs;
    p.status = h->d_status_bs;
    Timer tm(h, VP_KERNEL_EVALUATE);
    int rc;
    if (h->have_mrhs) {
        // S > 1: factor Phi once per problem, then ONE streaming pass over the S data columns
        p.mrhs_ws = &h->mrhs;
        p.mrhs_mode = 1;
        rc = h->kern->mrhs_factor(p);
        if (rc == VP_ERR_OK) rc = h->kern->mrhs_stream(p);
    } else {
        rc = h->kern->evaluate(p);
    }
    tm.stop();
    if (rc != VP_ERR_OK) return fail(rc, "evaluate kernel launch failed");
    return reduce_rhs(h);
}

// Every entry point runs with the handle's device current and leaves the calling thread's current device as it found
// it (a process that drives several GPUs -- torch with another current device, say -- must not see it change).
struct DeviceGuard {
    int prev = -1;
    bool armed = false;
    int enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            VP_HIP(hipSetDevice(device));
            armed = prev >= 0;
        }
        return 0;
    }
    ~DeviceGuard() {
        if (armed) (void)hipSetDevice(prev);
    }
};

#define VP_ENTER(h)                                                                                                   \
    if (!(h)) return fail(VP_ERR_INVALID, "null handle");                                                             \
    DeviceGuard dev_guard__;                                                                                          \
    if (int rc__ = dev_guard__.enter((h)->device)) return rc__

// == LevMarSolver::fit for problems with multiple right-hand sides (global fit): host-stepped loop of
// {factor, streaming reduction over Y, LM step} launches; all LM state stays on the device, the host only
// reads back one int (number of still-active problems) per iteration.
int mrhs_fit(vp_batch *h, const vp_lm_opts *opts, void *alpha_inout, void *C_out, vp_report *rep, double *trace_out,
             int trace_rows) {
    if (!h->have_mrhs && !h->kern->mrhs_fit_whole) return fail(VP_ERR_UNSUPPORTED, "no MRHS kernels for this (model, m)");
    vp_lm_opts o;
    if (opts) o = *opts;
    else vp_lm_opts_default(&o, h->dtype);
    const size_t ts = tsize(h->dtype);
    const hipMemcpyKind kin = device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // device-pointer handles with the MRHS kernel set: the kernels read the initial parameters from, and write the results
    // into, the CALLER's arrays through a pinned record of their addresses (MrhsIo) -- no staging copies around the fit
    const bool use_io = device_ptrs(h) && h->have_mrhs;
    if (!h->h_nactive) {
        VP_HIP(hipHostMalloc((void **)&h->h_nactive, 2 * sizeof(int32_t), hipHostMallocMapped));
        VP_HIP(hipHostGetDevicePointer((void **)&h->h_nactive_dev, h->h_nactive, 0));
        VP_HIP(hipHostMalloc((void **)&h->h_io, sizeof(MrhsIo), hipHostMallocMapped));
        VP_HIP(hipHostGetDevicePointer((void **)&h->h_io_dev, h->h_io, 0));
        std::memset(h->h_io, 0, sizeof(MrhsIo));
    }
    if (use_io) {
        h->h_io->alpha_in = alpha_inout;
        h->h_io->alpha_out = alpha_inout;
        h->h_io->C_out = C_out;
        h->h_io->rep_out = rep;
    } else {
        std::memset(h->h_io, 0, sizeof(MrhsIo));
        VP_HIP(hipMemcpyAsync(h->d_alpha, alpha_inout, (size_t)h->B * h->q * ts, kin, h->stream));
    }
    LaunchParams p;
    fill_params(h, p);
    p.mrhs_ws = &h->mrhs;
    p.opts = &o;
    OutBuf tr;
    const size_t tr_bytes = (size_t)h->B * (size_t)(trace_rows > 0 ? trace_rows : 0) * (h->q + 4) * sizeof(double);
    if (trace_out && trace_rows > 0) {
        if (int rc = tr.init(h, trace_out, tr_bytes)) return rc;
        VP_HIP(hipMemsetAsync(tr.dptr, 0xFF, tr_bytes, h->stream));
        p.trace = (double *)tr.dptr;
        p.trace_rows = trace_rows;
    }
    if (!h->have_mrhs) {
        // generic fallback kernels: the whole global fit is one launch (vp_generic.hpp, gen_mrhs_fit_kernel), which also
        // leaves the coefficients / cost / status of every column at the final point
        p.alpha_out = h->d_alpha;
        p.C_out = h->d_C;
        p.cost_out = h->d_cost_bs;
        p.status = h->d_status_bs;
        p.report = h->d_report;
        if (h->rhs_allreduce) {
            // right-hand sides sharded over ranks: the same kernel in phases -- init, then per evaluation {sums of this
            // rank's columns, the caller's all-reduce of B*(2 + q*q + q) doubles, LM step on the totals (every rank takes
            // bit-identical decisions)}, then the per-column results at the final point
            const int nacc = 2 + h->q * h->q + h->q;
            if (!h->d_gen_lm) VP_HIP(hipMalloc(&h->d_gen_lm, (size_t)h->B * h->kern->mrhs_state_bytes));
            if (!h->d_mrhs_tot) VP_HIP(hipMalloc((void **)&h->d_mrhs_tot, (size_t)h->B * nacc * sizeof(double)));
            if (!h->d_gen_nactive) VP_HIP(hipMalloc((void **)&h->d_gen_nactive, sizeof(int32_t)));
            VP_HIP(hipMemsetAsync(h->d_gen_nactive, 0, sizeof(int32_t), h->stream));
            p.gen_lm_state = h->d_gen_lm;
            p.gen_acc = h->d_mrhs_tot;
            p.gen_nactive = h->d_gen_nactive;
            p.mrhs_S_global = h->rhs_global;
            Timer tm(h, VP_KERNEL_FIT);
            p.gen_phase = 1;
            if (int rc = h->kern->mrhs_fit_whole(p)) return fail(rc, "generic global-fit (init) launch failed");
            const int max_iter = o.patience * (h->q + 1) + 2;
            for (int it = 0; it < max_iter; ++it) {
                p.gen_phase = 2;
                if (int rc = h->kern->mrhs_fit_whole(p)) return fail(rc, "generic global-fit (sums) launch failed");
                if (h->rhs_allreduce(h->d_mrhs_tot, h->B * nacc, (void *)h->stream, h->rhs_allreduce_user) != 0)
                    return fail(VP_ERR_INVALID, "the right-hand-side all-reduce callback reported an error");
                p.gen_phase = 3;
                if (int rc = h->kern->mrhs_fit_whole(p)) return fail(rc, "generic global-fit (step) launch failed");
                if ((it & 3) == 3 || it + 1 == max_iter) {
                    int32_t nact = 0;
                    VP_HIP(hipMemcpyAsync(&nact, h->d_gen_nactive, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
                    VP_HIP(hipStreamSynchronize(h->stream));
                    if (nact <= 0) break;
                }
            }
            p.gen_phase = 4;
            if (int rc = h->kern->mrhs_fit_whole(p)) return fail(rc, "generic global-fit (results) launch failed");
            tm.stop();
        } else {
            Timer tm(h, VP_KERNEL_FIT);
            if (int rc = h->kern->mrhs_fit_whole(p)) return fail(rc, "generic global-fit launch failed");
            tm.stop();
        }
        if (int rc = reduce_rhs(h)) return rc;
        h->have_params = true;
        h->r_valid = false;
        h->have_report = true;
        if (int rc = copy_out(h, alpha_inout, h->d_alpha, (size_t)h->B * h->q * ts)) return rc;
        if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
        if (int rc = copy_out(h, rep, h->d_report, (size_t)h->B * sizeof(vp_report))) return rc;
        if (int rc = tr.finish(h)) return rc;
        return VP_ERR_OK;
    }
    Timer tm(h, VP_KERNEL_FIT);
    const int max_iter = o.patience * (h->q + 1) + 2;
    // right-hand sides sharded over ranks: the reduced sums of this rank's columns are totalled in a fixed order,
    // summed over the ranks by the caller's collective (RCCL all-reduce of B*(1+n*n+p) doubles per evaluation) and
    // fed to the LM step as a single slot; every rank then takes bit-identical decisions.
    const int nacc = 1 + h->n * h->n + h->p;
    MrhsWs ws_tot = h->mrhs;
    if (h->rhs_allreduce) {
        if (!h->d_mrhs_tot) VP_HIP(hipMalloc((void **)&h->d_mrhs_tot, (size_t)h->B * nacc * sizeof(double)));
        ws_tot.acc = h->d_mrhs_tot;
        p.mrhs_S_global = h->rhs_global;
    }
    // the first launch: LM state from alpha0 (the active count is SET there, no memset) + factorisation of alpha0
    auto enqueue_init = [&](LaunchParams &lp) -> int {
        lp.mrhs_init = 1;
        lp.alpha = h->d_alpha;
        lp.mrhs_io = h->h_io_dev;
        const int rc = h->kern->mrhs_lm(lp);
        lp.mrhs_init = 0;
        return rc ? fail(rc, "mrhs_step (init) launch failed") : 0;
    };
    // one LM iteration = TWO launches: the streaming pass at the trial point, then the LM step on its sums fused with the
    // factorisation of the next trial point (mrhs_step_kernel).  The loop is device-driven: iterations are ENQUEUED with
    // no host synchronisation in between -- a problem whose LM loop has terminated is skipped on the device
    // (MrhsWs::done), so an iteration enqueued past the end costs two empty launches (~10 us).
    auto enqueue_iteration = [&](LaunchParams &lp) -> int {
        lp.alpha = h->mrhs.alpha_trial;
        lp.mrhs_mode = 0;
        if (int rc = h->kern->mrhs_stream(lp)) return fail(rc, "mrhs_stream launch failed");
        if (h->rhs_allreduce) {
            const int64_t total = h->B * nacc;
            hipLaunchKernelGGL(mrhs_reduce_partials_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                               lp.stream, (const double *)h->mrhs.acc, mrhs_gx(h->S, h->kern->mrhs_gx_cap > 0 ? h->kern->mrhs_gx_cap : 256), nacc, h->B,
                               h->d_mrhs_tot);
            VP_HIP(hipGetLastError());
            if (h->rhs_allreduce(h->d_mrhs_tot, total, (void *)lp.stream, h->rhs_allreduce_user) != 0)
                return fail(VP_ERR_INVALID, "the right-hand-side all-reduce callback reported an error");
            lp.mrhs_ws = &ws_tot;
            lp.mrhs_fws = &h->mrhs;
            lp.mrhs_gx = 1;
        }
        // the LM step on the sums of this pass + the factorisation of the next trial point (one launch)
        if (int rc = h->kern->mrhs_lm(lp)) return fail(rc, "mrhs_step launch failed");
        lp.mrhs_ws = &h->mrhs;
        lp.mrhs_fws = nullptr;
        lp.mrhs_gx = 0;
        return 0;
    };
    // final parameters + reports; coefficients, cost and status of every column at the final point come from the best
    // point's pass (double-buffered per-column results, MrhsWs::cbuf) -- no further pass over Y.  The m x S residual
    // matrix is produced on demand by vp_residuals, as after a single-RHS fit.  The finish kernel also leaves the active
    // count in pinned host memory (no copy kernel for the host's look at it).
    auto enqueue_finish = [&](LaunchParams &lp) -> int {
        lp.alpha_out = h->d_alpha;
        lp.report = h->d_report;
        lp.C_out = h->d_C;
        lp.cost_out = h->d_cost_bs;
        lp.status = h->d_status_bs;
        lp.mrhs_hflag = h->h_nactive_dev;
        lp.mrhs_io = h->h_io_dev;
        if (int rc = h->kern->mrhs_finish(lp)) return fail(rc, "mrhs_finish launch failed");
        return reduce_rhs(h, lp.stream, true); // per-problem cost / status from the per-column ones
    };
    // The WHOLE fit as ONE captured HIP graph: init, `iters` iterations, finish (no trace, no all-reduce callback: both
    // put host state into the launch sequence).  Captured on a private stream -- the handle's stream may be the null
    // stream, which cannot be captured -- and replayed on the handle's stream.  The graph holds the LM options by
    // value: it is re-captured when they change.  Its length follows the handle's previous fit (evaluations + 1 spare
    // iteration, at least 6; 12 for the first fit): repeated fits of similar data -- the streaming use this path is
    // built for -- enqueue almost no idle iterations, and a fit that needs more continues with further replays of the
    // iteration-only tail graph.
    const bool want_graph = !p.trace && !h->rhs_allreduce && !h->mrhs_graph_failed;
    int want_iters = h->mrhs_graph_iters > 0 ? h->mrhs_graph_iters : 12;
    auto capture = [&](hipGraphExec_t &exec, const bool head, const int iters) -> bool {
        bool ok = true;
        if (!h->cap_stream) ok = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
        hipGraph_t g = nullptr;
        if (ok) ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            LaunchParams gp = p;
            gp.stream = h->cap_stream;
            if (head) ok = enqueue_init(gp) == 0;
            for (int it = 0; it < iters && ok; ++it) ok = enqueue_iteration(gp) == 0;
            if (ok) ok = enqueue_finish(gp) == 0;
            const bool ended = hipStreamEndCapture(h->cap_stream, &g) == hipSuccess;
            ok = ok && ended && g != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        if (!ok) {
            (void)hipGetLastError();
            exec = nullptr;
        }
        return ok;
    };
    // Re-capture with hysteresis: on streaming data the longest fit moves by +-1 evaluation from one call to the next, and a
    // capture + instantiation costs as much as the fit it speeds up.  The head is re-captured when the options changed, when
    // the wanted length EXCEEDS the captured one (every further iteration would cost a replay of the tail graph) or falls
    // short of it by 4 or more (each idle iteration is two empty launches, ~10 us); the 12-iteration tail graph does not
    // depend on the length and is only re-captured with the options.
    const bool opts_changed = !h->mrhs_graph || std::memcmp(&h->mrhs_graph_opts, &o, sizeof(o)) != 0;
    // (a handle whose last VP_MRHS_EXACT_AFTER + 1 fits took the same number of evaluations drops the spare iteration -- two
    // empty launches, ~10 us of a 0.7 ms fit: want_iters is then exact and the head is re-captured ONCE, down to that length.
    // The exact length only ever SHRINKS the graph: a stream whose longest fit moves by +-1 every few calls (A A A B A A A B)
    // never reaches the run length, and after a miss the spare-iteration rule above decides alone -- no recapture per
    // fluctuation.)
    const bool exact = h->mrhs_same_count >= VP_MRHS_EXACT_AFTER;
    if (want_graph && (opts_changed || want_iters > h->mrhs_graph_len || want_iters + 4 <= h->mrhs_graph_len ||
                       (exact && want_iters < h->mrhs_graph_len))) {
        if (h->mrhs_graph) (void)hipGraphExecDestroy(h->mrhs_graph);
        h->mrhs_graph = nullptr;
        bool ok = capture(h->mrhs_graph, true, want_iters);
        if (ok && (opts_changed || !h->mrhs_graph_tail)) {
            if (h->mrhs_graph_tail) (void)hipGraphExecDestroy(h->mrhs_graph_tail);
            h->mrhs_graph_tail = nullptr;
            ok = capture(h->mrhs_graph_tail, false, 12);
        }
        if (ok) {
            h->mrhs_graph_opts = o;
            h->mrhs_graph_len = want_iters;
        } else {
            if (h->mrhs_graph) (void)hipGraphExecDestroy(h->mrhs_graph);
            if (h->mrhs_graph_tail) (void)hipGraphExecDestroy(h->mrhs_graph_tail);
            h->mrhs_graph = h->mrhs_graph_tail = nullptr;
            h->mrhs_graph_failed = true; // fall back to plain launches for the rest of this handle's life
        }
    }
    // device-pointer handles: the copies into the caller's arrays are enqueued right behind the graph, BEFORE the host
    // waits for the active count (a fit that outlasts the graph repeats them after the tail graph)
    bool outputs_done = use_io; // (written by the finish / gather kernels themselves)
    auto early_outputs = [&]() -> int {
        if (outputs_done || !device_ptrs(h) || tr.tmp) return 0;
        if (int rc = copy_out(h, alpha_inout, h->d_alpha, (size_t)h->B * h->q * ts)) return rc;
        if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
        if (int rc = copy_out(h, rep, h->d_report, (size_t)h->B * sizeof(vp_report))) return rc;
        outputs_done = true;
        return 0;
    };
    if (want_graph && h->mrhs_graph) {
        VP_HIP(hipGraphLaunch(h->mrhs_graph, h->stream));
        if (int rc = early_outputs()) return rc;
        VP_HIP(hipStreamSynchronize(h->stream));
        for (int it = h->mrhs_graph_len; *(volatile int32_t *)h->h_nactive > 0 && it < max_iter; it += 12) {
            VP_HIP(hipGraphLaunch(h->mrhs_graph_tail, h->stream));
            if (int rc = early_outputs()) return rc;
            VP_HIP(hipStreamSynchronize(h->stream));
        }
        // the next capture: as many iterations as this fit's longest problem took evaluations, plus one spare
        const int nfev_max = ((volatile int32_t *)h->h_nactive)[1];
        h->mrhs_same_count = (nfev_max == h->mrhs_prev_nfev) ? h->mrhs_same_count + 1 : 0;
        h->mrhs_prev_nfev = nfev_max;
        int it_next = nfev_max + (h->mrhs_same_count >= VP_MRHS_EXACT_AFTER ? 0 : 1);
        it_next = it_next < 6 ? 6 : (it_next > 24 ? 24 : it_next);
        h->mrhs_graph_iters = it_next;
    } else {
        if (int rc = enqueue_init(p)) return rc;
        int next_check = 12;
        for (int it = 0; it < max_iter;) {
            if (int rc = enqueue_iteration(p)) return rc;
            ++it;
            if (it < next_check && it < max_iter) continue;
            next_check += 24;
            int32_t nact = 0;
            VP_HIP(hipMemcpyAsync(&nact, h->mrhs.nactive, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            VP_HIP(hipStreamSynchronize(h->stream));
            if (nact <= 0) break;
        }
        if (int rc = enqueue_finish(p)) return rc;
        // device-pointer handles: the finish / gather kernels read the caller's array addresses from the pinned MrhsIo record
        // at EXECUTION time and nothing below waits for them on this path (no copies: they write the results themselves) --
        // a following vp_fit would overwrite the record under them
        if (use_io) VP_HIP(hipStreamSynchronize(h->stream));
    }
    tm.stop();
    h->have_params = true;
    h->r_valid = false;
    h->have_report = true;
    if (!outputs_done) {
        if (int rc = copy_out(h, alpha_inout, h->d_alpha, (size_t)h->B * h->q * ts)) return rc;
        if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
        if (int rc = copy_out(h, rep, h->d_report, (size_t)h->B * sizeof(vp_report))) return rc;
    }
    if (int rc = tr.finish(h)) return rc;
    return VP_ERR_OK;
}

} // namespace

// ---- C ABI -------------------------------------------------------------------------------------------
extern "C" {

const char *vp_last_error(void) { return g_err.c_str(); }
int vp_last_error_detail(void) { return g_detail; }
const char *vp_version(void) { return "varpro_hip 0.1.0 (gfx950)"; }

int vp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void vp_lm_opts_default(vp_lm_opts *o, int dtype) {
    const double eps = dtype == VP_F32 ? (double)FLT_EPSILON : DBL_EPSILON;
    o->ftol = 30.0 * eps;
    o->xtol = 30.0 * eps;
    o->gtol = 30.0 * eps;
    o->stepbound = 100.0;
    o->patience = 100;
    o->scale_diag = 1;
}

// a caller-evaluated model: dependency pairs (basis, parameter) in the caller's order
struct ExtSpec {
    int np;
    const int32_t *pb, *pp;
};
static int batch_create_impl(vp_batch **out, const vp_model_desc *model, int dtype, int64_t m, int64_t S, int64_t B,
                             const void *t, const void *Y, const void *w, double svd_epsilon, int flags, int device,
                             void *hip_stream, bool data_on_device, const ExtSpec *ext = nullptr);

// pad the row dimension of `blocks` blocks from m to mp rows with `fill` (host arrays of 4- or 8-byte elements)
static void pad_rows_host(const void *src, void *dst, size_t blocks, int64_t m, int64_t mp, size_t ts, double fill) {
    for (size_t blk = 0; blk < blocks; ++blk) {
        std::memcpy((char *)dst + blk * mp * ts, (const char *)src + blk * m * ts, (size_t)m * ts);
        for (int64_t i = m; i < mp; ++i) {
            if (ts == 8) ((double *)dst)[blk * mp + i] = fill;
            else ((float *)dst)[blk * mp + i] = (float)fill;
        }
    }
}

int vp_batch_create(vp_batch **out, const vp_model_desc *model, int dtype, int64_t m, int64_t S, int64_t B,
                    const void *t, const void *Y, const void *w, double svd_epsilon, int flags, int device,
                    void *hip_stream) {
    const bool dev_data = (flags & VP_FLAG_DEVICE_PTRS) != 0;
    if (!out || !model || !Y || !t || m <= 0 || S <= 0 || B <= 0 || (dtype != VP_F64 && dtype != VP_F32) || m >= model->n_basis ||
        model->n_basis > VP_MAX_BASIS)
        return batch_create_impl(out, model, dtype, m, S, B, t, Y, w, svd_epsilon, flags, device, hip_stream, dev_data);
    // m < n: the reference solves the underdetermined linear sub-problem by its truncated SVD (minimum-norm coefficients,
    // src/solvers/levmar/mod.rs:51-54).  Here: n - m extra rows of ZERO weight (grid value = the last sample, data 0) --
    // they change neither the minimum-norm solution, nor the residual, nor a non-zero singular value -- and the handle
    // strips them from every array that crosses the ABI (m_user).  Tiny problems by construction: padded on the host.
    const size_t ts = dtype == VP_F64 ? 8 : 4;
    const int64_t mp = model->n_basis;
    const size_t tb = (flags & VP_FLAG_T_PER_PROBLEM) ? (size_t)B : 1, wb = (flags & VP_FLAG_W_PER_PROBLEM) ? (size_t)B : 1;
    std::vector<char> th(tb * m * ts), yh((size_t)B * S * m * ts), wh(w ? wb * m * ts : 0);
    if (dev_data) {
        if (vp_device_count() <= 0) return fail(VP_ERR_NO_DEVICE, "no HIP device visible");
        DeviceGuard g__;
        if (int rc = g__.enter(device)) return rc;
        hipStream_t st = (flags & VP_FLAG_OWN_STREAM) ? nullptr : (hipStream_t)hip_stream;
        VP_HIP(hipMemcpyAsync(th.data(), t, th.size(), hipMemcpyDeviceToHost, st));
        VP_HIP(hipMemcpyAsync(yh.data(), Y, yh.size(), hipMemcpyDeviceToHost, st));
        if (w) VP_HIP(hipMemcpyAsync(wh.data(), w, wh.size(), hipMemcpyDeviceToHost, st));
        VP_HIP(hipStreamSynchronize(st));
    } else {
        std::memcpy(th.data(), t, th.size());
        std::memcpy(yh.data(), Y, yh.size());
        if (w) std::memcpy(wh.data(), w, wh.size());
    }
    std::vector<char> tp(tb * mp * ts), yp((size_t)B * S * mp * ts), wp(wb * mp * ts);
    for (size_t blk = 0; blk < tb; ++blk) { // grid: repeat the last sample
        const double last = ts == 8 ? ((const double *)th.data())[blk * m + m - 1] : (double)((const float *)th.data())[blk * m + m - 1];
        pad_rows_host(th.data() + blk * m * ts, tp.data() + blk * mp * ts, 1, m, mp, ts, last);
    }
    pad_rows_host(yh.data(), yp.data(), (size_t)B * S, m, mp, ts, 0.0);
    if (w) {
        pad_rows_host(wh.data(), wp.data(), wb, m, mp, ts, 0.0);
    } else { // unit weights on the caller's rows, zero on the padding
        std::vector<char> ones(wb * m * ts);
        for (size_t i = 0; i < wb * (size_t)m; ++i) {
            if (ts == 8) ((double *)ones.data())[i] = 1.0;
            else ((float *)ones.data())[i] = 1.0f;
        }
        pad_rows_host(ones.data(), wp.data(), wb, m, mp, ts, 0.0);
    }
    const int rc = batch_create_impl(out, model, dtype, mp, S, B, tp.data(), yp.data(), wp.data(), svd_epsilon, flags, device, hip_stream,
                                     false);
    if (rc == VP_ERR_OK) (*out)->m_user = m;
    return rc;
}

// == SeparableProblemBuilder::build (src/problem/builder.rs:278-324) for a model the caller evaluates: any
// SeparableNonlinearModel (src/model/mod.rs:239-363), known here by its shape and dependency-pair table only
int vp_batch_create_external(vp_batch **out, int32_t n_basis, int32_t n_params, int32_t n_pairs, const int32_t *pair_basis,
                             const int32_t *pair_param, int dtype, int64_t m, int64_t S, int64_t B, const void *Y,
                             const void *w, double svd_epsilon, int flags, int device, void *hip_stream) {
    if (!out) return fail(VP_ERR_INVALID, "null output handle");
    *out = nullptr;
    if (n_basis <= 0 || n_basis > VP_MAX_BASIS || n_params < 0 || n_params > VP_MAX_PARAMS)
        return fail(VP_ERR_INVALID, "model sizes out of range");
    if (n_pairs < 0 || n_pairs > VP_MAX_PAIRS || (n_pairs > 0 && (!pair_basis || !pair_param)))
        return fail(VP_ERR_INVALID, "too many dependency pairs (or a null pair table)");
    for (int i = 0; i < n_pairs; ++i) {
        if (pair_basis[i] < 0 || pair_basis[i] >= n_basis || pair_param[i] < 0 || pair_param[i] >= n_params)
            return fail(VP_ERR_INVALID, "dependency pair out of range");
        for (int k = 0; k < i; ++k)
            if (pair_basis[k] == pair_basis[i] && pair_param[k] == pair_param[i])
                return fail(VP_ERR_INVALID, "dependency pair listed twice");
    }
    if (flags & VP_FLAG_T_PER_PROBLEM) return fail(VP_ERR_INVALID, "a caller-evaluated model has no grid");
    vp_model_desc md;
    std::memset(&md, 0, sizeof(md));
    md.n_basis = n_basis;
    md.n_params = n_params;
    for (int j = 0; j < VP_MAX_BASIS; ++j) {
        md.kind[j] = j < n_basis ? VP_BASIS_EXTERNAL : 0;
        for (int a = 0; a < VP_MAX_BASIS_PARAMS; ++a) md.param[j][a] = -1;
    }
    const ExtSpec ext{n_pairs, pair_basis, pair_param};
    const bool dev_data = (flags & VP_FLAG_DEVICE_PTRS) != 0;
    if (!Y || m <= 0 || S <= 0 || B <= 0 || (dtype != VP_F64 && dtype != VP_F32) || m >= n_basis)
        return batch_create_impl(out, &md, dtype, m, S, B, nullptr, Y, w, svd_epsilon, flags, device, hip_stream, dev_data, &ext);
    // m < n: as vp_batch_create -- n - m extra rows of zero weight; the caller's Phi / dPhi keep THEIR m rows
    // (LaunchParams::ext_rows), the kernels read the missing rows as zeros
    const size_t ts = dtype == VP_F64 ? 8 : 4;
    const int64_t mp = n_basis;
    const size_t wb = (flags & VP_FLAG_W_PER_PROBLEM) ? (size_t)B : 1;
    std::vector<char> yh((size_t)B * S * m * ts), wh(wb * m * ts);
    if (dev_data) {
        if (vp_device_count() <= 0) return fail(VP_ERR_NO_DEVICE, "no HIP device visible");
        DeviceGuard g__;
        if (int rc = g__.enter(device)) return rc;
        hipStream_t st = (flags & VP_FLAG_OWN_STREAM) ? nullptr : (hipStream_t)hip_stream;
        VP_HIP(hipMemcpyAsync(yh.data(), Y, yh.size(), hipMemcpyDeviceToHost, st));
        if (w) VP_HIP(hipMemcpyAsync(wh.data(), w, wh.size(), hipMemcpyDeviceToHost, st));
        VP_HIP(hipStreamSynchronize(st));
    } else {
        std::memcpy(yh.data(), Y, yh.size());
        if (w) std::memcpy(wh.data(), w, wh.size());
    }
    if (!w)
        for (size_t i = 0; i < wb * (size_t)m; ++i) {
            if (ts == 8) ((double *)wh.data())[i] = 1.0;
            else ((float *)wh.data())[i] = 1.0f;
        }
    std::vector<char> yp((size_t)B * S * mp * ts), wp(wb * mp * ts);
    pad_rows_host(yh.data(), yp.data(), (size_t)B * S, m, mp, ts, 0.0);
    pad_rows_host(wh.data(), wp.data(), wb, m, mp, ts, 0.0);
    const int rc = batch_create_impl(out, &md, dtype, mp, S, B, nullptr, yp.data(), wp.data(), svd_epsilon, flags, device,
                                     hip_stream, false, &ext);
    if (rc == VP_ERR_OK) (*out)->m_user = m;
    return rc;
}

static int batch_create_impl(vp_batch **out, const vp_model_desc *model, int dtype, int64_t m, int64_t S, int64_t B,
                             const void *t, const void *Y, const void *w, double svd_epsilon, int flags, int device,
                             void *hip_stream, const bool data_on_device, const ExtSpec *ext) {
    if (!out) return fail(VP_ERR_INVALID, "null output handle");
    *out = nullptr;
    if (!model) return fail(VP_ERR_INVALID, "null model");
    if (dtype != VP_F64 && dtype != VP_F32) return fail(VP_ERR_INVALID, "bad dtype");
    // builder validation (src/problem/builder.rs:278-302)
    if (!Y) return fail(VP_ERR_INVALID, "Right hand side(s) not provided", VP_BUILD_Y_DATA_MISSING);
    if (m <= 0 || S <= 0 || B <= 0 || (!t && !ext))
        return fail(VP_ERR_INVALID, "x or y must have nonzero number of elements.", VP_BUILD_ZERO_LENGTH_VECTOR);
    if (model->n_basis <= 0 || model->n_basis > VP_MAX_BASIS || model->n_params < 0 ||
        model->n_params > VP_MAX_PARAMS)
        return fail(VP_ERR_INVALID, "model sizes out of range");
    int fa, fb, fc, npairs;
    if (ext) npairs = ext->np; // caller-evaluated model: shape and pair table only, nothing to classify
    else if (classify_model(*model, fa, fb, fc, npairs) < 0) return fail(VP_ERR_INVALID, "malformed model descriptor");
    if (npairs > VP_MAX_PAIRS) return fail(VP_ERR_INVALID, "too many dependency pairs");

    int ndev = vp_device_count();
    if (ndev <= 0) return fail(VP_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(VP_ERR_INVALID, "bad device index");
    DeviceGuard dev_guard__;
    if (int rc__ = dev_guard__.enter(device)) return rc__;
    hipDeviceProp_t prop;
    VP_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950");

    // a specialised (register-resident) kernel set if one is instantiated for this (dtype, model, m), else the generic
    // fallback kernels (vp_generic.hpp): any descriptor, any m -- slower, but never a CPU path and never "unsupported"
    const KernelEntry *kern = ext ? external_kernels(dtype, model->n_basis, model->n_params, ext->np, m, S)
                                  : find_kernels(dtype, *model, m, S, w != nullptr);
    if (!ext && (flags & VP_FLAG_STREAM_ROWS)) { // only the length-agnostic sets (capacity 2^26 rows) pass this length
        const KernelEntry *ks = find_kernels(dtype, *model, (int64_t)1 << 24, S, w != nullptr);
        if (ks) kern = ks;
    }
    if (!kern) kern = generic_kernels(dtype);
    // a global fit (S > 1) on a specialised set WITHOUT multiple-right-hand-side kernels (the multi-wave sets: double
    // exponential at 2048 < m <= 4096, the fp32 Gram shape) runs on the generic kernels as well
    if (!ext && S > 1 && !(kern->mrhs_factor && kern->mrhs_stream && kern->mrhs_lm && kern->mrhs_finish) && !kern->mrhs_fit_whole)
        kern = generic_kernels(dtype);

    vp_batch *h = new vp_batch();
    std::memset(h, 0, sizeof(*h));
    h->model = *model;
    h->dtype = dtype;
    h->m = m;
    h->S = S;
    h->B = B;
    h->n = model->n_basis;
    h->q = model->n_params;
    h->p = npairs;
    h->flags = flags;
    h->device = device;
    h->num_cus = prop.multiProcessorCount;
    const double meps = dtype == VP_F32 ? (double)FLT_EPSILON : DBL_EPSILON;
    h->eps = svd_epsilon < 0 ? meps : std::fabs(svd_epsilon); // src/problem/builder.rs:246-251, 282
    h->kern = kern;
    if (ext) {
        h->external = true;
        h->ext_np = ext->np;
        for (int i = 0; i < ext->np; ++i) {
            h->ext_pb[i] = ext->pb[i];
            h->ext_pp[i] = ext->pp[i];
        }
    }
    if (!(flags & VP_FLAG_OWN_STREAM)) {
        h->stream = (hipStream_t)hip_stream; // NULL == the null stream (PyTorch's default stream)
        h->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete h;
            return fail(VP_ERR_HIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    const size_t ts = tsize(dtype);
    const size_t t_elems = (size_t)((flags & VP_FLAG_T_PER_PROBLEM) ? B * m : m);
    const size_t w_elems = (size_t)((flags & VP_FLAG_W_PER_PROBLEM) ? B * m : m);
    const size_t y_elems = (size_t)B * S * m;
#define VP_TRY(expr)                                                                                                  \
    do {                                                                                                              \
        hipError_t e__ = (expr);                                                                                      \
        if (e__ != hipSuccess) {                                                                                      \
            std::string msg__ = std::string(#expr) + ": " + hipGetErrorString(e__);                                 \
            vp_batch_destroy(h);                                                                                      \
            return fail(VP_ERR_HIP, msg__);                                                                           \
        }                                                                                                             \
    } while (0)
    const hipMemcpyKind kin = data_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (t) { // (a caller-evaluated model has no grid)
        VP_TRY(hipMalloc(&h->d_t, t_elems * ts));
        VP_TRY(hipMemcpyAsync(h->d_t, t, t_elems * ts, kin, h->stream));
    }
    if (w) {
        VP_TRY(hipMalloc(&h->d_w, w_elems * ts));
        VP_TRY(hipMemcpyAsync(h->d_w, w, w_elems * ts, kin, h->stream));
    }
    int *&d_gflag = reinterpret_cast<int *&>(h->tmp_a);
    // (fp32 handles: only the Gram fit kernel, vp_fitg.hpp, uses the flag -- the other fp32 kernels have no recurrence)
    const bool try_uniform = t && m >= 3 && !(flags & VP_FLAG_NO_GRID_RECURRENCE);
    if (try_uniform) {
        const int one = 1;
        VP_TRY(hipMalloc((void **)&d_gflag, sizeof(int)));
        VP_TRY(hipMemcpyAsync(d_gflag, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
        const int64_t ngrids = (flags & VP_FLAG_T_PER_PROBLEM) ? B : 1;
        if (dtype == VP_F64)
            hipLaunchKernelGGL(grid_check_kernel<double>, dim3((unsigned)ngrids), dim3(256), 0, h->stream,
                               (const double *)h->d_t, (int)m, ngrids, d_gflag);
        else
            hipLaunchKernelGGL(grid_check_kernel<float>, dim3((unsigned)ngrids), dim3(256), 0, h->stream,
                               (const float *)h->d_t, (int)m, ngrids, d_gflag);
        VP_TRY(hipGetLastError());
    }
    VP_TRY(hipMalloc(&h->d_yw, y_elems * ts));
    {
        // Y_w = W * Y
        void *&ytmp = h->tmp_b;
        const void *ysrc = Y;
        if (!data_on_device) {
            VP_TRY(hipMalloc(&ytmp, y_elems * ts));
            VP_TRY(hipMemcpyAsync(ytmp, Y, y_elems * ts, hipMemcpyHostToDevice, h->stream));
            ysrc = ytmp;
        }
        const int64_t total = (int64_t)y_elems;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
        const int64_t wstride = (flags & VP_FLAG_W_PER_PROBLEM) ? m : 0;
        if (dtype == VP_F64)
            hipLaunchKernelGGL(weight_data_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (const double *)ysrc,
                               (const double *)h->d_w, (double *)h->d_yw, (int)m, S, wstride, total);
        else
            hipLaunchKernelGGL(weight_data_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (const float *)ysrc,
                               (const float *)h->d_w, (float *)h->d_yw, (int)m, S, wstride, total);
        VP_TRY(hipGetLastError());
        int gflag = 0;
        if (d_gflag) VP_TRY(hipMemcpyAsync(&gflag, d_gflag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        VP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(ytmp);
        ytmp = nullptr;
        (void)hipFree(d_gflag);
        d_gflag = nullptr;
        h->grid_uniform = gflag != 0;
    }
    VP_TRY(hipMalloc(&h->d_alpha, (size_t)std::max<int64_t>(1, B * h->q) * ts));
    VP_TRY(hipMalloc(&h->d_C, (size_t)B * S * h->n * ts));
    VP_TRY(hipMalloc((void **)&h->d_cost_bs, (size_t)B * S * sizeof(double)));
    VP_TRY(hipMalloc((void **)&h->d_status_bs, (size_t)B * S * sizeof(int32_t)));
    if (S == 1) {
        h->d_cost = h->d_cost_bs;
        h->d_status = h->d_status_bs;
    } else {
        VP_TRY(hipMalloc((void **)&h->d_cost, (size_t)B * sizeof(double)));
        VP_TRY(hipMalloc((void **)&h->d_status, (size_t)B * sizeof(int32_t)));
    }
    VP_TRY(hipMalloc((void **)&h->d_report, (size_t)B * sizeof(vp_report)));
    VP_TRY(hipMalloc((void **)&h->d_sum4, 4 * sizeof(double)));
    VP_TRY(hipMalloc((void **)&h->d_queue, sizeof(int)));
    if (kern->uses_gen_ws && !ext) {
        // (caller-evaluated models allocate it on first use: their resident kernels -- vp_ext.hpp -- need none)
        if (int rc = ensure_gen_ws(h)) {
            vp_batch_destroy(h);
            return rc;
        }
    }
    VP_TRY(hipEventCreate(&h->ev0));
    VP_TRY(hipEventCreate(&h->ev1));
    if (S > 1 && kern->mrhs_factor && kern->mrhs_stream && kern->mrhs_lm && kern->mrhs_finish) {
        const int n_ = h->n, p_ = h->p, q_ = h->q;
        VP_TRY(hipMalloc(&h->mrhs.qthin, (size_t)B * n_ * m * ts));
        VP_TRY(hipMalloc(&h->mrhs.g, (size_t)B * std::max(1, p_) * m * ts));
        VP_TRY(hipMalloc((void **)&h->mrhs.small, (size_t)B * mrhs_small_stride_rt(n_, p_) * sizeof(double)));
        VP_TRY(hipMalloc((void **)&h->mrhs.statusA, (size_t)B * sizeof(int32_t)));
        // partial-sum slots of the MODE 0 pass (the only writer): gx is fixed per handle -- min(ceil(S/8), the kernel set's
        // cap) -- not the upper bound VP_MRHS_GX_MAX (B = 70000, S = 2 needs 1 slot per problem, not 512)
        const int gx_acc = mrhs_gx(S, kern->mrhs_gx_cap > 0 ? kern->mrhs_gx_cap : 256);
        if (gx_acc > VP_MRHS_GX_MAX) {
            vp_batch_destroy(h);
            return fail(VP_ERR_INVALID, "internal: partial-sum slots per problem exceed VP_MRHS_GX_MAX");
        }
        VP_TRY(hipMalloc((void **)&h->mrhs.acc, (size_t)B * gx_acc * (1 + n_ * n_ + p_) * sizeof(double)));
        VP_TRY(hipMalloc(&h->mrhs.lm_state, (size_t)B * kern->mrhs_state_bytes));
        VP_TRY(hipMalloc((void **)&h->mrhs.nactive, 2 * sizeof(int32_t)));
        VP_TRY(hipMalloc((void **)&h->mrhs.done, (size_t)B * sizeof(int32_t)));
        VP_TRY(hipMemsetAsync(h->mrhs.done, 0, (size_t)B * sizeof(int32_t), h->stream));
        VP_TRY(hipMalloc(&h->mrhs.alpha_trial, (size_t)std::max<int64_t>(1, B * q_) * ts));
        for (int i = 0; i < 2; ++i) {
            VP_TRY(hipMalloc(&h->mrhs.cbuf[i], (size_t)B * S * n_ * ts));
            VP_TRY(hipMalloc((void **)&h->mrhs.costbuf[i], (size_t)B * S * sizeof(double)));
            VP_TRY(hipMalloc((void **)&h->mrhs.stbuf[i], (size_t)B * S * sizeof(int32_t)));
        }
        VP_TRY(hipMalloc((void **)&h->mrhs.widx, (size_t)B * sizeof(int32_t)));
        VP_TRY(hipMalloc((void **)&h->mrhs.bidx, (size_t)B * sizeof(int32_t)));
        VP_TRY(hipMalloc((void **)&h->mrhs.jcond, (size_t)B * sizeof(double)));
        VP_TRY(hipMemsetAsync(h->mrhs.jcond, 0, (size_t)B * sizeof(double), h->stream));
        h->have_mrhs = true;
    }
#undef VP_TRY
    for (int k = 0; k < 3; ++k) h->last_ms[k] = -1.f;
    *out = h;
    return VP_ERR_OK;
}

void vp_batch_destroy(vp_batch *h) {
    if (!h) return;
    DeviceGuard dev_guard__;
    (void)dev_guard__.enter(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_mrhs_tot);
    (void)hipFree(h->d_gen_lm);
    (void)hipFree(h->d_gen_nactive);
    (void)hipFree(h->d_t);
    (void)hipFree(h->d_w);
    (void)hipFree(h->d_yw);
    (void)hipFree(h->d_alpha);
    (void)hipFree(h->d_C);
    (void)hipFree(h->d_R);
    (void)hipFree(h->d_cost_bs);
    (void)hipFree(h->d_status_bs);
    if (h->S != 1) {
        (void)hipFree(h->d_cost);
        (void)hipFree(h->d_status);
    }
    (void)hipFree(h->d_report);
    (void)hipFree(h->d_sum4);
    (void)hipFree(h->d_queue);
    (void)hipFree(h->d_gen_ws);
    (void)hipFree(h->tmp_a);
    (void)hipFree(h->tmp_b);
    (void)hipFree(h->ext_phi_own);
    (void)hipFree(h->ext_dphi_own);
    (void)hipFree(h->d_xf_state);
    (void)hipFree(h->d_xf_trial);
    (void)hipFree(h->d_xf_want);
    (void)hipFree(h->d_xf_nactive);
    (void)hipFree(h->d_xf_ctrial);
    (void)hipFree(h->d_xf_active);
    (void)hipFree(h->d_rescue);
    (void)hipFree(h->d_rescue_ws);
    if (h->h_xf_nactive) (void)hipHostFree(h->h_xf_nactive);
    // (the struct is zero-initialised: freeing unconditionally also covers a create that failed half way)
    (void)hipFree(h->mrhs.qthin);
    (void)hipFree(h->mrhs.g);
    (void)hipFree(h->mrhs.small);
    (void)hipFree(h->mrhs.statusA);
    (void)hipFree(h->mrhs.done);
    (void)hipFree(h->mrhs.acc);
    (void)hipFree(h->mrhs.lm_state);
    (void)hipFree(h->mrhs.nactive);
    (void)hipFree(h->mrhs.alpha_trial);
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(h->mrhs.cbuf[i]);
        (void)hipFree(h->mrhs.costbuf[i]);
        (void)hipFree(h->mrhs.stbuf[i]);
    }
    (void)hipFree(h->mrhs.widx);
    (void)hipFree(h->mrhs.jcond);
    (void)hipFree(h->mrhs.bidx);
    if (h->mrhs_graph) (void)hipGraphExecDestroy(h->mrhs_graph);
    if (h->mrhs_graph_tail) (void)hipGraphExecDestroy(h->mrhs_graph_tail);
    if (h->h_nactive) (void)hipHostFree(h->h_nactive);
    if (h->h_io) (void)hipHostFree(h->h_io);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// ---- caller-evaluated models (vp_batch_create_external) -----------------------------------------------------------
#define VP_NOT_EXTERNAL(h, what)                                                                                       \
    if ((h)->external)                                                                                                 \
    return fail(VP_ERR_UNSUPPORTED, what ": the handle's model is evaluated by the caller (vp_set_params_with_basis / "   \
                                         "vp_evaluate_with_basis)")

// where the kernels find the columns of the current parameters: the caller's device arrays, or staged copies of host arrays
static int ext_stage(vp_batch *h, const void *user, int64_t cols, const void *&dev, void *&own) {
    if (!user) {
        dev = nullptr;
        return 0;
    }
    if (device_ptrs(h)) {
        dev = user;
        return 0;
    }
    const size_t bytes = (size_t)h->B * (size_t)cols * (size_t)(h->m_user ? h->m_user : h->m) * tsize(h->dtype);
    if (!own) VP_HIP(hipMalloc(&own, bytes ? bytes : 1));
    VP_HIP(hipMemcpyAsync(own, user, bytes, hipMemcpyHostToDevice, h->stream));
    dev = own;
    return 0;
}

int vp_set_params_with_basis(vp_batch *h, const void *alpha, const void *Phi, const void *dPhi) {
    VP_ENTER(h);
    if (!h->external) return fail(VP_ERR_UNSUPPORTED, "vp_set_params_with_basis needs a handle made by vp_batch_create_external");
    if (!alpha || !Phi) return fail(VP_ERR_INVALID, "null alpha / Phi");
    VP_HIP(hipMemcpyAsync(h->d_alpha, alpha, (size_t)h->B * h->q * tsize(h->dtype),
                          device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    if (int rc = ext_stage(h, Phi, h->n, h->ext_phi, h->ext_phi_own)) return rc;
    if (int rc = ext_stage(h, dPhi, h->ext_np, h->ext_dphi, h->ext_dphi_own)) return rc;
    if (int rc = ensure_R(h)) return rc;
    if (int rc = run_evaluate(h, h->d_R, nullptr, h->d_C)) return rc;
    h->have_params = true;
    h->r_valid = true;
    if (!device_ptrs(h)) VP_HIP(hipStreamSynchronize(h->stream)); // the caller may reuse its host arrays
    return VP_ERR_OK;
}

static int copy_status(vp_batch *h, int32_t *status);

int vp_jacobian_with_derivatives(vp_batch *h, const void *dPhi, void *J_out, int32_t *status) {
    VP_ENTER(h);
    if (!h->external) return fail(VP_ERR_UNSUPPORTED, "vp_jacobian_with_derivatives needs a handle made by vp_batch_create_external");
    if (!h->have_params) return copy_status(h, status) ? VP_ERR_HIP : VP_ERR_OK; // jacobian() before set_params(): None
    if (!dPhi && h->ext_np > 0) return fail(VP_ERR_INVALID, "null dPhi");
    if (int rc = ext_stage(h, dPhi, h->ext_np, h->ext_dphi, h->ext_dphi_own)) return rc;
    return vp_jacobian(h, J_out, status);
}

int vp_evaluate_with_basis(vp_batch *h, const void *alpha, const void *Phi, const void *dPhi, void *r_out, void *J_out,
                           void *C_out, double *cost_out, int32_t *status) {
    VP_ENTER(h);
    if (!h->external) return fail(VP_ERR_UNSUPPORTED, "vp_evaluate_with_basis needs a handle made by vp_batch_create_external");
    if (!alpha || !Phi) return fail(VP_ERR_INVALID, "null alpha / Phi");
    if (J_out && !dPhi && h->ext_np > 0) return fail(VP_ERR_INVALID, "a Jacobian needs the derivative columns dPhi");
    const size_t ts = tsize(h->dtype);
    VP_HIP(hipMemcpyAsync(h->d_alpha, alpha, (size_t)h->B * h->q * ts,
                          device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    if (int rc = ext_stage(h, Phi, h->n, h->ext_phi, h->ext_phi_own)) return rc;
    if (int rc = ext_stage(h, dPhi, h->ext_np, h->ext_dphi, h->ext_dphi_own)) return rc;
    RowOut r, J;
    if (int rc = r.init(h, r_out, (size_t)h->B * h->S)) return rc;
    if (int rc = J.init(h, J_out, (size_t)h->B * h->q * h->S)) return rc;
    if (int rc = run_evaluate(h, r.dptr, J.dptr, h->d_C)) return rc;
    h->have_params = true;
    h->r_valid = false;
    if (int rc = r.finish(h)) return rc;
    if (int rc = J.finish(h)) return rc;
    if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
    if (int rc = copy_out(h, cost_out, h->d_cost, (size_t)h->B * sizeof(double))) return rc;
    if (int rc = copy_status(h, status)) return rc;
    if (!device_ptrs(h)) VP_HIP(hipStreamSynchronize(h->stream));
    return VP_ERR_OK;
}

// ---- batched LM fit of a caller-evaluated model by reverse communication (vp_extfit.hpp) --------------------------
// == LevMarSolver::fit (src/solvers/levmar/mod.rs:238-254) over the trait surface (src/model/mod.rs:239-363)
int vp_fit_begin(vp_batch *h, const vp_lm_opts *opts, const void *alpha0, int flags) {
    VP_ENTER(h);
    if (!h->external)
        return fail(VP_ERR_UNSUPPORTED, "vp_fit_begin needs a handle made by vp_batch_create_external (descriptor models: vp_fit)");
    if (!alpha0) return fail(VP_ERR_INVALID, "null alpha0");
    if (flags & ~VP_FIT_DERIVATIVES_ON_ACCEPT) return fail(VP_ERR_INVALID, "unknown vp_fit_begin flag");
    if (h->q <= 0) return fail(VP_ERR_INVALID, "a fit needs at least one nonlinear parameter");
    if (h->m_user) return fail(VP_ERR_UNSUPPORTED, "the batched fit of caller-evaluated models needs m >= n");
    const size_t rec = external_fit_rec_bytes(h->dtype, h->n, h->ext_np, h->q, h->m);
    if (!rec) return fail(VP_ERR_UNSUPPORTED, "no LM step kernel for this number of parameters");
    const size_t ts = tsize(h->dtype);
    if (external_fit_generic(h->dtype, h->n, h->ext_np, h->q, h->m, h->S)) {
        // shapes outside the specialised tables / several right-hand sides: the generic step and its workspace
        if (!h->d_gen_ws) {
            const size_t slot = (size_t)(h->n + 1 + h->ext_np + h->q) * (size_t)h->m * ts;
            int64_t blocks = std::min<int64_t>(h->B, 1024);
            while (blocks > 1 && (size_t)blocks * slot > ((size_t)4 << 30)) blocks /= 2;
            h->gen_blocks = (int)blocks;
            VP_HIP(hipMalloc(&h->d_gen_ws, (size_t)blocks * slot));
        }
        if (h->S > 1 && !h->d_xf_ctrial) VP_HIP(hipMalloc(&h->d_xf_ctrial, (size_t)h->B * h->S * h->n * ts));
    }
    // (each buffer guarded on its own: an allocation that fails half way leaves the handle in a state the next call completes)
    if (!h->d_xf_state) VP_HIP(hipMalloc(&h->d_xf_state, (size_t)h->B * rec + 16));
    if (!h->d_xf_trial) VP_HIP(hipMalloc(&h->d_xf_trial, (size_t)h->B * h->q * ts));
    if (!h->d_xf_want) VP_HIP(hipMalloc((void **)&h->d_xf_want, (size_t)h->B * sizeof(int32_t)));
    if (!h->d_xf_nactive) VP_HIP(hipMalloc((void **)&h->d_xf_nactive, 2 * sizeof(int32_t)));
    if (!h->h_xf_nactive) VP_HIP(hipHostMalloc((void **)&h->h_xf_nactive, sizeof(int32_t), hipHostMallocDefault));
    if (!h->d_xf_active) {
        // both lists start as the identity: an entry beyond a step's count is then always a valid (finished) problem index
        VP_HIP(hipMalloc((void **)&h->d_xf_active, (size_t)2 * h->B * sizeof(int32_t)));
        std::vector<int32_t> iota((size_t)2 * h->B);
        for (int64_t i = 0; i < 2 * h->B; ++i) iota[(size_t)i] = (int32_t)(i % h->B);
        VP_HIP(hipMemcpyAsync(h->d_xf_active, iota.data(), iota.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        VP_HIP(hipStreamSynchronize(h->stream));
    }
    h->xf_known_active = h->B;
    if (opts) h->xf_opts = *opts;
    else vp_lm_opts_default(&h->xf_opts, h->dtype);
    VP_HIP(hipMemcpyAsync(h->d_alpha, alpha0, (size_t)h->B * h->q * ts,
                          device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    if (!device_ptrs(h)) VP_HIP(hipStreamSynchronize(h->stream));
    VP_HIP(hipMemsetAsync(h->d_xf_nactive, 0, 2 * sizeof(int32_t), h->stream));
    h->xf_running = true;
    h->xf_init = true;
    h->xf_flags = flags;
    h->xf_steps = 0;
    h->have_params = false;
    h->r_valid = false;
    h->have_report = false;
    return VP_ERR_OK;
}

int vp_fit_step_with_basis(vp_batch *h, const void *Phi, const void *dPhi, void *alpha_trial_out, int32_t *want_out,
                           int64_t *n_active_out) {
    VP_ENTER(h);
    if (!h->external || !h->xf_running) return fail(VP_ERR_INVALID, "vp_fit_step_with_basis without vp_fit_begin");
    if (!Phi) return fail(VP_ERR_INVALID, "null Phi");
    const bool lazy = (h->xf_flags & VP_FIT_DERIVATIVES_ON_ACCEPT) != 0;
    if (!dPhi && h->ext_np > 0 && (!lazy || h->xf_init))
        return fail(VP_ERR_INVALID, "null dPhi (only a VP_FIT_DERIVATIVES_ON_ACCEPT fit may omit it, and not in its first step)");
    if (int rc = ext_stage(h, Phi, h->n, h->ext_phi, h->ext_phi_own)) return rc;
    if (int rc = ext_stage(h, dPhi, h->ext_np, h->ext_dphi, h->ext_dphi_own)) return rc;
    const size_t ts = tsize(h->dtype);
    const bool direct = device_ptrs(h); // the kernel writes the caller's device arrays itself
    ExtFitParams p;
    std::memset(&p, 0, sizeof(p));
    p.dtype = h->dtype;
    p.n = h->n;
    p.q = h->q;
    p.np = h->ext_np;
    p.m = h->m;
    p.B = h->B;
    p.phi = h->ext_phi;
    p.dphi = h->ext_dphi;
    p.w = h->d_w;
    p.yw = h->d_yw;
    p.w_stride = (h->flags & VP_FLAG_W_PER_PROBLEM) ? h->m : 0;
    p.state = h->d_xf_state;
    p.alpha0 = h->d_alpha;
    p.alpha_best = h->d_alpha;
    p.C_best = h->d_C;
    p.cost = h->d_cost;
    p.status = h->d_status;
    p.report = h->d_report;
    p.alpha_trial = (direct && alpha_trial_out) ? alpha_trial_out : h->d_xf_trial;
    p.want = (direct && want_out) ? want_out : h->d_xf_want;
    p.nactive = h->d_xf_nactive;
    p.step = (int)(h->xf_steps & 1);
    p.pb = h->ext_pb;
    p.pp = h->ext_pp;
    p.eps = h->eps;
    p.opts = h->xf_opts;
    p.init = h->xf_init ? 1 : 0;
    p.lazy = lazy ? 1 : 0;
    p.S = h->S;
    p.gen_ws = h->d_gen_ws;
    p.gen_blocks = h->gen_blocks;
    p.C_trial = h->d_xf_ctrial;
    p.active_lists = h->d_xf_active;
    p.known_active = h->xf_known_active;
    p.stream = h->stream;
    Timer tm(h, VP_KERNEL_FIT);
    const int rc = external_fit_step(p);
    tm.stop();
    if (rc != VP_ERR_OK) return fail(rc, "fit step kernel launch failed");
    h->xf_init = false;
    h->xf_steps += 1;
    if (!direct) {
        if (alpha_trial_out)
            VP_HIP(hipMemcpyAsync(alpha_trial_out, h->d_xf_trial, (size_t)h->B * h->q * ts, hipMemcpyDeviceToHost, h->stream));
        if (want_out)
            VP_HIP(hipMemcpyAsync(want_out, h->d_xf_want, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    }
    if (n_active_out) {
        VP_HIP(hipMemcpyAsync(h->h_xf_nactive, h->d_xf_nactive + ((h->xf_steps - 1) & 1), sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        VP_HIP(hipStreamSynchronize(h->stream));
        *n_active_out = *h->h_xf_nactive;
        h->xf_known_active = *h->h_xf_nactive; // the next evaluation launch shrinks to it
    } else if (!direct) {
        VP_HIP(hipStreamSynchronize(h->stream)); // host arrays: the caller reads them (and reuses Phi / dPhi) right away
    }
    return VP_ERR_OK;
}

int vp_fit_active_set(vp_batch *h, int32_t *index_out, int32_t *count_out) {
    VP_ENTER(h);
    if (!h->external || !h->xf_running) return fail(VP_ERR_INVALID, "vp_fit_active_set without vp_fit_begin");
    if (h->xf_init) return fail(VP_ERR_INVALID, "vp_fit_active_set before the first vp_fit_step_with_basis");
    const int slot = (int)((h->xf_steps - 1) & 1); // the list and the counter the last step's LM kernel wrote
    if (int rc = copy_out(h, index_out, h->d_xf_active + (size_t)slot * h->B, (size_t)h->B * sizeof(int32_t))) return rc;
    if (int rc = copy_out(h, count_out, h->d_xf_nactive + slot, sizeof(int32_t))) return rc;
    return VP_ERR_OK;
}

int vp_fit_end(vp_batch *h, void *alpha_out, void *C_out, vp_report *rep) {
    VP_ENTER(h);
    if (!h->external || !h->xf_running) return fail(VP_ERR_INVALID, "vp_fit_end without vp_fit_begin");
    if (h->xf_init) return fail(VP_ERR_INVALID, "vp_fit_end before the first vp_fit_step_with_basis");
    const size_t ts = tsize(h->dtype);
    h->xf_running = false;
    h->have_params = true;
    h->r_valid = false;
    h->have_report = true;
    // the columns the handle last saw belong to a trial point, not necessarily to the fitted one
    h->ext_phi = nullptr;
    h->ext_dphi = nullptr;
    if (int rc = copy_out(h, alpha_out, h->d_alpha, (size_t)h->B * h->q * ts)) return rc;
    if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
    if (int rc = copy_out(h, rep, h->d_report, (size_t)h->B * sizeof(vp_report))) return rc;
    return VP_ERR_OK;
}

int vp_set_params(vp_batch *h, const void *alpha) {
    VP_ENTER(h);
    VP_NOT_EXTERNAL(h, "vp_set_params");
    if (!alpha) return fail(VP_ERR_INVALID, "null alpha");
    const size_t bytes = (size_t)h->B * h->q * tsize(h->dtype);
    VP_HIP(hipMemcpyAsync(h->d_alpha, alpha, bytes, device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          h->stream));
    if (int rc = ensure_R(h)) return rc;
    if (int rc = run_evaluate(h, h->d_R, nullptr, h->d_C)) return rc;
    h->have_params = true;
    h->r_valid = true;
    if (!device_ptrs(h)) VP_HIP(hipStreamSynchronize(h->stream));
    return VP_ERR_OK;
}

int vp_params(vp_batch *h, void *alpha_out) {
    VP_ENTER(h);
    if (!h->have_params) return fail(VP_ERR_INVALID, "no parameters set yet");
    return copy_out(h, alpha_out, h->d_alpha, (size_t)h->B * h->q * tsize(h->dtype));
}

static int copy_status(vp_batch *h, int32_t *status) {
    if (!status) return 0;
    if (!h->have_params) {
        // residuals()/jacobian() before set_params: cached is None
        std::vector<int32_t> tmp((size_t)h->B, VP_ST_NOT_EVALUATED);
        if (device_ptrs(h)) {
            VP_HIP(hipMemcpyAsync(status, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, h->stream));
            VP_HIP(hipStreamSynchronize(h->stream));
        } else {
            std::memcpy(status, tmp.data(), tmp.size() * 4);
        }
        return 0;
    }
    return copy_out(h, status, h->d_status, (size_t)h->B * sizeof(int32_t));
}

int vp_residuals(vp_batch *h, void *r_out, int32_t *status) {
    VP_ENTER(h);
    if (!h->have_params) return copy_status(h, status) ? VP_ERR_HIP : VP_ERR_OK;
    if (!h->r_valid) {
        if (int rc = ensure_R(h)) return rc;
        if (int rc = run_evaluate(h, h->d_R, nullptr, nullptr)) return rc;
        h->r_valid = true;
    }
    if (int rc = copy_out_rows(h, r_out, h->d_R, (size_t)h->B * h->S)) return rc;
    return copy_status(h, status);
}

int vp_jacobian(vp_batch *h, void *J_out, int32_t *status) {
    VP_ENTER(h);
    if (!h->have_params) return copy_status(h, status) ? VP_ERR_HIP : VP_ERR_OK;
    if (!J_out) return fail(VP_ERR_INVALID, "null J_out");
    if (h->external && h->ext_np > 0 && !h->ext_dphi)
        return fail(VP_ERR_INVALID, "no derivative columns at the current parameters: pass dPhi to vp_set_params_with_basis "
                                    "or call vp_jacobian_with_derivatives");
    RowOut J;
    if (int rc = J.init(h, J_out, (size_t)h->B * h->q * h->S)) return rc;
    if (int rc = run_evaluate(h, nullptr, J.dptr, nullptr)) return rc;
    if (int rc = J.finish(h)) return rc;
    return copy_status(h, status);
}

int vp_linear_coeffs(vp_batch *h, void *C_out, int32_t *status) {
    VP_ENTER(h);
    if (!h->have_params) return copy_status(h, status) ? VP_ERR_HIP : VP_ERR_OK;
    if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * tsize(h->dtype))) return rc;
    return copy_status(h, status);
}

int vp_weighted_data(vp_batch *h, void *Yw_out) {
    VP_ENTER(h);
    return copy_out_rows(h, Yw_out, h->d_yw, (size_t)h->B * h->S);
}

int vp_set_observations(vp_batch *h, const void *Y) {
    VP_ENTER(h);
    if (!Y) return fail(VP_ERR_INVALID, "Right hand side(s) not provided", VP_BUILD_Y_DATA_MISSING);
    const size_t ts = tsize(h->dtype);
    const size_t y_elems = (size_t)h->B * h->S * h->m;
    // Y_w = W * Y straight into the handle's buffer; host pointers are staged through the buffer itself
    const void *ysrc = Y;
    std::vector<char> ypad;
    if (h->m_user) { // m < n: pad the caller's rows on the host (tiny by construction), then as a host array
        const size_t nb = (size_t)h->B * h->S;
        std::vector<char> yh(nb * h->m_user * ts);
        if (device_ptrs(h)) {
            VP_HIP(hipMemcpyAsync(yh.data(), Y, yh.size(), hipMemcpyDeviceToHost, h->stream));
            VP_HIP(hipStreamSynchronize(h->stream));
        } else {
            std::memcpy(yh.data(), Y, yh.size());
        }
        ypad.resize(y_elems * ts);
        pad_rows_host(yh.data(), ypad.data(), nb, h->m_user, h->m, ts, 0.0);
        VP_HIP(hipMemcpyAsync(h->d_yw, ypad.data(), y_elems * ts, hipMemcpyHostToDevice, h->stream));
        VP_HIP(hipStreamSynchronize(h->stream));
        ysrc = h->d_yw;
    } else if (!device_ptrs(h)) {
        VP_HIP(hipMemcpyAsync(h->d_yw, Y, y_elems * ts, hipMemcpyHostToDevice, h->stream));
        ysrc = h->d_yw; // in place: every element is read once and written once by the same thread
    }
    if (h->d_w || ysrc != h->d_yw) {
        const int64_t total = (int64_t)y_elems;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65536);
        const int64_t wstride = (h->flags & VP_FLAG_W_PER_PROBLEM) ? h->m : 0;
        if (h->dtype == VP_F64)
            hipLaunchKernelGGL(weight_data_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (const double *)ysrc,
                               (const double *)h->d_w, (double *)h->d_yw, (int)h->m, h->S, wstride, total);
        else
            hipLaunchKernelGGL(weight_data_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (const float *)ysrc,
                               (const float *)h->d_w, (float *)h->d_yw, (int)h->m, h->S, wstride, total);
        VP_HIP(hipGetLastError());
    }
    if (!device_ptrs(h)) VP_HIP(hipStreamSynchronize(h->stream)); // the caller may reuse its host buffer
    // the cached evaluation / fit belongs to the old data
    h->have_params = false;
    h->r_valid = false;
    h->have_report = false;
    return VP_ERR_OK;
}

int vp_cost(vp_batch *h, double *cost_out) {
    VP_ENTER(h);
    if (!h->have_params) return fail(VP_ERR_INVALID, "no parameters set yet");
    return copy_out(h, cost_out, h->d_cost, (size_t)h->B * sizeof(double));
}

int vp_evaluate(vp_batch *h, const void *alpha, void *r_out, void *J_out, void *C_out, double *cost_out,
                int32_t *status) {
    VP_ENTER(h);
    VP_NOT_EXTERNAL(h, "vp_evaluate");
    if (!alpha) return fail(VP_ERR_INVALID, "null alpha");
    const size_t ts = tsize(h->dtype);
    VP_HIP(hipMemcpyAsync(h->d_alpha, alpha, (size_t)h->B * h->q * ts,
                          device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    RowOut r, J;
    if (int rc = r.init(h, r_out, (size_t)h->B * h->S)) return rc;
    if (int rc = J.init(h, J_out, (size_t)h->B * h->q * h->S)) return rc;
    if (int rc = run_evaluate(h, r.dptr, J.dptr, h->d_C)) return rc;
    h->have_params = true;
    h->r_valid = false;
    if (int rc = r.finish(h)) return rc;
    if (int rc = J.finish(h)) return rc;
    if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
    if (int rc = copy_out(h, cost_out, h->d_cost, (size_t)h->B * sizeof(double))) return rc;
    return copy_status(h, status);
}

int vp_basis(vp_batch *h, const void *alpha, void *Phi_out, void *dPhi_out, int flags) {
    VP_ENTER(h);
    VP_NOT_EXTERNAL(h, "vp_basis");
    if (!alpha) return fail(VP_ERR_INVALID, "null alpha");
    const size_t ts = tsize(h->dtype);
    int ncols = 0;
    for (int j = 0; j < h->n; ++j)
        if (!((flags & VP_BASIS_SKIP_INVARIANT) && h->model.kind[j] == VP_BASIS_CONST)) ++ncols;
    InBuf a;
    if (int rc = a.init(h, alpha, (size_t)h->B * h->q * ts)) return rc;
    RowOut phi, dphi;
    if (int rc = phi.init(h, Phi_out, (size_t)h->B * ncols)) return rc;
    if (int rc = dphi.init(h, dPhi_out, (size_t)h->B * h->p)) return rc;
    LaunchParams p;
    fill_params(h, p);
    p.alpha = a.dptr;
    p.Phi_out = phi.dptr;
    p.dPhi_out = dphi.dptr;
    p.basis_flags = flags;
    Timer tm(h, VP_KERNEL_BASIS);
    int rc = h->kern->basis(p);
    tm.stop();
    if (rc != VP_ERR_OK) return fail(rc, "basis kernel launch failed");
    if (int rc2 = phi.finish(h)) return rc2;
    if (int rc2 = dphi.finish(h)) return rc2;
    if (!device_ptrs(h)) VP_HIP(hipStreamSynchronize(h->stream));
    return VP_ERR_OK;
}

int vp_fit(vp_batch *h, const vp_lm_opts *opts, void *alpha_inout, void *C_out, vp_report *rep) {
    return vp_fit_trace(h, opts, alpha_inout, C_out, rep, nullptr, 0);
}

// ---- flag-and-refit (round 6; vp_fit.hpp jac_not_finite) --------------------------------------------------------------
// A fit kernel that finds the Jacobian factor of an accepted point non-finite after an evaluation that was ok ends that fit,
// appends the problem to h->d_rescue and leaves its initial guess in h->d_alpha.  rescue_refit launches the generic fit
// kernel (any descriptor, any m, weights) over that list with power-of-two column scaling -- the reference's order of
// operations, D_k c first (src/solvers/levmar/mod.rs:156-171) -- from alpha0; it overwrites every output of those
// problems.  kRescueBlocks persistent workgroups: a launch with an empty list ends in the time of the launch itself
// (~3 us on the stream, no host synchronisation anywhere).
namespace {
constexpr int kRescueBlocks = 8;
int rescue_prepare(vp_batch *h, LaunchParams &p) {
    if (h->rescue_off || h->kern->family == FAMILY_GENERIC || h->kern->gram_fit) return 0; // (the generic kernel scales by itself)
    if (!h->d_rescue) {
        VP_HIP(hipMalloc((void **)&h->d_rescue, (size_t)(2 + h->B) * sizeof(int32_t)));
        VP_HIP(hipMemsetAsync(h->d_rescue, 0, 2 * sizeof(int32_t), h->stream));
        h->rescue_slot = 0;
    }
    if (!h->d_rescue_ws) {
        const size_t slot = (size_t)(h->n + 1 + h->p + h->q) * (size_t)h->m * tsize(h->dtype);
        VP_HIP(hipMalloc(&h->d_rescue_ws, (size_t)kRescueBlocks * slot));
    }
    p.rescue = h->d_rescue;
    p.rescue_slot = h->rescue_slot;
    return 0;
}
int rescue_refit(vp_batch *h, const LaunchParams &fit_params) {
    if (!fit_params.rescue) return 0;
    LaunchParams p = fit_params;
    // the first kFitRescueGrid flagged problems on the set's own wave-per-problem kernel with scaled derivative columns (a
    // flagged fit at ~4 us per evaluation: it must not outlast the batch it came from); whatever is left -- more problems
    // than that, weights, per-problem grids, models without that kernel -- on the generic kernel, which also zeroes the
    // list's other counter
    p.gen_list_first = 0;
    if (launch_fn fast = find_fit_rescue(h->kern->fit_single)) {
        const int rc = fast(p);
        if (rc == VP_ERR_OK) p.gen_list_first = kFitRescueGrid;
        else if (rc != VP_ERR_UNSUPPORTED) return rc;
    }
    p.rescue = nullptr;
    p.gen_ws = h->d_rescue_ws;
    p.gen_blocks = kRescueBlocks;
    p.gen_list = h->d_rescue;
    p.gen_list_slot = h->rescue_slot;
    p.gen_scale_cols = 1;
    h->rescue_slot ^= 1;
    return generic_kernels(h->dtype)->fit_single(p);
}
} // namespace

int vp_fit_trace(vp_batch *h, const vp_lm_opts *opts, void *alpha_inout, void *C_out, vp_report *rep,
                 double *trace_out, int trace_rows) {
    VP_ENTER(h);
    VP_NOT_EXTERNAL(h, "vp_fit");
    if (!alpha_inout) return fail(VP_ERR_INVALID, "null alpha");
    if (h->m_user && (int64_t)h->q > h->m_user * h->S) {
        // m < n AND fewer residuals than nonlinear parameters (the padded handle would count its own n rows): the reference's
        // LM driver evaluates once and reports WrongDimensions
        const size_t ts = tsize(h->dtype);
        VP_HIP(hipMemcpyAsync(h->d_alpha, alpha_inout, (size_t)h->B * h->q * ts,
                              device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        if (int rc = run_evaluate(h, nullptr, nullptr, h->d_C)) return rc;
        OutBuf trw;
        if (trace_out && trace_rows > 0) { // rows never written read as NaN; row 0 = the initial point (ratio = NaN)
            const size_t trb = (size_t)h->B * (size_t)trace_rows * (h->q + 4) * sizeof(double);
            if (int rc = trw.init(h, trace_out, trb)) return rc;
            VP_HIP(hipMemsetAsync(trw.dptr, 0xFF, trb, h->stream));
        }
        hipLaunchKernelGGL(wrong_dimensions_report_kernel, dim3((unsigned)((h->B + 255) / 256)), dim3(256), 0, h->stream,
                           (const double *)h->d_cost, (const int32_t *)h->d_status, h->B, h->d_report,
                           trw.dptr ? (double *)trw.dptr : nullptr, trace_rows, h->q, h->dtype, (const void *)h->d_alpha);
        VP_HIP(hipGetLastError());
        h->have_params = true;
        h->r_valid = false;
        h->have_report = true;
        if (int rc = copy_out(h, C_out, h->d_C, (size_t)h->B * h->S * h->n * ts)) return rc;
        if (int rc = copy_out(h, rep, h->d_report, (size_t)h->B * sizeof(vp_report))) return rc;
        return trw.finish(h);
    }
    if (h->S != 1) return mrhs_fit(h, opts, alpha_inout, C_out, rep, trace_out, trace_rows);
    if (!h->kern->fit && !h->kern->fit_single) return fail(VP_ERR_UNSUPPORTED, "no fit kernel for this model");
    vp_lm_opts o;
    if (opts) o = *opts;
    else vp_lm_opts_default(&o, h->dtype);
    const size_t ts = tsize(h->dtype);
    VP_HIP(hipMemcpyAsync(h->d_alpha, alpha_inout, (size_t)h->B * h->q * ts,
                          device_ptrs(h) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    LaunchParams p;
    fill_params(h, p);
    p.alpha_out = h->d_alpha;
    p.C_out = h->d_C;
    p.cost_out = h->d_cost_bs;
    p.status = h->d_status_bs;
    p.report = h->d_report;
    p.opts = &o;
    OutBuf tr;
    const size_t tr_bytes = (size_t)h->B * (size_t)(trace_rows > 0 ? trace_rows : 0) * (h->q + 4) * sizeof(double);
    if (trace_out && trace_rows > 0) {
        if (int rc = tr.init(h, trace_out, tr_bytes)) return rc;
        VP_HIP(hipMemsetAsync(tr.dptr, 0xFF, tr_bytes, h->stream)); // NaN-fill: unused rows read as NaN
        p.trace = (double *)tr.dptr;
        p.trace_rows = trace_rows;
    }
    // kern->fit is the persistent slot kernel (vp_fit2.hpp); it falls back to the one-problem-per-wave kernel
    // (vp_fit.hpp) by itself for the cases it does not cover (weights, per-problem grids, models without a trailing
    // constant column, batches smaller than the device's resident wave slots).  vp_set_fit_kernel overrides.
    launch_fn fit_fn = h->kern->fit ? h->kern->fit : h->kern->fit_single;
    if (int rc0 = rescue_prepare(h, p)) return rc0;
    int list_used = 0; // (set by the launcher when its kernel may append to the list)
    p.rescue_used = &list_used;
    Timer tm(h, VP_KERNEL_FIT);
    int rc = fit_fn(p);
    if (rc == VP_ERR_OK && list_used) rc = rescue_refit(h, p); // (the problems the fit kernel flagged and did not re-fit itself)
    tm.stop();
    if (rc != VP_ERR_OK) return fail(rc, "fit kernel launch failed");
    h->have_params = true;
    h->r_valid = false;
    h->have_report = true;
    if (int rc2 = copy_out(h, alpha_inout, h->d_alpha, (size_t)h->B * h->q * ts)) return rc2;
    if (int rc2 = copy_out(h, C_out, h->d_C, (size_t)h->B * h->n * ts)) return rc2;
    if (int rc2 = copy_out(h, rep, h->d_report, (size_t)h->B * sizeof(vp_report))) return rc2;
    if (int rc2 = tr.finish(h)) return rc2;
    return VP_ERR_OK;
}

int vp_debug_set_refit(vp_batch *h, int enabled) {
    VP_ENTER(h);
    h->rescue_off = enabled == 0;
    return VP_ERR_OK;
}

int vp_debug_gram_evaluate(vp_batch *h, const void *alpha, double *out) {
    VP_ENTER(h);
    if (!alpha || !out) return fail(VP_ERR_INVALID, "null argument");
    if (h->external || h->S != 1 || !h->kern->gram_fit || !h->kern->fit)
        return fail(VP_ERR_UNSUPPORTED, "the handle's fit does not run on the Gram kernel (fp32, exponentials + offset beyond one wavefront)");
    const size_t ts = tsize(h->dtype);
    const int per = 1 + h->n + h->q + h->q * h->q;
    InBuf a;
    if (int rc = a.init(h, alpha, (size_t)h->B * h->q * ts)) return rc;
    OutBuf o;
    if (int rc = o.init(h, out, (size_t)h->B * per * sizeof(double))) return rc;
    vp_lm_opts opt;
    vp_lm_opts_default(&opt, h->dtype);
    LaunchParams p;
    fill_params(h, p);
    p.alpha_out = const_cast<void *>(a.dptr); // read only in this mode
    p.opts = &opt;
    p.gram_dbg = (double *)o.dptr;
    if (int rc = h->kern->fit(p)) return fail(rc, "Gram evaluation launch failed");
    return o.finish(h);
}

extern "C++" {
namespace {
// one lane per record: the Gram fit kernel's trust-region sub-problem (lmpar_chol, vp_fit.hpp) exactly as
// slot_scalar_phase<..., GRAM> instantiates it
template <int Q>
__global__ void debug_lmpar_gram_kernel(int64_t B, const double *Rj, const int32_t *ipvt, const double *diag, const double *qtb,
                                        const double *delta, const double *par_in, double *out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double r[Q][Q], dg[Q], qb[Q], step[Q];
    int ip[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        dg[i] = diag[b * Q + i];
        qb[i] = qtb[b * Q + i];
        ip[i] = ipvt[b * Q + i];
        step[i] = 0.0;
#pragma unroll
        for (int j = 0; j < Q; ++j) r[i][j] = Rj[(b * Q + i) * Q + j];
    }
    double dxnorm = 0.0;
    const double par = vp::lmpar_chol<double, Q, false, (Q > 3)>(r, ip, dg, qb, delta[b], par_in[b], step, dxnorm);
    double *o = out + b * (Q + 2);
    o[0] = par;
    o[1] = dxnorm;
#pragma unroll
    for (int i = 0; i < Q; ++i) o[2 + i] = step[i];
}
} // namespace
} // extern "C++"

int vp_debug_lmpar_gram(int64_t B, int q, const double *Rj, const int32_t *ipvt, const double *diag, const double *qtb,
                        const double *delta, const double *par_in, double *out) {
    if (B <= 0 || !Rj || !ipvt || !diag || !qtb || !delta || !par_in || !out) return fail(VP_ERR_INVALID, "null argument");
    if (q != 2 && q != 3 && q != 5) return fail(VP_ERR_UNSUPPORTED, "q must be 2, 3 or 5");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VP_ERR_NO_DEVICE, "no device");
    const size_t nq = (size_t)B * q;
    const size_t sizes[7] = {nq * q * sizeof(double), nq * sizeof(int32_t), nq * sizeof(double), nq * sizeof(double),
                             (size_t)B * sizeof(double), (size_t)B * sizeof(double), (size_t)B * (q + 2) * sizeof(double)};
    const void *src[6] = {Rj, ipvt, diag, qtb, delta, par_in};
    void *d[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = VP_ERR_OK;
    for (int i = 0; i < 7 && rc == VP_ERR_OK; ++i)
        if (hipMalloc(&d[i], sizes[i]) != hipSuccess) rc = VP_ERR_HIP;
    for (int i = 0; i < 6 && rc == VP_ERR_OK; ++i)
        if (hipMemcpy(d[i], src[i], sizes[i], hipMemcpyHostToDevice) != hipSuccess) rc = VP_ERR_HIP;
    if (rc == VP_ERR_OK) {
        const dim3 grid((unsigned)((B + 63) / 64)), block(64);
        auto go = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            hipLaunchKernelGGL((debug_lmpar_gram_kernel<Q>), grid, block, 0, 0, B, (const double *)d[0], (const int32_t *)d[1],
                               (const double *)d[2], (const double *)d[3], (const double *)d[4], (const double *)d[5], (double *)d[6]);
        };
        if (q == 2) go(std::integral_constant<int, 2>());
        else if (q == 3) go(std::integral_constant<int, 3>());
        else go(std::integral_constant<int, 5>());
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d[6], sizes[6], hipMemcpyDeviceToHost) != hipSuccess) rc = VP_ERR_HIP;
    }
    for (int i = 0; i < 7; ++i)
        if (d[i]) (void)hipFree(d[i]);
    return rc == VP_ERR_OK ? VP_ERR_OK : fail(rc, "vp_debug_lmpar_gram: HIP call failed");
}

int vp_set_rhs_allreduce(vp_batch *h, vp_allreduce_fn fn, void *user, int64_t global_rhs_count) {
    VP_ENTER(h);
    if (fn) {
        if (h->S <= 1 || (!h->have_mrhs && !h->kern->mrhs_fit_whole))
            return fail(VP_ERR_UNSUPPORTED, "right-hand-side sharding needs a handle with S > 1");
        if (global_rhs_count < h->S) return fail(VP_ERR_INVALID, "global_rhs_count smaller than the local S");
    }
    h->rhs_allreduce = fn;
    h->rhs_allreduce_user = user;
    h->rhs_global = fn ? global_rhs_count : 0;
    return VP_ERR_OK;
}

int vp_best_fit(vp_batch *h, void *fit_out) {
    VP_ENTER(h);
    if (!h->have_params) return fail(VP_ERR_INVALID, "no parameters set yet");
    if (!h->kern->best_fit) return fail(VP_ERR_UNSUPPORTED, "no best_fit kernel for this model");
    RowOut f;
    if (int rc = f.init(h, fit_out, (size_t)h->B * h->S)) return rc;
    LaunchParams p;
    fill_params(h, p);
    p.C_out = h->d_C; // input here
    p.r_out = f.dptr; // output
    int rc = h->kern->best_fit(p);
    if (rc != VP_ERR_OK) return fail(rc, "best_fit kernel launch failed");
    return f.finish(h);
}

int vp_statistics(vp_batch *h, void *cov_out, double *reduced_chi2_out, void *conf_sigma_out, int32_t *status) {
    VP_ENTER(h);
    if (!h->have_params) return fail(VP_ERR_INVALID, "no parameters set yet");
    if (h->S != 1) // src/solvers/levmar/mod.rs:271-273: statistics are single-RHS only
        return fail(VP_ERR_UNSUPPORTED, "fit statistics are only supported for a single right-hand side");
    if (!h->kern->stats) return fail(VP_ERR_UNSUPPORTED, "no statistics kernel for this model");
    if (h->external && h->ext_np > 0 && !h->ext_dphi)
        return fail(VP_ERR_INVALID, "fit statistics need the derivative columns at the current parameters (vp_set_params_with_basis with dPhi)");
    if (int rc = ensure_gen_ws(h)) return rc;
    if (!cov_out || !reduced_chi2_out) return fail(VP_ERR_INVALID, "null output");
    const size_t ts = tsize(h->dtype);
    const int k = h->n + h->q;
    OutBuf cov, chi2, st;
    RowOut sig;
    if (int rc = cov.init(h, cov_out, (size_t)h->B * k * k * ts)) return rc;
    if (int rc = chi2.init(h, reduced_chi2_out, (size_t)h->B * sizeof(double))) return rc;
    if (int rc = sig.init(h, conf_sigma_out, (size_t)h->B)) return rc;
    void *st_tmp = nullptr;
    int32_t *st_dev = status && device_ptrs(h) ? status : nullptr;
    if (!st_dev) {
        VP_HIP(hipMalloc(&st_tmp, (size_t)h->B * sizeof(int32_t)));
        st_dev = (int32_t *)st_tmp;
    }
    LaunchParams p;
    fill_params(h, p);
    p.C_out = h->d_C;
    p.cost_out = h->d_cost;
    p.status = h->d_status;
    p.Phi_out = cov.dptr;
    p.dPhi_out = chi2.dptr;
    p.r_out = sig.dptr;
    p.J_out = st_dev;
    int rc = h->kern->stats(p);
    if (rc == VP_ERR_OK) rc = cov.finish(h);
    if (rc == VP_ERR_OK) rc = chi2.finish(h);
    if (rc == VP_ERR_OK) rc = sig.finish(h);
    if (rc == VP_ERR_OK && status && !device_ptrs(h)) {
        hipError_t e = hipMemcpyAsync(status, st_dev, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = VP_ERR_HIP;
    }
    if (st_tmp) (void)hipFree(st_tmp);
    if (rc != VP_ERR_OK) return fail(rc, "statistics kernel failed");
    return VP_ERR_OK;
}

int vp_summary(vp_batch *h, double out[4]) {
    VP_ENTER(h);
    if (!h->have_report) return fail(VP_ERR_INVALID, "vp_summary requires a completed vp_fit");
    VP_HIP(hipMemsetAsync(h->d_sum4, 0, 4 * sizeof(double), h->stream));
    const unsigned grid = (unsigned)std::min<int64_t>((h->B + 255) / 256, 1024);
    hipLaunchKernelGGL(summary_kernel, dim3(grid), dim3(256), 0, h->stream, h->d_report, h->B, h->d_sum4);
    VP_HIP(hipGetLastError());
    VP_HIP(hipMemcpyAsync(out, h->d_sum4, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    VP_HIP(hipStreamSynchronize(h->stream));
    return VP_ERR_OK;
}

// cond(J D^-1) as the Gram-based LM step of the last global fit saw it at its worst (MrhsWs::jcond)
int vp_global_fit_condition(vp_batch *h, double *cond_out) {
    VP_ENTER(h);
    if (!cond_out) return fail(VP_ERR_INVALID, "null output");
    if (!h->have_mrhs || !h->mrhs.jcond || !h->have_report)
        return fail(VP_ERR_INVALID, "vp_global_fit_condition requires a completed vp_fit on a handle with several right-hand sides "
                                    "(specialised kernel set)");
    return copy_out(h, cond_out, h->mrhs.jcond, (size_t)h->B * sizeof(double));
}

int vp_summary_device(vp_batch *h, double *dev_out4) {
    VP_ENTER(h);
    if (!h->have_report) return fail(VP_ERR_INVALID, "vp_summary_device requires a completed vp_fit");
    if (!dev_out4) return fail(VP_ERR_INVALID, "null output");
    VP_HIP(hipMemsetAsync(dev_out4, 0, 4 * sizeof(double), h->stream));
    const unsigned grid = (unsigned)std::min<int64_t>((h->B + 255) / 256, 1024);
    hipLaunchKernelGGL(summary_kernel, dim3(grid), dim3(256), 0, h->stream, h->d_report, h->B, dev_out4);
    VP_HIP(hipGetLastError());
    return VP_ERR_OK;
}

// == the vp_reduce_cost of SURVEY.md 8(b): local aggregates + ONE RCCL all-reduce of 4 doubles on the handle's stream.
// RCCL is resolved at run time (the library itself does not link it).
typedef int (*vp_nccl_allreduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
static vp_nccl_allreduce_t resolve_nccl_allreduce() {
    static vp_nccl_allreduce_t fn = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllReduce"); // the host process links / has loaded RCCL
        if (!sym) {
            void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) sym = dlsym(lib, "ncclAllReduce");
        }
        fn = reinterpret_cast<vp_nccl_allreduce_t>(sym);
    }
    return fn;
}

int vp_reduce_cost(vp_batch *h, void *rccl_comm, double out[4]) {
    VP_ENTER(h);
    if (!out) return fail(VP_ERR_INVALID, "null output");
    if (!h->have_report) return fail(VP_ERR_INVALID, "vp_reduce_cost requires a completed vp_fit");
    VP_HIP(hipMemsetAsync(h->d_sum4, 0, 4 * sizeof(double), h->stream));
    const unsigned grid = (unsigned)std::min<int64_t>((h->B + 255) / 256, 1024);
    hipLaunchKernelGGL(summary_kernel, dim3(grid), dim3(256), 0, h->stream, h->d_report, h->B, h->d_sum4);
    VP_HIP(hipGetLastError());
    if (rccl_comm) {
        vp_nccl_allreduce_t ar = resolve_nccl_allreduce();
        if (!ar) return fail(VP_ERR_UNSUPPORTED, "ncclAllReduce not found: link librccl or make librccl.so.1 loadable");
        const int rc = ar(h->d_sum4, h->d_sum4, 4, /*ncclDouble*/ 8, /*ncclSum*/ 0, rccl_comm, h->stream);
        if (rc != 0) return fail(VP_ERR_HIP, "ncclAllReduce failed with code " + std::to_string(rc));
    }
    VP_HIP(hipMemcpyAsync(out, h->d_sum4, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    VP_HIP(hipStreamSynchronize(h->stream));
    return VP_ERR_OK;
}

int vp_set_fit_kernel(vp_batch *h, int which) {
    if (!h) return fail(VP_ERR_INVALID, "null handle");
    if (which != VP_FIT_KERNEL_AUTO && which != VP_FIT_KERNEL_WAVE && which != VP_FIT_KERNEL_SLOTS)
        return fail(VP_ERR_INVALID, "vp_set_fit_kernel: unknown kernel selector");
    h->fit_kernel = which;
    return VP_ERR_OK;
}

int vp_set_timing(vp_batch *h, int enable) {
    if (!h) return fail(VP_ERR_INVALID, "null handle");
    h->timing = enable != 0;
    return VP_ERR_OK;
}

int vp_last_kernel_ms(vp_batch *h, int which, float *ms) {
    if (!h || !ms || which < 0 || which > 2) return fail(VP_ERR_INVALID, "bad argument");
    *ms = h->last_ms[which];
    return VP_ERR_OK;
}

int vp_synchronize(vp_batch *h) {
    VP_ENTER(h);
    VP_HIP(hipStreamSynchronize(h->stream));
    return VP_ERR_OK;
}

} // extern "C"

// ---- registry ------------------------------------------------------------------------------------------
namespace vp {

std::vector<RescueEntry> &rescue_registry() {
    static std::vector<RescueEntry> r;
    return r;
}
std::vector<KernelEntry> &registry() {
    static std::vector<KernelEntry> r;
    return r;
}

static int kind_arity(int kind) {
    switch (kind) {
    case VP_BASIS_CONST: return 0;
    case VP_BASIS_EXP_DECAY:
    case VP_BASIS_EXP_RATE: return 1;
    case VP_BASIS_EXP_COS:
    case VP_BASIS_SIN_PHASE: return 2;
    default: return -1; // (VP_BASIS_EXTERNAL never reaches a descriptor: vp_batch_create_external only)
    }
}

int classify_model(const vp_model_desc &d, int &a, int &b, int &c, int &p_out) {
    int pairs = 0;
    for (int j = 0; j < d.n_basis; ++j) {
        const int ar = kind_arity(d.kind[j]);
        if (ar < 0) return -1;
        for (int k = 0; k < VP_MAX_BASIS_PARAMS; ++k) {
            const int pi = d.param[j][k];
            if (k < ar) {
                if (pi < 0 || pi >= d.n_params) return -1;
                ++pairs;
            } else if (pi >= 0) {
                return -1;
            }
        }
    }
    p_out = pairs;
    // multi-exponential family: exp(-t/alpha_j) for j = 0..q-1 in order, optional trailing constant
    bool multiexp = d.n_params >= 1 && (d.n_basis == d.n_params || d.n_basis == d.n_params + 1);
    if (multiexp) {
        for (int j = 0; j < d.n_params; ++j)
            if (d.kind[j] != VP_BASIS_EXP_DECAY || d.param[j][0] != j) multiexp = false;
        if (multiexp && d.n_basis == d.n_params + 1 && d.kind[d.n_params] != VP_BASIS_CONST) multiexp = false;
    }
    if (multiexp) {
        a = d.n_params;
        b = d.n_basis - d.n_params;
        c = 0;
        return FAMILY_MULTIEXP;
    }
    a = d.n_basis;
    b = d.n_params;
    c = pairs;
    return FAMILY_RT;
}

const KernelEntry *find_kernels(int dtype, const vp_model_desc &d, int64_t m, int64_t S, bool weighted) {
    int a, b, c, p;
    const int fam = classify_model(d, a, b, c, p);
    if (fam < 0) return nullptr;
    const KernelEntry *best = nullptr;
    for (int pass = 0; pass < 2 && !best; ++pass) {
        // pass 0: exact family; pass 1: a multi-exponential model may also run on the runtime-model kernels
        int f = fam, ka = a, kb = b, kc = c;
        if (pass == 1) {
            if (fam != FAMILY_MULTIEXP) break;
            f = FAMILY_RT;
            ka = d.n_basis;
            kb = d.n_params;
            kc = p;
        }
        for (const KernelEntry &e : registry()) {
            if (e.dtype != dtype || e.family != f || e.a != ka || e.b != kb || e.c != kc) continue;
            if ((int64_t)64 * e.R * e.W < m) continue;
            if (S == 1 && !e.evaluate) continue; // (a set that only carries multiple-right-hand-side kernels)
            // weighted problems: grid + weights + data column of 64 R W rows each must fit the 160 KiB of LDS the
            // one-problem-per-group fit kernel stages them in (KernelEntry::fit_lds_w: launch_fit's own expression)
            if (weighted && e.fit_lds_w > (size_t)160 * 1024) continue;
            // smallest capacity first.  Among equal capacities: a multiple-RHS handle takes the set that has MRHS kernels;
            // a single-RHS handle the one whose columns fit the registers (R <= 16 rows per lane -- the R = 32 sets spill
            // hundreds of VGPRs: triple exponential at m = 2048, 0.71 vs 0.10 ms per 16384 evaluations at m = 1024), else
            // the fewest waves per problem
            auto better = [&](const KernelEntry &x, const KernelEntry &y) {
                if (x.R * x.W != y.R * y.W) return x.R * x.W < y.R * y.W;
                if (S > 1 && (x.mrhs_stream != nullptr) != (y.mrhs_stream != nullptr)) return x.mrhs_stream != nullptr;
                if (S == 1 && (x.R <= 16) != (y.R <= 16)) return x.R <= 16;
                return x.W < y.W;
            };
            if (!best || better(e, *best)) best = &e;
        }
    }
    return best;
}

} // namespace vp
