// kernel set of caller-evaluated models (vp_batch_create_external): the resident evaluate kernels of vp_ext.hpp where the
// shape is in their table, the generic kernels (vp_generic.hpp, reading the caller's columns) everywhere else
#include <cstring>

#include "vp_ext.hpp"
#include "vp_extfit.hpp"
#include "vp_generic.hpp"
#include "vp_gen_extfit.hpp"
#include "vp_registry.hpp"
#include "vp_extfit_api.hpp"

namespace vp {
namespace {
int ext_evaluate_f64(const LaunchParams &p) { return ext::launch_evaluate<double>(p, &gen::launch_evaluate<double>); }
int ext_evaluate_f32(const LaunchParams &p) { return ext::launch_evaluate<float>(p, &gen::launch_evaluate<float>); }
} // namespace

const KernelEntry *external_kernels(int dtype, int n, int q, int np, int64_t m, int64_t S) {
    (void)n;
    (void)q;
    (void)np;
    (void)m;
    (void)S;
    // (no basis / fit entries: the device cannot evaluate the model; vp_api.hip refuses those calls)
    static const KernelEntry f64{VP_F64, FAMILY_GENERIC, 0, 0, 0, 0, 1, &ext_evaluate_f64, nullptr, nullptr, nullptr,
                                 &gen::launch_best_fit<double>, nullptr, nullptr, nullptr, nullptr, 0, &gen::launch_stats<double>, nullptr, 0, 0, 0, 1};
    static const KernelEntry f32{VP_F32, FAMILY_GENERIC, 0, 0, 0, 0, 1, &ext_evaluate_f32, nullptr, nullptr, nullptr,
                                 &gen::launch_best_fit<float>, nullptr, nullptr, nullptr, nullptr, 0, &gen::launch_stats<float>, nullptr, 0, 0, 0, 1};
    return dtype == VP_F64 ? &f64 : &f32;
}

namespace {
template <typename T> int ext_fit_step_t(const ExtFitParams &p) {
    const ext::ExtFitEntry<T> *e = p.S > 1 ? nullptr : ext::find_extfit<T>(p.n, p.np, p.q, p.m);
    const ext::ExtFitLmEntry<T> *l = ext::find_extfit_lm<T>(p.q);
    if (!l) return VP_ERR_UNSUPPORTED;
    if (!e && (!p.gen_ws || (p.S > 1 && !p.C_trial))) return VP_ERR_INVALID;
    ext::ExtFitArgs<T> a;
    a.S = (int)(p.S > 1 ? p.S : 1);
    a.C_trial = (const T *)p.C_trial;
    // the evaluation covers the active set the previous step's LM kernel compacted (none before the first step)
    const bool have_list = !p.init && p.active_lists != nullptr;
    a.active_in = have_list ? p.active_lists + (size_t)((p.step + 1) & 1) * (size_t)p.B : nullptr;
    a.active_count = have_list ? p.nactive + ((p.step + 1) & 1) : nullptr;
    a.active_out = p.active_lists ? p.active_lists + (size_t)(p.step & 1) * (size_t)p.B : nullptr;
    a.grid_problems = have_list ? (p.known_active < p.B ? (p.known_active > 0 ? p.known_active : 1) : p.B) : p.B;
    a.phi = (const T *)p.phi;
    a.dphi = (const T *)p.dphi;
    a.w = (const T *)p.w;
    a.yw = (const T *)p.yw;
    a.state = p.state;
    a.alpha0 = (const T *)p.alpha0;
    a.alpha_best = (T *)p.alpha_best;
    a.C_best = (T *)p.C_best;
    a.cost = p.cost;
    a.status = p.status;
    a.report = p.report;
    a.alpha_trial = (T *)p.alpha_trial;
    a.want = p.want;
    a.nactive = p.nactive;
    a.step = p.step;
    // sweep order: the invariant basis functions (no derivative pair) first, each group in the model's order (ExtFitArgs::perm)
    int inv[VP_MAX_BASIS];
    {
        bool varies[VP_MAX_BASIS] = {};
        for (int i = 0; i < p.np; ++i)
            if (p.pb[i] >= 0 && p.pb[i] < VP_MAX_BASIS) varies[p.pb[i]] = true;
        int k = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < p.n && j < VP_MAX_BASIS; ++j)
                if (varies[j] == (pass == 1)) a.perm[k++] = j;
        for (; k < VP_MAX_BASIS; ++k) a.perm[k] = k;
        for (int j = 0; j < VP_MAX_BASIS; ++j) inv[a.perm[j]] = j;
    }
    for (int i = 0; i < VP_MAX_PAIRS; ++i) {
        a.pb[i] = i < p.np ? inv[p.pb[i]] : 0;
        a.pp[i] = i < p.np ? p.pp[i] : -1;
    }
    a.np = p.np;
    a.n = p.n;
    a.m = (int)p.m;
    a.B = p.B;
    a.w_stride = p.w_stride;
    a.eps = (T)p.eps;
    a.o.ftol = (T)p.opts.ftol;
    a.o.xtol = (T)p.opts.xtol;
    a.o.gtol = (T)p.opts.gtol;
    a.o.stepbound = (T)p.opts.stepbound;
    a.o.patience = p.opts.patience;
    a.o.scale_diag = p.opts.scale_diag;
    a.init = p.init;
    a.lazy = p.lazy;
    a.vec = host_aligned<T>((int)p.m, {p.phi, p.dphi, p.w, p.yw}) ? 1 : 0;
    if (a.B <= 0) return VP_ERR_OK;
    // the evaluation of every active problem (one wavefront each), then the LM drivers (one lane each)
    if (e) {
        if (int rc = e->launch(a, p.stream)) return rc;
    } else {
        // no specialised kernel for this (n, pairs, q, m) or several right-hand sides: the generic step (vp_gen_extfit.hpp)
        gen::GenExtFitArgs<T> x;
        std::memset(&x, 0, sizeof(x));
        x.g.mdl.n_basis = p.n;
        x.g.mdl.n_params = p.q;
        x.g.P = p.np;
        for (int i = 0; i < p.np && i < VP_MAX_PAIRS; ++i) {
            x.g.pb[i] = p.pb[i];
            x.g.pa[i] = 0;
            x.g.pp[i] = p.pp[i];
        }
        x.g.ext = 1;
        x.g.ext_phi = (const T *)p.phi;
        x.g.ext_dphi = (const T *)p.dphi;
        x.g.ext_rows = (int)p.m;
        x.g.w = (const T *)p.w;
        x.g.yw = (const T *)p.yw;
        x.g.S = a.S;
        x.g.ws = (T *)p.gen_ws;
        x.g.ws_cols = p.n + 1 + p.np + p.q;
        x.g.m = (int)p.m;
        x.g.B = p.B;
        x.g.w_stride = p.w_stride;
        x.g.eps = (T)p.eps;
        x.state = p.state;
        x.C_trial = (T *)p.C_trial;
        x.init = p.init;
        x.q = p.q;
        x.active_in = a.active_in;
        x.active_count = a.active_count;
        const int blocks = (int)(a.grid_problems < p.gen_blocks ? a.grid_problems : p.gen_blocks);
        const size_t lds = gen::gen_lds_for<T>(x.g, &gen::gen_extfit_eval_kernel<T>);
        hipLaunchKernelGGL((gen::gen_extfit_eval_kernel<T>), dim3((unsigned)blocks), dim3(gen::TB), lds, p.stream, x);
        if (hipGetLastError() != hipSuccess) return VP_ERR_HIP;
    }
    return l->launch(a, p.stream);
}
} // namespace

// (every admitted shape has a step: the LM kernel of its q, and the generic evaluation where no specialised one exists)
size_t external_fit_rec_bytes(int dtype, int n, int np, int q, int64_t m) {
    (void)n;
    (void)np;
    (void)m;
    if (dtype == VP_F64) {
        const ext::ExtFitLmEntry<double> *l = ext::find_extfit_lm<double>(q);
        return l ? l->rec_bytes : 0;
    }
    const ext::ExtFitLmEntry<float> *l = ext::find_extfit_lm<float>(q);
    return l ? l->rec_bytes : 0;
}
bool external_fit_generic(int dtype, int n, int np, int q, int64_t m, int64_t S) {
    if (S > 1) return true;
    return dtype == VP_F64 ? ext::find_extfit<double>(n, np, q, m) == nullptr : ext::find_extfit<float>(n, np, q, m) == nullptr;
}

int external_fit_step(const ExtFitParams &p) {
    return p.dtype == VP_F64 ? ext_fit_step_t<double>(p) : ext_fit_step_t<float>(p);
}

bool external_resident(int dtype, int n, int np, int64_t m, int64_t ext_rows, bool with_d) {
    if (dtype != VP_F64 || ext_rows != m || m < n || m > (1 << 20)) return false;
    return ext::find_ext<double>(n, np, (int)m, with_d && np > 0) != nullptr;
}
} // namespace vp
