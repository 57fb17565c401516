// vp_registry.hpp -- table of compiled kernel instantiations and its lookup.
#pragma once
#include <vector>

#include "vp_kernels.hpp"

namespace vp {

typedef int (*launch_fn)(const LaunchParams &);

enum { FAMILY_MULTIEXP = 1, FAMILY_RT = 2, FAMILY_GENERIC = 3 };

struct KernelEntry {
    int dtype;  // VP_F64 / VP_F32
    int family; // FAMILY_*
    int a, b, c; // MULTIEXP: (nexp, offset, 0)   RT: (n, q, p)
    int R;      // rows per lane
    int W;      // waves per problem; handles m <= 64*R*W
    launch_fn evaluate;
    launch_fn basis;
    launch_fn fit;      // persistent slot LM kernel (vp_fit2.hpp; falls back to fit_single by itself); may be null
    launch_fn fit_single; // one-wavefront(-group)-per-problem LM (vp_fit.hpp)
    launch_fn best_fit; // may be null
    // multiple-right-hand-side path (vp_mrhs.hpp); all null if not instantiated
    launch_fn mrhs_factor, mrhs_stream, mrhs_lm, mrhs_finish;
    size_t mrhs_state_bytes;
    launch_fn stats; // batched fit statistics (vp_stats.hpp)
    launch_fn mrhs_fit_whole; // S > 1 fit as ONE launch (generic fallback kernels only; null elsewhere)
    int gram_fit;       // 1: `fit` is the fp64-Gram kernel (vp_fitg.hpp), which also serves vp_debug_gram_evaluate
    int mrhs_gx_cap;    // MRHS streaming kernel: max workgroups (partial-sum slots) per problem; 0 = 256
    size_t fit_lds_w;   // dynamic LDS the single-RHS fit kernel needs for a WEIGHTED problem (0: not recorded, fits)
    int uses_gen_ws;    // 1: evaluate / stats / global fits of this set run on the generic kernels (vp_generic.hpp) and need
                        // their global-memory workspace (the generic set itself; the length-agnostic sets of vp_block.hpp)
};

std::vector<KernelEntry> &registry();

struct Registrar {
    explicit Registrar(const KernelEntry &e) { registry().push_back(e); }
};

// the fast re-fit launch of a set's flagged problems (fit_kernel<..., RESCUE>, vp_fit.hpp), keyed by the set's fit_single
// launcher (a side table: KernelEntry initialisers stay as they are)
struct RescueEntry {
    launch_fn fit_single, rescue;
};
std::vector<RescueEntry> &rescue_registry();
struct RescueRegistrar {
    RescueRegistrar(launch_fn fit_single, launch_fn rescue) { rescue_registry().push_back(RescueEntry{fit_single, rescue}); }
};
inline launch_fn find_fit_rescue(launch_fn fit_single) {
    for (const RescueEntry &e : rescue_registry())
        if (e.fit_single == fit_single) return e.rescue;
    return nullptr;
}

// classify a public descriptor; returns family and its key (a,b,c); p_out = number of dependency pairs
int classify_model(const vp_model_desc &d, int &a, int &b, int &c, int &p_out);

// S: right-hand sides of the handle -- among sets of equal capacity a single-RHS handle prefers the one whose columns fit
// the registers (R <= 16: more waves per problem), a multiple-RHS handle the one that has MRHS kernels
// weighted: sets whose fit kernel cannot stage grid + weights + data of a weighted problem in the CU's LDS are skipped
const KernelEntry *find_kernels(int dtype, const vp_model_desc &d, int64_t m, int64_t S = 1, bool weighted = false);
// kernel set of a caller-evaluated model (vp_batch_create_external) of shape (n, q, np pairs)
const KernelEntry *external_kernels(int dtype, int n, int q, int np, int64_t m, int64_t S);
// true: an evaluation of this shape runs on a register-resident kernel (vp_ext.hpp) and needs no generic workspace
bool external_resident(int dtype, int n, int np, int64_t m, int64_t ext_rows, bool with_d);
// the generic fallback set (vp_generic.hpp): any descriptor, any m, single right-hand side fits
const KernelEntry *generic_kernels(int dtype);

} // namespace vp
