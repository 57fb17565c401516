// kernel set of caller-evaluated models (vp_batch_create_external): the resident evaluate kernels of vp_ext.hpp where the
// shape is in their table, the generic kernels (vp_generic.hpp, reading the caller's columns) everywhere else
#include "vp_ext.hpp"
#include "vp_generic.hpp"
#include "vp_registry.hpp"

namespace vp {
namespace {
int ext_evaluate_f64(const LaunchParams &p) { return ext::launch_evaluate<double>(p, &gen::launch_evaluate<double>); }
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
    static const KernelEntry f32{VP_F32, FAMILY_GENERIC, 0, 0, 0, 0, 1, &gen::launch_evaluate<float>, nullptr, nullptr, nullptr,
                                 &gen::launch_best_fit<float>, nullptr, nullptr, nullptr, nullptr, 0, &gen::launch_stats<float>, nullptr, 0, 0, 0, 1};
    return dtype == VP_F64 ? &f64 : &f32;
}

bool external_resident(int dtype, int n, int np, int64_t m, int64_t ext_rows, bool with_d) {
    if (dtype != VP_F64 || ext_rows != m || m < n || m > (1 << 20)) return false;
    return ext::find_ext<double>(n, np, (int)m, with_d && np > 0) != nullptr;
}
} // namespace vp
