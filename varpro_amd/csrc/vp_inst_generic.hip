// generic fallback kernels (any descriptor, any m; vp_generic.hpp), f64 and f32
#include "vp_generic.hpp"
#include "vp_registry.hpp"

namespace vp {
const KernelEntry *generic_kernels(int dtype) {
    static const KernelEntry f64{VP_F64, FAMILY_GENERIC, 0, 0, 0, 0, 1, &gen::launch_evaluate<double>, &gen::launch_basis<double>, nullptr,
                                 &gen::launch_fit<double>, &gen::launch_best_fit<double>, nullptr, nullptr, nullptr, nullptr, gen::mrhs_lm_state_bytes<double>(), &gen::launch_stats<double>, &gen::launch_mrhs_fit<double>, 0, 0, 0, 1};
    static const KernelEntry f32{VP_F32, FAMILY_GENERIC, 0, 0, 0, 0, 1, &gen::launch_evaluate<float>, &gen::launch_basis<float>, nullptr,
                                 &gen::launch_fit<float>, &gen::launch_best_fit<float>, nullptr, nullptr, nullptr, nullptr, gen::mrhs_lm_state_bytes<float>(), &gen::launch_stats<float>, &gen::launch_mrhs_fit<float>, 0, 0, 0, 1};
    return dtype == VP_F64 ? &f64 : &f32;
}
} // namespace vp
