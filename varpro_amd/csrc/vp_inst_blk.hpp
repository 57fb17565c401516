// vp_inst_blk.hpp -- registration of the LENGTH-AGNOSTIC kernel sets (vp_block.hpp): the set a single-RHS fit lands on when
// no register-resident set is long enough (capacity 2^26 rows: it loses against every resident set that covers m and wins
// where none does).  the trait-level evaluation streams as well (blk_evaluate_kernel); basis / best_fit / statistics / global fits run on the generic kernels.
#pragma once
#include "vp_block.hpp"
#include "vp_generic.hpp"
#include "vp_registry.hpp"

#define VP_BLK_CAT_(a, b) a##b
#define VP_BLK_CAT(a, b) VP_BLK_CAT_(a, b)
namespace vp {
namespace blk {
template <typename T, class M> int launch_evaluate_entry(const LaunchParams &p) {
    return launch_evaluate<T, M>(p, &::vp::gen::launch_evaluate<T>);
}
} // namespace blk
} // namespace vp
#define VP_BLK_ENTRY(T, DT, FAM, KA, KB, KC, MODEL)                                                                    \
    static ::vp::Registrar VP_BLK_CAT(vp_blk_reg_, __COUNTER__)(::vp::KernelEntry{                                    \
        DT, FAM, KA, KB, KC, 1 << 20, 1, &::vp::blk::launch_evaluate_entry<T, MODEL>, &::vp::gen::launch_basis<T>, nullptr, \
        &::vp::blk::launch_fit<T, MODEL>, &::vp::gen::launch_best_fit<T>, nullptr, nullptr, nullptr, nullptr,          \
        ::vp::gen::mrhs_lm_state_bytes<T>(), &::vp::gen::launch_stats<T>, &::vp::gen::launch_mrhs_fit<T>, 0, 0, 0, 1});
#define VP_REGISTER_BLOCKED_MULTIEXP(T, DT, NEXP, OFF)                                                                 \
    VP_BLK_ENTRY(T, DT, ::vp::FAMILY_MULTIEXP, NEXP, OFF, 0, VP_BLK_ME(NEXP, OFF))                                     \
    static ::vp::RescueRegistrar VP_BLK_CAT(vp_blk_resc_, __COUNTER__)(                                               \
        &::vp::blk::launch_fit<T, VP_BLK_ME(NEXP, OFF)>,                                                              \
        ::vp::fit_rescue_v<T, VP_BLK_ME(NEXP, OFF)> ? &::vp::blk::launch_fit_rescue<T, VP_BLK_ME(NEXP, OFF)> : nullptr);
#define VP_BLK_ME(NEXP, OFF) ::vp::MultiExpModel<NEXP, (OFF) != 0>
#define VP_REGISTER_BLOCKED_RT(T, DT, NN, QQ, PP) VP_BLK_ENTRY(T, DT, ::vp::FAMILY_RT, NN, QQ, PP, VP_BLK_RT(NN, QQ, PP))
#define VP_BLK_RT(NN, QQ, PP) ::vp::RtModel<NN, QQ, PP>
