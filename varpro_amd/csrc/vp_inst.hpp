// vp_inst.hpp -- macros that instantiate one (dtype, model, R) kernel set and register it.
#pragma once
#include "vp_fit2.hpp"
#include "vp_fitg.hpp"
#include "vp_lm_core.hpp"
#include "vp_mrhs.hpp"
#include "vp_stats.hpp"
#include "vp_registry.hpp"

#define VP_CAT_(a, b) a##b
#define VP_CAT(a, b) VP_CAT_(a, b)

#define VP_REGISTER_MULTIEXP(T, DT, NEXP, OFF, RR)                                                                     \
    static ::vp::Registrar VP_CAT(vp_reg_, __COUNTER__)(::vp::KernelEntry{                                            \
        DT, ::vp::FAMILY_MULTIEXP, NEXP, OFF, 0, RR, 1, &::vp::launch_evaluate<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>, \
        &::vp::launch_basis<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                            \
        &::vp::launch_fit2<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                           \
        &::vp::launch_fit<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                              \
        &::vp::launch_best_fit<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                         \
        &::vp::launch_mrhs_factor<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                      \
        &::vp::launch_mrhs_stream<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                      \
        &::vp::launch_mrhs_lm<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                          \
        &::vp::launch_mrhs_finish<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                      \
        ::vp::mrhs_state_bytes<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>>(),                                          \
        &::vp::launch_stats<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>, nullptr, 0, ::vp::mrhs_gx_cap<T, RR>()});            \
    VP_REGISTER_FIT_RESCUE(T, NEXP, OFF, RR, 1)

// the scaled re-fit kernel of the set's flagged problems (models with an offset: fit_rescue_v)
#define VP_REGISTER_FIT_RESCUE(T, NEXP, OFF, RR, WW)                                                                   \
    static ::vp::RescueRegistrar VP_CAT(vp_resc_, __COUNTER__)(                                                       \
        &::vp::launch_fit<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                          \
        ::vp::fit_rescue_v<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>>                                                  \
            ? &::vp::launch_fit_rescue<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>                              \
            : nullptr);

#define VP_REGISTER_RT(T, DT, NN, QQ, PP, RR)                                                                          \
    static ::vp::Registrar VP_CAT(vp_reg_, __COUNTER__)(::vp::KernelEntry{                                            \
        DT, ::vp::FAMILY_RT, NN, QQ, PP, RR, 1, &::vp::launch_evaluate<T, ::vp::RtModel<NN, QQ, PP>, RR>,             \
        &::vp::launch_basis<T, ::vp::RtModel<NN, QQ, PP>, RR>, &::vp::launch_fit2<T, ::vp::RtModel<NN, QQ, PP>, RR>, \
        &::vp::launch_fit<T, ::vp::RtModel<NN, QQ, PP>, RR>,                                                          \
        &::vp::launch_best_fit<T, ::vp::RtModel<NN, QQ, PP>, RR>,                                                     \
        &::vp::launch_mrhs_factor<T, ::vp::RtModel<NN, QQ, PP>, RR>,                                                  \
        &::vp::launch_mrhs_stream<T, ::vp::RtModel<NN, QQ, PP>, RR>, &::vp::launch_mrhs_lm<T, ::vp::RtModel<NN, QQ, PP>, RR>, \
        &::vp::launch_mrhs_finish<T, ::vp::RtModel<NN, QQ, PP>, RR>, ::vp::mrhs_state_bytes<T, ::vp::RtModel<NN, QQ, PP>>(), \
        &::vp::launch_stats<T, ::vp::RtModel<NN, QQ, PP>, RR>, nullptr, 0, ::vp::mrhs_gx_cap<T, RR>()});

// a set that only serves handles with several right-hand sides (S > 1): the MRHS kernels (+ basis / best_fit).  The
// single-RHS kernels of the triple exponential at 32 rows per lane spilled 400-940 VGPRs and were never selected
// (find_kernels hands single-RHS handles of that length the 4-wave set of 8 rows per lane)
#define VP_REGISTER_MULTIEXP_MRHS_ONLY(T, DT, NEXP, OFF, RR)                                                           \
    static ::vp::Registrar VP_CAT(vp_reg_, __COUNTER__)(::vp::KernelEntry{                                            \
        DT, ::vp::FAMILY_MULTIEXP, NEXP, OFF, 0, RR, 1, nullptr, &::vp::launch_basis<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>, \
        nullptr, nullptr, &::vp::launch_best_fit<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                       \
        &::vp::launch_mrhs_factor<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                      \
        &::vp::launch_mrhs_stream<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                      \
        &::vp::launch_mrhs_lm<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                          \
        &::vp::launch_mrhs_finish<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR>,                                      \
        ::vp::mrhs_state_bytes<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>>(), nullptr, nullptr, 0, ::vp::mrhs_gx_cap<T, RR>()});

// multi-wave groups (WW waves per problem): problems whose columns do not fit the registers of one wave
#define VP_REGISTER_MULTIEXP_W(T, DT, NEXP, OFF, RR, WW)                                                               \
    static ::vp::Registrar VP_CAT(vp_reg_, __COUNTER__)(::vp::KernelEntry{                                            \
        DT, ::vp::FAMILY_MULTIEXP, NEXP, OFF, 0, RR, WW,                                                               \
        &::vp::launch_evaluate<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                     \
        &::vp::launch_basis<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                        \
        &::vp::launch_fit2<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                         \
        &::vp::launch_fit<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                          \
        &::vp::launch_best_fit<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>, nullptr, nullptr, nullptr, nullptr, 0, \
        &::vp::launch_stats<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>, nullptr, 0, 0,                         \
        ::vp::fit_lds_bytes<T, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>(true)});                                \
    VP_REGISTER_FIT_RESCUE(T, NEXP, OFF, RR, WW)

// fp32, many columns x many rows: the FIT runs on the fp64 Gram matrix (vp_fitg.hpp: any grid, any weights; there is
// no fp32 Householder fit kernel for these shapes -- it lost 15 % of the fits and spilled 230-250 VGPRs);
// evaluate / basis / best_fit / statistics stay on the WW-wave Householder kernels
namespace vp {
template <class M> int launch_fitg_entry(const LaunchParams &p) { return launch_fitg<M>(p, p.gram_dbg); }
} // namespace vp
#define VP_REGISTER_MULTIEXP_W_GRAM(NEXP, OFF, RR, WW)                                                                 \
    static ::vp::Registrar VP_CAT(vp_reg_, __COUNTER__)(::vp::KernelEntry{                                            \
        VP_F32, ::vp::FAMILY_MULTIEXP, NEXP, OFF, 0, RR, WW,                                                           \
        &::vp::launch_evaluate<float, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                 \
        &::vp::launch_basis<float, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>,                                    \
        &::vp::launch_fitg_entry<::vp::MultiExpModel<NEXP, (OFF) != 0>>, nullptr,                                     \
        &::vp::launch_best_fit<float, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>, nullptr, nullptr, nullptr, nullptr, 0, \
        &::vp::launch_stats<float, ::vp::MultiExpModel<NEXP, (OFF) != 0>, RR, WW>, nullptr, 1});

// run-time-descriptor models at larger m (WW waves per problem): single-RHS kernel set only
#define VP_REGISTER_RT_W(T, DT, NN, QQ, PP, RR, WW)                                                                    \
    static ::vp::Registrar VP_CAT(vp_reg_, __COUNTER__)(::vp::KernelEntry{                                            \
        DT, ::vp::FAMILY_RT, NN, QQ, PP, RR, WW, &::vp::launch_evaluate<T, ::vp::RtModel<NN, QQ, PP>, RR, WW>,        \
        &::vp::launch_basis<T, ::vp::RtModel<NN, QQ, PP>, RR, WW>, nullptr,                                           \
        &::vp::launch_fit<T, ::vp::RtModel<NN, QQ, PP>, RR, WW>,                                                      \
        &::vp::launch_best_fit<T, ::vp::RtModel<NN, QQ, PP>, RR, WW>, nullptr, nullptr, nullptr, nullptr, 0,         \
        &::vp::launch_stats<T, ::vp::RtModel<NN, QQ, PP>, RR, WW>, nullptr, 0, 0,                                     \
        ::vp::fit_lds_bytes<T, ::vp::RtModel<NN, QQ, PP>, RR, WW>(true)});
