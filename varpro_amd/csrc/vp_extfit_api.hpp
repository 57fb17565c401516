// vp_extfit_api.hpp -- host-side, type-erased entry of the batched reverse-communication LM fit of caller-evaluated models
// (vp_extfit.hpp): what vp_api.hip calls and vp_inst_ext.hip implements.  Kept out of vp_registry.hpp: every instantiation
// file includes that one.
#pragma once
#include "vp_registry.hpp"

namespace vp {

// One step of the batched reverse-communication LM fit of a caller-evaluated model (vp_extfit.hpp; vp_fit_begin /
// vp_fit_step_with_basis / vp_fit_end).  Type-erased; every pointer is a device pointer except the pair table.
struct ExtFitParams {
    int dtype, n, q, np;
    int64_t m, B;
    const void *phi, *dphi; // [B][n][m], [B][np][m] (dphi may be null when no problem wants derivatives)
    const void *w, *yw;
    int64_t w_stride;
    void *state;            // [B] records of external_fit_rec_bytes() each
    const void *alpha0;     // [B][q], read when init
    void *alpha_best, *C_best;
    double *cost;
    int32_t *status;
    vp_report *report;
    void *alpha_trial;      // [B][q]
    int32_t *want;          // [B]
    int32_t *nactive;       // [2] device counters, see ExtFitArgs
    int step;               // steps taken since vp_fit_begin
    const int32_t *pb, *pp; // [np] host
    double eps;
    vp_lm_opts opts;
    int init, lazy;
    // the generic step (vp_gen_extfit.hpp): shapes outside the specialised tables and several right-hand sides
    int64_t S;              // right-hand sides per problem (yw: [B][S][m], C_best: [B][S][n])
    void *gen_ws;           // gen_blocks workspace slots of (n + 1 + np + q) columns x m
    int gen_blocks;
    void *C_trial;          // [B][S][n] scratch (S > 1)
    // compacted active set (ExtFitArgs::active_in / active_out): two lists of B indices, used alternately
    int32_t *active_lists;  // [2][B]
    int64_t known_active;   // the host's last known number of active problems (an upper bound; B before the first step)
    hipStream_t stream;
};
// true: a step of this shape runs on the generic kernel and needs gen_ws (and C_trial when S > 1)
bool external_fit_generic(int dtype, int n, int np, int q, int64_t m, int64_t S);
// bytes of one LM record of the step kernel that covers this shape; 0 = no kernel (the caller reports VP_ERR_UNSUPPORTED)
size_t external_fit_rec_bytes(int dtype, int n, int np, int q, int64_t m);
int external_fit_step(const ExtFitParams &p);

} // namespace vp
