// batched LM fit of caller-evaluated models (vp_extfit.hpp): evaluation kernels for LONG problems, f64 (four / eight waves per problem); written by gen_extfit_inst.py
#include "vp_extfit.hpp"

VP_REGISTER_EXTFIT_W(double, 1, 1, 1, 16, 4)
VP_REGISTER_EXTFIT_W(double, 1, 2, 2, 16, 4)
VP_REGISTER_EXTFIT_W(double, 1, 3, 3, 16, 4)
VP_REGISTER_EXTFIT_W(double, 1, 4, 2, 16, 4)
VP_REGISTER_EXTFIT_W(double, 1, 4, 4, 16, 4)
VP_REGISTER_EXTFIT_W(double, 2, 1, 1, 16, 4)
VP_REGISTER_EXTFIT_W(double, 2, 2, 2, 16, 4)
VP_REGISTER_EXTFIT_W(double, 2, 3, 3, 16, 4)
VP_REGISTER_EXTFIT_W(double, 2, 4, 2, 16, 4)
VP_REGISTER_EXTFIT_W(double, 2, 4, 4, 16, 4)
VP_REGISTER_EXTFIT_W(double, 3, 2, 2, 16, 4)
VP_REGISTER_EXTFIT_W(double, 3, 3, 3, 16, 4)
VP_REGISTER_EXTFIT_W(double, 3, 4, 2, 16, 4)
VP_REGISTER_EXTFIT_W(double, 3, 4, 4, 16, 4)
VP_REGISTER_EXTFIT_W(double, 4, 3, 3, 16, 4)
