// batched LM fit of caller-evaluated models (vp_extfit.hpp): the LM drivers, one lane per problem, one kernel per parameter count
#include "vp_extfit.hpp"

VP_REGISTER_EXTFIT_LM(double, 1)
VP_REGISTER_EXTFIT_LM(double, 2)
VP_REGISTER_EXTFIT_LM(double, 3)
VP_REGISTER_EXTFIT_LM(double, 4)
VP_REGISTER_EXTFIT_LM(double, 5)
VP_REGISTER_EXTFIT_LM(double, 6)
VP_REGISTER_EXTFIT_LM(double, 7)
VP_REGISTER_EXTFIT_LM(double, 8)
VP_REGISTER_EXTFIT_LM(float, 1)
VP_REGISTER_EXTFIT_LM(float, 2)
VP_REGISTER_EXTFIT_LM(float, 3)
VP_REGISTER_EXTFIT_LM(float, 4)
VP_REGISTER_EXTFIT_LM(float, 5)
VP_REGISTER_EXTFIT_LM(float, 6)
VP_REGISTER_EXTFIT_LM(float, 7)
VP_REGISTER_EXTFIT_LM(float, 8)
