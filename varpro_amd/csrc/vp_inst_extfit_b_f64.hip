// batched LM step of caller-evaluated models (vp_extfit.hpp), f64, n = 3
#include "vp_extfit.hpp"

VP_REGISTER_EXTFIT(double, 3, 2, 2, 4)
VP_REGISTER_EXTFIT(double, 3, 2, 2, 16)
VP_REGISTER_EXTFIT(double, 3, 3, 3, 4)
VP_REGISTER_EXTFIT(double, 3, 3, 3, 16)
VP_REGISTER_EXTFIT(double, 3, 4, 2, 4)
VP_REGISTER_EXTFIT(double, 3, 4, 2, 16)
VP_REGISTER_EXTFIT(double, 3, 4, 4, 4)
VP_REGISTER_EXTFIT(double, 3, 4, 4, 16)
VP_REGISTER_EXTFIT(double, 3, 6, 3, 4)
VP_REGISTER_EXTFIT(double, 3, 6, 3, 16)
VP_REGISTER_EXTFIT(double, 3, 6, 4, 4)
VP_REGISTER_EXTFIT(double, 3, 6, 4, 16)
VP_REGISTER_EXTFIT(double, 3, 6, 6, 4)
VP_REGISTER_EXTFIT(double, 3, 6, 6, 16)
