// batched LM step of caller-evaluated models (vp_extfit.hpp), f64, n = 4, 5, 6
#include "vp_extfit.hpp"

VP_REGISTER_EXTFIT(double, 4, 3, 3, 4)
VP_REGISTER_EXTFIT(double, 4, 3, 3, 16)
VP_REGISTER_EXTFIT(double, 4, 4, 4, 4)
VP_REGISTER_EXTFIT(double, 4, 4, 4, 16)
VP_REGISTER_EXTFIT(double, 4, 6, 3, 4)
VP_REGISTER_EXTFIT(double, 4, 6, 3, 16)
VP_REGISTER_EXTFIT(double, 4, 6, 4, 4)
VP_REGISTER_EXTFIT(double, 4, 6, 4, 16)
VP_REGISTER_EXTFIT(double, 4, 6, 6, 4)
VP_REGISTER_EXTFIT(double, 4, 6, 6, 16)
VP_REGISTER_EXTFIT(double, 5, 4, 4, 4)
VP_REGISTER_EXTFIT(double, 5, 4, 4, 16)
VP_REGISTER_EXTFIT(double, 5, 6, 6, 4)
VP_REGISTER_EXTFIT(double, 6, 5, 5, 4)
VP_REGISTER_EXTFIT(double, 6, 6, 6, 4)
