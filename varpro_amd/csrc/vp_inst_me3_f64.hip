// triple exponential (+offset) fp64 (BASELINE.json configs[2] model family)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 16)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 0, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 32)
