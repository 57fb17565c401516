// double-exponential (+offset) fp64: the headline configuration (BASELINE.json configs[0,1,3])
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 16)
