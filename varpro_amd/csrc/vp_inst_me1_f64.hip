// single exponential (+offset) fp64
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 1, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 1, 16)
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 0, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 0, 16)
