// caller-evaluated models (vp_ext.hpp): resident evaluate kernels, f64, N = 7, 8
#include "vp_ext.hpp"

VP_REGISTER_EXT0(double, 7, 4)
VP_REGISTER_EXT(double, 7, 2, 4)
VP_REGISTER_EXT(double, 7, 4, 4)
VP_REGISTER_EXT(double, 7, 8, 4)
VP_REGISTER_EXT(double, 7, 16, 4)
VP_REGISTER_EXT0(double, 8, 4)
VP_REGISTER_EXT(double, 8, 2, 4)
VP_REGISTER_EXT(double, 8, 4, 4)
VP_REGISTER_EXT(double, 8, 8, 4)
