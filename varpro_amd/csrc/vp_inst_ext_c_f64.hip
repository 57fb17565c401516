// caller-evaluated models (vp_ext.hpp): resident evaluate kernels, f64, N = 4
#include "vp_ext.hpp"

VP_REGISTER_EXT0(double, 4, 16)
VP_REGISTER_EXT(double, 4, 2, 16)
VP_REGISTER_EXT(double, 4, 4, 16)
VP_REGISTER_EXT0(double, 4, 4)
VP_REGISTER_EXT(double, 4, 2, 4)
VP_REGISTER_EXT(double, 4, 4, 4)
VP_REGISTER_EXT(double, 4, 8, 4)
VP_REGISTER_EXT(double, 4, 16, 4)
