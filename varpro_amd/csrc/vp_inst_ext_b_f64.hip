// caller-evaluated models (vp_ext.hpp): resident evaluate kernels, f64, N = 3
#include "vp_ext.hpp"

VP_REGISTER_EXT0(double, 3, 16)
VP_REGISTER_EXT(double, 3, 2, 16)
VP_REGISTER_EXT(double, 3, 4, 16)
VP_REGISTER_EXT(double, 3, 6, 16)
VP_REGISTER_EXT0(double, 3, 4)
VP_REGISTER_EXT(double, 3, 2, 4)
VP_REGISTER_EXT(double, 3, 4, 4)
VP_REGISTER_EXT(double, 3, 8, 4)
VP_REGISTER_EXT(double, 3, 16, 4)
