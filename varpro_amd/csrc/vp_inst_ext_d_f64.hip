// caller-evaluated models (vp_ext.hpp): resident evaluate kernels, f64, N = 5, 6
#include "vp_ext.hpp"

VP_REGISTER_EXT0(double, 5, 16)
VP_REGISTER_EXT(double, 5, 2, 16)
VP_REGISTER_EXT(double, 5, 4, 16)
VP_REGISTER_EXT0(double, 5, 4)
VP_REGISTER_EXT(double, 5, 2, 4)
VP_REGISTER_EXT(double, 5, 4, 4)
VP_REGISTER_EXT(double, 5, 8, 4)
VP_REGISTER_EXT(double, 5, 16, 4)
VP_REGISTER_EXT0(double, 6, 16)
VP_REGISTER_EXT0(double, 6, 4)
VP_REGISTER_EXT(double, 6, 2, 4)
VP_REGISTER_EXT(double, 6, 4, 4)
VP_REGISTER_EXT(double, 6, 8, 4)
VP_REGISTER_EXT(double, 6, 16, 4)
