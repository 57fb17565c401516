// caller-evaluated models (vp_ext.hpp): resident evaluate kernels, f64, EIGHT waves per problem (m <= 8192: the
// workgroup fills a CU at two waves per SIMD, 256 VGPRs per lane: shapes of up to six columns)
#include "vp_ext.hpp"

VP_REGISTER_EXT0_W(double, 1, 16, 8)
VP_REGISTER_EXT0_W(double, 2, 16, 8)
VP_REGISTER_EXT0_W(double, 3, 16, 8)
VP_REGISTER_EXT0_W(double, 4, 16, 8)
VP_REGISTER_EXT_W(double, 1, 2, 16, 8)
VP_REGISTER_EXT_W(double, 1, 4, 16, 8)
VP_REGISTER_EXT_W(double, 2, 2, 16, 8)
VP_REGISTER_EXT_W(double, 3, 2, 16, 8)
