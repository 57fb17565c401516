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
// (round 6, measured: sixteen waves x 8 rows per lane instead of eight x 16 -- 128 VGPRs per lane, 116 spilled -- 1.60 ms against
// 1.09-1.13 ms per 8 192 problems of 8 192 rows)
VP_REGISTER_EXT_W(double, 3, 2, 16, 8)
