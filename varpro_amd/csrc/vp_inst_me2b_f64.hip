// double-exponential fp64: m <= 2048 (32 rows per lane) and the variants without offset
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 32)
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 0, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 0, 16)
