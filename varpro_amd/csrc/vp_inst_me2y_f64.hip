// double exponential + offset (and the single exponential + offset), fp64, beyond 4096 rows on four waves per problem:
// 24 / 32 rows per lane (m <= 6144 / 8192), single-RHS kernel sets -- beyond 4096 rows every model ran on the generic kernels
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP_W(double, VP_F64, 2, 1, 24, 4)
VP_REGISTER_MULTIEXP_W(double, VP_F64, 2, 1, 32, 4)
VP_REGISTER_MULTIEXP_W(double, VP_F64, 1, 1, 32, 4)
