// triple exponential (+offset) fp64 (BASELINE.json configs[2] model family)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 16)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 0, 2)
VP_REGISTER_MULTIEXP_MRHS_ONLY(double, VP_F64, 3, 1, 32)
// single-RHS problems at 1024 < m <= 2048: 4 waves per problem, 8 rows per lane (the R = 32 set above stays for its MRHS kernels)
VP_REGISTER_MULTIEXP_W(double, VP_F64, 3, 1, 8, 4)
