// runtime-descriptor models, fp64, beyond 1024 rows: 4 waves per problem, 16 rows per lane (m <= 4096), single-RHS kernel
// set -- without it these shapes dropped to the generic kernels there (O'Leary at m = 1100: 0.21 M fits/s against 3.1 M at
// m = 1024)
#include "vp_inst.hpp"
VP_REGISTER_RT_W(double, VP_F64, 3, 2, 2, 16, 4)
VP_REGISTER_RT_W(double, VP_F64, 2, 3, 4, 16, 4)
