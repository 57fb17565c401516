// triple exponential + offset, fp64, beyond 2048 rows: 4 waves per problem, 12 / 16 rows per lane (m <= 3072 / 4096),
// single-RHS kernel set -- without it the model dropped to the generic kernels there
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP_W(double, VP_F64, 3, 1, 12, 4)
VP_REGISTER_MULTIEXP_W(double, VP_F64, 3, 1, 16, 4)
