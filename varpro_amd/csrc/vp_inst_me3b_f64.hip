// triple exponential + offset, fp64: 20 / 24 rows per lane (1024 < m <= 1280 / 1536), single- and multi-RHS -- between the
// 1024-row set and the 4-wave set that serves single-RHS handles up to 2048 rows
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 20)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 24)
