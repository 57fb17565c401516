// triple exponential + offset, fp64: 20 rows per lane (1024 < m <= 1280; the 24-row set was within 13 % of the streamed fit and spilled 230-245 VGPRs), single- and multi-RHS -- between the
// 1024-row set and the 4-wave set that serves single-RHS handles up to 2048 rows
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 20)
