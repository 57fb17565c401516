// triple exponential + offset, fp64: 12 rows per lane (512 < m <= 768)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 12)
