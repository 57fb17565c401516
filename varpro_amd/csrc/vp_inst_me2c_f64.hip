// double exponential + offset, fp64: intermediate sizes -- 4 and 8 rows per lane (m <= 256, m <= 512), so that a
// mid-size problem does not pay for the padding rows of the 1024-row kernel
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 4)
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 8)
