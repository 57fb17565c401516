// single and triple exponential (+ offset), fp64: 8 rows per lane (m <= 512) next to the 128- and 1024-row kernels
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 1, 8)
VP_REGISTER_MULTIEXP(double, VP_F64, 3, 1, 8)
