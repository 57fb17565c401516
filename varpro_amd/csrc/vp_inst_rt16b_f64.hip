// runtime-descriptor models, fp64, m <= 1024 (one wave, 16 rows per lane): full kernel set (single- and multi-RHS)
#include "vp_inst.hpp"
VP_REGISTER_RT(double, VP_F64, 2, 3, 4, 16)
VP_REGISTER_RT(double, VP_F64, 1, 2, 2, 16)
VP_REGISTER_RT(double, VP_F64, 2, 4, 4, 16)
