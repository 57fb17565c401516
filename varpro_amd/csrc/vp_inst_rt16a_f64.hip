// runtime-descriptor models, fp64, m <= 1024 (one wave, 16 rows per lane): full kernel set (single- and multi-RHS)
#include "vp_inst.hpp"
VP_REGISTER_RT(double, VP_F64, 3, 2, 2, 16)
VP_REGISTER_RT(double, VP_F64, 1, 1, 1, 16)
VP_REGISTER_RT(double, VP_F64, 2, 2, 2, 16)
