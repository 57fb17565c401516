// runtime-descriptor models, fp64 (reference unit-test model with swapped columns; O'Leary example; misc.)
#include "vp_inst.hpp"
VP_REGISTER_RT(double, VP_F64, 3, 2, 2, 2)
VP_REGISTER_RT(double, VP_F64, 2, 3, 4, 2)
VP_REGISTER_RT(double, VP_F64, 1, 1, 1, 2)
VP_REGISTER_RT(double, VP_F64, 2, 2, 2, 2)
VP_REGISTER_RT(double, VP_F64, 1, 2, 2, 2)
VP_REGISTER_RT(double, VP_F64, 2, 4, 4, 2)
