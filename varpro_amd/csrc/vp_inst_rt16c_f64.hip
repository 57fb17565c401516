// runtime-descriptor models, fp64, m <= 1024: further shapes (see vp_inst_rtc_f64.hip)
#include "vp_inst.hpp"
VP_REGISTER_RT(double, VP_F64, 2, 1, 1, 16)
VP_REGISTER_RT(double, VP_F64, 3, 3, 3, 16)
