// runtime-descriptor models, fp64: further (n, q, p) shapes -- rate-form exponentials with an offset, three- and
// four-term mixes of the descriptor kinds (every basis depending on its own parameters)
#include "vp_inst.hpp"
VP_REGISTER_RT(double, VP_F64, 2, 1, 1, 2)
VP_REGISTER_RT(double, VP_F64, 3, 3, 3, 2)
VP_REGISTER_RT(double, VP_F64, 4, 3, 3, 2)
VP_REGISTER_RT(double, VP_F64, 3, 4, 4, 2)
VP_REGISTER_RT(double, VP_F64, 4, 4, 4, 2)
