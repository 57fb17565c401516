// length-agnostic fit kernels (vp_block.hpp), run-time-descriptor models (kinds and dependency table at run time), f64
#define VP_INLINE_SINCOS 1 // (vp_model.hpp: tsincos inlined in these kernels)
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_RT(double, VP_F64, 3, 2, 2)
VP_REGISTER_BLOCKED_RT(double, VP_F64, 2, 3, 4)
VP_REGISTER_BLOCKED_RT(double, VP_F64, 1, 1, 1)
VP_REGISTER_BLOCKED_RT(double, VP_F64, 2, 2, 2)
