// length-agnostic fit kernels (vp_block.hpp), run-time-descriptor models, f64 (third file: parallel build)
#define VP_INLINE_SINCOS 1 // (vp_model.hpp: tsincos inlined in these kernels)
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_RT(double, VP_F64, 4, 3, 3)
VP_REGISTER_BLOCKED_RT(double, VP_F64, 3, 4, 4)
VP_REGISTER_BLOCKED_RT(double, VP_F64, 4, 4, 4)
