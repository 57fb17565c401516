// length-agnostic fit kernels (vp_block.hpp), multi-exponential models, f64
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 2, 1)
VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 2, 0)
