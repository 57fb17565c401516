// length-agnostic fit kernels (vp_block.hpp), single and triple exponentials, f64
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 1, 1)
VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 1, 0)
VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 3, 1)
VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 3, 0)
