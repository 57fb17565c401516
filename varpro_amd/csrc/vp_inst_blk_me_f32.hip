// length-agnostic fit kernels (vp_block.hpp), multi-exponential models, f32 (the five-exponential shape keeps its fp64-Gram
// fit up to 4096 rows -- vp_fitg.hpp -- and streams beyond)
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_MULTIEXP(float, VP_F32, 2, 1)
VP_REGISTER_BLOCKED_MULTIEXP(float, VP_F32, 5, 1)
