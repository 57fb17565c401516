// length-agnostic fit / evaluate kernels (vp_block.hpp), four exponentials (+ offset), f64 (round 5: the shape ran on the
// generic kernels at every length; five exponentials: vp_inst_blk_me5_f64.hip)
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 4, 1)
VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 4, 0)
