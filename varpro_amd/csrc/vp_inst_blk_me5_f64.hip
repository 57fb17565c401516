// length-agnostic fit / evaluate kernels (vp_block.hpp), five exponentials (+ offset), f64: twelve columns at one wave per
// SIMD (blk_waves) -- the shape of BASELINE configs[4] in double precision, at every length (the generic kernels before)
#include "vp_inst_blk.hpp"

VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 5, 1)
VP_REGISTER_BLOCKED_MULTIEXP(double, VP_F64, 5, 0)
