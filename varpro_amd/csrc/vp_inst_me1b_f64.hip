// single exponential + offset, fp64, beyond 1024 rows: 24 / 32 rows per lane (m <= 1536 / 2048) -- three columns (one
// exponential, the data, one derivative) fit the registers at 32 rows per lane; without these sets a 1100-row single
// exponential dropped to the generic kernels
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 1, 24)
VP_REGISTER_MULTIEXP(double, VP_F64, 1, 1, 32)
// (beyond 2048 rows: the length-agnostic set, vp_inst_blk_me13_f64.hip)
