// runtime-descriptor models, fp64: in-between sizes for the two shapes the reference's own examples use -- the builder-made
// double exponential + offset (n, q, p) = (3, 2, 2) and the O'Leary exp*cos pair (2, 3, 4): 4 and 8 rows per lane
// (m <= 256, m <= 512) and 12 (m <= 768), so that a 200-row problem does not pay for the 1024 rows of the next set (tools/rt_m_sweep.py:
// O'Leary at m = 129 ... 1024 ran at 2.5-3.1 M fits/s whatever its length, 12.8 M at m = 128)
#include "vp_inst.hpp"
VP_REGISTER_RT(double, VP_F64, 3, 2, 2, 4)
VP_REGISTER_RT(double, VP_F64, 3, 2, 2, 8)
VP_REGISTER_RT(double, VP_F64, 2, 3, 4, 4)
VP_REGISTER_RT(double, VP_F64, 2, 3, 4, 8)
VP_REGISTER_RT(double, VP_F64, 3, 2, 2, 12)
VP_REGISTER_RT(double, VP_F64, 2, 3, 4, 12)
