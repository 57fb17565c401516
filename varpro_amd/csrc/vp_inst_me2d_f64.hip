// double exponential + offset, fp64: 12 rows per lane (512 < m <= 768) -- a length just above the 512-row set otherwise pays
// for the 1024-row set's rows (profiles/r03_m_sweep.json: m = 520 at 21 M fits/s against 30 M at m = 512)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 12)
