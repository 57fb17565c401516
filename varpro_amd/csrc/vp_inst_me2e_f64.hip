// double exponential + offset, fp64: 20 rows per lane (1024 < m <= 1280) -- between the 1024-row set and the 2048-row set, whose 32
// rows per lane keep 320 VGPRs of columns in 256 registers (profiles/r03_m_sweep.json: m = 1100 at 5.2 M fits/s against 19.8 M
// at m = 1024; with this set 13 M)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 1, 20)
