// double exponential + offset, fp32: 8 rows per lane (m <= 512) and 32 rows per lane (m <= 2048: five fp32 columns of 32 rows
// are 160 VGPRs) -- the fp32 handle had the 128- and the 1024-row set only and ran on the generic kernels above 1024 rows
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(float, VP_F32, 2, 1, 8)
VP_REGISTER_MULTIEXP(float, VP_F32, 2, 1, 32)
