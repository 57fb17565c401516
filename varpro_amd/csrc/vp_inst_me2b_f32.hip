// double exponential + offset, fp32: 8 rows per lane (m <= 512) and, for handles with several right-hand sides, 32 rows per lane (m <= 2048);
// single-RHS handles above 1024 rows run on the streamed kernels (faster than the 32-row resident set at m = 2048, within 11 % at 1500:
// profiles/r05_prune_probe.json)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(float, VP_F32, 2, 1, 8)
VP_REGISTER_MULTIEXP_MRHS_ONLY(float, VP_F32, 2, 1, 32)
