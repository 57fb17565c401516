// double-exponential fp64: m <= 2048 (32 rows per lane) and the variants without offset
#include "vp_inst.hpp"
// 32 rows per lane (m <= 2048): the multiple-right-hand-side kernels only -- single-RHS handles above 1280 rows run on the
// streamed kernels (vp_block.hpp), which match the resident 24 / 28 / 32-row sets to 6-20 % (tools/prune_probe.py,
// profiles/r05_prune_probe.json) without their 175-289 spilled VGPRs
VP_REGISTER_MULTIEXP_MRHS_ONLY(double, VP_F64, 2, 1, 32)
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 0, 2)
VP_REGISTER_MULTIEXP(double, VP_F64, 2, 0, 16)
