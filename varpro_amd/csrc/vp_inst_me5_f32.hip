// five exponentials + offset, fp32 (BASELINE.json configs[4]: n = 6, q = 5, m = 4096): the 12 resident columns
// need 4 waves per problem (16 rows per lane)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP_W_GRAM(5, 1, 16, 4)
VP_REGISTER_MULTIEXP(float, VP_F32, 5, 1, 2)
