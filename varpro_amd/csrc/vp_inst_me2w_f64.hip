// double exponential + offset fp64 with 4 waves per problem: m <= 4096
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP_W(double, VP_F64, 2, 1, 16, 4)
