// double exponential + offset, fp64, beyond 2048 rows on TWO waves per problem: 20 / 24 / 28 rows per lane (m <= 2560 / 3072 /
// 3584) -- the 4-wave set costs the same 4.7 ms per 16384 fits whatever the length (2100 rows: 3.4 M fits/s against 9.6 M at
// 2048 rows)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP_W(double, VP_F64, 2, 1, 20, 2)
VP_REGISTER_MULTIEXP_W(double, VP_F64, 2, 1, 24, 2)
VP_REGISTER_MULTIEXP_W(double, VP_F64, 2, 1, 28, 2)
