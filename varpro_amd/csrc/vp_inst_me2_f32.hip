// double exponential + offset in fp32 (ScalarType = f32 is supported by the reference: src/model/builder/mod.rs:66)
#include "vp_inst.hpp"
VP_REGISTER_MULTIEXP(float, VP_F32, 2, 1, 2)
VP_REGISTER_MULTIEXP(float, VP_F32, 2, 1, 16)
