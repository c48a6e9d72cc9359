// Frame-resident (LDS) message-passing kernels, every check degree 8, every variable degree 2
// (so m / n = 2 / 8 and a thread with 8 variables has 2 checks).
#include "bp_lds.hpp"

namespace ldpc {
hipError_t bp_lds_launch_8_2(const LdsLaunch& l, const LdsParams& p) { return bp_lds_launch_impl<8, 2, kLdsVpt>(l, p); }
unsigned bp_lds_8_2_params_bytes() { return (unsigned)sizeof(LdsParams); }

}  // namespace ldpc
