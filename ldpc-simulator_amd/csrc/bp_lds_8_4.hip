// Frame-resident (LDS) message-passing kernels, every check degree 8, every variable degree 4
// (GenerateMatrix.gallagerCheckMatrix(n, 8, 4), the code of LDPC_GallagerLoopSimu.java:19-21: m / n = 4 / 8, a thread
// with 8 variables has 4 checks).
#include "bp_lds.hpp"

namespace ldpc {
hipError_t bp_lds_launch_8_4(const LdsLaunch& l, const LdsParams& p) { return bp_lds_launch_impl<8, 4, kLdsVpt>(l, p); }
unsigned bp_lds_8_4_params_bytes() { return (unsigned)sizeof(LdsParams); }

}  // namespace ldpc
