// Frame-resident (LDS) message-passing kernels, every check degree 6, every variable degree 3
// (so m / n = 3 / 6 and a thread with 8 variables has 4 checks).
#include "bp_lds.hpp"

namespace ldpc {
hipError_t bp_lds_launch_6_3(const LdsLaunch& l, const LdsParams& p) { return bp_lds_launch_impl<6, 3, 4>(l, p); }
unsigned bp_lds_6_3_params_bytes() { return (unsigned)sizeof(LdsParams); }

}  // namespace ldpc
