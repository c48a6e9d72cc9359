// Frame-resident (LDS) message-passing kernels, every check degree 8, every variable degree 2.
#include "bp_lds.hpp"

namespace ldpc {
hipError_t bp_lds_launch_8_2(const LdsLaunch& l, const LdsParams& p) { return bp_lds_launch_impl<8, 2>(l, p); }
}  // namespace ldpc
