// Frame-resident (LDS) message-passing kernels, every check degree 6, every variable degree 3
// (so m / n = 3 / 6 and a thread with 8 variables has 4 checks, one with 12 has 6, one with 16 has 8).
#include "bp_lds.hpp"

namespace ldpc {
hipError_t bp_lds_launch_6_3(const LdsLaunch& l, const LdsParams& p) {
    return l.vpt == kLdsVptWide ? bp_lds_launch_impl<6, 3, kLdsVptWide>(l, p)
           : l.vpt == kLdsVptMid ? bp_lds_launch_impl<6, 3, kLdsVptMid>(l, p) : bp_lds_launch_impl<6, 3, kLdsVpt>(l, p);
}
unsigned bp_lds_6_3_params_bytes() { return (unsigned)sizeof(LdsParams); }

}  // namespace ldpc
