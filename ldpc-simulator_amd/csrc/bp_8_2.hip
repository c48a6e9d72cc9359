// Message-passing kernels, specialisation DC=8 DV=2 (0 = any degree); see bp_kernels.hpp.
#include "bp_kernels.hpp"

namespace ldpc {
void bp_launch_8_2(const BpLaunch& l, const BpParams& p) { bp_launch_impl<8, 2>(l, p); }
unsigned bp_8_2_params_bytes() { return (unsigned)sizeof(BpParams); }

}  // namespace ldpc
