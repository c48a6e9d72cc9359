// Message-passing kernels, specialisation DC=6 DV=3 (0 = any degree); see bp_kernels.hpp.
#include "bp_kernels.hpp"

namespace ldpc {
void bp_launch_6_3(const BpLaunch& l, const BpParams& p) { bp_launch_impl<6, 3>(l, p); }
unsigned bp_6_3_params_bytes() { return (unsigned)sizeof(BpParams); }

}  // namespace ldpc
