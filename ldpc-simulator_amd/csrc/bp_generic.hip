// Message-passing kernels, specialisation DC=0 DV=0 (0 = any degree); see bp_kernels.hpp.
#include "bp_kernels.hpp"

namespace ldpc {
void bp_launch_generic(const BpLaunch& l, const BpParams& p) { bp_launch_impl<0, 0>(l, p); }
unsigned bp_generic_params_bytes() { return (unsigned)sizeof(BpParams); }

}  // namespace ldpc
