// Message-passing kernels, specialisation DC=6 DV=3 (0 = any degree); see bp_kernels.hpp.
#include "bp_kernels.hpp"

namespace ldpc {
void bp_launch_6_3(const BpLaunch& l, const BpParams& p) { bp_launch_impl<6, 3>(l, p); }
}  // namespace ldpc
