// Frame-resident kernels with two frames per workgroup, half an iteration apart (bp_lds2.hpp), for (6,3) codes:
// 4 variables and 2 checks per thread (16 wavefronts at n = 4096, four per SIMD) or 8 and 4 (8 wavefronts).
#include "bp_lds2.hpp"

namespace ldpc {
// frame block of the n = 4096 code: 12,288 message words + 128 decision words + 4 + 64 syndrome words
constexpr int kFb4096 = 4 * (12288 + 128 + 4 + 64);
hipError_t bp_lds2_launch_6_3(const LdsLaunch& l, const LdsParams& p) {
    if (l.vpt == 4) return bp_lds2_launch_impl<6, 3, 4, 4, kFb4096>(l, p);
    if (l.vpt == 8) return bp_lds2_launch_impl<6, 3, 8, 2, kFb4096>(l, p);
    return hipErrorInvalidConfiguration;
}
unsigned bp_lds2_6_3_params_bytes() { return (unsigned)sizeof(LdsParams); }

}  // namespace ldpc
