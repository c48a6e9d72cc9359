// Frame-resident (LDS) message-passing kernels, every check degree 10, every variable degree 5: the quasi-cyclic
// codes of GenerateMatrix.generateQC(z, 5, 10) (LDPC_QCIntStreamSimu.java:20-22, :65) whose base matrix has no empty
// block, about two draws in three.  4 variables and 2 checks per thread: a check pair alone holds 60 registers.
#include "bp_lds.hpp"

namespace ldpc {
// The same kernels also take the draws with an empty block (one in three): checks of degree 9 - 10, variables of degree
// 4 - 5, completed with constant extra edges (bp_lds_kernel, DEF).
hipError_t bp_lds_launch_10_5(const LdsLaunch& l, const LdsParams& p) { return bp_lds_launch_impl<10, 5, 4, true>(l, p); }
unsigned bp_lds_10_5_params_bytes() { return (unsigned)sizeof(LdsParams); }

}  // namespace ldpc
