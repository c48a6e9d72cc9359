// Frame-resident engine for any code whose messages fit in LDS (see bp_lds_gen.hpp).
#define LDPC_LDS_GEN_KERNELS
// checks up to degree 20 and variables up to degree 32 in registers; the host sends codes above that
// to the streaming engine (kLdsGenMaxDc / kLdsGenMaxDv)
#define LDPC_REG_DEG_CN 20
#define LDPC_REG_DEG_VN 32
#include "bp_lds_gen.hpp"

namespace ldpc {

static_assert(kRegDegCn == kLdsGenMaxDc && kRegDegVn == kLdsGenMaxDv, "host limits follow the register code");

hipError_t bp_lds_gen_launch(const LdsGenLaunch& l, const LdsGenParams& p) {
    const dim3 grid((unsigned)p.frames), block((unsigned)l.threads);
#define LDPC_GEN(R, E)                                                                                       \
    do {                                                                                                     \
        auto k = bp_lds_gen_kernel<R, E>;                                                                    \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                           (int)l.lds_bytes);                                                \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL(k, grid, block, l.lds_bytes, l.stream, p);                                        \
    } while (0)
    switch (l.rule) {
        case kCnMinSum: if (l.early) LDPC_GEN(kCnMinSum, true); else LDPC_GEN(kCnMinSum, false); break;
        case kCnSumProduct: if (l.early) LDPC_GEN(kCnSumProduct, true); else LDPC_GEN(kCnSumProduct, false); break;
        default: if (l.early) LDPC_GEN(kCnComplement, true); else LDPC_GEN(kCnComplement, false); break;
    }
#undef LDPC_GEN
    return hipGetLastError();
}

unsigned bp_lds_gen_params_bytes() { return (unsigned)sizeof(LdsGenParams); }

}  // namespace ldpc
