// Node-parallel cooperative engine for very small batches (see bp_nodes.hpp).
#define LDPC_NODES_KERNELS
// checks up to degree 20 and variables up to degree 32 in registers; the host sends codes above that
// to the streaming engine (kNodesMaxDc / kNodesMaxDv)
#define LDPC_REG_DEG_CN 20
#define LDPC_REG_DEG_VN 32

#include <algorithm>

#include "bp_nodes.hpp"

namespace ldpc {

static_assert(kRegDegCn == kNodesMaxDc && kRegDegVn == kNodesMaxDv, "host limits follow the register code");

namespace {

const void* kernel_of(int rule) {
    switch (rule) {
        case kCnMinSum: return (const void*)bp_nodes_kernel<kCnMinSum>;
        case kCnSumProduct: return (const void*)bp_nodes_kernel<kCnSumProduct>;
        default: return (const void*)bp_nodes_kernel<kCnComplement>;
    }
}

}  // namespace

hipError_t bp_nodes_resident_blocks(int* blocks) {
    int dev = 0, cus = 0, least = 1 << 30;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    for (int rule : {kCnSumProduct, kCnMinSum, kCnComplement}) {
        int per_cu = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel_of(rule), 256, 0);
        if (e != hipSuccess) return e;
        least = std::min(least, per_cu);
    }
    *blocks = least * cus;
    return hipSuccess;
}

hipError_t bp_nodes_launch(const NodesLaunch& l, const NodesParams& p) {
    if (p.frames < 1 || p.frames > kNodesMaxFrames || p.max_iters < 1 || l.max_blocks < 1) return hipErrorInvalidValue;
    const long long items = (long long)p.frames * (((long long)std::max(p.m, p.n) + 63) / 64 * 64);
    const long long want = (items + 255) / 256;
    const dim3 grid((unsigned)(l.force_blocks > 0 ? l.force_blocks : std::max<long long>(1, std::min<long long>(want, l.max_blocks)))),
        block(256);
    NodesParams q = p;
    void* args[] = {&q};
    return hipLaunchCooperativeKernel(kernel_of(l.rule), grid, block, args, 0, l.stream);
}

unsigned bp_nodes_params_bytes() { return (unsigned)sizeof(NodesParams); }

}  // namespace ldpc
