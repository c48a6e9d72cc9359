// Row-parallel sweep launches (see bp_rows.hpp).
#define LDPC_ROWS_KERNELS
#define LDPC_MSG_NT 1
#define LDPC_LANE_IN_ACCESS 1
// these kernels keep checks up to degree 20 and variables up to degree 32 in registers
#define LDPC_REG_DEG_CN 20
#define LDPC_REG_DEG_VN 32
#include "bp_rows.hpp"

namespace ldpc {

namespace {

template <int FPL, int DC>
void launch_cn(const RowsLaunch& l, const RowsParams& p, dim3 grid, dim3 block, bool first) {
    constexpr bool kCanFuse = FPL == 1;
#define LDPC_CN(R)                                                                                                    \
    if (first) hipLaunchKernelGGL((rows_cn_kernel<FPL, R, true, DC, false>), grid, block, l.lds_bytes, l.stream, p);      \
    else if (kCanFuse && p.fused)                                                                                     \
        hipLaunchKernelGGL((rows_cn_kernel<FPL, R, false, DC, kCanFuse>), grid, block, l.lds_bytes, l.stream, p);         \
    else hipLaunchKernelGGL((rows_cn_kernel<FPL, R, false, DC, false>), grid, block, l.lds_bytes, l.stream, p)
    switch (l.rule) {
        case kCnMinSum: LDPC_CN(kCnMinSum); break;
        case kCnSumProduct: LDPC_CN(kCnSumProduct); break;
        default: LDPC_CN(kCnComplement); break;
    }
#undef LDPC_CN
}

template <int FPL, int DC, int DV>
hipError_t run(const RowsLaunch& l, RowsParams p) {
    const dim3 block((unsigned)(64 * l.wpb));
    const dim3 grid((unsigned)((l.chunks + l.wpb - 1) / l.wpb), (unsigned)p.b.tiles);
    const int slots = p.b.tiles * kWave * FPL;
    p.slots = slots;
    const dim3 sgrid((unsigned)((slots + 255) / 256));
    for (int it = 0; it < p.b.max_iters; ++it) {
        p.iter = it;
        const bool last = it + 1 == p.b.max_iters;
        // fixed iterations: the sweeps of iterations 1 .. L - 2 leave the marked leaves alone (BpParams::leaf_row); the
        // variable-node sweep of iteration 0 has written lambda to their slots, the last iteration decides them
        p.b.leaf_skip = (p.b.leaf_row && !p.early && it >= 1 && it + 2 <= p.b.max_iters) ? 1 : 0;
        launch_cn<FPL, DC>(l, p, grid, block, it == 0);
        const int write_beta = (p.early || !last) ? 1 : 0;
        if (FPL == 1 && p.fused && it > 0)
            hipLaunchKernelGGL((rows_vn_kernel<FPL, DV, FPL == 1>), grid, block, l.lds_bytes, l.stream, p, write_beta);
        else
            hipLaunchKernelGGL((rows_vn_kernel<FPL, DV, false>), grid, block, l.lds_bytes, l.stream, p, write_beta);
        if (last || (p.early && !p.fused)) {  // fused: the launches above carry the syndrome of the iteration before
            hipLaunchKernelGGL((rows_synd_kernel<FPL>), grid, block, 0, l.stream, p);
            hipLaunchKernelGGL(rows_finish_kernel, sgrid, dim3(256), 0, l.stream, p, slots, last ? 1 : 0);
        }
    }
    hipLaunchKernelGGL(rows_export_kernel, sgrid, dim3(256), 0, l.stream, p, slots);
    return hipGetLastError();
}

template <int DC, int DV>
hipError_t run_fpl(const RowsLaunch& l, const RowsParams& p) {
    return l.fpl == 4 ? run<4, DC, DV>(l, p) : run<1, DC, DV>(l, p);
}

}  // namespace

hipError_t bp_rows_run(const RowsLaunch& l, RowsParams p) {
    if (l.dc == 6 && l.dv == 3) return run_fpl<6, 3>(l, p);
    if (l.dc == 8 && l.dv == 2) return run_fpl<8, 2>(l, p);
    return run_fpl<0, 0>(l, p);
}

unsigned bp_rows_params_bytes() { return (unsigned)sizeof(RowsParams); }

}  // namespace ldpc
