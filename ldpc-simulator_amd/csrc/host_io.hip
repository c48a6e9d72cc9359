// Launches around the decode: BSC magnitudes, channel and codeword generators, error accounting.
#include "host_ctx.hpp"
#include "io_kernels.hpp"

using namespace ldpc;

namespace ldpc {
namespace host {

int load_lambda(ldpc_amd_ctx* c, const SrcArgs& src, const BpParams& p, float* ws_lam, int T, long long frames, hipStream_t s) {
    const int n = c->n;
    const dim3 grid((unsigned)(p.tiles * (T / 64)), (unsigned)((n + 63) / 64)), block(256);
    if (src.kind == Src::Float) {
        hipLaunchKernelGGL((load_lambda_kernel<FloatLlr>), grid, block, 0, s, FloatLlr{src.llr, (float)llr_scale(c)}, ws_lam,
                           p.lam_tile, p.lam_row, n, T, p.frame_base, frames);
    } else {
        hipLaunchKernelGGL((load_lambda_kernel<BscLlr>), grid, block, 0, s,
                           BscLlr{src.r, (float)(src.l0 * llr_scale(c)), (float)(src.l1 * llr_scale(c)), n - c->m}, ws_lam,
                           p.lam_tile, p.lam_row, n, T, p.frame_base, frames);
    }
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int unpack_hard(ldpc_amd_ctx* c, const BpParams& p, const uint32_t* ws_bits, int T, long long frames_here, uint8_t* d_hard,
                hipStream_t s) {
    const int n = c->n;
    const long long threads = frames_here * ((n + 3) / 4);
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipLaunchKernelGGL(unpack_hard_kernel, grid, block, 0, s, ws_bits, p.bits_tile, p.bits_row, n, T, p.frame_base,
                       frames_here, d_hard);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

bool generator_fits(int k) { return gen_codewords_lds_bytes(k) <= 160 * 1024; }

int bsc_magnitudes(ldpc_amd_ctx* c, double e, int llr_mode, double* l0, double* l1) {
    if (!(e > 0.0) || !(e < 1.0)) return fail(c, LDPC_AMD_EINVAL, "crossover probability must be in (0,1)");
    const double d0 = std::log((1 - e) / e);  // LogDecoder.java:46
    double d1 = d0;
    if (llr_mode == LDPC_AMD_LLR_PARITY_LN1E20) d1 = std::log((1 - 0.00000000000000000001) / 0.00000000000000000001);
    else if (llr_mode == LDPC_AMD_LLR_PARITY_1EM7) d1 = std::log((1 - 0.0000001) / 0.0000001);
    else if (llr_mode != LDPC_AMD_LLR_UNIFORM) return fail(c, LDPC_AMD_EINVAL, "unknown llr_mode");
    *l0 = d0;
    *l1 = d1;
    return LDPC_AMD_OK;
}

uint32_t bsc_threshold(double e) {
    if (!(e > 0.0)) return 0u;
    const double t = std::floor(e * 4294967296.0);
    return (t >= 4294967295.0) ? 4294967295u : (uint32_t)t;
}

int count_launch(ldpc_amd_ctx* c, long long frames, const uint8_t* d_hard, const uint8_t* d_cw, const uint8_t* d_rx,
                 const float* d_rx_llr,
                 const int32_t* d_iters, const int32_t* d_synd, int max_iters, unsigned long long* d_counters,
                 hipStream_t s) {
    if (frames <= 0) return LDPC_AMD_OK;
    const dim3 grid((unsigned)((frames + 63) / 64)), block(256);
    const size_t sh = (size_t)(16 + max_iters) * sizeof(unsigned long long);
    const int k = c->n - c->m;
    if (d_rx) {
        hipLaunchKernelGGL((count_errors_kernel<RxBytes, true>), grid, block, sh, s, c->n, k, frames, d_hard,
                           d_cw, RxBytes{d_rx}, d_iters, d_synd, max_iters, d_counters);
    } else if (d_rx_llr) {
        hipLaunchKernelGGL((count_errors_kernel<RxLlrSign, true>), grid, block, sh, s, c->n, k, frames, d_hard,
                           d_cw, RxLlrSign{d_rx_llr}, d_iters, d_synd, max_iters, d_counters);
    } else {
        hipLaunchKernelGGL((count_errors_kernel<RxNone, false>), grid, block, sh, s, c->n, k, frames, d_hard, d_cw,
                           RxNone{}, d_iters, d_synd, max_iters, d_counters);
    }
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int gen_codewords(ldpc_amd_ctx* c, uint64_t seed, long long first_frame, long long frames, uint8_t* d_cw,
                  hipStream_t s) {
    if (frames <= 0) return LDPC_AMD_OK;
    if (!c->d_gen) {  // no generator matrix: the all-zero codeword
        LDPC_HIP(c, hipMemsetAsync(d_cw, 0, (size_t)frames * c->n, s));
        return LDPC_AMD_OK;
    }
    const size_t lds = gen_codewords_lds_bytes(c->gen_k);
    if (!generator_fits(c->gen_k)) return fail(c, LDPC_AMD_ELIMIT, "gen_codewords: generator matrices of more than 10,208 rows are not supported");
    auto kern = gen_codewords_kernel;
    LDPC_HIP(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((frames + kCwFrames - 1) / kCwFrames)), dim3(kCwThreads), lds, s, c->n, c->gen_k,
                       c->d_gen, seed, first_frame, frames, d_cw);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int gen_bsc(ldpc_amd_ctx* c, int channel, double e, uint64_t seed, long long first_frame, long long frames,
            const uint8_t* d_cw, uint8_t* d_r, hipStream_t s) {
    if (channel != LDPC_AMD_CH_BSC_INFO && channel != LDPC_AMD_CH_BSC_ALL)
        return fail(c, LDPC_AMD_EINVAL, "gen_bsc: channel must be a BSC channel");
    if (frames <= 0) return LDPC_AMD_OK;
    const int flip_end = (channel == LDPC_AMD_CH_BSC_INFO) ? (c->n - c->m) : c->n;
    const long long threads = frames * ((c->n + 3) / 4);
    hipLaunchKernelGGL(gen_bsc_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, c->n, flip_end,
                       bsc_threshold(e), seed, first_frame, frames, d_cw, d_r);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int gen_awgn(ldpc_amd_ctx* c, double ebn0_db, uint64_t seed, long long first_frame, long long frames,
             const uint8_t* d_cw, float* d_llr, hipStream_t s) {
    if (frames <= 0) return LDPC_AMD_OK;
    if (c->n <= c->m) return fail(c, LDPC_AMD_EINVAL, "gen_awgn: code rate (n-m)/n must be positive");
    const double rate = (double)(c->n - c->m) / (double)c->n;
    const double sigma2 = 1.0 / (2.0 * rate * std::pow(10.0, ebn0_db / 10.0));
    const float a = (float)(2.0 / sigma2), b = (float)(2.0 / std::sqrt(sigma2));
    const long long threads = frames * ((c->n + 3) / 4);
    hipLaunchKernelGGL(gen_awgn_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, c->n, a, b, seed,
                       first_frame, frames, d_cw, d_llr);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

}  // namespace host
}  // namespace ldpc
