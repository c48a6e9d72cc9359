// Frame-resident flooding belief propagation for any code whose messages fit in LDS: one workgroup
// decodes one frame, msg[E] (CSR edge order), the channel LLRs and the packed decisions stay in LDS
// for all iterations (see bp_lds.hpp for the specialised regular-code version of the same idea).
//
// Degrees are per node here.  The host sorts checks and variables by degree and deals them out in
// that order to the positions p = t + s * nt (thread t, step s), so the 64 lanes of a wave at one step
// hold nodes of (nearly) one degree and the per-degree code of cn_dispatch / vn_dispatch
// (bp_kernels.hpp: the same fully unrolled register code as the streaming engines, hence the same
// bits) runs with little divergence.  Positions past the last node are marked empty.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernels.hpp"

namespace ldpc {

// limits of the register code this engine is compiled with (bp_lds_gen.hip); codes above them stream
constexpr int kLdsGenMaxDc = 20, kLdsGenMaxDv = 32;

struct LdsGenParams {
    int m, n, nwp;          // nwp: words of decisions = vn_steps * nt / 32
    long long E;
    int cn_steps, vn_steps; // positions per thread
    const int* row_ptr;     // [m+1]
    const int* col_edge;    // [E] CSR edge ids of a variable, ascending check
    const int* edge_col;    // [E] variable of CSR edge e
    const int2* cn_desc;    // [cn_steps*nt] check at CN position p: (first edge, degree); degree 0 pads
    const int4* vn_desc;    // [vn_steps*nt] variable at VN position p: (j, first entry in col_edge, degree, 0); j = -1 pads
    const int* pos_of_var;  // [n]
    const int* edge_pos;    // [E] VN position of the variable of CSR edge e (its decision bit index)
    const float* llr;       // [frames][n] channel LLRs, or null
    const uint8_t* r;       // [frames][n] hard received words (then lambda = +-l0 / +-l1), or null
    float l0, l1;           // already in the rule's LLR unit
    float scale;            // channel LLR unit conversion
    int g_length;           // n - m
    long long frames;
    int max_iters;
    uint8_t* hard;          // [frames][n]
    int32_t* out_iters;     // [frames] or null
    int32_t* out_synd;      // [frames] or null
};

#ifdef LDPC_LDS_GEN_KERNELS  // compiled in bp_lds_gen.hip only

// number of unsatisfied checks of the decisions in `bits` (sumSyndro, LogDecoder.java:108-116)
__device__ __forceinline__ int lds_gen_syndrome(const LdsGenParams& p, const uint32_t* bits, int* red, int t, int nt) {
    int cnt = 0;
    for (int i = t; i < p.m; i += nt) {
        uint32_t par = 0u;
        for (int e = p.row_ptr[i]; e < p.row_ptr[i + 1]; ++e) {
            const int q = p.edge_pos[e];
            par ^= bits[q >> 5] >> (q & 31);
        }
        cnt += (int)(par & 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((t & 63) == 0 && cnt) atomicAdd(red, cnt);
    __syncthreads();
    const int total = *red;
    __syncthreads();
    if (t == 0) *red = 0;
    return total;
}

template <int RULE, bool EARLY>
__global__ void __launch_bounds__(512, 4) bp_lds_gen_kernel(LdsGenParams p) {
    extern __shared__ __align__(16) float smem_gen[];
    float* msg = smem_gen;                                      // [E]
    float* lam = smem_gen + p.E;                                // [n], by variable
    uint32_t* bits = reinterpret_cast<uint32_t*>(lam + p.n);    // [nwp], by VN position
    int* red = reinterpret_cast<int*>(bits + p.nwp);            // syndrome accumulator
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63;
    const long long frame = blockIdx.x;
    if (frame >= p.frames) return;

    // channel LLRs (LogDecoder.java:49-54 for a hard received word), then beta of the first
    // iteration: lambda + the zero-initialised alpha (LogDecoder.java:57, :70-77)
    if (p.llr) {
        const float* src = p.llr + frame * p.n;
        for (int j = t; j < p.n; j += nt) lam[j] = src[j] * p.scale;
    } else {
        const uint8_t* src = p.r + frame * p.n;
        for (int j = t; j < p.n; j += nt) {
            const float mag = (j < p.g_length) ? p.l0 : p.l1;
            lam[j] = (src[j] == 0) ? mag : -mag;
        }
    }
    for (int w = t; w < p.nwp; w += nt) bits[w] = 0u;
    if (t == 0) *red = 0;
    __syncthreads();
    for (int e = t; e < (int)p.E; e += nt) msg[e] = lam[p.edge_col[e]] + 0.0f;
    __syncthreads();

    int iters = p.max_iters, synd = 0;
    for (int it = 0; it < p.max_iters; ++it) {
        // check nodes: beta -> alpha in place (LogDecoder.java:82-95)
        for (int s = 0; s < p.cn_steps; ++s) {
            const int2 cd = p.cn_desc[t + s * nt];
            if (cd.y > 0) cn_dispatch<1, RULE, false>(cd.y, msg + cd.x, nullptr, nullptr, 1, 1, nullptr, 0, lane);
        }
        __syncthreads();
        // variable nodes: alpha -> beta in place, posterior sign -> decision bit (:67-79, :98-105)
        const bool write_beta = EARLY || (it + 1 < p.max_iters);
        for (int s = 0; s < p.vn_steps; ++s) {
            const int pos = t + s * nt;
            const int4 vd = p.vn_desc[pos];
            const int j = vd.x;
            float g[1] = {0.0f};
            if (j >= 0) vn_dispatch<1>(vd.z, msg, lam + j, p.col_edge + vd.y, 1, write_beta, g, nullptr, 0, lane);
            if (EARLY || it + 1 == p.max_iters) {
                const unsigned long long mask = __ballot(j >= 0 && g[0] < 0.0f);  // "(gamma[j] < 0) ? 1 : 0", :104
                if (lane == 0) {  // pos is a multiple of 64 here
                    bits[pos >> 5] = (uint32_t)mask;
                    bits[(pos >> 5) + 1] = (uint32_t)(mask >> 32);
                }
            }
        }
        __syncthreads();
        if constexpr (EARLY) {
            synd = lds_gen_syndrome(p, bits, red, t, nt);
            if (synd == 0) {  // LogDecoder.java:117-119
                iters = it + 1;
                break;
            }
        }
    }
    if constexpr (!EARLY) synd = lds_gen_syndrome(p, bits, red, t, nt);

    uint8_t* out = p.hard + frame * p.n;
    for (int j = t; j < p.n; j += nt) {
        const int q = p.pos_of_var[j];
        out[j] = (uint8_t)((bits[q >> 5] >> (q & 31)) & 1u);
    }
    if (t == 0) {
        if (p.out_iters) p.out_iters[frame] = iters;
        if (p.out_synd) p.out_synd[frame] = synd;
    }
}

#endif  // LDPC_LDS_GEN_KERNELS

struct LdsGenLaunch {
    int rule;
    bool early;
    int threads;
    size_t lds_bytes;
    hipStream_t stream;
};

hipError_t bp_lds_gen_launch(const LdsGenLaunch& l, const LdsGenParams& p);
unsigned bp_lds_gen_params_bytes();  // see bp_kernels.hpp

}  // namespace ldpc
