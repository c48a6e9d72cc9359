// Frame-resident flooding belief propagation for any code whose messages fit in LDS: one workgroup
// decodes one frame; its messages, the channel LLRs and the packed decisions stay in LDS for all
// iterations (see bp_lds.hpp for the specialised regular-code version of the same idea).
//
// Message slots: the row of a check is contiguous, and the rows of the lanes that run one degree's code together
// follow each other with a word of skew wherever their banks would repeat, so a check-node access of a wave
// meets no bank conflict, whatever the degree, at constant offsets from the row's first slot.
// The variable side reaches its messages through a slot table in place of CSR edge ids.
//
// Degrees are per node here.  The host sorts checks and variables by degree and pairs neighbours of the sorted
// list into work items, so the two nodes of an item nearly always have the same degree, in which case they share
// packed registers (cn_complement<v2f, D> / vn_compute_v<v2f, D>: the same operations as the scalar code of the
// streaming engines, hence the same bits).  Items whose two degrees differ, and the other check-node rules, go
// through cn_dispatch / vn_dispatch one node at a time.  A wave runs the code of every degree present among its
// 64 lanes of a step in turn; the host places the items on the slots t + u * nt (thread t, step u) so that each
// wave meets few degrees and the waves of a workgroup have equal work (build_generic_resident_tables).  Slots
// past the last node are marked empty.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernels.hpp"

namespace ldpc {

// limits of the register code this engine is compiled with (bp_lds_gen.hip); codes above them stream
constexpr int kLdsGenMaxDc = 20, kLdsGenMaxDv = 32;
// degrees up to which two nodes share packed registers (beyond that the register file does not hold a pair)
constexpr int kLdsGenPairDc = 10, kLdsGenPairDv = 8;
constexpr int kGenRowStride = 1;    // words between consecutive positions of a check's row

struct LdsGenParams {
    int m, n, nwp;          // nwp: words of decisions = 2 * vn_steps * nt / 32
    long long E;
    int slots;              // message words in LDS (E plus the words of skew between rows of even degree)
    int cn_steps, vn_steps; // work items (pairs of nodes) per thread
    const int* row_ptr;     // [m+1]
    const int* col_slot;    // [E] message slots of a variable's edges, ascending check (CSC order)
    const int* edge_col;    // [E] variable of CSR edge e
    const int* edge_slot;   // [E] message slot of CSR edge e
    const int* col_check;   // [E] check of a variable's edges, ascending (CSC order)
    const int4* cn_desc;    // [cn_steps*nt] two checks: (first slot a, degree a, first slot b, degree b); degree 0 = none;
                            // position k of a row is kGenRowStride words after its first slot
    const int4* vn_desc;    // [vn_steps*nt][2] two variables: (j, first entry in col_slot, degree, 0) each; j = -1 = none
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

// two checks of degree D in packed registers
template <int D>
__device__ __forceinline__ void gen_cn_pair(float* msg, int ea, int eb) {
    v2f x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = v2f{msg[ea + k * kGenRowStride], msg[eb + k * kGenRowStride]};
    cn_complement<v2f, D>(x);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        msg[ea + k * kGenRowStride] = x[k].x;
        msg[eb + k * kGenRowStride] = x[k].y;
    }
}

// two variables of degree D in packed registers; returns the two posteriors
template <int D>
__device__ __forceinline__ v2f gen_vn_pair(float* msg, const float* lam, const int* col_slot, int4 a, int4 b, bool write_beta) {
    const v2f l = v2f{lam[a.x], lam[b.x]};
    if constexpr (D == 0) {
        return l;
    } else {
        int ia[D], ib[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            ia[k] = col_slot[a.y + k];
            ib[k] = col_slot[b.y + k];
        }
        v2f x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = v2f{msg[ia[k]], msg[ib[k]]};
        v2f g;
        vn_compute_v<v2f, D>(x, l, g);
        if (write_beta) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                msg[ia[k]] = x[k].x;
                msg[ib[k]] = x[k].y;
            }
        }
        return g;
    }
}

#define LDPC_GEN_CN(D) case D: gen_cn_pair<D>(msg, cd.x, cd.z); break;
#define LDPC_GEN_VN(D) case D: g = gen_vn_pair<D>(msg, lam, p.col_slot, va, vb, write_beta); break;

template <int RULE, bool EARLY>
__global__ void __launch_bounds__(512, 4) bp_lds_gen_kernel(LdsGenParams p) {
    extern __shared__ __align__(16) float smem_gen[];
    float* msg = smem_gen;                                      // [slots]
    float* lam = smem_gen + p.slots;                            // [n], by variable
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
    for (int e = t; e < (int)p.E; e += nt) msg[p.edge_slot[e]] = lam[p.edge_col[e]] + 0.0f;
    __syncthreads();

    // early termination keeps the syndrome itself, one bit per check (all decisions 0 and all syndrome
    // bits 0 is the consistent state to start from); bit 2 s + h of mydec is the current decision of
    // variable h of this thread's item in step s
    uint32_t* par = reinterpret_cast<uint32_t*>(red + 4);  // [ceil(m / 32)]
    const int mw = (p.m + 31) >> 5;
    if constexpr (EARLY) {
        for (int w = t; w < mw; w += nt) par[w] = 0u;
        __syncthreads();
    }
    uint32_t mydec = 0u;
    bool converged = false;
    int iters = p.max_iters, synd = 0;
    int4 cd_next = p.cn_desc[t];
    for (int it = 0; it < p.max_iters; ++it) {
        // check nodes: beta -> alpha in place (LogDecoder.java:82-95)
        // the descriptor of the next step is requested before this step's work: a step is a chain of dependent
        // reads (descriptor, then slots) and few waves share a SIMD here
        for (int s = 0; s < p.cn_steps; ++s) {
            const int4 cd = cd_next;
            if (s + 1 < p.cn_steps) cd_next = p.cn_desc[t + (s + 1) * nt];
            if (RULE == kCnComplement && cd.y == cd.w && cd.y <= kLdsGenPairDc) {
                switch (cd.y) {
                    case 0: break;
                    LDPC_GEN_CN(1) LDPC_GEN_CN(2) LDPC_GEN_CN(3) LDPC_GEN_CN(4) LDPC_GEN_CN(5)
                    LDPC_GEN_CN(6) LDPC_GEN_CN(7) LDPC_GEN_CN(8) LDPC_GEN_CN(9) LDPC_GEN_CN(10)
                    default: break;
                }
            } else {
                if (cd.y > 0) cn_dispatch<1, RULE, false>(cd.y, msg + cd.x, nullptr, nullptr, kGenRowStride, 1, nullptr, 0, lane);
                if (cd.w > 0) cn_dispatch<1, RULE, false>(cd.w, msg + cd.z, nullptr, nullptr, kGenRowStride, 1, nullptr, 0, lane);
            }
        }
        // the first descriptors of a phase are requested before the barrier that precedes it
        int4 va_next = p.vn_desc[2 * t], vb_next = p.vn_desc[2 * t + 1];
        __syncthreads();
        // variable nodes: alpha -> beta in place, posterior sign -> decision bit (:67-79, :98-105)
        const bool write_beta = EARLY || (it + 1 < p.max_iters);
        uint32_t newdec = 0u;
        for (int s = 0; s < p.vn_steps; ++s) {
            const int4 va = va_next, vb = vb_next;
            if (s + 1 < p.vn_steps) {
                const int item_next = t + (s + 1) * nt;
                va_next = p.vn_desc[2 * item_next];
                vb_next = p.vn_desc[2 * item_next + 1];
            }
            v2f g = v2f{0.0f, 0.0f};
            if (va.x >= 0 && vb.x >= 0 && va.z == vb.z && va.z <= kLdsGenPairDv) {
                switch (va.z) {
                    LDPC_GEN_VN(0) LDPC_GEN_VN(1) LDPC_GEN_VN(2) LDPC_GEN_VN(3) LDPC_GEN_VN(4) LDPC_GEN_VN(5)
                    LDPC_GEN_VN(6) LDPC_GEN_VN(7) LDPC_GEN_VN(8)
                    default: break;
                }
            } else {
                float g1[1] = {0.0f};
                if (va.x >= 0) {
                    vn_dispatch<1>(va.z, msg, lam + va.x, p.col_slot + va.y, 1, write_beta, g1, nullptr, 0, lane);
                    g.x = g1[0];
                }
                if (vb.x >= 0) {
                    vn_dispatch<1>(vb.z, msg, lam + vb.x, p.col_slot + vb.y, 1, write_beta, g1, nullptr, 0, lane);
                    g.y = g1[0];
                }
            }
            // "(gamma[j] < 0) ? 1 : 0", :104
            const uint32_t da = (va.x >= 0 && g.x < 0.0f) ? 1u : 0u, db = (vb.x >= 0 && g.y < 0.0f) ? 1u : 0u;
            newdec |= (da | (db << 1)) << (2 * s);
            if constexpr (EARLY) {
                // the syndrome is kept bit by bit; a changed decision toggles the bits of its checks
                const uint32_t ch = ((mydec >> (2 * s)) & 3u) ^ (da | (db << 1));
                if (ch & 1u)
                    for (int k = 0; k < va.z; ++k) {
                        const int i = p.col_check[va.y + k];
                        atomicXor(&par[i >> 5], 1u << (i & 31));
                    }
                if (ch & 2u)
                    for (int k = 0; k < vb.z; ++k) {
                        const int i = p.col_check[vb.y + k];
                        atomicXor(&par[i >> 5], 1u << (i & 31));
                    }
            }
        }
        mydec = newdec;
        cd_next = p.cn_desc[t];
        __syncthreads();
        if constexpr (EARLY) {
            // every wave scans all the words for itself: the same verdict everywhere without a reduction or
            // a second barrier (the next toggles come after the barrier that ends the next check-node phase)
            uint32_t any = 0u;
            for (int w = lane; w < mw; w += 64) any |= par[w];
            if (__ballot(any != 0u) == 0ull) {  // "sumSyndro == 0", LogDecoder.java:108-119
                iters = it + 1;
                converged = true;
                break;
            }
        }
    }
    // decisions -> packed words; position of an item's first variable: 128 * (item / 64) + item % 64,
    // of its second: 64 more
    for (int s = 0; s < p.vn_steps; ++s) {
        const int item = t + s * nt;
        const unsigned long long ma = __ballot((mydec >> (2 * s)) & 1u), mb = __ballot((mydec >> (2 * s + 1)) & 1u);
        if (lane == 0) {  // item is a multiple of 64 here
            uint32_t* w = bits + 4 * (item >> 6);
            w[0] = (uint32_t)ma;
            w[1] = (uint32_t)(ma >> 32);
            w[2] = (uint32_t)mb;
            w[3] = (uint32_t)(mb >> 32);
        }
    }
    __syncthreads();
    if constexpr (EARLY) {
        if (!converged) {  // weight of the last iteration's syndrome
            int cnt = 0;
            for (int w = t; w < mw; w += nt) cnt += __popc(par[w]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0 && cnt) atomicAdd(red, cnt);
            __syncthreads();
            synd = *red;
        }
    } else {
        synd = lds_gen_syndrome(p, bits, red, t, nt);
    }

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

#undef LDPC_GEN_CN
#undef LDPC_GEN_VN

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
