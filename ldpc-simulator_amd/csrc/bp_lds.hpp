// Frame-resident flooding belief propagation for gfx950: one workgroup decodes one frame whose whole
// message state lives in LDS, so HBM sees only the channel LLRs in and the hard decisions out.
//
// Used for regular codes whose messages fit on chip (E floats <= ~78 KiB, two workgroups per CU;
// the channel LLRs of a thread's own variables stay in its registers).
// The arithmetic is the same code as the streaming kernels (cn_compute1 / vn_compute in
// bp_kernels.hpp): same neighbour order, same prefix x suffix leave-one-out, so both engines return
// identical bits.  What changes is the mapping:
//   * thread t owns checks t, t+NT, ... in the CN phase and variables t, t+NT, ... in the VN phase
//     (LogDecoder.java:82-95 and :67-79, :98-105), with a workgroup barrier between the phases;
//   * msg[e] (CSR edge order) is read/written in place: a check's DC messages are DC consecutive
//     floats, a variable's DV messages are gathered through its edge ids, which stay in registers for
//     the whole decode;
//   * a frame that reaches a zero syndrome leaves at once (LogDecoder.java:117-119): the workgroup
//     ends and the next frame's workgroup takes its place, so early termination costs no idle lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernels.hpp"

namespace ldpc {

constexpr int kLdsVpt = 8;  // variables per thread; threads per frame = n / 8, checks per thread = 8 m / n

struct LdsParams {
    int m, n, nw;
    long long E;
    // Tables built once per code on the host (ldpc_amd.hip, build_resident_tables):
    const int* perm;       // [n]  variable handled at VN position p (thread p % nt, step p / nt)
    const int* pos_of_var; // [n]  inverse of perm
    const int* vn_off;     // [n*DV] byte offset in msg of edge k of the variable at position p, [p*DV + k]
    const uint32_t* tog;   // [(kLdsVpt/2)*DV][nt] syndrome bit of the check on edge k of the variables at positions
                           // t + v*nt (low half) and t + (v+1)*nt (high half), v even, at [(v/2*DV + k)*nt + t]
    const float* llr;      // [frames][n] channel LLRs, or null
    const uint8_t* r;      // [frames][n] hard received words (then lambda = +-l0 / +-l1), or null
    float l0, l1;          // already in the rule's LLR unit
    float scale;           // channel LLR unit conversion (see FloatLlr)
    int g_length;          // n - m
    long long frames;
    int max_iters;
    uint8_t* hard;         // [frames][n]
    int32_t* out_iters;    // [frames] or null
    int32_t* out_synd;     // [frames] or null
};

// LDS slot of CSR edge e = (check i, position k): the rows of the two checks a thread processes
// together (i = t + c*nt for c = 2q, 2q+1) are interleaved element by element, so that
// (check 2q position k, check 2q+1 position k) is one aligned 8-byte pair = one packed register.
__host__ __device__ inline int lds_slot(int e, int dc, int nt) {
    const int i = e / dc, k = e - i * dc;
    const int c = i / nt, t = i - c * nt;
    return 2 * dc * (t + (c >> 1) * nt) + 2 * k + (c & 1);
}

// LDS accesses by 32-bit LDS address
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ uint32_t lds_address(const void* generic) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)generic;
}
__device__ __forceinline__ float lds_load(uint32_t a) { return *(const lds_float*)(uintptr_t)a; }
__device__ __forceinline__ void lds_store(uint32_t a, float v) { *(lds_float*)(uintptr_t)a = v; }

// CPT (even) checks and kLdsVpt variables per thread, exactly: m = CPT * blockDim.x, n = kLdsVpt * blockDim.x.
// WPE = wavefronts per SIMD the register budget is set for: 6 (80 registers, three 8-wave workgroups
// per CU) or 4 (128 registers, two).
template <int DC, int DV, int RULE, bool EARLY, int CPT, int WPE>
__global__ void __launch_bounds__(512, WPE) bp_lds_kernel(LdsParams p) {
    static_assert(CPT % 2 == 0 && kLdsVpt % 2 == 0, "checks and variables are processed in pairs");
    extern __shared__ __align__(16) float smem[];
    float* msg = smem;                                        // [E]
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem + p.E); // [nw], by VN position
    int* red = reinterpret_cast<int*>(bits + p.nw);           // syndrome accumulator
    const int t = threadIdx.x, nt = blockDim.x;
    char* msgb = reinterpret_cast<char*>(msg);

    // LDS addresses of the edge slots of this thread's variables: fixed for the whole decode, and
    // kept opaque so that they stay in registers instead of being re-formed as base + offset per use
    uint32_t off[kLdsVpt][DV];
#pragma unroll
    for (int v = 0; v < kLdsVpt; ++v)
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            off[v][k] = lds_address(msgb + p.vn_off[(long long)(t + v * nt) * DV + k]);
            asm volatile("" : "+v"(off[v][k]));
        }
    // With early termination a workgroup decodes frames blockIdx.x, blockIdx.x + gridDim.x, ... in turn:
    // the addresses above are formed once for all of them, which counts when a frame takes a handful of
    // iterations.  At fixed iterations it takes one frame (the launch gives it no second one), and the
    // compiler sees a body that runs once and keeps nothing else alive across it.
    for (long long frame = blockIdx.x; frame < p.frames; frame += gridDim.x) {
    // channel LLR of variable j of this frame (LogDecoder.java:49-54 for a hard received word)
    const float* src_llr = p.llr ? p.llr + frame * p.n : nullptr;
    const uint8_t* src_r = p.r ? p.r + frame * p.n : nullptr;
    auto lam_of = [&](int j) -> float {
        if (src_llr) return src_llr[j] * p.scale;
        const float mag = (j < p.g_length) ? p.l0 : p.l1;
        return (src_r[j] == 0) ? mag : -mag;
    };
    if (t == 0) *red = 0;
    // The syndrome is kept as one bit per check.  With early termination it is maintained every iteration
    // by toggling the checks of the variables whose decision changed since the previous one (few, after
    // the first iterations); at fixed iterations it is formed once at the end, from the variables decided
    // 1.  All decisions 0 and all syndrome bits 0 is the consistent state to start from.
    uint32_t* par = reinterpret_cast<uint32_t*>(red + 4);   // [m / 32]
    const int mw = p.m >> 5;
    for (int w = t; w < mw; w += nt) par[w] = 0u;
    uint32_t mydec = 0u;  // bit v: the current decision of this thread's variable v

    // channel LLRs of this thread's own variables stay in registers for the whole decode
    v2f vl[kLdsVpt / 2];
#pragma unroll
    for (int v = 0; v < kLdsVpt; v += 2) {
        float a = lam_of(p.perm[t + v * nt]), b = lam_of(p.perm[t + (v + 1) * nt]);
        asm volatile("" : "+v"(a), "+v"(b));  // loaded once, here
        vl[v / 2] = v2f{a, b};
    }
    // beta of the first iteration: lambda + the zero-initialised alpha (LogDecoder.java:57, :70-77),
    // written from the variable side: the slots of a variable's edges are the addresses held above
#pragma unroll
    for (int v = 0; v < kLdsVpt; v += 2)
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            lds_store(off[v][k], vl[v / 2].x + 0.0f);
            lds_store(off[v + 1][k], vl[v / 2].y + 0.0f);
        }
    __syncthreads();

    // toggles the syndrome bits of the checks of those of this thread's variables whose bit is set in
    // `which`.  The bit numbers come from a table (12 coalesced words per lane, read when a toggle is due,
    // from a 24 KiB array that every workgroup shares): deriving them from the slot addresses cost a division
    // by 2*DC per edge and made the first iterations, where most lanes toggle, twice as long as the later ones.
    auto toggle_checks = [&](uint32_t which) {
        const uint32_t* tog = p.tog + t;
        asm volatile("" : "+v"(tog));  // read here, when needed: 12 more loop invariants do not fit
        uint32_t x[kLdsVpt / 2][DV];
#pragma unroll
        for (int h = 0; h < kLdsVpt / 2; ++h)
#pragma unroll
            for (int k = 0; k < DV; ++k) x[h][k] = tog[(h * DV + k) * nt];
#pragma unroll
        for (int v = 0; v < kLdsVpt; ++v) {
            if ((which >> v) & 1u) {
#pragma unroll
                for (int k = 0; k < DV; ++k) {
                    const uint32_t b = (v & 1) ? (x[v / 2][k] >> 16) : (x[v / 2][k] & 0xffffu);
                    atomicXor(&par[b >> 5], 1u << (b & 31u));
                }
            }
        }
    };

    int iters = p.max_iters, synd = 0;
    bool converged = false;
    for (int it = 0; it < p.max_iters; ++it) {
        // check nodes: beta -> alpha in place, two checks per step in the halves of packed registers
#pragma unroll
        for (int q = 0; q < CPT / 2; ++q) {
            v2f* row = reinterpret_cast<v2f*>(msg + 2 * DC * (t + q * nt));
            v2f x[DC];
#pragma unroll
            for (int k = 0; k < DC; ++k) x[k] = row[k];
            if constexpr (RULE == kCnComplement) {
                cn_complement<v2f, DC>(x);
            } else {
                float xa[DC], xb[DC];
#pragma unroll
                for (int k = 0; k < DC; ++k) { xa[k] = x[k].x; xb[k] = x[k].y; }
                cn_compute1<DC, RULE>(xa);
                cn_compute1<DC, RULE>(xb);
#pragma unroll
                for (int k = 0; k < DC; ++k) x[k] = v2f{xa[k], xb[k]};
            }
#pragma unroll
            for (int k = 0; k < DC; ++k) row[k] = x[k];
        }
        __syncthreads();
        // variable nodes: alpha -> beta in place, posterior sign -> decision bit
        // (all gathers are issued before any scatter: distinct variables never share a slot)
        const bool write_beta = EARLY || (it + 1 < p.max_iters);
        v2f va[kLdsVpt / 2][DV];
#pragma unroll
        for (int v = 0; v < kLdsVpt; v += 2)
#pragma unroll
            for (int k = 0; k < DV; ++k) va[v / 2][k] = v2f{lds_load(off[v][k]), lds_load(off[v + 1][k])};
        uint32_t newdec = 0u;
#pragma unroll
        for (int v = 0; v < kLdsVpt; v += 2) {
            v2f g;
            vn_compute_v<v2f, DV>(va[v / 2], vl[v / 2], g);   // two variables per packed register
            if (write_beta) {
#pragma unroll
                for (int k = 0; k < DV; ++k) {
                    lds_store(off[v][k], va[v / 2][k].x);
                    lds_store(off[v + 1][k], va[v / 2][k].y);
                }
            }
            // "(gamma[j] < 0) ? 1 : 0", :104
            newdec |= (g.x < 0.0f ? 1u : 0u) << v | (g.y < 0.0f ? 1u : 0u) << (v + 1);
        }
        if constexpr (EARLY) {
            const uint32_t changed = newdec ^ mydec;
            mydec = newdec;
            if (changed) toggle_checks(changed);  // rare after the first iterations
            __syncthreads();
            // "sumSyndro == 0" (LogDecoder.java:108-119): no bit set in any word.  Every wave scans all the
            // words for itself, so all reach the same verdict without a reduction or a second barrier (the
            // next toggles come after the barrier that ends the next check-node phase).
            const uint32_t any = ((t & 63) < mw) ? par[t & 63] : 0u;  // m <= 2048 here: at most 64 words
            const bool bad = __ballot(any != 0u) != 0ull;
            if (!bad) {
                iters = it + 1;
                converged = true;
                break;
            }
        } else {
            mydec = newdec;
            __syncthreads();
        }
    }
    // decisions -> packed words by VN position (q is a multiple of 64 in lane 0, n a multiple of 64)
#pragma unroll
    for (int v = 0; v < kLdsVpt; ++v) {
        const unsigned long long mask = __ballot((mydec >> v) & 1u);
        if ((t & 63) == 0) {
            const int q = t + v * nt;
            bits[q >> 5] = (uint32_t)mask;
            bits[(q >> 5) + 1] = (uint32_t)(mask >> 32);
        }
    }
    __syncthreads();
    if constexpr (!EARLY) {  // "sumSyndro" of the last decisions (LogDecoder.java:108-116)
        toggle_checks(mydec);
        __syncthreads();
    }
    if (!converged) {  // weight of the syndrome
        int cnt = 0;
        for (int w = t; w < mw; w += nt) cnt += __popc(par[w]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((t & 63) == 0 && cnt) atomicAdd(red, cnt);
        __syncthreads();
        synd = *red;
    }

    uint8_t* out = p.hard + frame * p.n;
    if ((p.n & 3) == 0) {
        for (int j4 = t; j4 < (p.n >> 2); j4 += nt) {
            const int4 q = reinterpret_cast<const int4*>(p.pos_of_var)[j4];
            reinterpret_cast<uchar4*>(out)[j4] =
                make_uchar4((bits[q.x >> 5] >> (q.x & 31)) & 1u, (bits[q.y >> 5] >> (q.y & 31)) & 1u,
                            (bits[q.z >> 5] >> (q.z & 31)) & 1u, (bits[q.w >> 5] >> (q.w & 31)) & 1u);
        }
    } else {
        for (int j = t; j < p.n; j += nt) {
            const int q = p.pos_of_var[j];
            out[j] = (uint8_t)((bits[q >> 5] >> (q & 31)) & 1u);
        }
    }
    if (t == 0) {
        if (p.out_iters) p.out_iters[frame] = iters;
        if (p.out_synd) p.out_synd[frame] = synd;
    }
    if constexpr (!EARLY) break;
    __syncthreads();  // the next frame reuses the LDS
    }
}

struct LdsLaunch {
    int rule;
    bool early;
    int frames_per_wg;  // frames one workgroup decodes in turn (>= 1; early termination only)
    int wpe;  // 4 or 6, see bp_lds_kernel
    int threads;
    size_t lds_bytes;
    hipStream_t stream;
};

template <int DC, int DV, int CPT>
inline hipError_t bp_lds_launch_impl(const LdsLaunch& l, const LdsParams& p) {
    const long long fpw = (l.early && l.frames_per_wg > 0) ? l.frames_per_wg : 1;
    const dim3 grid((unsigned)((p.frames + fpw - 1) / fpw)), block((unsigned)l.threads);
#define LDPC_LDS2(R, E, W)                                                                                   \
    do {                                                                                                     \
        auto k = bp_lds_kernel<DC, DV, R, E, CPT, W>;                                                        \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                           (int)l.lds_bytes);                                                \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL(k, grid, block, l.lds_bytes, l.stream, p);                                        \
    } while (0)
#define LDPC_LDS(R, E)                                                                                       \
    do {                                                                                                     \
        if (l.wpe >= 6) LDPC_LDS2(R, E, 6); else LDPC_LDS2(R, E, 4);                                         \
    } while (0)
    switch (l.rule) {
        case kCnMinSum: if (l.early) LDPC_LDS(kCnMinSum, true); else LDPC_LDS(kCnMinSum, false); break;
        case kCnSumProduct: if (l.early) LDPC_LDS(kCnSumProduct, true); else LDPC_LDS(kCnSumProduct, false); break;
        default: if (l.early) LDPC_LDS(kCnComplement, true); else LDPC_LDS(kCnComplement, false); break;
    }
#undef LDPC_LDS2
#undef LDPC_LDS
    return hipGetLastError();
}

hipError_t bp_lds_launch_6_3(const LdsLaunch& l, const LdsParams& p);
unsigned bp_lds_6_3_params_bytes();  // see bp_kernels.hpp
unsigned bp_lds_8_2_params_bytes();
hipError_t bp_lds_launch_8_2(const LdsLaunch& l, const LdsParams& p);

}  // namespace ldpc
