// Frame-resident flooding belief propagation with TWO frames per workgroup, half an iteration apart.
//
// bp_lds_kernel (bp_lds.hpp) gives a workgroup one frame: every iteration is a check-node phase (vector
// arithmetic) and a variable-node phase (LDS gathers and scatters) with a workgroup barrier after each, and
// the waves of a workgroup have nothing to issue while they wait at either.  Here the same threads hold two
// frames, A and B, and between two barriers every thread runs the check-node phase of one and the
// variable-node phase of the other (LogDecoder.java:82-95 of one frame beside :67-79, :98-105 of the other):
//
//      CN(A) | VN(A) + CN(B) | CN(A) + VN(B) | VN(A) + CN(B) | ... | VN(B)
//
// One barrier per frame-iteration instead of two, and between barriers every wave has the gathers of one frame
// in flight while it issues the arithmetic of the other.  The two frames take twice the LDS, so one workgroup
// fills a CU.  Same tables, same slot layout, same arithmetic code in the same order as bp_lds_kernel:
// identical bits.  Fixed iteration count only (no per-frame exit), default check-node rule only.
#pragma once
#include "bp_lds.hpp"

namespace ldpc {

// LDS per frame: messages | packed decisions | 4 words | one syndrome bit per check, as in bp_lds_kernel
__host__ __device__ inline size_t lds2_frame_words(int msg_words, int nw, int m) {
    return (size_t)msg_words + (size_t)nw + 4 + (size_t)((m + 31) >> 5);
}

// FB: bytes from frame A's LDS block to frame B's when known at compile time (then it travels in the offset field
// of the LDS instructions; 0: taken from the parameters, one addition per access to frame B)
template <int DC, int DV, int VPT, int WPE, int FB>
__global__ void __launch_bounds__(4096 / VPT, WPE) bp_lds2_kernel(LdsParams p) {
    constexpr int CPT = VPT * DV / DC;
    static_assert(CPT * DC == VPT * DV && CPT % 2 == 0 && VPT % 4 == 0, "checks and variables are processed in pairs");
    extern __shared__ __align__(16) float smem[];
    const int t = threadIdx.x, nt = blockDim.x;
    const uint32_t frame_words = (uint32_t)lds2_frame_words(p.msg_words, p.nw, p.m);
    // 16-byte aligned: rows are read with b128 accesses
    const uint32_t frame_bytes = FB ? (uint32_t)FB : 4u * ((frame_words + 3u) & ~3u);
    char* const base0 = reinterpret_cast<char*>(smem);

    // LDS addresses of the edge slots of this thread's variables in frame A; frame B's lie frame_bytes further on
    uint32_t off[VPT][DV];
    {
        int raw[VPT][DV];
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int k = 0; k < DV; ++k) raw[v][k] = p.vn_off[(long long)(t + v * nt) * DV + k];
#pragma unroll
        for (int v = 0; v < VPT; ++v)
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                off[v][k] = lds_address(base0 + raw[v][k]);
                asm volatile("" : "+v"(off[v][k]));
            }
    }
    const long long frame0 = 2ll * blockIdx.x;
    const int nf = (frame0 + 1 < p.frames) ? 2 : 1;  // the last workgroup of an odd batch holds one frame and a dummy

    // channel LLRs of this thread's variables, both frames (a dummy frame or variable: positive LLR, bit 0)
    v2f vl[2][VPT / 2];
    {
        int pj[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) pj[v] = p.perm[t + v * nt];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float lam[VPT];
            const long long frame = frame0 + (f < nf ? f : 0);
            if (p.llr) {
                const float* src = p.llr + frame * p.n;
#pragma unroll
                for (int v = 0; v < VPT; ++v) lam[v] = src[pj[v] < p.n ? pj[v] : 0];
#pragma unroll
                for (int v = 0; v < VPT; ++v) lam[v] = (pj[v] < p.n && f < nf) ? lam[v] : 8.0f;
#pragma unroll
                for (int v = 0; v < VPT; ++v) lam[v] *= p.scale;
            } else {
                const uint8_t* src = p.r + frame * p.n;
                uint8_t rb[VPT];
#pragma unroll
                for (int v = 0; v < VPT; ++v) rb[v] = src[pj[v] < p.n ? pj[v] : 0];
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const float mag = (pj[v] < p.g_length) ? p.l0 : p.l1;
                    lam[v] = (rb[v] == 0 || pj[v] >= p.n || f >= nf) ? mag : -mag;
                }
            }
#pragma unroll
            for (int v = 0; v < VPT; v += 2) {
                float a = lam[v], b = lam[v + 1];
                asm volatile("" : "+v"(a), "+v"(b));
                vl[f][v / 2] = v2f{a, b};
            }
        }
    }
    // syndrome bits and accumulators of both frames to zero; beta of the first iteration = lambda + 0.0 (LogDecoder.java:57, :70-77)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        uint32_t* bits = reinterpret_cast<uint32_t*>(base0 + f * frame_bytes) + p.msg_words;
        int* red = reinterpret_cast<int*>(bits + p.nw);
        uint32_t* par = reinterpret_cast<uint32_t*>(red + 4);
        if (t == 0) *red = 0;
        for (int w = t; w < (p.m >> 5); w += nt) par[w] = 0u;
#pragma unroll
        for (int v = 0; v < VPT; v += 2)
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                lds_store(off[v][k] + f * frame_bytes, vl[f][v / 2].x + 0.0f);
                lds_store(off[v + 1][k] + f * frame_bytes, vl[f][v / 2].y + 0.0f);
            }
    }
    __syncthreads();

    // check nodes of one frame: beta -> alpha in place, two checks per step in the halves of packed registers
    auto cn_phase = [&](uint32_t fb) {
#pragma unroll
        for (int q = 0; q < CPT / 2; ++q) {
            v2f* row = reinterpret_cast<v2f*>(base0 + fb) + (size_t)(lds_pair_stride(DC) / 2) * (size_t)(t + q * nt);
            v2f x[DC];
#pragma unroll
            for (int k = 0; k < DC; ++k) x[k] = row[k];
            cn_complement<v2f, DC>(x);
#pragma unroll
            for (int k = 0; k < DC; ++k) row[k] = x[k];
        }
    };
    // variable nodes of one frame, in two halves so that the gathers can be issued ahead of the other frame's
    // check-node arithmetic and the sums and scatters follow it
    auto vn_gather = [&](uint32_t fb, v2f (&va)[VPT / 2][DV]) {
#pragma unroll
        for (int v = 0; v < VPT; v += 2)
#pragma unroll
            for (int k = 0; k < DV; ++k) va[v / 2][k] = v2f{lds_load(off[v][k] + fb), lds_load(off[v + 1][k] + fb)};
    };
    auto vn_finish = [&](uint32_t fb, v2f (&va)[VPT / 2][DV], const v2f (&lam)[VPT / 2], bool write_beta, auto decide,
                         unsigned long long (&newm)[VPT]) {
#pragma unroll
        for (int v = 0; v < VPT; v += 2) {
            v2f g;
            vn_compute_v<v2f, DV>(va[v / 2], lam[v / 2], g);
            if (write_beta) {
#pragma unroll
                for (int k = 0; k < DV; ++k) {
                    lds_store(off[v][k] + fb, va[v / 2][k].x);
                    lds_store(off[v + 1][k] + fb, va[v / 2][k].y);
                }
            }
            if constexpr (decltype(decide)::value) {  // "(gamma[j] < 0) ? 1 : 0", LogDecoder.java:104
                newm[v] = __ballot(g.x < 0.0f);
                newm[v + 1] = __ballot(g.y < 0.0f);
            }
        }
    };
    unsigned long long dec[2][VPT];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int v = 0; v < VPT; ++v) dec[f][v] = 0ull;
    const uint32_t fbA = 0u, fbB = frame_bytes;
    const int L = p.max_iters;
    v2f va[VPT / 2][DV];

    cn_phase(fbA);
    __syncthreads();
    for (int it = 0; it + 1 < L; ++it) {
        // A: variable nodes of iteration it; B: check nodes of iteration it
        vn_gather(fbA, va);
        cn_phase(fbB);
        vn_finish(fbA, va, vl[0], true, LdsNo{}, dec[0]);
        __syncthreads();
        // A: check nodes of iteration it + 1; B: variable nodes of iteration it
        vn_gather(fbB, va);
        cn_phase(fbA);
        vn_finish(fbB, va, vl[1], true, LdsNo{}, dec[1]);
        __syncthreads();
    }
    // the last iteration forms the decisions and writes no beta (LogDecoder.java:98-105, :122)
    vn_gather(fbA, va);
    cn_phase(fbB);
    vn_finish(fbA, va, vl[0], false, LdsYes{}, dec[0]);
    __syncthreads();
    vn_gather(fbB, va);
    vn_finish(fbB, va, vl[1], false, LdsYes{}, dec[1]);

    // per frame: decisions -> packed words by position, syndrome bits by toggling the checks of the variables
    // decided 1 (LogDecoder.java:108-116), weight, hard word in the caller's order
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        char* const fbase = base0 + f * frame_bytes;
        uint32_t* bits = reinterpret_cast<uint32_t*>(fbase) + p.msg_words;
        int* red = reinterpret_cast<int*>(bits + p.nw);
        uint32_t* par = reinterpret_cast<uint32_t*>(red + 4);
        const uint32_t lds_base = lds_address(base0);
        if ((t & 63) == 0) {
            uint32_t* b0 = bits + (t >> 5);
            const int step = nt >> 5;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                b0[v * step] = (uint32_t)dec[f][v];
                b0[v * step + 1] = (uint32_t)(dec[f][v] >> 32);
            }
        }
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (dec[f][v] == 0ull) continue;
            if (__builtin_amdgcn_inverse_ballot_w64(dec[f][v])) {
#pragma unroll
                for (int k = 0; k < DV; ++k) {
                    const uint32_t slot = (off[v][k] - lds_base) >> 2;
                    const uint32_t b = 2u * (slot / (uint32_t)lds_pair_stride(DC)) + (slot & 1u);
                    atomicXor(&par[b >> 5], 1u << (b & 31u));
                }
            }
        }
    }
    __syncthreads();
    for (int f = 0; f < nf; ++f) {
        char* const fbase = base0 + f * frame_bytes;
        uint32_t* bits = reinterpret_cast<uint32_t*>(fbase) + p.msg_words;
        int* red = reinterpret_cast<int*>(bits + p.nw);
        uint32_t* par = reinterpret_cast<uint32_t*>(red + 4);
        const long long frame = frame0 + f;
        int cnt = 0;
        for (int w = t; w < (p.m >> 5); w += nt) cnt += __popc(par[w]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if ((t & 63) == 0 && cnt) atomicAdd(red, cnt);
        uint8_t* out = p.hard + frame * p.n;
        const int4* pos4 = reinterpret_cast<const int4*>(p.pos_of_var) + t;
        auto bit = [&](int q) -> unsigned char { return (unsigned char)((bits[q >> 5] >> (q & 31)) & 1u); };
#pragma unroll
        for (int u = 0; u < VPT / 4; ++u)
            if (4 * (t + u * nt) < p.n) {
                const int4 q = pos4[u * nt];
                reinterpret_cast<uchar4*>(out)[t + u * nt] = make_uchar4(bit(q.x), bit(q.y), bit(q.z), bit(q.w));
            }
        __syncthreads();
        if (t == 0) {
            if (p.out_iters) p.out_iters[frame] = L;
            if (p.out_synd) p.out_synd[frame] = *red;
        }
    }
}

__host__ __device__ inline uint32_t lds2_frame_bytes(int msg_words, int nw, int m) {
    return 4u * (((uint32_t)lds2_frame_words(msg_words, nw, m) + 3u) & ~3u);
}

template <int DC, int DV, int VPT, int WPE, int FB>
inline hipError_t bp_lds2_launch_one(const LdsLaunch& l, const LdsParams& p) {
    const dim3 grid((unsigned)((p.frames + 1) / 2)), block((unsigned)l.threads);
    auto k = bp_lds2_kernel<DC, DV, VPT, WPE, FB>;
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, l.lds_bytes, l.stream, p);
    return hipGetLastError();
}

// FB_KNOWN: the frame block size of the code length the kernel is specialised for (n = 4096 for the (6,3) kernels)
template <int DC, int DV, int VPT, int WPE, int FB_KNOWN>
inline hipError_t bp_lds2_launch_impl(const LdsLaunch& l, const LdsParams& p) {
    if (l.threads * VPT < p.n || l.threads > 4096 / VPT || l.threads % 64 || l.early || l.count || l.rule != kCnComplement)
        return hipErrorInvalidConfiguration;
    if (l.lds_bytes < 2 * (size_t)lds2_frame_bytes(p.msg_words, p.nw, p.m)) return hipErrorInvalidValue;
    if (lds2_frame_bytes(p.msg_words, p.nw, p.m) == (uint32_t)FB_KNOWN) return bp_lds2_launch_one<DC, DV, VPT, WPE, FB_KNOWN>(l, p);
    return bp_lds2_launch_one<DC, DV, VPT, WPE, 0>(l, p);
}

// two frames per workgroup, (6,3) codes: l.vpt = 4 (16 wavefronts at n = 4096) or 8 (8 wavefronts)
hipError_t bp_lds2_launch_6_3(const LdsLaunch& l, const LdsParams& p);
unsigned bp_lds2_6_3_params_bytes();

}  // namespace ldpc
