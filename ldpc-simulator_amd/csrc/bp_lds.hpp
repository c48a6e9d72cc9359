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

constexpr int kLdsVpt = 8;   // variables per thread; threads per frame = n / 8, checks per thread = 8 m / n
constexpr int kLdsVptWide = 16;  // the second tiling: half as many threads per frame, each with twice the nodes
constexpr int kLdsVptMid = 12;   // a third one for the (6,3) kernels: n = 3072 in 4 wavefronts instead of 6 (or 1536 in 2 for 3)

struct LdsParams {
    int m, n, nw;          // n: the caller's code length (stride of its frames); m, nw: checks and decision words of the
                           // tiling, which is larger than the code when n is not (variables per thread) x blockDim.x
    long long E;
    int msg_words;         // floats of LDS the messages take: (m / 2) * lds_pair_stride(DC)
    // Tables built once per code on the host (host_tables.hip, build_resident_tables):
    const int* perm;       // [n]  variable handled at VN position p (thread p % nt, step p / nt)
    const int* pos_of_var; // [n]  inverse of perm
    const int* vn_off;     // [n*DV] byte offset in msg of edge k of the variable at position p, [p*DV + k]
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
    // Error accounting inside the kernel (a Monte-Carlo leg, COUNT kernels only): the decisions are compared with
    // the transmitted word where they are formed and only the counters of count_errors_kernel leave the
    // chip - no hard[] rows, no iteration and syndrome arrays, no second pass over the batch.
    const uint8_t* cw;             // [frames][n] transmitted words, or null = the all-zero word
    unsigned long long* counters;  // [16 + max_iters], added to with atomics
    int count_rx;                  // also the channel-side counters (errors the channel made, bits decoded wrong that it had not touched)
};

// LDS slot of CSR edge e = (check i, position k): the rows of the two checks a thread processes
// together (i = t + c*nt for c = 2q, 2q+1) are interleaved element by element, so that
// (check 2q position k, check 2q+1 position k) is one aligned 8-byte pair = one packed register.
// Floats from one row pair to the next.  The lanes of a wave read their pairs with 16-byte accesses, which the
// LDS serves in groups of 16 lanes (reads, 64 banks) or 8 lanes (writes, 32 banks): the group is conflict-free
// when (stride / 4) is odd.  2 * 6 = 12 is; 2 * 8 = 16 puts every fourth lane on the same banks (4-way: the (8,2)
// kernel ran with its LDS 98 % busy, two thirds of that conflicts), so its rows are 4 floats apart.
__host__ __device__ constexpr int lds_pair_stride(int dc) { return 2 * dc + ((dc / 2) % 2 == 0 ? 4 : 0); }

__host__ __device__ inline int lds_slot(int e, int dc, int nt) {
    const int i = e / dc, k = e - i * dc;
    const int c = i / nt, t = i - c * nt;
    return lds_pair_stride(dc) * (t + (c >> 1) * nt) + 2 * k + (c & 1);
}

// LDS accesses by 32-bit LDS address
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ uint32_t lds_address(const void* generic) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)generic;
}
__device__ __forceinline__ float lds_load(uint32_t a) { return *(const lds_float*)(uintptr_t)a; }
__device__ __forceinline__ void lds_store(uint32_t a, float v) { *(lds_float*)(uintptr_t)a = v; }

// byte offset x of a message slot -> its pair row: x / (4 * lds_pair_stride(dc)) == (x * magic) >> 22 for every slot of a
// code these kernels take (at most 1024 pair rows), the product below 2^32
__host__ __device__ constexpr uint32_t pair_row_magic(int dc) {
    return ((1u << 22) + 4u * (uint32_t)lds_pair_stride(dc) - 1u) / (4u * (uint32_t)lds_pair_stride(dc));
}
__host__ __device__ constexpr bool pair_row_magic_exact(int dc) {
    // both sides are non-decreasing in x, and x / bytes steps only at the multiples of `bytes`: looking at each
    // multiple and at its predecessor covers every x
    const uint32_t bytes = 4u * (uint32_t)lds_pair_stride(dc), magic = pair_row_magic(dc);
    for (uint32_t row = 1; row <= 1024u; ++row) {
        const uint32_t x = row * bytes;
        if (((x - 1u) * magic) >> 22 != row - 1u) return false;
        if (row < 1024u && (x * magic) >> 22 != row) return false;
    }
    return (unsigned long long)(1024u * bytes - 1u) * magic < (1ull << 32) && magic < (1u << 24);
}
template <int DC> constexpr uint32_t kPairRowMagic = pair_row_magic(DC);
static_assert(pair_row_magic_exact(6) && pair_row_magic_exact(8) && pair_row_magic_exact(10), "slot -> pair row by multiplication");

template <int V> struct LdsInt { static constexpr int value = V; };
struct LdsYes { static constexpr bool value = true; };
struct LdsNo { static constexpr bool value = false; };

// VPT variables and CPT = VPT * DV / DC (even) checks per thread, exactly: the tiling has VPT * blockDim.x variables
// and CPT * blockDim.x checks (p.m); a shorter code fills the rest with dummy nodes (LdsParams::n).
// WPE = wavefronts per SIMD the register budget is set for.  VPT = 8: 6 (80 registers, three 8-wave workgroups
// per CU) or 4 (128 registers, two).  VPT = 16: 3 (168 registers, three 4-wave workgroups per CU at n = 4096).
// VPT = 12: 4 (103 registers).
// COUNT: error accounting in place of the outputs (LdsParams::counters; LDPC_GallagerIntStreamSimu.java:136-183).
//
// DEF: codes that are one edge short of regular in places (checks of degree DC or DC - 1, variables of degree DV or
// DV - 1, as many short checks as short variables: a quasi-cyclic code with an empty block in its base matrix,
// GenerateMatrix.java:100).  The host completes the graph with one extra edge per short node, from a short
// variable to a short check, and marks it in the slot table (bit 0 of the variable's last slot offset).  The
// extra edge carries constants that leave both nodes' arithmetic exactly as the reference's shorter loops have
// it: the check reads +Inf on it (a factor tanh = 1: complement q = 0, which the prefix and suffix steps pass
// through unchanged, sign +), the variable reads 0.0 on it as the last term of its sums (x + 0.0 = x).  The
// variable-node phase owns both constants: it replaces what the check wrote there by 0.0 when it gathers, and
// writes +Inf instead of a sum when it scatters; the syndrome toggles skip the edge.  No message ever travels
// between the two nodes, so nothing the reference keeps apart is joined (a real connection would let a NaN
// or an infinity of one node reach the other).
template <int DC, int DV, int RULE, bool EARLY, int VPT, int WPE, bool COUNT = false, bool DEF = false>
__global__ void __launch_bounds__(4096 / VPT, WPE) bp_lds_kernel(LdsParams p) {
    constexpr int CPT = VPT * DV / DC;
    static_assert(!COUNT || VPT <= 8, "the accounting keeps three 8-bit masks of a thread's variables in one register");
    constexpr int kLdsVpt = VPT;  // shadows the namespace constant: everything below is written for VPT
    static_assert(CPT * DC == VPT * DV && CPT % 2 == 0 && VPT % 4 == 0, "checks and variables are processed in pairs");
    extern __shared__ __align__(16) float smem[];
    float* msg = smem;                                        // [msg_words]
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem + p.msg_words); // [nw], by VN position
    int* red = reinterpret_cast<int*>(bits + p.nw);           // syndrome accumulator
    const int t = threadIdx.x, nt = blockDim.x;
    char* msgb = reinterpret_cast<char*>(msg);

    // LDS addresses of the edge slots of this thread's variables: fixed for the whole decode, and
    // kept opaque so that they stay in registers instead of being re-formed as base + offset per use
    const uint32_t lds_base = lds_address(msgb);
    uint32_t off[kLdsVpt][DV];
    // DEF: lane masks of the short variables among this thread's (their last edge is the extra one)
    unsigned long long shortm[DEF ? kLdsVpt : 1];
    auto is_short = [&](int v) -> bool {
        if constexpr (DEF) return __builtin_amdgcn_inverse_ballot_w64(shortm[v]);
        return false;
    };
    {
        // all table reads first, one wait for the lot (made opaque one by one, each read was waited for in turn)
        int raw[kLdsVpt][DV];
#pragma unroll
        for (int v = 0; v < kLdsVpt; ++v)
#pragma unroll
            for (int k = 0; k < DV; ++k) raw[v][k] = p.vn_off[(long long)(t + v * nt) * DV + k];
        if constexpr (DEF) {
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) {
                shortm[v] = __ballot((raw[v][DV - 1] & 1) != 0);
                raw[v][DV - 1] &= ~1;
            }
        }
#pragma unroll
        for (int v = 0; v < kLdsVpt; ++v)
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                off[v][k] = lds_address(msgb + raw[v][k]);
                asm volatile("" : "+v"(off[v][k]));
            }
    }
    // With early termination a workgroup decodes frames blockIdx.x, blockIdx.x + gridDim.x, ... in turn:
    // the addresses above are formed once for all of them, which counts when a frame takes a handful of
    // iterations.  At fixed iterations it takes one frame (the launch gives it no second one), and the
    // compiler sees a body that runs once and keeps nothing else alive across it.
    // accounting: two sums per wavefront of the frame in flight, then the workgroup's share of the counters (flushed when it ends)
    unsigned* fsum = reinterpret_cast<unsigned*>(red + 4 + ((p.m + 31) >> 5));
    unsigned long long* wcnt = reinterpret_cast<unsigned long long*>(fsum + 2 * (4096 / 64 / VPT));
    const int nc = 16 + p.max_iters;
    if constexpr (COUNT)
        for (int i = t; i < nc; i += nt) wcnt[i] = 0ull;
    for (long long frame = blockIdx.x; frame < p.frames; frame += gridDim.x) {
    if (t == 0) *red = 0;
    // The syndrome is kept as one bit per check.  With early termination it is maintained every iteration
    // by toggling the checks of the variables whose decision changed since the previous one (few, after
    // the first iterations); at fixed iterations it is formed once at the end, from the variables decided
    // 1.  All decisions 0 and all syndrome bits 0 is the consistent state to start from.
    uint32_t* par = reinterpret_cast<uint32_t*>(red + 4);   // [m / 32]
    const int mw = p.m >> 5;
    for (int w = t; w < mw; w += nt) par[w] = 0u;

    // Channel LLRs of this thread's own variables (LogDecoder.java:49-54 for a hard received word); they stay
    // in registers for the whole decode.  The eight table reads go out together and so do the eight gathers
    // that depend on them: two memory round trips per frame (one branch per variable made it sixteen, each
    // waited for in turn, and that was most of what a frame cost beyond its iterations).  tt: the per-frame
    // address arithmetic is kept inside the frame loop, where its registers are free again afterwards.
    int tt = t;
    asm volatile("" : "+v"(tt));
    float lam8[kLdsVpt];
    // accounting: bit v = transmitted bit of variable v, bit 8 + v = its received hard bit, bit 16 + v = it is an
    // information position
    uint32_t side = 0u;
    {
        const int* permp = p.perm + tt;
        int pj[kLdsVpt];
#pragma unroll
        for (int v = 0; v < kLdsVpt; ++v) pj[v] = permp[v * nt];
        // a variable past the end of the code (a code tiled as a larger one) reads nothing: positive LLR, bit 0
        int pr[kLdsVpt];
#pragma unroll
        for (int v = 0; v < kLdsVpt; ++v) pr[v] = pj[v] < p.n ? pj[v] : 0;
        uint8_t cb[kLdsVpt];
        if constexpr (COUNT) {
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) cb[v] = 0;
            if (p.cw) {
                const uint8_t* src_cw = p.cw + frame * p.n;
#pragma unroll
                for (int v = 0; v < kLdsVpt; ++v) cb[v] = src_cw[pr[v]];
#pragma unroll
                for (int v = 0; v < kLdsVpt; ++v) cb[v] = pj[v] < p.n ? cb[v] : (uint8_t)0;
            }
        }
        uint32_t rx = 0u;  // received hard bits: "llr < 0" of the caller's value, or the received word itself
        if (p.llr) {
            const float* src_llr = p.llr + frame * p.n;
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) lam8[v] = src_llr[pr[v]];
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) lam8[v] = pj[v] < p.n ? lam8[v] : 8.0f;
            if constexpr (COUNT) {
#pragma unroll
                for (int v = 0; v < kLdsVpt; ++v) rx |= (lam8[v] < 0.0f ? 1u : 0u) << v;
            }
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) lam8[v] *= p.scale;
        } else {
            const uint8_t* src_r = p.r + frame * p.n;
            uint8_t rb[kLdsVpt];
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) rb[v] = src_r[pr[v]];
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) rb[v] = pj[v] < p.n ? rb[v] : (uint8_t)0;
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) {
                const float mag = (pj[v] < p.g_length) ? p.l0 : p.l1;
                lam8[v] = (rb[v] == 0) ? mag : -mag;
                if constexpr (COUNT) rx |= (rb[v] != 0 ? 1u : 0u) << v;
            }
        }
        if constexpr (COUNT) {
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) side |= (pj[v] < p.g_length ? 1u : 0u) << (16 + v) | (cb[v] != 0 ? 1u : 0u) << v;
            side |= rx << 8;
            asm volatile("" : "+v"(side));  // packed here, not where it is used: the loaded bytes must not live until then
        }
    }
    v2f vl[kLdsVpt / 2];
#pragma unroll
    for (int v = 0; v < kLdsVpt; v += 2) {
        float a = lam8[v], b = lam8[v + 1];
        asm volatile("" : "+v"(a), "+v"(b));  // loaded once, here
        vl[v / 2] = v2f{a, b};
    }
    // beta of the first iteration: lambda + the zero-initialised alpha (LogDecoder.java:57, :70-77),
    // written from the variable side: the slots of a variable's edges are the addresses held above
#pragma unroll
    for (int v = 0; v < kLdsVpt; v += 2)
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            float bx = vl[v / 2].x + 0.0f, by = vl[v / 2].y + 0.0f;
            if constexpr (DEF) {
                if (k == DV - 1) {
                    bx = is_short(v) ? __builtin_inff() : bx;
                    by = is_short(v + 1) ? __builtin_inff() : by;
                }
            }
            lds_store(off[v][k], bx);
            lds_store(off[v + 1][k], by);
        }
    __syncthreads();

    // Decisions are kept as lane masks, one 64-bit scalar per variable of the thread (bit l: the decision of lane l's
    // variable v): a sign test writes such a mask directly, "did any lane change" is scalar arithmetic, and the
    // lanes that have to act are selected by the mask itself.  (As bits of a vector register, eight compares,
    // selects and merges per iteration and eight more tests to take them apart cost 9 % of an iteration.)
    // toggle_var(v, lanes): toggles the syndrome bits of the checks of variable v in the lanes of the mask (static
    // indices into off[][]: a run-time index would move the array to scratch memory)
    auto toggle_var = [&](auto vc, unsigned long long lanes) {
        constexpr int v = decltype(vc)::value;
        if (lanes == 0ull) return;
        if (__builtin_amdgcn_inverse_ballot_w64(lanes)) {
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                // slot -> syndrome bit: 2 * (pair row) + (which check of the pair); any fixed
                // one-to-one numbering of the checks serves
                uint32_t o = off[v][k];
                asm volatile("" : "+v"(o));  // formed here, when needed: 24 more loop invariants do not fit
                // byte offset / (bytes of a pair row) by a 24-bit multiplication (full rate; the 32-bit division the
                // compiler forms takes a quarter-rate v_mul_hi_u32 per edge)
                const uint32_t x = o - lds_base;
                const uint32_t b = 2u * ((uint32_t)__umul24(x, kPairRowMagic<DC>) >> 22) + ((x >> 2) & 1u);
                if constexpr (DEF) {
                    if (k == DV - 1 && is_short(v)) continue;  // the extra edge is not an edge of the code
                }
                atomicXor(&par[b >> 5], 1u << (b & 31u));
            }
        }
    };
    auto toggle_all = [&](const unsigned long long (&lanes)[kLdsVpt]) {
        toggle_var(LdsInt<0>{}, lanes[0]); toggle_var(LdsInt<1>{}, lanes[1]);
        toggle_var(LdsInt<2>{}, lanes[2]); toggle_var(LdsInt<3>{}, lanes[3]);
        if constexpr (kLdsVpt > 4) {
            toggle_var(LdsInt<4>{}, lanes[4]); toggle_var(LdsInt<5>{}, lanes[5]);
            toggle_var(LdsInt<6>{}, lanes[6]); toggle_var(LdsInt<7>{}, lanes[7]);
        }
        if constexpr (kLdsVpt > 8) {
            toggle_var(LdsInt<8>{}, lanes[8]); toggle_var(LdsInt<9>{}, lanes[9]);
            toggle_var(LdsInt<10>{}, lanes[10]); toggle_var(LdsInt<11>{}, lanes[11]);
        }
        if constexpr (kLdsVpt > 12) {
            toggle_var(LdsInt<12>{}, lanes[12]); toggle_var(LdsInt<13>{}, lanes[13]);
            toggle_var(LdsInt<14>{}, lanes[14]); toggle_var(LdsInt<15>{}, lanes[15]);
        }
    };
    unsigned long long decm[kLdsVpt];  // the current decisions
#pragma unroll
    for (int v = 0; v < kLdsVpt; ++v) decm[v] = 0ull;

    int iters = p.max_iters, synd = 0;
    bool converged = false;
    // One iteration up to the barrier that ends its variable-node phase.  DECIDE: also form the decisions; at
    // fixed iterations only the last iteration's are ever read (LogDecoder.java:98-105 forms them every time,
    // LogDecoder.java:122 returns the last), so the others skip the posterior sum and the sign tests.
    auto iteration = [&](auto decide, bool write_beta, unsigned long long (&newm)[kLdsVpt]) {
        constexpr bool DECIDE = decltype(decide)::value;
        // check nodes: beta -> alpha in place, two checks per step in the halves of packed registers
#pragma unroll
        for (int q = 0; q < CPT / 2; ++q) {
            v2f* row = reinterpret_cast<v2f*>(msg + lds_pair_stride(DC) * (t + q * nt));
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
        v2f va[kLdsVpt / 2][DV];
#pragma unroll
        for (int v = 0; v < kLdsVpt; v += 2)
#pragma unroll
            for (int k = 0; k < DV; ++k) va[v / 2][k] = v2f{lds_load(off[v][k]), lds_load(off[v + 1][k])};
        if constexpr (DEF) {  // what a short variable's extra edge holds is the check's business: the variable adds 0.0
#pragma unroll
            for (int v = 0; v < kLdsVpt; v += 2) {
                va[v / 2][DV - 1].x = is_short(v) ? 0.0f : va[v / 2][DV - 1].x;
                va[v / 2][DV - 1].y = is_short(v + 1) ? 0.0f : va[v / 2][DV - 1].y;
            }
        }
#pragma unroll
        for (int v = 0; v < kLdsVpt; v += 2) {
            v2f g;
            vn_compute_v<v2f, DV>(va[v / 2], vl[v / 2], g);   // two variables per packed register
            if (write_beta) {
                if constexpr (DEF) {  // ... and the check reads a factor of exactly 1 from it
                    va[v / 2][DV - 1].x = is_short(v) ? __builtin_inff() : va[v / 2][DV - 1].x;
                    va[v / 2][DV - 1].y = is_short(v + 1) ? __builtin_inff() : va[v / 2][DV - 1].y;
                }
#pragma unroll
                for (int k = 0; k < DV; ++k) {
                    lds_store(off[v][k], va[v / 2][k].x);
                    lds_store(off[v + 1][k], va[v / 2][k].y);
                }
            }
            // "(gamma[j] < 0) ? 1 : 0", :104
            if constexpr (DECIDE) {
                newm[v] = __ballot(g.x < 0.0f);
                newm[v + 1] = __ballot(g.y < 0.0f);
            }
        }
    };
    if constexpr (EARLY) {
        for (int it = 0; it < p.max_iters; ++it) {
            unsigned long long newm[kLdsVpt];
            iteration(LdsYes{}, true, newm);
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) {
                const unsigned long long now = newm[v];
                newm[v] ^= decm[v];     // the lanes whose decision changed: few after the first iterations
                decm[v] = now;
            }
            toggle_all(newm);
            __syncthreads();
            // "sumSyndro == 0" (LogDecoder.java:108-119): no bit set in any word.  Every wave scans all the
            // words for itself, so all reach the same verdict without a reduction or a second barrier (the
            // next toggles come after the barrier that ends the next check-node phase).
            // (the lane index afresh: the address of the word would otherwise sit in a register of its own for the
            // whole decode, or in scratch memory)
            int tl = t;
            asm volatile("" : "+v"(tl));
            const uint32_t any = ((tl & 63) < mw) ? par[tl & 63] : 0u;  // m <= 2048 here: at most 64 words
            const bool bad = __ballot(any != 0u) != 0ull;
            if (!bad) {
                iters = it + 1;
                converged = true;
                break;
            }
        }
    } else {
        for (int it = 0; it + 1 < p.max_iters; ++it) {
            iteration(LdsNo{}, true, decm);
            __syncthreads();
        }
        iteration(LdsYes{}, false, decm);
        __syncthreads();
    }
    // decisions -> packed words by VN position: position t + v*nt of lane 0 of a wave is a multiple of 64, so its
    // word is (t >> 5) + v * (nt >> 5)
    if constexpr (!COUNT) {
        if ((tt & 63) == 0) {
            uint32_t* b0 = bits + (tt >> 5);
            const int step = nt >> 5;
#pragma unroll
            for (int v = 0; v < kLdsVpt; ++v) {
                b0[v * step] = (uint32_t)decm[v];
                b0[v * step + 1] = (uint32_t)(decm[v] >> 32);
            }
        }
    }
    __syncthreads();
    if constexpr (!EARLY) {  // "sumSyndro" of the last decisions (LogDecoder.java:108-116)
        toggle_all(decm);
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

    if constexpr (COUNT) {
        // The sums of count_errors_kernel from the masks every thread holds of its own variables: decisions that
        // differ from the transmitted word at information / parity positions, channel errors at information
        // positions, wrong decisions where the channel had made none.  Each at most 8 per thread and n per
        // frame: two of them share a word.
        uint32_t mydec = 0u;  // this thread's decisions, bit v = variable v
#pragma unroll
        for (int v = 0; v < kLdsVpt; ++v) mydec |= __builtin_amdgcn_inverse_ballot_w64(decm[v]) ? (1u << v) : 0u;
        const uint32_t sent = side & 0xffu;
        const uint32_t chan = ((side >> 8) & 0xffu) ^ sent, info = side >> 16;
        const uint32_t err = (mydec ^ sent) & 0xffu;
        unsigned a = (unsigned)__popc(err & info) | (unsigned)__popc(err & ~info) << 16;
        unsigned b = (unsigned)__popc(chan & info) | (unsigned)__popc(err & info & ~chan) << 16;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
        }
        // the lane index afresh: addresses and lane tests formed from `t` would be hoisted out of the frame loop and
        // live in registers through every iteration of every frame
        int te = t;
        asm volatile("" : "+v"(te));
        if ((te & 63) == 0) {  // one slot per wavefront, plain stores
            fsum[2 * (te >> 6)] = a;
            fsum[2 * (te >> 6) + 1] = b;
        }
        __syncthreads();
        // one counter per thread (one thread updating them all in turn holds a dozen 64-bit sums in registers
        // at once, on top of everything that lives across frames)
        if (te < 13) {
            unsigned sa = 0u, sb = 0u;
            for (int w = 0; w < (nt >> 6); ++w) {
                sa += fsum[2 * w];
                sb += fsum[2 * w + 1];
            }
            const unsigned e_info = sa & 0xffffu, e_chk = sa >> 16, ch = sb & 0xffffu, mis = sb >> 16;
            const bool correct = (e_info + e_chk) == 0u;
            const unsigned rxc = p.count_rx ? 1u : 0u;
            unsigned long long inc = 1ull;                                         // 0: frames
            inc = te == 1 ? (correct ? 0u : 1u) : inc;                              // 1: frame errors
            inc = te == 2 ? e_info : inc;                                           // 2: information bit errors
            inc = te == 3 ? e_chk : inc;                                            // 3: parity bit errors
            inc = te == 4 ? (e_info ? 1u : 0u) : inc;                               // 4, 5: frames with such errors
            inc = te == 5 ? (e_chk ? 1u : 0u) : inc;
            inc = te == 6 ? ((synd == 0 && !correct) ? 1u : 0u) : inc;              // 6: undetected
            inc = te == 7 ? mis * rxc : inc;                                        // 7: decoded wrong, received right
            inc = te == 8 ? (correct ? (unsigned)iters : 0u) : inc;                 // 8, 9: iterations of right / wrong frames
            inc = te == 9 ? (correct ? 0u : (unsigned)iters) : inc;
            inc = te == 10 ? ch * rxc : inc;                                        // 10, 11: channel errors, squared
            inc = te == 11 ? (unsigned long long)ch * ch * rxc : inc;
            int slot = te;
            if (te == 12) {                                                         // iteration histogram
                slot = 16 + iters - 1;
                if (iters < 1 || iters > p.max_iters) { slot = 12; inc = 0ull; }
            }
            wcnt[slot] += inc;
        }
    } else {
    // hard[frame][j] for j = 4 (tt + u nt) .. + 3, u = 0 .. VPT / 4 - 1: n = VPT nt, the table reads go out together
    {
        uint8_t* out = p.hard + frame * p.n;
        const int4* pos4 = reinterpret_cast<const int4*>(p.pos_of_var) + tt;
        int4 q[VPT / 4];
#pragma unroll
        for (int u = 0; u < VPT / 4; ++u) q[u] = pos4[u * nt];
        auto bit = [&](int q) -> unsigned char { return (unsigned char)((bits[q >> 5] >> (q & 31)) & 1u); };
#pragma unroll
        for (int u = 0; u < VPT / 4; ++u)
            if (4 * (tt + u * nt) < p.n)  // n is a multiple of 4; past it lie the dummy variables of a tiled code
                reinterpret_cast<uchar4*>(out)[tt + u * nt] = make_uchar4(bit(q[u].x), bit(q[u].y), bit(q[u].z), bit(q[u].w));
    }
    if (t == 0) {
        if (p.out_iters) p.out_iters[frame] = iters;
        if (p.out_synd) p.out_synd[frame] = synd;
    }
    }
    if constexpr (!EARLY) break;
    __syncthreads();  // the next frame reuses the LDS
    }
    if constexpr (COUNT) {
        __syncthreads();
        for (int i = t; i < nc; i += nt)
            if (wcnt[i]) atomicAdd(&p.counters[i], wcnt[i]);
    }
}

struct LdsLaunch {
    int rule;
    bool early;
    int frames_per_wg;  // frames one workgroup decodes in turn (>= 1; early termination only)
    int wpe;  // 4 or 6, see bp_lds_kernel
    int vpt;  // variables per thread: kLdsVpt or kLdsVptWide
    bool count;  // error accounting inside the kernel (default check-node rule, at most 8 variables per thread)
    bool def = false;  // the code is one edge short of regular in places (bp_lds_kernel, DEF); default rule only
    int threads;
    size_t lds_bytes;
    hipStream_t stream;
};

template <int DC, int DV, int VPT, bool ALLOW_DEF = false>
inline hipError_t bp_lds_launch_impl(const LdsLaunch& l, const LdsParams& p) {
    if (l.threads * VPT < p.n || l.threads > 4096 / VPT || l.threads % 64) return hipErrorInvalidConfiguration;
    if (l.def && (!ALLOW_DEF || l.rule != kCnComplement)) return hipErrorInvalidConfiguration;
    const long long fpw = (l.early && l.frames_per_wg > 0) ? l.frames_per_wg : 1;
    const dim3 grid((unsigned)((p.frames + fpw - 1) / fpw)), block((unsigned)l.threads);
#define LDPC_LDS2(R, E, W, ...)                                                                              \
    do {                                                                                                     \
        auto k = bp_lds_kernel<DC, DV, R, E, VPT, W, ##__VA_ARGS__>;                                         \
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                           (int)l.lds_bytes);                                                \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL(k, grid, block, l.lds_bytes, l.stream, p);                                        \
    } while (0)
#define LDPC_LDS(R, E)                                                                                       \
    do {                                                                                                     \
        if constexpr (VPT == kLdsVptWide) LDPC_LDS2(R, E, 3);                                                \
        else if constexpr (VPT == kLdsVptMid) LDPC_LDS2(R, E, 4);                                            \
        else if (l.wpe >= 6) LDPC_LDS2(R, E, 6); else LDPC_LDS2(R, E, 4);                                    \
    } while (0)
    if constexpr (ALLOW_DEF) {
        if (l.def) {
            if (l.count) {
                if (!p.counters) return hipErrorInvalidValue;
                if (l.early) { if (l.wpe >= 6) LDPC_LDS2(kCnComplement, true, 6, true, true); else LDPC_LDS2(kCnComplement, true, 4, true, true); }
                else { if (l.wpe >= 6) LDPC_LDS2(kCnComplement, false, 6, true, true); else LDPC_LDS2(kCnComplement, false, 4, true, true); }
            } else {
                if (l.early) { if (l.wpe >= 6) LDPC_LDS2(kCnComplement, true, 6, false, true); else LDPC_LDS2(kCnComplement, true, 4, false, true); }
                else { if (l.wpe >= 6) LDPC_LDS2(kCnComplement, false, 6, false, true); else LDPC_LDS2(kCnComplement, false, 4, false, true); }
            }
            return hipGetLastError();
        }
    }
    if (l.count) {
        if constexpr (VPT <= kLdsVpt) {
            if (l.rule != kCnComplement || !p.counters) return hipErrorInvalidValue;
            if (l.early) { if (l.wpe >= 6) LDPC_LDS2(kCnComplement, true, 6, true); else LDPC_LDS2(kCnComplement, true, 4, true); }
            else { if (l.wpe >= 6) LDPC_LDS2(kCnComplement, false, 6, true); else LDPC_LDS2(kCnComplement, false, 4, true); }
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
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
unsigned bp_lds_8_4_params_bytes();
hipError_t bp_lds_launch_8_4(const LdsLaunch& l, const LdsParams& p);
unsigned bp_lds_10_5_params_bytes();
hipError_t bp_lds_launch_10_5(const LdsLaunch& l, const LdsParams& p);

}  // namespace ldpc
