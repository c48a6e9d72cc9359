// Flooding belief-propagation kernels for gfx950 (wave64).
//
// Mapping: one wavefront owns one tile of T = 64*FPL frames for the whole decode; lane l
// carries frames l*FPL .. l*FPL+FPL-1 of the tile.  Every message array is frame-major,
// msg[edge][T], so a wave's access to one edge is one contiguous T*4-byte row whatever the
// structure of H, and the CSR/CSC indices are wave-uniform (scalar loads).  Frames never
// interact, so a wave needs no barrier and no other wave's data: the loop
// LogDecoder.java:64-120 runs start to end inside one launch.
//
// One iteration (reference order: VN :67-79, CN :82-95, decide :98-105, syndrome :108-116)
// is restated as  CN sweep -> VN sweep that also forms gamma/decisions -> syndrome.  The
// VN sweep of iteration l produces the beta of iteration l+1 from the same alpha reads that
// give gamma of iteration l; the first CN sweep reads beta = lambda (+ 0.0, the sum of the
// zero-initialised alpha of :57) straight from the channel LLRs.
//
// A single message array is updated in place: the CN sweep turns row beta[e] into
// alpha[e], the VN sweep turns it back.  Leave-one-out sums and products are prefix x
// suffix combinations; nothing is ever subtracted or divided out, so +-Inf and NaN behave
// as in the reference (Inf - Inf never appears where the reference has a finite value).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace ldpc {

constexpr int kWave = 64;
constexpr int kCnSumProduct = 0;
constexpr int kCnMinSum = 1;
constexpr int kCnComplement = 2;
// node degrees handled in registers with 1 frame per lane (larger rows are staged through LDS);
// a translation unit may raise them for its own kernels
#ifndef LDPC_REG_DEG_CN
#define LDPC_REG_DEG_CN 16
#endif
#ifndef LDPC_REG_DEG_VN
#define LDPC_REG_DEG_VN 16
#endif
constexpr int kRegDegCn = LDPC_REG_DEG_CN;
constexpr int kRegDegVn = LDPC_REG_DEG_VN;
constexpr int kRegDegFpl1 = 16;  // the one-launch kernels' limit for both
constexpr int kRegDegFpl4 = 8;   // ... and with 2 or 4 frames per lane

// Host code and kernels live in different objects.  Every translation unit that holds kernels also
// exports the size of the parameter struct it was compiled with; ldpc_amd_create compares those
// with its own, so a stale object cannot run with pointers read from the wrong offsets.
unsigned bp_generic_params_bytes();
unsigned bp_6_3_params_bytes();
unsigned bp_8_2_params_bytes();

struct BpParams {
    int m, n, nw;            // checks, variables, 32-bit decision words per frame
    const int* row_ptr;      // [m+1] CSR: edges of check i, ascending column
    const int* edge_col;     // [E]
    const int* col_ptr;      // [n+1] CSC
    const int* col_edge;     // [E]   CSR edge id of each entry of B(j), ascending row
    float* ws_msg;           // [tiles][E][T]
    const float* ws_lam;     // [tiles][n][T]
    uint32_t* ws_bits;       // [tiles][nw][T] packed decisions, bit (j & 31) of word j >> 5
    long long msg_tile, lam_tile, bits_tile;  // element offset from one tile to the next
    long long msg_row, lam_row, bits_row;     // element offset from one edge / variable / word row to the next
    int tiles;               // tiles in this launch
    long long frame_base;    // batch index of frame 0 of tile 0
    long long frames_total;  // frames in the batch
    int max_iters;
    int32_t* out_iters;      // [frames_total] or null
    int32_t* out_synd;       // [frames_total] or null
    int lds_maxd;            // rows of LDS staging per wave and array (0 = no LDS path)
    int bits_bytes;          // 1: a wave's share of a decision word is one or more whole bytes (8 variables), stored as bytes
    // Leaves: a variable of degree 1 sends its check lambda in every iteration and takes no part in any other
    // message, and what the check sends back only enters its own decision.  Where the decisions of an iteration
    // are not needed (fixed iterations, all but the last) a sweep can leave the leaves alone: the check-node sweep
    // does not store alpha on a leaf edge, so that the lambda the variable-node sweep of iteration 0 wrote there
    // stays, and the variable-node sweep skips the leaf - 4 bytes of traffic per leaf edge and iteration instead
    // of 20, the same bits at the end.  Only leaves whose edge is the last of a register-resident row are marked.
    const uint8_t* leaf_row;  // [m] 1: the last edge of the check leads to a marked leaf; null = no leaves marked
    const uint8_t* leaf_var;  // [n] 1: marked leaf
    int leaf_skip;            // 1: this sweep leaves the marked leaves alone
};

// ---- vector access helpers: FPL consecutive floats / words per lane -------------------
// LDPC_MSG_NT (set by a translation unit before this header): rows of four frames per lane (1 KiB per wave
// access, the fixed-iteration sweeps of codes with degrees <= 8) are read and written with non-temporal
// accesses.  A sweep touches every message of a tile once and the next touch is a whole sweep later, far
// beyond what the caches hold, so nothing is gained by keeping the lines: in-place read-modify-write streams
// measured 5.6 TB/s with non-temporal accesses against 5.2 TB/s without (tools/membw.hip), the sweeps +2 % to
// +3.5 %.  One-frame-per-lane rows stay on the default policy: with early termination the lanes of finished
// frames sit out, and partial rows written past the caches measured 6 - 11 % slower.
#ifndef LDPC_MSG_NT
#define LDPC_MSG_NT 0
#endif
// LDPC_LANE_IN_ACCESS (set by a translation unit before this header): the pointers handed to ldv / stv / ldw / stw are
// the same in every lane of the wavefront - the address of a row - and the lane's own FPL elements are addressed
// inside the access.  A row address built from wave-uniform values (tile, edge index from a scalar load) then
// lives in scalar registers and the access is "scalar base + 32-bit lane offset": no pair of vector registers
// per row in flight (a variable of degree 30 held 60 of them), no 64-bit vector additions.  Without it the
// caller's pointer already includes the lane.
#ifndef LDPC_LANE_IN_ACCESS
#define LDPC_LANE_IN_ACCESS 0
#endif
template <int FPL>
__device__ __forceinline__ unsigned lane_bytes() {  // byte offset of this lane's first element in a row of 4-byte elements
    return LDPC_LANE_IN_ACCESS ? (unsigned)__lane_id() * (unsigned)(4 * FPL) : 0u;
}
template <class T>
__device__ __forceinline__ T* at_lane(T* p, unsigned bytes) {
    char* row = reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p));
    if (LDPC_LANE_IN_ACCESS) {
        // The row address as a scalar, stated explicitly: left to itself the compiler folds the lane offset into the
        // tile base once (it does not change from row to row) and then forms every row address with 64-bit vector
        // additions, one pair of vector registers per row in flight.
        const unsigned long long a = reinterpret_cast<unsigned long long>(row);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        // (rebuilt as a pointer into global memory, so that the access stays a global one)
        typedef __attribute__((address_space(1))) char global_char;
        global_char* g = reinterpret_cast<global_char*>(((unsigned long long)hi << 32) | lo);
        return reinterpret_cast<T*>((char*)(g + bytes));
    }
    return reinterpret_cast<T*>(row + bytes);
}
// An edge index the compiler must take from a scalar register as it is, without connecting it to where it came
// from (LDPC_LANE_IN_ACCESS only: there the index is the same in every lane).  A row address is formed from it
// on the scalar unit right where it is used, for the load and again for the store; connected, the compiler
// keeps the 64-bit address of every row of a node in registers from its load to its store (60 registers for a
// variable of degree 30) instead of the 32-bit index.
__device__ __forceinline__ int fresh_index(int e) {
#if LDPC_LANE_IN_ACCESS
    e = __builtin_amdgcn_readfirstlane(e);
    asm volatile("" : "+s"(e));
#endif
    return e;
}
// An entry of one of the graph's index arrays (row and column pointers, edge lists, leaf marks).  With
// LDPC_LANE_IN_ACCESS the index is the same in every lane and the arrays are never written while a kernel runs:
// read through the constant address space the load is a scalar one (scalar cache, no vector memory
// instruction, no round trip through a vector register before the row address can be formed).
template <class T>
__device__ __forceinline__ T idx(const T* p, long long i) {
#if LDPC_LANE_IN_ACCESS
    typedef const __attribute__((address_space(4))) T constant_T;
    return ((constant_T*)p)[i];
#else
    return p[i];
#endif
}
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <int FPL>
__device__ __forceinline__ void ldv(const float* p, float (&v)[FPL]) {
    p = at_lane(p, lane_bytes<FPL>());
    if constexpr (FPL == 1) {
        v[0] = p[0];
    } else if constexpr (FPL == 2) {
        const f2v t = *reinterpret_cast<const f2v*>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
        const f4v t = LDPC_MSG_NT ? __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)) : *reinterpret_cast<const f4v*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}
template <int FPL>
__device__ __forceinline__ void stv(float* p, const float (&v)[FPL]) {
    p = at_lane(p, lane_bytes<FPL>());
    if constexpr (FPL == 1) {
        p[0] = v[0];
    } else if constexpr (FPL == 2) {
        *reinterpret_cast<f2v*>(p) = f2v{v[0], v[1]};
    } else {
        const f4v t = {v[0], v[1], v[2], v[3]};
        if (LDPC_MSG_NT) __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p)); else *reinterpret_cast<f4v*>(p) = t;
    }
}
template <int FPL>
__device__ __forceinline__ void ldw(const uint32_t* p, uint32_t (&v)[FPL]) {
    p = at_lane(p, lane_bytes<FPL>());
    if constexpr (FPL == 1) {
        v[0] = p[0];
    } else if constexpr (FPL == 2) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}
template <int FPL>
__device__ __forceinline__ void stw(uint32_t* p, const uint32_t (&v)[FPL]) {
    p = at_lane(p, lane_bytes<FPL>());
    if constexpr (FPL == 1) {
        p[0] = v[0];
    } else if constexpr (FPL == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(v[0], v[1]);
    } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// ---- check-node arithmetic -----------------------------------------------------------
// tanh(b/2), the factor of LogDecoder.java:90.  1 - 2/(e^b + 1) on the hardware exp2/rcp;
// +-Inf give +-1 exactly and NaN stays NaN, as Math.tanh does.  Below |b| = 0.25 the
// quotient cancels, so the odd Taylor polynomial is used there (error < 2e-9 relative).
__device__ __forceinline__ float tanh_half(float b) {
#pragma clang fp contract(off)  // fused operations only where written: every engine must round alike
    const float e = __builtin_amdgcn_exp2f(b * 1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    const float big = fmaf(-2.0f, r, 1.0f);
    const float h = 0.5f * b;
    const float h2 = h * h;
    const float poly = h * fmaf(h2, fmaf(h2, fmaf(h2, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
    return (fabsf(b) < 0.25f) ? poly : big;
}

// 2*atanh(p) = ln((1+p)/(1-p)), LogDecoder.java:9-11 and :93.  p = +-1 gives +-Inf
// (2/0 and ln 0), NaN stays NaN.  Below |p| = 0.15 the logarithm cancels, so the series
// 2(p + p^3/3 + ... + p^9/9) is used there.
__device__ __forceinline__ float atanh2(float p) {
#pragma clang fp contract(off)  // 1 + p must not fuse with the product the caller formed p from
    const float q = (1.0f + p) * __builtin_amdgcn_rcpf(1.0f - p);
    const float big = __builtin_amdgcn_logf(q) * 0.6931471805599453f;
    const float p2 = p * p;
    const float ser =
        2.0f * p * fmaf(p2, fmaf(p2, fmaf(p2, fmaf(p2, 1.0f / 9.0f, 1.0f / 7.0f), 1.0f / 5.0f), 1.0f / 3.0f), 1.0f);
    return (fabsf(p) < 0.15f) ? ser : big;
}

// Math.signum: +-0 and NaN map to themselves (MinSumDecoder.java:84)
__device__ __forceinline__ float signum(float x) { return (x > 0.0f) ? 1.0f : (x < 0.0f) ? -1.0f : x; }

// ---- the sum-product rule on complements ------------------------------------------------------
// 2 atanh(prod tanh(|beta_k|/2)) with every factor t_k represented by its complement
// q_k = 1 - t_k = 2w/(1 + w), w = e^-|beta_k|, which is formed directly (no cancellation), and every
// partial product P represented by ITS complement Q = 1 - P through
//     Q(P t) = Q + q (1 - Q)          and          Q(P1 P2) = Q1 + (1 - Q1) Q2,
// sums of non-negative terms only.  Then alpha = ln((1 + P)/(1 - P)) = ln(2/Q - 1) keeps full
// relative accuracy in fp32 where alpha is large (Q small: 1 - P is never formed from a value rounded
// next to 1) and an absolute error of an ulp of 1 where alpha is small, over the whole range in which
// the double-precision reference is finite.  Q = 0 (every other member saturated) gives +Inf; a
// member with beta = 0 has q = 1, which makes Q exactly 1 and alpha exactly 0; NaN stays NaN.
// Per edge this costs exp2, two rcp and log2.
//
// Everything below is written once for V = float and V = v2f (two independent checks per lane): on
// gfx950 a plain fp32 VALU instruction and a packed one (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// cost the same issue cycles, so the packed form halves the cost of the bookkeeping arithmetic.
// Both instantiate the same operations in the same order (fused multiply-adds only where written),
// so they return identical bits per element.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ v2f vexp2(v2f x) { return v2f{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ float vlog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ v2f vlog2(v2f x) { return v2f{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
__device__ __forceinline__ float vrcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ v2f vrcp(v2f x) { return v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// (x < t) ? a : b, per element
__device__ __forceinline__ float vsel_lt(float x, float t, float a, float b) { return (x < t) ? a : b; }
__device__ __forceinline__ v2f vsel_lt(v2f x, float t, v2f a, v2f b) {
    return v2f{(x.x < t) ? a.x : b.x, (x.y < t) ? a.y : b.y};
}
// (x > t) ? a : b, per element
__device__ __forceinline__ float vsel_gt(float x, float t, float a, float b) { return (x > t) ? a : b; }
__device__ __forceinline__ v2f vsel_gt(v2f x, float t, v2f a, v2f b) {
    return v2f{(x.x > t) ? a.x : b.x, (x.y > t) ? a.y : b.y};
}
// +-1.0 carrying the sign bit of x (so that x * sign1(x) = |x| exactly, also for -0 and NaN)
__device__ __forceinline__ float vsign1(float x) {
    return __uint_as_float((__float_as_uint(x) & 0x80000000u) | 0x3f800000u);
}
__device__ __forceinline__ v2f vsign1(v2f x) { return v2f{vsign1(x.x), vsign1(x.y)}; }
// The sign bits of a value over the lanes of the wavefront, one 64-bit lane mask per element.  Lane masks live
// in scalar registers: products of signs are s_xor_b64 on the scalar unit, which issues beside the vector
// instructions, and a sign is applied with one v_cndmask_b32 (negated source), where +-1.0 factors cost a
// v_bfi_b32 and three multiplications per message.  Inactive lanes read as 0 and are not written.
struct SignMask1 { unsigned long long m; };
struct SignMask2 { unsigned long long x, y; };
__device__ __forceinline__ unsigned long long sign_lanes(float x) {
    return __builtin_amdgcn_ballot_w64((int32_t)__float_as_uint(x) < 0);  // the sign bit itself: -0 and NaN included
}
__device__ __forceinline__ SignMask1 vsignmask(float x) { return SignMask1{sign_lanes(x)}; }
__device__ __forceinline__ SignMask2 vsignmask(v2f x) { return SignMask2{sign_lanes(x.x), sign_lanes(x.y)}; }
__device__ __forceinline__ SignMask1 operator^(SignMask1 a, SignMask1 b) { return SignMask1{a.m ^ b.m}; }
__device__ __forceinline__ SignMask2 operator^(SignMask2 a, SignMask2 b) { return SignMask2{a.x ^ b.x, a.y ^ b.y}; }
// x with its sign flipped in the lanes of the mask: x * (-1.0) there, x * 1.0 elsewhere
__device__ __forceinline__ float vflip(float x, SignMask1 s) { return __builtin_amdgcn_inverse_ballot_w64(s.m) ? -x : x; }
__device__ __forceinline__ v2f vflip(v2f x, SignMask2 s) {
    return v2f{__builtin_amdgcn_inverse_ballot_w64(s.x) ? -x.x : x.x, __builtin_amdgcn_inverse_ballot_w64(s.y) ? -x.y : x.y};
}
template <class V> struct SignMaskOf;
template <> struct SignMaskOf<float> { typedef SignMask1 type; static __device__ __forceinline__ SignMask1 none() { return SignMask1{0ull}; } };
template <> struct SignMaskOf<v2f> { typedef SignMask2 type; static __device__ __forceinline__ SignMask2 none() { return SignMask2{0ull, 0ull}; } };
__device__ __forceinline__ float vabs(float x) { return fabsf(x); }
__device__ __forceinline__ v2f vabs(v2f x) { return v2f{fabsf(x.x), fabsf(x.y)}; }
template <class V> __device__ __forceinline__ V vsplat(float x);
template <> __device__ __forceinline__ float vsplat<float>(float x) { return x; }
template <> __device__ __forceinline__ v2f vsplat<v2f>(float x) { return v2f{x, x}; }

// Units: with this rule every LLR and message is held in BITS (natural-log value / ln 2; the layout
// kernels scale the channel LLRs once on the way in, decisions only need signs).  Then 2^|x| is
// the bare hardware exp2 and alpha comes straight out of the hardware log2, with no constants.
//
// |beta| above 55 ln 2 nats = 55 bits has tanh(beta/2) == 1.0 in double (1 - tanh < 2^-54): the
// reference's factor is exactly 1 there, so q is exactly 0.
constexpr float kDoubleTanhSaturation = 55.0f;   // bits
constexpr float kBitsPerNat = 1.4426950408889634f;  // 1 / ln 2: scale of the channel LLRs for this rule

// q = 1 - tanh(|beta| ln2 / 2) = 2 / (2^|beta| + 1), beta in bits; beta = 0 gives exactly 1.
// Saturation costs one multiplication: 2^|beta| is carried scaled by 2^73, which overflows to +Inf
// from |beta| = 55 on, and 1/Inf is the exact 0 the reference's factor of exactly 1 stands for; below
// the threshold the scaling is undone exactly inside the fused multiply-add.  |beta| is a source
// modifier of the exp2.
template <class V>
__device__ __forceinline__ V tanh_complement(V beta) {
#pragma clang fp contract(off)
    static_assert(kDoubleTanhSaturation == 55.0f, "2^73 * 2^55 = 2^128 is the first fp32 overflow");
    const V e = vexp2(vabs(beta)) * vsplat<V>(9.4447329657392904e21f);             // 2^73
    return vrcp(vfma(e, vsplat<V>(5.2939559203393771e-23f), vsplat<V>(0.5f)));     // 2^-74
}

// |alpha| in bits from the complement Q of a leave-one-out product: log2((2 - Q)/Q) = log2(2/Q - 1).
// Q = 1 gives exactly 0, Q = 0 gives +Inf.
template <class V>
__device__ __forceinline__ V atanh2_complement(V Q) {
#pragma clang fp contract(off)
    return vlog2(vfma(vrcp(Q), vsplat<V>(2.0f), vsplat<V>(-1.0f)));
}

// beta[0..D) -> alpha[0..D) in place.  Products ascend over k like those of LogDecoder.java:86-91;
// leave-one-out is (prefix of k) * (suffix of k).  The sign of alpha_k is the product of the other
// members' signs, formed as (product of all) * (own) on lane masks of the sign bits.
template <class V, int D>
__device__ __forceinline__ void cn_complement(V (&b)[D]) {
#pragma clang fp contract(off)
    const V one = vsplat<V>(1.0f);
    typedef typename SignMaskOf<V>::type S;
    V q[D], pq[D];
    S sg[D], all = SignMaskOf<V>::none();
    V Q = vsplat<V>(0.0f);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        sg[k] = vsignmask(b[k]);
        q[k] = tanh_complement<V>(b[k]);
        pq[k] = Q;
        // the first step is fma(q, 1 - 0, 0) = q exactly (q is never -0; NaN stays NaN): no instruction
        Q = (k == 0) ? q[k] : vfma(q[k], one - Q, Q);
        all = all ^ sg[k];
    }
    V sQ = vsplat<V>(0.0f);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        // for the last member the suffix is empty: fma(0, 1 - pq, pq) = pq exactly (pq lies in [0, 1] or is NaN); for
        // the first the prefix is: fma(sQ, 1 - 0, 0) = sQ exactly (sQ is never -0)
        const V mag = atanh2_complement<V>((k == D - 1) ? pq[k] : (k == 0) ? sQ : vfma(sQ, one - pq[k], pq[k]));
        b[k] = vflip(mag, all ^ sg[k]);
        sQ = (k == D - 1) ? q[k] : vfma(q[k], one - sQ, sQ);
    }
}

// beta[0..D) -> alpha[0..D) in place, D a compile-time degree.  The FPL frames of a lane are
// independent; they are processed one after the other so that only one frame's prefix array is live.
template <int D, int RULE>
__device__ __forceinline__ void cn_compute1(float (&b)[D]) {
    if constexpr (RULE == kCnSumProduct) {
        float pre[D];
        pre[0] = 1.0f;  // "double product = 1.0", LogDecoder.java:86
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float t = tanh_half(b[k]);
            b[k] = t;
            if (k + 1 < D) pre[k + 1] = pre[k] * t;
        }
        float suf = 1.0f;
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const float p = pre[k] * suf;
            suf *= b[k];
            b[k] = atanh2(p);
        }
    } else if constexpr (RULE == kCnComplement) {
        cn_complement<float, D>(b);
    } else {
        // MinSumDecoder.java:79-93: alpha = prod signum(beta) * min|beta| * 0.7 over k != j;
        // "minbata > |beta|" never selects a NaN, and Double.MAX_VALUE becomes FLT_MAX.
        float ps[D], pm[D];
        ps[0] = 1.0f;
        pm[0] = 3.402823466e+38f;
#pragma unroll
        for (int k = 0; k + 1 < D; ++k) {
            const float a = fabsf(b[k]);
            ps[k + 1] = ps[k] * signum(b[k]);
            pm[k + 1] = (pm[k] > a) ? a : pm[k];
        }
        float ss = 1.0f, sm = 3.402823466e+38f;
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const float a = fabsf(b[k]);
            const float sg = signum(b[k]);
            const float mn = (pm[k] > sm) ? sm : pm[k];
            b[k] = ps[k] * ss * mn * 0.7f;
            ss *= sg;
            sm = (sm > a) ? a : sm;
        }
    }
}

template <int FPL, int D, int RULE>
__device__ __forceinline__ void cn_compute(float (&b)[D][FPL]) {
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
        float x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = b[k][f];
        cn_compute1<D, RULE>(x);
#pragma unroll
        for (int k = 0; k < D; ++k) b[k][f] = x[k];
    }
}

// One check of compile-time degree D.  `row` points at this lane's slot of the check's first
// edge row; FIRST reads beta = lambda + 0.0 through the column indices instead.
template <int FPL, int D, int RULE, bool FIRST>
__device__ __forceinline__ void cn_row(float* row, const float* lam, const int* cols, long long T, long long LT,
                                       bool skip_last = false) {
    float b[D][FPL];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if constexpr (FIRST) {
            ldv<FPL>(lam + (long long)idx(cols, k) * LT, b[k]);
#pragma unroll
            for (int f = 0; f < FPL; ++f) b[k][f] += 0.0f;
        } else {
            ldv<FPL>(row + (long long)k * T, b[k]);
        }
    }
    cn_compute<FPL, D, RULE>(b);
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (k + 1 < D || !skip_last) stv<FPL>(row + (long long)k * T, b[k]);  // skip_last: see BpParams::leaf_row
}

// one element of this lane (one frame per lane): the staged paths below
__device__ __forceinline__ float ld1(const float* p) { float v[1]; ldv<1>(p, v); return v[0]; }
__device__ __forceinline__ void st1(float* p, float x) { const float v[1] = {x}; stv<1>(p, v); }

// Any degree up to the LDS capacity, one frame per lane: tanh values and exclusive prefix
// products of the row are staged in LDS as [k][lane] (conflict-free), then combined with
// the running suffix product on the way back.
template <int RULE, bool FIRST>
__device__ __forceinline__ void cn_row_lds(float* row, const float* lam, const int* cols, long long T, long long LT,
                                           int d, float* lds, int cap, int lane) {
    float* st = lds + lane;                 // [k][64]: tanh (or beta for min-sum)
    float* sp = lds + (long long)cap * 64 + lane;  // [k][64]: exclusive prefix product / prefix min
    if constexpr (RULE == kCnSumProduct) {
        float pre = 1.0f;
        for (int k = 0; k < d; ++k) {
            float b = FIRST ? (ld1(lam + (long long)idx(cols, k) * LT) + 0.0f) : ld1(row + (long long)k * T);
            const float t = tanh_half(b);
            st[k * 64] = t;
            sp[k * 64] = pre;
            pre *= t;
        }
        float suf = 1.0f;
        for (int k = d - 1; k >= 0; --k) {
            const float p = sp[k * 64] * suf;
            suf *= st[k * 64];
            st1(row + (long long)k * T, atanh2(p));
        }
    } else if constexpr (RULE == kCnComplement) {
#pragma clang fp contract(off)
        // staged: q_k with sign(beta_k) in its sign bit; the exclusive prefix complement with the prefix sign
        float Q = 0.0f;
        uint32_t psg = 0u;
        for (int k = 0; k < d; ++k) {
            const float b = FIRST ? (ld1(lam + (long long)idx(cols, k) * LT) + 0.0f) : ld1(row + (long long)k * T);
            const uint32_t sg = __float_as_uint(b) & 0x80000000u;
            const float qk = tanh_complement<float>(b);
            st[k * 64] = __uint_as_float(__float_as_uint(qk) | sg);
            sp[k * 64] = __uint_as_float(__float_as_uint(Q) | psg);
            Q = fmaf(qk, 1.0f - Q, Q);
            psg ^= sg;
        }
        float sQ = 0.0f;
        uint32_t ssg = 0u;
        for (int k = d - 1; k >= 0; --k) {
            const float sv = st[k * 64], pv = sp[k * 64];
            const float qk = fabsf(sv), pq = fabsf(pv);
            const float mag = atanh2_complement<float>(fmaf(sQ, 1.0f - pq, pq));
            st1(row + (long long)k * T, __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(pv) & 0x80000000u) ^ ssg));
            sQ = fmaf(qk, 1.0f - sQ, sQ);
            ssg ^= __float_as_uint(sv) & 0x80000000u;
        }
    } else {
        // beta and the exclusive prefix minimum are staged; the exclusive prefix sign product
        // of row k is rebuilt from the staged beta on the way back (this path is rare).
        float pm = 3.402823466e+38f;
        for (int k = 0; k < d; ++k) {
            float b = FIRST ? (ld1(lam + (long long)idx(cols, k) * LT) + 0.0f) : ld1(row + (long long)k * T);
            st[k * 64] = b;
            sp[k * 64] = pm;
            const float a = fabsf(b);
            pm = (pm > a) ? a : pm;
        }
        float ss = 1.0f, sm = 3.402823466e+38f;
        for (int k = d - 1; k >= 0; --k) {
            const float b = st[k * 64];
            float psign = 1.0f;
            for (int q = 0; q < k; ++q) psign *= signum(st[q * 64]);
            const float pmin = sp[k * 64];
            const float mn = (pmin > sm) ? sm : pmin;
            st1(row + (long long)k * T, psign * ss * mn * 0.7f);
            ss *= signum(b);
            const float a = fabsf(b);
            sm = (sm > a) ? a : sm;
        }
    }
}

// ---- variable-node arithmetic --------------------------------------------------------
// alpha[0..D) -> beta[0..D) in place; g = lambda + sum alpha, accumulated in ascending
// order from lambda exactly as gamma is (LogDecoder.java:100-103).
template <int FPL, int D>
__device__ __forceinline__ void vn_compute(float (&a)[D][FPL], const float (&lam)[FPL], float (&g)[FPL]) {
    float pre[D + 1][FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) pre[0][f] = lam[f];  // "double product = lambda[j]", :70
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
        for (int f = 0; f < FPL; ++f) pre[k + 1][f] = pre[k][f] + a[k][f];
#pragma unroll
    for (int f = 0; f < FPL; ++f) g[f] = pre[D][f];
    float suf[FPL];
#pragma unroll
    for (int k = D - 1; k >= 0; --k)
#pragma unroll
        for (int f = 0; f < FPL; ++f) {
            const float ak = a[k][f];
            if (k == D - 1) {
                a[k][f] = pre[k][f];
                suf[f] = ak;
            } else {
                a[k][f] = pre[k][f] + suf[f];
                suf[f] = ak + suf[f];
            }
        }
}

// the same sums on one value type (float, or v2f = two variables per lane in packed registers)
template <class V, int D>
__device__ __forceinline__ void vn_compute_v(V (&a)[D], V lam, V& g) {
    V pre[D + 1];
    pre[0] = lam;
#pragma unroll
    for (int k = 0; k < D; ++k) pre[k + 1] = pre[k] + a[k];
    g = pre[D];
    V suf = a[D - 1];
    a[D - 1] = pre[D - 1];
#pragma unroll
    for (int k = D - 2; k >= 0; --k) {
        const V ak = a[k];
        a[k] = pre[k] + suf;
        suf = ak + suf;
    }
}

template <int FPL, int D>
__device__ __forceinline__ void vn_col(float* msg, const float* lamrow, const int* edges, long long T, bool write_beta,
                                       float (&g)[FPL]) {
    float lam[FPL];
    ldv<FPL>(lamrow, lam);
    if constexpr (D == 0) {
#pragma unroll
        for (int f = 0; f < FPL; ++f) g[f] = lam[f];
    } else {
        float a[D][FPL];
        int ek[D];
#pragma unroll
        for (int k = 0; k < D; ++k) ek[k] = idx(edges, k);
#pragma unroll
        for (int k = 0; k < D; ++k) ldv<FPL>(msg + (long long)ek[k] * T, a[k]);
        vn_compute<FPL, D>(a, lam, g);
        if (write_beta) {
#pragma unroll
            for (int k = 0; k < D; ++k) stv<FPL>(msg + (long long)fresh_index(ek[k]) * T, a[k]);
        }
    }
}

__device__ __forceinline__ float vn_col_lds(float* msg, const float* lamrow, const int* edges, long long T, int d,
                                            bool write_beta, float* lds, int cap, int lane) {
    float* sa = lds + lane;                        // alpha
    float* sp = lds + (long long)cap * 64 + lane;  // prefix sums
    float pre = ld1(lamrow);
    for (int k = 0; k < d; ++k) {
        const float a = ld1(msg + (long long)idx(edges, k) * T);
        sa[k * 64] = a;
        sp[k * 64] = pre;
        pre += a;
    }
    if (write_beta) {
        float suf = 0.0f;
        for (int k = d - 1; k >= 0; --k) {
            const float a = sa[k * 64];
            st1(msg + (long long)idx(edges, k) * T, (k == d - 1) ? sp[k * 64] : (sp[k * 64] + suf));
            suf = (k == d - 1) ? a : (a + suf);
        }
    }
    return pre;
}

// ---- degree dispatch -------------------------------------------------------------------
#define LDPC_CASE_CN(D)                                                              \
    case D:                                                                          \
        if constexpr (D <= (FPL == 1 ? kRegDegCn : kRegDegFpl4))                     \
            cn_row<FPL, D, RULE, FIRST>(row, lam, cols, T, LT, skip_last);               \
        else if constexpr (FPL == 1)                                                 \
            cn_row_lds<RULE, FIRST>(row, lam, cols, T, LT, d, lds, cap, lane);           \
        break;
#define LDPC_CASE_VN(D)                                                              \
    case D:                                                                          \
        if constexpr (D <= (FPL == 1 ? kRegDegVn : kRegDegFpl4))                     \
            vn_col<FPL, D>(msg, lamrow, edges, T, write_beta, g);                    \
        else if constexpr (FPL == 1)                                                 \
            g[0] = vn_col_lds(msg, lamrow, edges, T, d, write_beta, lds, cap, lane); \
        break;

template <int FPL, int RULE, bool FIRST>
__device__ __forceinline__ void cn_dispatch(int d, float* row, const float* lam, const int* cols, long long T,
                                            long long LT, float* lds, int cap, int lane, bool skip_last = false) {
    switch (d) {
        case 0: break;
        LDPC_CASE_CN(1) LDPC_CASE_CN(2) LDPC_CASE_CN(3) LDPC_CASE_CN(4)
        LDPC_CASE_CN(5) LDPC_CASE_CN(6) LDPC_CASE_CN(7) LDPC_CASE_CN(8)
        default:
            if constexpr (FPL == 1) {
                switch (d) {
                    LDPC_CASE_CN(9) LDPC_CASE_CN(10) LDPC_CASE_CN(11) LDPC_CASE_CN(12)
                    LDPC_CASE_CN(13) LDPC_CASE_CN(14) LDPC_CASE_CN(15) LDPC_CASE_CN(16)
                    LDPC_CASE_CN(17) LDPC_CASE_CN(18) LDPC_CASE_CN(19) LDPC_CASE_CN(20)
                    LDPC_CASE_CN(21) LDPC_CASE_CN(22) LDPC_CASE_CN(23) LDPC_CASE_CN(24)
                    LDPC_CASE_CN(25) LDPC_CASE_CN(26) LDPC_CASE_CN(27) LDPC_CASE_CN(28)
                    LDPC_CASE_CN(29) LDPC_CASE_CN(30) LDPC_CASE_CN(31) LDPC_CASE_CN(32)
                    default: cn_row_lds<RULE, FIRST>(row, lam, cols, T, LT, d, lds, cap, lane); break;
                }
            }
            break;
    }
}

template <int FPL>
__device__ __forceinline__ void vn_dispatch(int d, float* msg, const float* lamrow, const int* edges, long long T,
                                            bool write_beta, float (&g)[FPL], float* lds, int cap, int lane) {
    switch (d) {
        LDPC_CASE_VN(0) LDPC_CASE_VN(1) LDPC_CASE_VN(2) LDPC_CASE_VN(3) LDPC_CASE_VN(4)
        LDPC_CASE_VN(5) LDPC_CASE_VN(6) LDPC_CASE_VN(7) LDPC_CASE_VN(8)
        default:
            if constexpr (FPL == 1) {
                switch (d) {
                    LDPC_CASE_VN(9) LDPC_CASE_VN(10) LDPC_CASE_VN(11) LDPC_CASE_VN(12)
                    LDPC_CASE_VN(13) LDPC_CASE_VN(14) LDPC_CASE_VN(15) LDPC_CASE_VN(16)
                    LDPC_CASE_VN(17) LDPC_CASE_VN(18) LDPC_CASE_VN(19) LDPC_CASE_VN(20)
                    LDPC_CASE_VN(21) LDPC_CASE_VN(22) LDPC_CASE_VN(23) LDPC_CASE_VN(24)
                    LDPC_CASE_VN(25) LDPC_CASE_VN(26) LDPC_CASE_VN(27) LDPC_CASE_VN(28)
                    LDPC_CASE_VN(29) LDPC_CASE_VN(30) LDPC_CASE_VN(31) LDPC_CASE_VN(32)
                    default: g[0] = vn_col_lds(msg, lamrow, edges, T, d, write_beta, lds, cap, lane); break;
                }
            }
            break;
    }
}
#undef LDPC_CASE_CN
#undef LDPC_CASE_VN

// ---- sweeps ------------------------------------------------------------------------------
// Regular codes (every check of degree D): rows of check i are edges i*D .. i*D+D-1.  The loads of
// check i+1 are issued before check i is computed and stored, so a wave always has one row set in
// flight while it computes, and the wait for those loads never has to wait for the stores issued
// after them (vector memory operations retire in issue order).
template <int FPL, int D, bool FIRST>
__device__ __forceinline__ void cn_load_at(const BpParams& p, const float* msg, const float* lam, long long e0,
                                           float (&b)[D][FPL]) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if constexpr (FIRST) ldv<FPL>(lam + (long long)idx(p.edge_col, e0 + k) * p.lam_row, b[k]);
        else                 ldv<FPL>(msg + (e0 + k) * p.msg_row, b[k]);
    }
}

template <int FPL, int D, int RULE, bool FIRST>
__device__ __forceinline__ void cn_finish_at(const BpParams& p, float* msg, long long e0, float (&b)[D][FPL],
                                             bool skip_last = false) {
    if constexpr (FIRST) {
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int f = 0; f < FPL; ++f) b[k][f] += 0.0f;  // lambda + the zero-initialised alpha of :57
    }
    cn_compute<FPL, D, RULE>(b);
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (k + 1 < D || !skip_last) stv<FPL>(msg + (e0 + k) * p.msg_row, b[k]);
}

template <int FPL, int D, bool FIRST>
__device__ __forceinline__ void cn_load(const BpParams& p, const float* msg, const float* lam, int i,
                                        float (&b)[D][FPL]) {
    cn_load_at<FPL, D, FIRST>(p, msg, lam, (long long)i * D, b);
}

template <int FPL, int D, int RULE, bool FIRST>
__device__ __forceinline__ void cn_finish(const BpParams& p, float* msg, int i, float (&b)[D][FPL]) {
    cn_finish_at<FPL, D, RULE, FIRST>(p, msg, (long long)i * D, b);
}

// Rows whose loads a wave issues together before it uses the first of them (at most 48 message
// registers in flight): what hides the memory latency when a wave's rows are short.
template <int FPL, int D> constexpr int rows_group() {
    return 48 / (D * FPL) >= 4 ? 4 : 48 / (D * FPL) >= 1 ? 48 / (D * FPL) : 1;
}
// ... for codes of any degrees, where the row addresses are not multiples of one constant and take
// registers of their own: 24 message registers per group of checks, 48 registers of messages, 64-bit
// edge offsets and channel LLRs per group of variables
template <int FPL, int D> constexpr int rows_group_any() {
    return 24 / (D * FPL) >= 4 ? 4 : 24 / (D * FPL) >= 1 ? 24 / (D * FPL) : 1;
}
template <int FPL, int D> constexpr int cols_group() {
    return 48 / (D * (FPL + 2) + FPL) >= 4 ? 4 : 48 / (D * (FPL + 2) + FPL) >= 1 ? 48 / (D * (FPL + 2) + FPL) : 1;
}

// Any degrees: checks i0 .. i1-1.  A run of consecutive checks of one degree (the Z rows of a block
// row of a quasi-cyclic H, or a regular stretch of any H) is taken as a group like the rows of a
// regular code; everything else goes row by row.  Same per-row arithmetic either way.
template <int FPL, int D, int RULE, bool FIRST>
__device__ __forceinline__ int cn_run(const BpParams& p, float* msg, const float* lam, int i, int i1, int e0) {
    constexpr int G = rows_group_any<FPL, D>();
    if constexpr (G > 1) {
        bool run = i + G <= i1;
        if (run) {
#pragma unroll
            for (int g = 2; g <= G; ++g) run = run && (idx(p.row_ptr, i + g) - e0 == g * D);
        }
        if (run) {
            float buf[G][D][FPL];
#pragma unroll
            for (int g = 0; g < G; ++g) cn_load_at<FPL, D, FIRST>(p, msg, lam, (long long)e0 + g * D, buf[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cn_finish_at<FPL, D, RULE, FIRST>(p, msg, (long long)e0 + g * D, buf[g], p.leaf_skip && idx(p.leaf_row, i + g));
                __builtin_amdgcn_sched_barrier(0);  // one row's temporaries at a time
            }
            return G;
        }
    }
    cn_row<FPL, D, RULE, FIRST>(msg + (long long)e0 * p.msg_row, lam, p.edge_col + e0, p.msg_row, p.lam_row,
                                p.leaf_skip && idx(p.leaf_row, i));
    return 1;
}

#define LDPC_RUN_CN(D)                                                         \
    case D:                                                                    \
        if constexpr (D <= (FPL == 1 ? kRegDegCn : kRegDegFpl4)) {             \
            i += cn_run<FPL, D, RULE, FIRST>(p, msg, lam, i, i1, e0);          \
            continue;                                                          \
        }                                                                      \
        break;

template <int FPL, int RULE, bool FIRST>
__device__ __forceinline__ void cn_rows_any(const BpParams& p, float* msg, const float* lam, int i0, int i1, float* lds,
                                            int lane) {
    int i = i0;
    while (i < i1) {
        const int e0 = idx(p.row_ptr, i);
        const int d = idx(p.row_ptr, i + 1) - e0;
        switch (d) {
            LDPC_RUN_CN(1) LDPC_RUN_CN(2) LDPC_RUN_CN(3) LDPC_RUN_CN(4) LDPC_RUN_CN(5) LDPC_RUN_CN(6)
            LDPC_RUN_CN(7) LDPC_RUN_CN(8) LDPC_RUN_CN(9) LDPC_RUN_CN(10) LDPC_RUN_CN(11) LDPC_RUN_CN(12)
            LDPC_RUN_CN(13) LDPC_RUN_CN(14) LDPC_RUN_CN(15) LDPC_RUN_CN(16) LDPC_RUN_CN(17) LDPC_RUN_CN(18)
            LDPC_RUN_CN(19) LDPC_RUN_CN(20) LDPC_RUN_CN(21) LDPC_RUN_CN(22) LDPC_RUN_CN(23) LDPC_RUN_CN(24)
            default: break;
        }
        cn_dispatch<FPL, RULE, FIRST>(d, msg + (long long)e0 * p.msg_row, lam, p.edge_col + e0, p.msg_row, p.lam_row, lds,
                                      p.lds_maxd, lane, p.leaf_skip && idx(p.leaf_row, i));
        ++i;
    }
}
#undef LDPC_RUN_CN

template <int FPL, int D, int RULE, bool FIRST>
__device__ __forceinline__ void cn_sweep_regular(const BpParams& p, float* msg, const float* lam) {
    float A[D][FPL], B[D][FPL];
    const int m = p.m;
    if (m > 0) cn_load<FPL, D, FIRST>(p, msg, lam, 0, A);
    int i = 0;
    for (; i + 1 < m; i += 2) {
        cn_load<FPL, D, FIRST>(p, msg, lam, i + 1, B);
        cn_finish<FPL, D, RULE, FIRST>(p, msg, i, A);
        if (i + 2 < m) cn_load<FPL, D, FIRST>(p, msg, lam, i + 2, A);
        cn_finish<FPL, D, RULE, FIRST>(p, msg, i + 1, B);
    }
    if (i < m) cn_finish<FPL, D, RULE, FIRST>(p, msg, i, A);
}

template <int FPL, int RULE, bool FIRST, int DC>
__device__ __forceinline__ void cn_sweep(const BpParams& p, float* msg, const float* lam, float* lds, int lane) {
    if constexpr (DC > 0) {
        cn_sweep_regular<FPL, DC, RULE, FIRST>(p, msg, lam);
        return;
    }
    cn_rows_any<FPL, RULE, FIRST>(p, msg, lam, 0, p.m, lds, lane);
}

// Decision bits collected in acc go to the packed words when variable j closes a word, or closes a byte when
// the variables of a word are shared out among several waves (small batches, bp_rows.hpp), or is the last one.
template <int FPL>
__device__ __forceinline__ void flush_decisions(const BpParams& p, uint32_t* bits, int j, uint32_t (&acc)[FPL]) {
    const int mask = p.bits_bytes ? 7 : 31;
    if ((j & mask) != mask && j != p.n - 1) return;
    uint32_t* word = bits + (long long)(j >> 5) * p.bits_row;
    if (p.bits_bytes) {
#pragma unroll
        for (int f = 0; f < FPL; ++f) reinterpret_cast<uint8_t*>(at_lane(word, lane_bytes<FPL>()) + f)[(j >> 3) & 3] = (uint8_t)(acc[f] >> (j & 24));
    } else {
        stw<FPL>(word, acc);
    }
#pragma unroll
    for (int f = 0; f < FPL; ++f) acc[f] = 0u;
}

// beta for the next iteration + tentative decisions of this one (LogDecoder.java:67-79, :98-105)
template <int FPL, int D>
struct VnBuf {
    float a[D][FPL];
    float lam[FPL];
    int e[D];  // CSR edge ids of the variable's messages
};

template <int FPL, int D>
__device__ __forceinline__ void vn_load_at(const BpParams& p, const float* msg, const float* lam, int j, long long c0,
                                           VnBuf<FPL, D>& b) {
#pragma unroll
    for (int k = 0; k < D; ++k) b.e[k] = idx(p.col_edge, c0 + k);
#pragma unroll
    for (int k = 0; k < D; ++k) ldv<FPL>(msg + (long long)b.e[k] * p.msg_row, b.a[k]);
    ldv<FPL>(lam + (long long)j * p.lam_row, b.lam);
}

template <int FPL, int D>
__device__ __forceinline__ void vn_load(const BpParams& p, const float* msg, const float* lam, int j, VnBuf<FPL, D>& b) {
    vn_load_at<FPL, D>(p, msg, lam, j, (long long)j * D, b);
}

template <int FPL, int D>
__device__ __forceinline__ void vn_finish(const BpParams& p, float* msg, uint32_t* bits, int j, bool write_beta,
                                          VnBuf<FPL, D>& b, uint32_t (&acc)[FPL]) {
    float g[FPL];
    vn_compute<FPL, D>(b.a, b.lam, g);
    if (write_beta) {
#pragma unroll
        for (int k = 0; k < D; ++k) stv<FPL>(msg + (long long)fresh_index(b.e[k]) * p.msg_row, b.a[k]);
    }
#pragma unroll
    for (int f = 0; f < FPL; ++f) acc[f] |= (g[f] < 0.0f ? 1u : 0u) << (j & 31);  // "(gamma[j] < 0) ? 1 : 0", :104
    flush_decisions<FPL>(p, bits, j, acc);
}

template <int FPL, int D>
__device__ __forceinline__ void vn_sweep_regular(const BpParams& p, float* msg, const float* lam, uint32_t* bits,
                                                 bool write_beta) {
    VnBuf<FPL, D> A, B;
    uint32_t acc[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) acc[f] = 0u;
    const int n = p.n;
    if (n > 0) vn_load<FPL, D>(p, msg, lam, 0, A);
    int j = 0;
    for (; j + 1 < n; j += 2) {
        vn_load<FPL, D>(p, msg, lam, j + 1, B);
        vn_finish<FPL, D>(p, msg, bits, j, write_beta, A, acc);
        if (j + 2 < n) vn_load<FPL, D>(p, msg, lam, j + 2, A);
        vn_finish<FPL, D>(p, msg, bits, j + 1, write_beta, B, acc);
    }
    if (j < n) vn_finish<FPL, D>(p, msg, bits, j, write_beta, A, acc);
}

// Any degrees: variables j0 .. j1-1 (j0 a multiple of 32, or of 8 with bits_bytes; acc the decision bits collected so far), with runs
// of one degree grouped like the checks in cn_rows_any.
template <int FPL, int D>
__device__ __forceinline__ int vn_run(const BpParams& p, float* msg, const float* lam, uint32_t* bits, bool write_beta,
                                      int j, int j1, int c0, uint32_t (&acc)[FPL]) {
    constexpr int G = cols_group<FPL, D>();
    bool run = G > 1 && j + G <= j1;
    if (run) {
#pragma unroll
        for (int g = 2; g <= G; ++g) run = run && (idx(p.col_ptr, j + g) - c0 == g * D);
    }
    if (run) {
        VnBuf<FPL, D> buf[G];
#pragma unroll
        for (int g = 0; g < G; ++g) vn_load_at<FPL, D>(p, msg, lam, j + g, (long long)c0 + g * D, buf[g]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            vn_finish<FPL, D>(p, msg, bits, j + g, write_beta, buf[g], acc);
            __builtin_amdgcn_sched_barrier(0);  // one variable's temporaries at a time
        }
        return G;
    }
    VnBuf<FPL, D> one;
    vn_load_at<FPL, D>(p, msg, lam, j, (long long)c0, one);
    vn_finish<FPL, D>(p, msg, bits, j, write_beta, one, acc);
    return 1;
}

#define LDPC_RUN_VN(D)                                                               \
    case D:                                                                          \
        if constexpr (D <= (FPL == 1 ? kRegDegVn : kRegDegFpl4)) {                   \
            j += vn_run<FPL, D>(p, msg, lam, bits, write_beta, j, j1, c0, acc);      \
            continue;                                                                \
        }                                                                            \
        break;

template <int FPL>
__device__ __forceinline__ void vn_cols_any(const BpParams& p, float* msg, const float* lam, uint32_t* bits,
                                            bool write_beta, int j0, int j1, float* lds, int lane) {
    uint32_t acc[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) acc[f] = 0u;
    int j = j0;
    while (j < j1) {
        const int c0 = idx(p.col_ptr, j);
        const int d = idx(p.col_ptr, j + 1) - c0;
        if (p.leaf_skip && idx(p.leaf_var, j)) {  // its slot keeps lambda and nobody reads its decision (BpParams::leaf_row)
            flush_decisions<FPL>(p, bits, j, acc);
            ++j;
            continue;
        }
        switch (d) {
            LDPC_RUN_VN(1) LDPC_RUN_VN(2) LDPC_RUN_VN(3) LDPC_RUN_VN(4) LDPC_RUN_VN(5) LDPC_RUN_VN(6)
            LDPC_RUN_VN(7) LDPC_RUN_VN(8) LDPC_RUN_VN(9) LDPC_RUN_VN(10) LDPC_RUN_VN(11) LDPC_RUN_VN(12)
            default: break;
        }
        float g[FPL];
        vn_dispatch<FPL>(d, msg, lam + (long long)j * p.lam_row, p.col_edge + c0, p.msg_row, write_beta, g, lds,
                         p.lds_maxd, lane);
#pragma unroll
        for (int f = 0; f < FPL; ++f) acc[f] |= (g[f] < 0.0f ? 1u : 0u) << (j & 31);  // "(gamma[j] < 0) ? 1 : 0", :104
        flush_decisions<FPL>(p, bits, j, acc);
        ++j;
    }
}
#undef LDPC_RUN_VN

template <int FPL, int DV>
__device__ __forceinline__ void vn_sweep(const BpParams& p, float* msg, const float* lam, uint32_t* bits,
                                         bool write_beta, float* lds, int lane) {
    if constexpr (DV > 0) {
        vn_sweep_regular<FPL, DV>(p, msg, lam, bits, write_beta);
        return;
    }
    vn_cols_any<FPL>(p, msg, lam, bits, write_beta, 0, p.n, lds, lane);
}

// number of unsatisfied checks of the packed decisions (sumSyndro, LogDecoder.java:108-116)
template <int FPL>
__device__ __forceinline__ void syndrome_sweep(const BpParams& p, const uint32_t* bits, int (&w)[FPL]) {
#pragma unroll
    for (int f = 0; f < FPL; ++f) w[f] = 0;
    for (int i = 0; i < p.m; ++i) {
        const int e0 = p.row_ptr[i], e1 = p.row_ptr[i + 1];
        uint32_t par[FPL];
#pragma unroll
        for (int f = 0; f < FPL; ++f) par[f] = 0u;
        for (int e = e0; e < e1; ++e) {
            const int j = p.edge_col[e];
            uint32_t v[FPL];
            ldw<FPL>(bits + (long long)(j >> 5) * p.bits_row, v);
#pragma unroll
            for (int f = 0; f < FPL; ++f) par[f] ^= v[f] >> (j & 31);
        }
#pragma unroll
        for (int f = 0; f < FPL; ++f) w[f] += (int)(par[f] & 1u);
    }
}

// ---- the decoder: one wave = one tile, all iterations in one launch -------------------------
// DC / DV > 0: every check / variable has exactly that degree (pipelined sweeps); 0: any degrees.
// WPS: waves per SIMD the register allocation must allow (4 -> 128 VGPRs, 2 -> 256 VGPRs).
template <int FPL, int RULE, bool EARLY, int DC, int DV, int WPS>
__global__ void __launch_bounds__(256, WPS) bp_decode_kernel(BpParams p) {
    extern __shared__ float lds_all[];
    static_assert(!EARLY || FPL == 1, "per-frame early termination masks whole lanes");
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int tile = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    if (tile >= p.tiles) return;
    constexpr int T = kWave * FPL;
    float* lds = lds_all + (long long)wave_in_block * 2 * p.lds_maxd * 64;
    float* msg = p.ws_msg + (long long)tile * p.msg_tile + lane * FPL;
    const float* lam = p.ws_lam + (long long)tile * p.lam_tile + lane * FPL;
    uint32_t* bits = p.ws_bits + (long long)tile * p.bits_tile + lane * FPL;
    const long long frame0 = p.frame_base + (long long)tile * T + lane * FPL;

    if constexpr (EARLY) {
        bool active = frame0 < p.frames_total;
        int iters = p.max_iters, synd = 0;
        for (int it = 0; it < p.max_iters; ++it) {
            if (active) {
                if (it == 0) cn_sweep<FPL, RULE, true, DC>(p, msg, lam, lds, lane);
                else         cn_sweep<FPL, RULE, false, DC>(p, msg, lam, lds, lane);
                vn_sweep<FPL, DV>(p, msg, lam, bits, true, lds, lane);
                int w[1];
                syndrome_sweep<1>(p, bits, w);
                synd = w[0];
                if (synd == 0) {  // "if(sumSyndro == 0) return ... l + 1", LogDecoder.java:117-119
                    iters = it + 1;
                    active = false;
                }
            }
            if (__ballot(active) == 0ull) break;
        }
        if (frame0 < p.frames_total) {
            if (p.out_iters) p.out_iters[frame0] = iters;
            if (p.out_synd) p.out_synd[frame0] = synd;
        }
    } else {
        for (int it = 0; it < p.max_iters; ++it) {
            if (it == 0) cn_sweep<FPL, RULE, true, DC>(p, msg, lam, lds, lane);
            else         cn_sweep<FPL, RULE, false, DC>(p, msg, lam, lds, lane);
            vn_sweep<FPL, DV>(p, msg, lam, bits, it + 1 < p.max_iters, lds, lane);
        }
        int w[FPL];
        syndrome_sweep<FPL>(p, bits, w);
#pragma unroll
        for (int f = 0; f < FPL; ++f) {
            if (frame0 + f < p.frames_total) {
                if (p.out_iters) p.out_iters[frame0 + f] = p.max_iters;
                if (p.out_synd) p.out_synd[frame0 + f] = w[f];
            }
        }
    }
}

// One launcher per (DC, DV) specialisation, each compiled in its own translation unit.
struct BpLaunch {
    int fpl;     // frames per lane: 1, 2 or 4
    int rule;    // kCnComplement / kCnSumProduct / kCnMinSum
    bool early;  // per-frame early termination (fpl == 1 only)
    int wps;     // 2 or 4 waves per SIMD (2 only with fpl == 4)
    dim3 grid, block;
    size_t lds_bytes;
    hipStream_t stream;
};

template <int DC, int DV>
inline void bp_launch_impl(const BpLaunch& l, const BpParams& p) {
#define LDPC_LAUNCH_W(F, R, E, W) \
    hipLaunchKernelGGL((bp_decode_kernel<F, R, E, DC, DV, W>), l.grid, l.block, l.lds_bytes, l.stream, p)
#define LDPC_LAUNCH(F, R, E)                                         \
    if (F == 4 && l.wps == 2) { LDPC_LAUNCH_W(F, R, E, (F == 4 ? 2 : 4)); } else { LDPC_LAUNCH_W(F, R, E, 4); }
#define LDPC_BY_RULE(F, E)                                       \
    switch (l.rule) {                                            \
        case kCnMinSum: LDPC_LAUNCH(F, kCnMinSum, E); break;     \
        case kCnSumProduct: LDPC_LAUNCH(F, kCnSumProduct, E); break; \
        default: LDPC_LAUNCH(F, kCnComplement, E); break;           \
    }
    if (l.fpl == 1) {
        if (l.early) { LDPC_BY_RULE(1, true) } else { LDPC_BY_RULE(1, false) }
    } else if (l.fpl == 2) {
        LDPC_BY_RULE(2, false)
    } else {
        LDPC_BY_RULE(4, false)
    }
#undef LDPC_BY_RULE
#undef LDPC_LAUNCH
#undef LDPC_LAUNCH_W
}

void bp_launch_generic(const BpLaunch& l, const BpParams& p);  // any degrees
void bp_launch_6_3(const BpLaunch& l, const BpParams& p);      // every check degree 6, every variable degree 3
void bp_launch_8_2(const BpLaunch& l, const BpParams& p);      // every check degree 8, every variable degree 2

}  // namespace ldpc
