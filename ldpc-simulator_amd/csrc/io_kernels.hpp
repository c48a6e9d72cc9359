// Layout changes around the decoder, the on-device channel and the error accounting.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace ldpc {

// ---- frame-row input -> frame-major lambda tiles ---------------------------------------------
// in[frame][j] (row per frame, what a caller holds) becomes lam[tile][j][t], t = frame % T:
// the layout in which lane t of a wave reads its own frame with one coalesced row per j.
// A 64 x 64 block goes through LDS (row stride 65 words, conflict-free both ways).
// `scale` converts the caller's natural-log LLRs into the unit the check-node rule works in
// (1/ln 2 for the default sum-product rule, 1 otherwise); decisions depend on signs only.
struct FloatLlr {
    const float* in;
    float scale;
    __device__ float operator()(long long idx, int /*j*/) const { return in[idx] * scale; }
};
// lambda from a hard received word: LogDecoder.java:45-54 / LogDecoderBefore.java:45-50
struct BscLlr {
    const uint8_t* in;
    float l0, l1;
    int g_length;  // n - m, LogDecoder.java:24
    __device__ float operator()(long long idx, int j) const {
        const float mag = (j < g_length) ? l0 : l1;
        return (in[idx] == 0) ? mag : -mag;
    }
};

template <class Src>
__global__ void __launch_bounds__(256) load_lambda_kernel(Src src, float* ws_lam, long long lam_tile, long long lam_row,
                                                          int n, int T, long long frame_base, long long frames_total) {
    __shared__ float tile[64][65];
    const int chunks_per_tile = T >> 6;
    const int tile_id = blockIdx.x / chunks_per_tile;
    const int t0 = (blockIdx.x % chunks_per_tile) << 6;
    const int j0 = blockIdx.y << 6;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long f0 = frame_base + (long long)tile_id * T + t0;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int fr = ty * 16 + r;
        const long long frame = f0 + fr;
        const int j = j0 + tx;
        float v = 0.0f;
        if (frame < frames_total && j < n) v = src(frame * n + j, j);
        tile[fr][tx] = v;
    }
    __syncthreads();
    float* out = ws_lam + (long long)tile_id * lam_tile;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int jj = ty * 16 + r;
        const int j = j0 + jj;
        if (j < n) out[(long long)j * lam_row + t0 + tx] = tile[tx][jj];
    }
}

// ---- packed decisions -> hard[frame][j] bytes ---------------------------------------------------
__global__ void __launch_bounds__(256) unpack_hard_kernel(const uint32_t* ws_bits, long long bits_tile, long long bits_row,
                                                          int n, int T, long long frame_base,
                                                          long long frames_in_launch, uint8_t* hard) {
    const int n4 = (n + 3) >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long fl = idx / n4;  // frame within the launch
    if (fl >= frames_in_launch) return;
    const int j = (int)(idx % n4) << 2;
    const long long tile = fl / T;
    const int t = (int)(fl % T);
    const uint32_t w = ws_bits[tile * bits_tile + (long long)(j >> 5) * bits_row + t] >> (j & 31);
    uint8_t* o = hard + (frame_base + fl) * n + j;
    if ((n & 3) == 0) {
        *reinterpret_cast<uchar4*>(o) = make_uchar4(w & 1u, (w >> 1) & 1u, (w >> 2) & 1u, (w >> 3) & 1u);
    } else {
        for (int k = 0; k < 4 && j + k < n; ++k) o[k] = (uint8_t)((w >> k) & 1u);
    }
}

// ---- channel: all-zero codeword -------------------------------------------------------------------
// Channel.java:5-17 / :28-39: "random.nextDouble() < e" becomes u32 < floor(e * 2^32) on one
// Philox word per position; positions >= flip_end are copied unchanged (error-free parity part).
__global__ void __launch_bounds__(256) gen_bsc_kernel(int n, int flip_end, uint32_t thr, uint64_t seed,
                                                      long long first_frame, long long frames, const uint8_t* cw,
                                                      uint8_t* r) {
    const int n4 = (n + 3) >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long f = idx / n4;
    if (f >= frames) return;
    const int g = (int)(idx % n4);
    const int j = g << 2;
    const u32x4 x = frame_block(seed, (uint64_t)(first_frame + f), (uint32_t)g, 0u);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
    uint8_t b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        b[k] = (j + k < flip_end && xs[k] < thr) ? 1 : 0;
        if (cw && j + k < n) b[k] ^= cw[f * n + j + k];  // r = c with flips (Channel.java:9-13); null = all-zero word
    }
    uint8_t* o = r + f * n + j;
    if ((n & 3) == 0) {
        *reinterpret_cast<uchar4*>(o) = make_uchar4(b[0], b[1], b[2], b[3]);
    } else {
        for (int k = 0; k < 4 && j + k < n; ++k) o[k] = b[k];
    }
}

// BPSK over AWGN, all-zero codeword (+1 sent): llr = 2*y/sigma^2 = a + b*z, a = 2/sigma^2,
// b = 2/sigma, z ~ N(0,1) by Box-Muller on two Philox words per pair of positions.
__global__ void __launch_bounds__(256) gen_awgn_kernel(int n, float a, float b, uint64_t seed, long long first_frame,
                                                       long long frames, const uint8_t* cw, float* llr) {
    const int n4 = (n + 3) >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long f = idx / n4;
    if (f >= frames) return;
    const int g = (int)(idx % n4);
    const int j = g << 2;
    const u32x4 x = frame_block(seed, (uint64_t)(first_frame + f), (uint32_t)g, 1u);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)xs[2 * h] + 0.5f) * 2.3283064365386963e-10f;
        const float u2 = (float)xs[2 * h + 1] * 2.3283064365386963e-10f;
        // the hardware's own log2, square root, sine and cosine (the last two take their argument in revolutions, which
        // is what u2 is): about 1e-6 of relative error in z, a quarter of the instructions of the library functions
        // (with those this kernel was bound by their arithmetic, at a third of the rate the LLR stores allow)
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1)
        z[2 * h] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
    // bit 0 is sent as +1, bit 1 as -1: llr = 2 y / sigma^2 with y = (1 - 2c) + sigma z
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sa = (cw && j + k < n && cw[f * n + j + k]) ? -a : a;
        v[k] = sa + b * z[k];
    }
    float* o = llr + f * n + j;
    if ((n & 3) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int k = 0; k < 4 && j + k < n; ++k) o[k] = v[k];
    }
}

// ---- source: random codewords u * G (GenerateC.java:12-24) --------------------------------------------
// Message bit i of a frame is bit (i & 127) of the Philox block (frame, i >> 7, stream 2).
// A block encodes 64 frames at once, bit-sliced over the frames: U[i] holds message bit i of the 64 frames
// in one 64-bit word, and the 64 code bits at position j are the XOR of U[i] over the ones of column j of
// G.  The XOR is taken four message bits at a time: T[g][s], s = 0..15, is the XOR of the U words of
// group g (message bits 4g .. 4g+3) selected by the bits of s, built once per block in LDS, and a position
// looks up one entry per nibble of its column - k/4 branch-free steps instead of one step per set bit
// (about k/2 in a parity column), each a 64-bit LDS read whose 64 lanes fall into 16 consecutive words
// (no bank conflicts).  GT[kw64][n] is G by columns, 64 rows to a word, word-major, so that the lanes of a
// wave (consecutive positions) read consecutive words; a zero word (most words of a column of the identity
// part of a systematic G) is skipped; the words of a column are read eight at a time (one wait per eight).
// T covers kCwWordsPerPass words of the columns at a time (32 KiB, so that two blocks share a CU: 5.4 -> 4.5 ns
// per frame at k = 2048 against 64 KiB and one block); longer columns take several passes and rebuild it.
constexpr int kCwFrames = 64;          // frames per block
constexpr int kCwThreads = 1024;       // positions per round
constexpr int kCwWordsPerPass = 16;    // 64-bit words of a column per table

// bytes of dynamic LDS: U | T | stage, the raw message words of the first phase lie over T and stage
__host__ __device__ inline size_t gen_codewords_lds_bytes(int k) {
    const size_t kw32 = ((size_t)k + 31) >> 5, kw64 = ((size_t)k + 63) >> 6;
    const size_t wp = kw64 < (size_t)kCwWordsPerPass ? kw64 : (size_t)kCwWordsPerPass;
    const size_t tab = (wp * 256 + kCwThreads) * sizeof(unsigned long long);
    const size_t raw = (size_t)kCwFrames * (kw32 | 1) * sizeof(uint32_t);
    return kw64 * 64 * sizeof(unsigned long long) + (tab > raw ? tab : raw);
}

__global__ void __launch_bounds__(kCwThreads) gen_codewords_kernel(int n, int k, const unsigned long long* GT, uint64_t seed,
                                                                   long long first_frame, long long frames, uint8_t* cw) {
    extern __shared__ unsigned long long cw_lds[];
    const int kw32 = (k + 31) >> 5, kw64 = (k + 63) >> 6, groups = (k + 127) >> 7;
    const int wp = kw64 < kCwWordsPerPass ? kw64 : kCwWordsPerPass;
    const int raw_stride = kw32 | 1;                       // odd: the 64 lanes of a transposing read hit 64 banks
    unsigned long long* U = cw_lds;                        // [kw64 * 64] message bit i of the 64 frames
    unsigned long long* T = U + (size_t)kw64 * 64;         // [wp * 16 groups][16] XOR combinations of a group
    unsigned long long* stage = T + (size_t)wp * 256;      // [kCwThreads] code bits of the positions in flight
    uint32_t* raw = reinterpret_cast<uint32_t*>(T);        // [64][raw_stride] message words by frame (first phase only)
    const long long f0 = (long long)blockIdx.x * kCwFrames;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < kCwFrames * groups; idx += kCwThreads) {
        const int fr = idx / groups, g = idx - fr * groups;
        const u32x4 x = frame_block(seed, (uint64_t)(first_frame + f0 + fr), (uint32_t)g, 2u);
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
        for (int q = 0; q < 4; ++q)
            if (4 * g + q < kw32) raw[fr * raw_stride + 4 * g + q] = xs[q];
    }
    __syncthreads();
    for (int i = wave; i < kw64 * 64; i += kCwThreads / 64) {   // lane = frame: one ballot transposes bit i
        const bool bit = i < k && ((raw[lane * raw_stride + (i >> 5)] >> (i & 31)) & 1u);
        const unsigned long long mask = __ballot(bit);
        if (lane == 0) U[i] = mask;
    }
    __syncthreads();
    const bool one_pass = kw64 <= wp;
    for (int j0 = 0; j0 < n; j0 += kCwThreads) {
        const int j = j0 + threadIdx.x;
        unsigned long long acc = 0ull;
        for (int w0 = 0; w0 < kw64; w0 += wp) {
            const int words = (kw64 - w0 < wp) ? kw64 - w0 : wp;
            if (!one_pass || j0 == 0) {
                if (!one_pass) __syncthreads();            // the lookups of the previous pass are over
                for (int e = threadIdx.x; e < words * 256; e += kCwThreads) {
                    const unsigned long long* u = U + (size_t)w0 * 64 + 4 * (e >> 4);
                    unsigned long long v = 0ull;
                    if (e & 1) v ^= u[0];
                    if (e & 2) v ^= u[1];
                    if (e & 4) v ^= u[2];
                    if (e & 8) v ^= u[3];
                    T[e] = v;
                }
                __syncthreads();
            }
            if (j < n) {
                // the words of the column eight at a time: one wait per eight reads instead of one per word
                for (int wb = 0; wb < words; wb += 8) {
                    unsigned long long col[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) col[u] = (wb + u < words) ? GT[(size_t)(w0 + wb + u) * n + j] : 0ull;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned long long bits = col[u];
                        if (bits == 0ull) continue;
                        const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
                        const unsigned long long* Tw = T + (wb + u) * 256;
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc ^= Tw[q * 16 + ((lo >> (4 * q)) & 15u)];
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc ^= Tw[(8 + q) * 16 + ((hi >> (4 * q)) & 15u)];
                    }
                }
            }
        }
        stage[threadIdx.x] = acc;
        __syncthreads();
        // bit fr of stage[p] -> byte cw[f0 + fr][j0 + p]: a wave writes 256 consecutive bytes of one frame
        for (int item = wave; item < kCwFrames * (kCwThreads / 256); item += kCwThreads / 64) {
            const int fr = item / (kCwThreads / 256), p = (item % (kCwThreads / 256)) * 256 + 4 * lane;
            const long long f = f0 + fr;
            if (f >= frames) break;
            if (j0 + p >= n) continue;
            uint8_t* o = cw + f * n + j0 + p;
            uint32_t b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = (uint32_t)((stage[p + q] >> fr) & 1ull);
            if (j0 + p + 3 < n && ((n | j0) & 3) == 0) {
                *reinterpret_cast<uint32_t*>(o) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            } else {
                for (int q = 0; q < 4 && j0 + p + q < n; ++q) o[q] = (uint8_t)b[q];
            }
        }
        __syncthreads();
    }
}

// ---- error accounting (LDPC_GallagerIntStreamSimu.java:136-183, c = all-zero) -------------------------
// One wave per frame at a time, lanes stride the positions; a block handles 64 frames and keeps its
// sums in LDS so that each block issues at most (16 + max_iters) global atomics.
// rx(i): received hard bit at position i of the batch; four(i): the bits at i .. i+3 in bits 0, 8, 16, 24
// (i a multiple of 4, rows 4-byte aligned)
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) {  // 0x01 in every byte of w that is not zero
    return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7 & 0x01010101u;
}
struct RxNone {
    __device__ int operator()(long long) const { return 0; }
    __device__ uint32_t four(long long) const { return 0u; }
    __device__ bool aligned() const { return true; }
};
struct RxBytes {
    const uint8_t* p;
    __device__ int operator()(long long i) const { return p[i] != 0; }
    __device__ uint32_t four(long long i) const { return nonzero_bytes(*reinterpret_cast<const uint32_t*>(p + i)); }
    __device__ bool aligned() const { return ((uintptr_t)p & 3) == 0; }
};
struct RxLlrSign {
    const float* p;
    __device__ int operator()(long long i) const { return p[i] < 0.0f; }
    __device__ uint32_t four(long long i) const {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        return (v.x < 0.0f ? 1u : 0u) | (v.y < 0.0f ? 0x100u : 0u) | (v.z < 0.0f ? 0x10000u : 0u) |
               (v.w < 0.0f ? 0x1000000u : 0u);
    }
    __device__ bool aligned() const { return ((uintptr_t)p & 15) == 0; }
};

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <class Rx, bool HAS_RX>
__global__ void __launch_bounds__(256) count_errors_kernel(int n, int k_info, long long frames, const uint8_t* hard,
                                                           const uint8_t* cw, Rx rx, const int32_t* iters,
                                                           const int32_t* synd, int max_iters,
                                                           unsigned long long* counters) {
    extern __shared__ unsigned long long acc[];  // 16 + max_iters
    const int nc = 16 + max_iters;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) acc[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long fb = (long long)blockIdx.x * 64;
    const bool vec = (n & 3) == 0 && (((uintptr_t)hard | (uintptr_t)cw) & 3) == 0 && (!HAS_RX || rx.aligned());
    for (int q = wave; q < 64; q += 4) {
        const long long f = fb + q;
        if (f >= frames) break;
        unsigned info = 0, chk = 0, mis = 0, ch = 0;
        const uint8_t* d = hard + f * n;
        if (vec) {  // four positions per lane and step: one 4-byte (16-byte for LLRs) load each
            for (int j = 4 * lane; j < n; j += 256) {
                const uint32_t c4 = cw ? nonzero_bytes(*reinterpret_cast<const uint32_t*>(cw + f * n + j)) : 0u;
                const uint32_t d4 = nonzero_bytes(*reinterpret_cast<const uint32_t*>(d + j)) ^ c4;
                // bytes of positions below k_info
                const int left = k_info - j;
                const uint32_t im = left >= 4 ? 0x01010101u : left <= 0 ? 0u : (0x01010101u >> (8 * (4 - left)));
                info += __popc(d4 & im);
                chk += __popc(d4 & ~im);
                if (HAS_RX) {
                    if (left > 0) {
                        const uint32_t r4 = (rx.four(f * n + j) ^ c4) & im;
                        ch += __popc(r4);
                        mis += __popc(d4 & im & ~r4);
                    }
                }
            }
        } else
        for (int j = lane; j < n; j += 64) {
            const unsigned cj = cw ? (cw[f * n + j] != 0) : 0u;  // transmitted bit; null = all-zero word
            const unsigned dj = (unsigned)(d[j] != 0) ^ cj;       // decoded bit differs from it
            if (j < k_info) {
                info += dj;
                if (HAS_RX) {
                    const unsigned rj = (unsigned)rx(f * n + j) ^ cj;  // channel error on this bit
                    ch += rj;
                    mis += (rj == 0u && dj) ? 1u : 0u;
                }
            } else {
                chk += dj;
            }
        }
        info = wave_sum(info); chk = wave_sum(chk); mis = wave_sum(mis); ch = wave_sum(ch);
        if (lane == 0) {
            const bool correct = (info + chk) == 0;
            const int it = iters[f];
            atomicAdd(&acc[0], 1ull);
            if (!correct) atomicAdd(&acc[1], 1ull);
            if (info) { atomicAdd(&acc[2], (unsigned long long)info); atomicAdd(&acc[4], 1ull); }
            if (chk) { atomicAdd(&acc[3], (unsigned long long)chk); atomicAdd(&acc[5], 1ull); }
            if (synd[f] == 0 && !correct) atomicAdd(&acc[6], 1ull);
            if (HAS_RX) {
                if (mis) atomicAdd(&acc[7], (unsigned long long)mis);
                if (ch) {
                    atomicAdd(&acc[10], (unsigned long long)ch);
                    atomicAdd(&acc[11], (unsigned long long)ch * ch);
                }
            }
            atomicAdd(&acc[correct ? 8 : 9], (unsigned long long)it);
            if (it >= 1 && it <= max_iters) atomicAdd(&acc[16 + it - 1], 1ull);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += blockDim.x)
        if (acc[i]) atomicAdd(&counters[i], acc[i]);
}

}  // namespace ldpc
