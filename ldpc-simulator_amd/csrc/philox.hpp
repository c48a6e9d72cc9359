// Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw, SC'11), device side.
// Replaces the unseeded java.util.Random of src/LDPC/Channel.java:7, :30: the counter is
// (frame, position / 4, stream) and the key is the user's seed, so a frame's channel noise
// is the same on any GPU and in any batch split.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldpc {

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// One 128-bit block for positions 4*group .. 4*group+3 of `frame`.
// stream 0 = BSC flips, stream 1 = AWGN noise.
__host__ __device__ inline u32x4 frame_block(uint64_t seed, uint64_t frame, uint32_t group, uint32_t stream) {
    u32x4 c{(uint32_t)frame, (uint32_t)(frame >> 32), group, stream};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

}  // namespace ldpc
