// Row-parallel sweeps: the streaming engine for batches too small to give every CU its own tiles.
//
// bp_decode_kernel gives one wavefront one tile of frames for the whole decode, which needs thousands
// of tiles to fill 256 CUs.  A long code with a small batch (n = 64800 with 64 frames is ONE tile) has
// the parallelism elsewhere: the checks of one sweep are independent of each other, and so are the
// variables.  Here every sweep is its own launch over (tile, row chunk) pairs -- the launch boundary is
// the grid-wide barrier between the CN and the VN sweep -- on the same frame-major arrays and with the
// same per-row code (cn_dispatch / vn_dispatch of bp_kernels.hpp), so the bits are identical to the
// one-launch kernel's.  Per-frame early termination lives in small state arrays between launches.
//
// Early termination costs no launch of its own: the syndrome of the decisions of iteration l-1
// (LogDecoder.java:108-116) is formed by the check-node launch of iteration l, whose waves walk the same
// rows anyway, into the accumulator of parity (l-1) & 1; the variable-node launch of iteration l, which
// starts after that launch has ended, reads the complete weights: a frame whose weight is 0 had converged
// at iteration l-1, its lanes sit out from here on and its decisions stay as that iteration left them
// (the check-node update its messages have just received is never used).  The first wave of every tile
// records that (done, l, weight 0) and clears the other accumulator for the next iteration.  Only after
// the last iteration do the syndrome and the closing launch run on their own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernels.hpp"

namespace ldpc {

// The waves of a launch add their rows' share of a frame's syndrome weight into that frame's accumulator.  With one
// accumulator per frame every wave of a tile sends its atomic to the same two cache lines, and a small batch (one
// tile, thousands of waves) waits for them in a row: there the accumulators are kept in kSyndParts parts in different
// lines (RowsParams::synd_parts = kSyndParts; 1 for the large batches, whose tiles spread the traffic by themselves
// and whose variable-node waves would only pay for reading the parts).
constexpr int kSyndParts = 16;

struct RowsParams {
    BpParams b;          // graph, workspace arrays, strides, tiles, frame_base, frames_total, max_iters
    int rows_per_chunk;  // checks per wave in a CN / syndrome launch
    int vars_per_chunk;  // variables per wave in a VN launch, a multiple of 32
    int iter;            // iteration this launch belongs to (0-based)
    int early;           // per-frame early termination
    int fused;           // early termination without launches of its own (see above)
    int32_t* st_done;    // [tiles*T] 0 = still iterating (early mode)
    int32_t* st_acc;     // [2][synd_parts][slots] syndrome weight accumulators, by parity of the iteration they belong to;
                         // a wave adds to part (its chunk mod synd_parts), the readers sum the parts
    int synd_parts;      // 1 or kSyndParts (a power of two)
    int slots;           // tiles * T
    int32_t* st_iters;   // [tiles*T]
    int32_t* st_synd;    // [tiles*T]
};

#ifdef LDPC_ROWS_KERNELS  // the kernels are compiled in bp_rows.hip only
static_assert(LDPC_LANE_IN_ACCESS == 1, "these kernels hand wave-uniform row addresses to the accesses");

template <int FPL>
__device__ __forceinline__ bool rows_lane_active(const RowsParams& p, int tile, int lane) {
    // FPL > 1 only runs without early termination; with FPL == 1 a finished frame's lane sits out
    if constexpr (FPL == 1) return !p.early || p.st_done[(long long)tile * kWave + lane] == 0;
    return true;
}

// number of unsatisfied checks among rows i0 .. i1-1 of the packed decisions (sumSyndro, LogDecoder.java:108-116),
// for the check-node launch that carries the syndrome of the previous iteration.  A small batch waits on every
// round trip, so the decision words of four rows of at most eight members each are requested together (32 loads in
// flight); a longer row goes eight loads at a time.  Row pointers and members are the same in every lane: scalar.
template <int FPL>
__device__ __forceinline__ void rows_syndrome_row(const BpParams& b, const uint32_t* bits, int e0, int e1, int (&w)[FPL]) {
    uint32_t par[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) par[f] = 0u;
    for (int e = e0; e < e1; e += 8) {
        int j[8];
        uint32_t v[8][FPL];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = (e + u < e1) ? idx(b.edge_col, e + u) : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j[u] >= 0) ldw<FPL>(bits + (long long)(j[u] >> 5) * b.bits_row, v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j[u] >= 0) {
#pragma unroll
                for (int f = 0; f < FPL; ++f) par[f] ^= v[u][f] >> (j[u] & 31);
            }
    }
#pragma unroll
    for (int f = 0; f < FPL; ++f) w[f] += (int)(par[f] & 1u);
}
template <int FPL>
__device__ __forceinline__ void rows_syndrome(const BpParams& b, const uint32_t* bits, int i0, int i1, int (&w)[FPL]) {
#pragma unroll
    for (int f = 0; f < FPL; ++f) w[f] = 0;
    constexpr int R = 4, D = 8;
    for (int i = i0; i < i1; i += R) {
        int rp[R + 1];
#pragma unroll
        for (int g = 0; g <= R; ++g) rp[g] = idx(b.row_ptr, min(i + g, i1));
        bool short_rows = true;
#pragma unroll
        for (int g = 0; g < R; ++g) short_rows = short_rows && rp[g + 1] - rp[g] <= D;
        if (short_rows) {
            int j[R][D];
            uint32_t v[R][D][FPL];
#pragma unroll
            for (int g = 0; g < R; ++g)
#pragma unroll
                for (int u = 0; u < D; ++u) j[g][u] = (rp[g] + u < rp[g + 1]) ? idx(b.edge_col, rp[g] + u) : -1;
#pragma unroll
            for (int g = 0; g < R; ++g)
#pragma unroll
                for (int u = 0; u < D; ++u)
                    if (j[g][u] >= 0) ldw<FPL>(bits + (long long)(j[g][u] >> 5) * b.bits_row, v[g][u]);
#pragma unroll
            for (int g = 0; g < R; ++g) {
                uint32_t par[FPL];
#pragma unroll
                for (int f = 0; f < FPL; ++f) par[f] = 0u;
#pragma unroll
                for (int u = 0; u < D; ++u)
                    if (j[g][u] >= 0) {
#pragma unroll
                        for (int f = 0; f < FPL; ++f) par[f] ^= v[g][u][f] >> (j[g][u] & 31);
                    }
#pragma unroll
                for (int f = 0; f < FPL; ++f) w[f] += (int)(par[f] & 1u);
            }
        } else {
#pragma unroll
            for (int g = 0; g < R; ++g) rows_syndrome_row<FPL>(b, bits, rp[g], rp[g + 1], w);
        }
    }
}

// A wave's rows are handled in groups whose loads are all issued before the first of them is used
// (rows_group / cols_group in bp_kernels.hpp), so that a small batch, where a launch is a few thousand
// short waves and nothing else hides the memory latency, waits once per group instead of once per row:
// every group of a regular code (DC / DV > 0 = the degree of every check / variable), runs of equal
// degree otherwise.  Same per-row arithmetic, same bits.
template <int FPL, int RULE, bool FIRST, int DC, bool FUSED>
__global__ void __launch_bounds__(256, 4) rows_cn_kernel(RowsParams p) {
    static_assert(!FUSED || (FPL == 1 && !FIRST), "the syndrome of the previous iteration, one frame per lane");
    extern __shared__ float lds_all[];
    float* lds = lds_all + (long long)(threadIdx.x >> 6) * 2 * p.b.lds_maxd * 64;
    const int lane = threadIdx.x & 63;
    // the same in every lane of a wavefront: as a scalar, everything indexed by it (row and column pointers, edge lists,
    // row addresses) is scalar loads and scalar arithmetic, and a row address costs no vector registers
    const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int tile = blockIdx.y;
    const int i0 = chunk * p.rows_per_chunk;
    if (i0 >= p.b.m) return;
    const int i1 = min(p.b.m, i0 + p.rows_per_chunk);
    // row addresses, the same in every lane: the accesses add the lane (LDPC_LANE_IN_ACCESS)
    float* msg = p.b.ws_msg + (long long)tile * p.b.msg_tile;
    const float* lam = p.b.ws_lam + (long long)tile * p.b.lam_tile;
    if (rows_lane_active<FPL>(p, tile, lane)) {
        if constexpr (FUSED) {  // weight of the previous iteration's syndrome over this wave's rows
            int w[1];
            rows_syndrome<1>(p.b, p.b.ws_bits + (long long)tile * p.b.bits_tile, i0, i1, w);
            if (w[0])
                atomicAdd(&p.st_acc[(long long)(((p.iter - 1) & 1) * p.synd_parts + (chunk & (p.synd_parts - 1))) * p.slots +
                                    (long long)tile * kWave + lane], w[0]);
        }
        if constexpr (DC > 0) {
            constexpr int G = rows_group<FPL, DC>();
            for (int i = i0; i < i1; i += G) {
                float buf[G][DC][FPL];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (i + g < i1) cn_load<FPL, DC, FIRST>(p.b, msg, lam, i + g, buf[g]);
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (i + g < i1) cn_finish<FPL, DC, RULE, FIRST>(p.b, msg, i + g, buf[g]);
            }
        } else {
            cn_rows_any<FPL, RULE, FIRST>(p.b, msg, lam, i0, i1, lds, lane);
        }
    }
}

template <int FPL, int DV, bool FUSED>
__global__ void __launch_bounds__(256, 4) rows_vn_kernel(RowsParams p, int write_beta) {
    static_assert(!FUSED || FPL == 1, "early termination runs with one frame per lane");
    extern __shared__ float lds_all[];
    float* lds = lds_all + (long long)(threadIdx.x >> 6) * 2 * p.b.lds_maxd * 64;
    const int lane = threadIdx.x & 63;
    // the same in every lane of a wavefront: as a scalar, everything indexed by it (row and column pointers, edge lists,
    // row addresses) is scalar loads and scalar arithmetic, and a row address costs no vector registers
    const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int tile = blockIdx.y;
    const int j0 = chunk * p.vars_per_chunk;
    if (j0 >= p.b.n) return;
    const int j1 = min(p.b.n, j0 + p.vars_per_chunk);
    float* msg = p.b.ws_msg + (long long)tile * p.b.msg_tile;
    const float* lam = p.b.ws_lam + (long long)tile * p.b.lam_tile;
    uint32_t* bits = p.b.ws_bits + (long long)tile * p.b.bits_tile;
    bool active = rows_lane_active<FPL>(p, tile, lane);
    if constexpr (FUSED) {
        {
            // the check-node launch of this iteration has left the weight of the previous iteration's syndrome
            const long long slot = (long long)tile * kWave + lane;
            int w = 0;
            if (p.synd_parts == 1) {
                w = p.st_acc[(long long)((p.iter - 1) & 1) * p.slots + slot];
            } else {  // all parts requested together
#pragma unroll
                for (int k = 0; k < kSyndParts; ++k) w += p.st_acc[(long long)(((p.iter - 1) & 1) * kSyndParts + k) * p.slots + slot];
            }
            if (chunk == 0) {  // one wave per tile keeps the books
                if (active) {
                    p.st_synd[slot] = w;
                    if (w == 0) {  // "if(sumSyndro == 0) return ... l + 1", LogDecoder.java:117-119
                        p.st_done[slot] = 1;
                        p.st_iters[slot] = p.iter;
                    }
                }
                for (int k = 0; k < p.synd_parts; ++k)
                    p.st_acc[(long long)((p.iter & 1) * p.synd_parts + k) * p.slots + slot] = 0;  // last read one launch pair ago
            }
            active = active && w != 0;
        }
    }
    if (!active) return;
    uint32_t acc[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) acc[f] = 0u;
    if constexpr (DV > 0) {
        constexpr int G = rows_group<FPL, DV>();
        for (int j = j0; j < j1; j += G) {
            VnBuf<FPL, DV> buf[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (j + g < j1) vn_load<FPL, DV>(p.b, msg, lam, j + g, buf[g]);
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (j + g < j1) vn_finish<FPL, DV>(p.b, msg, bits, j + g, write_beta != 0, buf[g], acc);
        }
        return;
    }
    vn_cols_any<FPL>(p.b, msg, lam, bits, write_beta != 0, j0, j1, lds, lane);
}

template <int FPL>
__global__ void __launch_bounds__(256, 4) rows_synd_kernel(RowsParams p) {
    const int lane = threadIdx.x & 63;
    // the same in every lane of a wavefront: as a scalar, everything indexed by it (row and column pointers, edge lists,
    // row addresses) is scalar loads and scalar arithmetic, and a row address costs no vector registers
    const int chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int tile = blockIdx.y;
    const int i0 = chunk * p.rows_per_chunk;
    if (i0 >= p.b.m) return;
    const int i1 = min(p.b.m, i0 + p.rows_per_chunk);
    const uint32_t* bits = p.b.ws_bits + (long long)tile * p.b.bits_tile;
    if (!rows_lane_active<FPL>(p, tile, lane)) return;
    int w[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) w[f] = 0;
    for (int i = i0; i < i1; ++i) {
        const int e0 = idx(p.b.row_ptr, i), e1 = idx(p.b.row_ptr, i + 1);
        uint32_t par[FPL];
#pragma unroll
        for (int f = 0; f < FPL; ++f) par[f] = 0u;
        int e = e0;
        for (; e + 4 <= e1; e += 4) {  // four loads in flight: the row is a chain of cache hits otherwise
            int j[4];
            uint32_t v[4][FPL];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                j[u] = idx(p.b.edge_col, e + u);
                ldw<FPL>(bits + (long long)(j[u] >> 5) * p.b.bits_row, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int f = 0; f < FPL; ++f) par[f] ^= v[u][f] >> (j[u] & 31);
        }
        for (; e < e1; ++e) {
            const int j = idx(p.b.edge_col, e);
            uint32_t v[FPL];
            ldw<FPL>(bits + (long long)(j >> 5) * p.b.bits_row, v);
#pragma unroll
            for (int f = 0; f < FPL; ++f) par[f] ^= v[f] >> (j & 31);
        }
#pragma unroll
        for (int f = 0; f < FPL; ++f) w[f] += (int)(par[f] & 1u);
    }
    int32_t* acc = p.st_acc + (long long)((p.iter & 1) * p.synd_parts + (chunk & (p.synd_parts - 1))) * p.slots +
                   (long long)tile * (kWave * FPL) + lane * FPL;
#pragma unroll
    for (int f = 0; f < FPL; ++f)
        if (w[f]) atomicAdd(&acc[f], w[f]);
}

// one thread per frame slot: close the iteration (LogDecoder.java:115-119)
__global__ void __launch_bounds__(256) rows_finish_kernel(RowsParams p, int slots, int last) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    if (p.early && p.st_done[s]) return;
    int w = 0;
    for (int k = 0; k < p.synd_parts; ++k) {
        w += p.st_acc[(long long)((p.iter & 1) * p.synd_parts + k) * p.slots + s];
        p.st_acc[(long long)((p.iter & 1) * p.synd_parts + k) * p.slots + s] = 0;
    }
    p.st_synd[s] = w;
    if (p.early && w == 0) {
        p.st_done[s] = 1;
        p.st_iters[s] = p.iter + 1;
    } else if (last) {
        p.st_iters[s] = p.b.max_iters;
    }
}

__global__ void __launch_bounds__(256) rows_export_kernel(RowsParams p, int slots) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    const long long frame = p.b.frame_base + s;
    if (frame >= p.b.frames_total) return;
    if (p.b.out_iters) p.b.out_iters[frame] = p.st_iters[s];
    if (p.b.out_synd) p.b.out_synd[frame] = p.st_synd[s];
}

#endif  // LDPC_ROWS_KERNELS

constexpr int kRowsRegDegCn = 20, kRowsRegDegVn = 32;  // register limits of the kernels in bp_rows.hip

struct RowsLaunch {
    int dc, dv;         // degrees of a regular code the kernels are specialised for, else 0, 0
    int fpl;            // 1 or 4
    int rule;
    int chunks;         // row / variable chunks per tile
    int wpb;            // waves per block (1 when rows are staged through LDS)
    size_t lds_bytes;   // dynamic LDS per block
    hipStream_t stream;
};

// All launches of one decode of the tiles in p.b (state arrays zeroed by the caller).
hipError_t bp_rows_run(const RowsLaunch& l, RowsParams p);
unsigned bp_rows_params_bytes();  // see bp_kernels.hpp

}  // namespace ldpc
