// Node-parallel engine for very small batches: one thread per check / per variable of a frame.
//
// The other engines put one frame on a lane, which leaves 63 of 64 lanes idle when a caller brings a
// single frame of a long code, and spend about 80 dependent launches on it.  Here a batch of at most 32
// frames runs in ONE cooperative launch: the threads of the grid walk over (frame, check) pairs in the
// check-node phase and over (frame, variable) pairs in the variable-node phase, a grid-wide barrier
// stands where the sweep launches had their launch boundary, and the messages of a frame lie in CSR
// edge order (msg[frame][E]), so a check's row is contiguous in its thread and the variables gather.
// The per-node arithmetic is the code of the other engines (cn_dispatch / vn_dispatch with one frame per
// lane and row stride 1): same operations in the same order, identical bits.
//
// Early termination (LogDecoder.java:108-119): the syndrome of iteration l is formed at the start of
// the check-node phase of iteration l+1, from the decisions the variable-node phase of l has stored,
// into a counter of its own (synd_acc[frame][l], zeroed by the host, written only in that phase and
// read only after the barrier that ends it).  Every thread reads the same counters after the same
// barrier, so `done` holds the same bits in every thread and the loop exits are grid-uniform; a frame
// found converged keeps its decisions (its nodes are skipped from then on).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernels.hpp"

namespace ldpc {

constexpr int kNodesMaxFrames = 32;                   // one bit per frame in a register
constexpr int kNodesMaxDc = 20, kNodesMaxDv = 32;     // degrees the register code of bp_nodes.hip covers
constexpr int kNodesBarWords = 18;                    // words of grid-barrier state (NodesParams::bar)

struct NodesParams {
    int m, n;
    long long E;
    const int* row_ptr;
    const int* edge_col;
    const int* col_ptr;
    const int* col_edge;
    const float* llr;    // [frames][n] channel LLRs, or null
    const uint8_t* r;    // [frames][n] hard received word when llr is null (LogDecoder.java:49-54)
    float l0, l1, scale;
    int g_length;
    int frames, max_iters, early;
    int flat_barrier;    // 1: every grid barrier is the one-level one (nodes_grid_barrier_flat); results identical
    float* msg;          // [frames][E]
    float* lam;          // [frames][n]
    int32_t* synd_acc;   // [frames][max_iters], zero on entry
    unsigned* bar;       // [kNodesBarWords] grid barrier state (nodes_grid_barrier); zero on entry
    uint8_t* hard;       // [frames][n]
    int32_t* out_iters;
    int32_t* out_synd;
};

#ifdef LDPC_NODES_KERNELS  // the kernel is compiled in bp_nodes.hip only

// number of unsatisfied checks among this wave's (frame f, check i) items -> synd_acc (one atomic per wave)
__device__ __forceinline__ void nodes_syndrome(const NodesParams& p, int f, int i, int32_t* acc, int lane) {
    bool odd = false;
    if (i < p.m) {
        const uint8_t* dec = p.hard + (long long)f * p.n;
        uint32_t par = 0u;
        for (int e = p.row_ptr[i]; e < p.row_ptr[i + 1]; ++e) par ^= dec[p.edge_col[e]];
        odd = (par & 1u) != 0u;
    }
    const int cnt = __popcll(__ballot(odd));
    if (lane == 0 && cnt) atomicAdd(acc, cnt);
}

// Grid-wide barrier.  The phase's plain stores sit in the L2 of the XCD that wrote them; an agent-scope release
// writes that whole L2 back, and when every workgroup issues one (the simple barrier, `nodes_grid_barrier_flat`)
// those write-backs are what the barrier costs (measured on one frame of n = 64800, 64 workgroups: 9 of 23 us
// per iteration).  The barrier used between the phases therefore has two levels: every wave drains its stores
// (s_waitcnt vmcnt(0): they have reached the L2), the workgroup meets, one lane adds 1 to the counter of its
// XCD; the workgroup that completes its XCD's count issues the release for all of them - one write-back per XCD
// and barrier - and only then adds 1 to the grid counter, on which every workgroup polls.  After the poll has
// matched the lane acquires at agent scope (this CU's L1 and the L2 lines of other XCDs' data are dropped) and
// the workgroup meets again before anyone loads.
// bar[0] grid counter, bar[1] time-out flag, bar[2..9] arrivals per XCD, bar[10..17] workgroups per XCD; all
// zero on entry, all monotonic.  The grid is launched cooperatively, so all workgroups are resident and the
// waits end; they are bounded all the same (about 0.2 s): a time-out raises bar[1], every later barrier falls
// through at once, the kernel finishes with meaningless results, writes -1 into every frame's iteration count and
// syndrome weight, and the host reports the flag as an error.
__device__ __forceinline__ void nodes_poll(unsigned* bar, unsigned target) {
    unsigned spins = 0;
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 21) || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

__device__ __forceinline__ void nodes_grid_barrier_flat(unsigned* bar, unsigned& target, unsigned blocks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        target += blocks;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nodes_poll(bar, target);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

struct NodesBarrier {
    unsigned* bar;
    unsigned grid_target;   // value of bar[0] once everyone has passed the next barrier
    unsigned xcd_target;    // value of bar[2 + xcd] once every workgroup of this XCD has arrived
    unsigned xcd, xcd_blocks, xcds;
    unsigned flat, blocks;  // flat: the one-level barrier over all `blocks` workgroups instead (A/B and tests)
};

__device__ __forceinline__ void nodes_grid_barrier(NodesBarrier& b) {
    if (b.flat) {  // grid-uniform
        nodes_grid_barrier_flat(b.bar, b.grid_target, b.blocks);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        b.grid_target += b.xcds;
        b.xcd_target += b.xcd_blocks;
        const unsigned before = __hip_atomic_fetch_add(&b.bar[2 + b.xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == b.xcd_target) {  // the last workgroup of this XCD: everyone's stores are in its L2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&b.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        nodes_poll(b.bar, b.grid_target);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int RULE>
__global__ void __launch_bounds__(256) bp_nodes_kernel(NodesParams p) {
    // which XCD this workgroup runs on (XCC_ID, bits 3:0 of the hardware register); the workgroups count
    // themselves per XCD before the first barrier, which is the flat one
    NodesBarrier nb;
    nb.bar = p.bar;
    {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        nb.xcd = id & 7u;
    }
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&p.bar[10 + nb.xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long gsize = (long long)gridDim.x * blockDim.x;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // items are (frame, node) pairs with the node count padded to whole waves: a wave stays in one frame
    const int mp = (p.m + 63) & ~63, np = (p.n + 63) & ~63;
    const long long cn_items = (long long)p.frames * mp, vn_items = (long long)p.frames * np;
    const int L = p.max_iters;

    for (long long item = gtid; item < vn_items; item += gsize) {
        const int f = (int)(item / np), j = (int)(item - (long long)f * np);
        if (j >= p.n) continue;
        const long long at = (long long)f * p.n + j;
        float v;
        if (p.llr) {
            v = p.llr[at] * p.scale;
        } else {
            const float mag = (j < p.g_length) ? p.l0 : p.l1;
            v = (p.r[at] == 0) ? mag : -mag;
        }
        p.lam[at] = v;
        p.hard[at] = 0;
    }
    nb.grid_target = 0u;
    nodes_grid_barrier_flat(p.bar, nb.grid_target, gridDim.x);
    nb.xcd_target = 0u;
    nb.xcds = 0u;
    nb.flat = p.flat_barrier ? 1u : 0u;
    nb.blocks = gridDim.x;
    for (unsigned x = 0; x < 8u; ++x) {
        const unsigned cnt = __hip_atomic_load(&p.bar[10 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nb.xcds += cnt != 0u;
        if (x == nb.xcd) nb.xcd_blocks = cnt;
    }

    const uint32_t all = p.frames >= 32 ? 0xffffffffu : ((1u << p.frames) - 1u);
    uint32_t done = 0u;  // bit f: frame f has converged; the same in every thread of the grid
    for (int it = 0; it < L; ++it) {
        // syndrome of the previous iteration's decisions, then check nodes (LogDecoder.java:82-95)
        for (long long item = gtid; item < cn_items; item += gsize) {
            const int f = (int)(item / mp), i = (int)(item - (long long)f * mp);
            if ((done >> f) & 1u) continue;
            if (p.early && it > 0) nodes_syndrome(p, f, i, p.synd_acc + (long long)f * L + (it - 1), lane);
            if (i < p.m) {
                const int e0 = p.row_ptr[i];
                const int d = p.row_ptr[i + 1] - e0;
                float* row = p.msg + (long long)f * p.E + e0;
                const float* lam = p.lam + (long long)f * p.n;
                if (it == 0) cn_dispatch<1, RULE, true>(d, row, lam, p.edge_col + e0, 1, 1, nullptr, 0, lane);
                else         cn_dispatch<1, RULE, false>(d, row, lam, p.edge_col + e0, 1, 1, nullptr, 0, lane);
            }
        }
        nodes_grid_barrier(nb);
        if (p.early && it > 0) {
            for (int f = 0; f < p.frames; ++f) {
                if (((done >> f) & 1u) == 0u && p.synd_acc[(long long)f * L + (it - 1)] == 0) {  // "sumSyndro == 0", :117
                    done |= 1u << f;
                    if (gtid == 0) {
                        if (p.out_iters) p.out_iters[f] = it;  // iteration index it-1, "l + 1"
                        if (p.out_synd) p.out_synd[f] = 0;
                    }
                }
            }
            if (done == all) break;
        }
        // variable nodes: next beta and the decisions of this iteration (:67-79, :98-105)
        const bool write_beta = p.early || it + 1 < L;
        for (long long item = gtid; item < vn_items; item += gsize) {
            const int f = (int)(item / np), j = (int)(item - (long long)f * np);
            if (((done >> f) & 1u) || j >= p.n) continue;
            const int c0 = p.col_ptr[j];
            const int d = p.col_ptr[j + 1] - c0;
            float g[1];
            vn_dispatch<1>(d, p.msg + (long long)f * p.E, p.lam + (long long)f * p.n + j, p.col_edge + c0, 1, write_beta, g,
                           nullptr, 0, lane);
            p.hard[(long long)f * p.n + j] = (g[0] < 0.0f) ? 1 : 0;  // "(gamma[j] < 0) ? 1 : 0", :104
        }
        nodes_grid_barrier(nb);
    }
    if (done != all) {  // frames that ran all iterations: weight of the last syndrome (:115, :122)
        for (long long item = gtid; item < cn_items; item += gsize) {
            const int f = (int)(item / mp), i = (int)(item - (long long)f * mp);
            if ((done >> f) & 1u) continue;
            nodes_syndrome(p, f, i, p.synd_acc + (long long)f * L + (L - 1), lane);
        }
        nodes_grid_barrier(nb);
        if (gtid == 0) {
            for (int f = 0; f < p.frames; ++f) {
                if ((done >> f) & 1u) continue;
                if (p.out_iters) p.out_iters[f] = L;
                if (p.out_synd) p.out_synd[f] = p.synd_acc[(long long)f * L + (L - 1)];
            }
        }
    }
    // a barrier timed out somewhere: nothing this launch wrote can be trusted, and a caller that only looks at its
    // own buffers must see that too
    if (gtid == 0 && __hip_atomic_load(&p.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        for (int f = 0; f < p.frames; ++f) {
            if (p.out_iters) p.out_iters[f] = -1;
            if (p.out_synd) p.out_synd[f] = -1;
        }
    }
}

#endif  // LDPC_NODES_KERNELS

struct NodesLaunch {
    int rule;
    int max_blocks;  // blocks of 256 threads that can be resident at once on the device (0: ask)
    int force_blocks = 0;  // tests only: launch exactly this many blocks, also when they cannot be resident
    hipStream_t stream;
};

// One cooperative launch; fails with hipErrorCooperativeLaunchTooLarge rather than run a grid that could not
// be resident as a whole.
hipError_t bp_nodes_launch(const NodesLaunch& l, const NodesParams& p);
// blocks of 256 threads of the widest kernel variant that fit on the current device at once
hipError_t bp_nodes_resident_blocks(int* blocks);
unsigned bp_nodes_params_bytes();  // see bp_kernels.hpp

}  // namespace ldpc
