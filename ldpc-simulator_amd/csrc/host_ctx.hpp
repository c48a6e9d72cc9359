// Host side of the library, shared by its translation units: the context behind the opaque handle of
// include/ldpc_amd.h, its per-context switches, and the functions the units call in each other.
//   host_tables.hip   Tanner graph (A(i) / B(j)), tables of the frame-resident engines
//   host_decode.hip   engine choice and launch sequencing of one decode
//   host_io.hip       channel, codeword and accounting launches
//   ldpc_amd.hip      the C ABI
#pragma once
#include "../../include/ldpc_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "bp_kernels.hpp"
#include "bp_lds.hpp"
#include "bp_lds_gen.hpp"
#include "bp_nodes.hpp"
#include "bp_rows.hpp"

namespace ldpc {
namespace host {

// Per-context switches (ldpc_amd_set_option); 0 in a tuning field = the library's own choice.
struct Options {
    int engine = LDPC_AMD_ENGINE_AUTO;
    int stream_kernel = 0;
    int nodes = -1;
    int sim_overlap = 1;
    int sim_fused = 1;
    int rows_leaves = 1;
    int sim_chunk_mb = 0;
    int verbose = 0;
    int lds_wpe = 0, lds_fpw = 0, generic_lds_kb = 0, nodes_blocks = 0, fpl = 0, layout = 0, rows_waves = 0, wps = 0,
        nopipe = 0, rows_fused = -1, generic_threads = 0, lds_vpt = 0;
    int test_nodes_oversize = 0, test_nodes_timeout = 0, nodes_flat = 0;
    int small_call_kb = 0;
};

}  // namespace host
}  // namespace ldpc

struct ldpc_amd_ctx {
    int device = 0;
    ldpc::host::Options opt;
    int m = 0, n = 0, nw = 0;
    long long E = 0;
    int max_dc = 0, max_dv = 0;
    int reg_dc = 0, reg_dv = 0;  // the common degree when every check / variable has it, else 0
    // one edge short of regular in places: every check has near_dc or near_dc - 1 edges, every variable near_dv or
    // near_dv - 1, as many short checks as short variables (a quasi-cyclic code with empty blocks in distinct rows
    // and columns of its base matrix); else 0
    int near_dc = 0, near_dv = 0;
    int cn_rule = LDPC_AMD_CN_SUMPRODUCT;
    uint8_t* d_leaf = nullptr;  // leaf_row[m] | leaf_var[n] (BpParams), null when the code has no leaf to mark
    int* d_row_ptr = nullptr;
    int* d_edge_col = nullptr;
    int* d_col_ptr = nullptr;
    int* d_col_edge = nullptr;
    std::vector<int> h_row_ptr, h_edge_col, h_col_ptr, h_col_edge;
    // decode workspace: [tiles][E][T] messages | [tiles][n][T] lambda | [tiles][nw][T] decisions
    void* ws = nullptr;
    size_t ws_bytes = 0;
    void* scratch_ptr = nullptr;  // this device's copy of the static scratch array (see g_scratch_ptr)
    unsigned long long ws_limit = 0;
    // staging for the host-pointer entry points and for simulate
    // (slots 4..7 are the second set of the double-buffered host-pointer path)
    void* stage[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // page-locked host memory the device reads and writes in place: small host-pointer calls (LDPC_AMD_OPT_TUNE_SMALL_CALL_KB)
    void* small_host = nullptr;
    void* small_dev = nullptr;  // the same memory as the device addresses it
    size_t small_bytes = 0;
    hipStream_t io_stream[2] = {nullptr, nullptr};
    hipEvent_t io_decoded[2] = {nullptr, nullptr};
    hipEvent_t sim_enter = nullptr;  // ldpc_amd_simulate_dev: the caller's stream at the time of the call
    hipEvent_t sim_generated[2] = {nullptr, nullptr}, sim_decoded[2] = {nullptr, nullptr};  // simulate: a buffer set is full / free again
    hipEvent_t sim_begin = nullptr;                                      // simulate: the decode stream at the start of a leg
    hipStream_t sim_cw_stream = nullptr;                                 // simulate: random codewords, two chunks ahead
    hipEvent_t sim_cw_done[3] = {nullptr, nullptr, nullptr}, sim_cw_free[3] = {nullptr, nullptr, nullptr};
    unsigned long long* d_counters = nullptr;
    size_t counters_len = 0;
    // generator matrix for random codewords, by columns: n columns of ceil(k / 64) words, row i at bit i & 63
    // of word i >> 6 (null = all-zero word)
    int gen_k = 0;
    unsigned long long* d_gen = nullptr;
    void* stage_cw = nullptr;
    size_t stage_cw_bytes = 0;
    // tables of the frame-resident engine (built on first use, see build_resident_tables)
    int res_vpt = 0;
    int res_nt = 0;
    int* d_res = nullptr;  // perm[n] | pos_of_var[n] | vn_off[E]
    double res_conflict = 0.0;  // mean LDS bank multiplicity of the VN gathers (1 = conflict free)
    int res_clean_groups = 0;   // half wavefronts whose gathers are all conflict-free
    // tables of the generic frame-resident engine: cn_desc | vn_desc | pos_of_var[n] | edge_pos[E] | edge_slot[E] | col_slot[E] | col_check[E]
    int gen_nt = 0, gen_cn_steps = 0, gen_vn_steps = 0, gen_slots = 0;
    // node-parallel cooperative engine: blocks that can be resident at once (0: not asked yet, -1: not available)
    int nodes_blocks = 0;
    long long nodes_refusals = 0;   // cooperative launches the runtime refused
    int nodes_refused_in_a_row = 0;
    int nodes_last_error = 0;       // hipError_t of the last refusal
    unsigned* nodes_bar = nullptr;  // barrier words of the last node-parallel launch (bar[1]: time-out flag)
    int* d_genres = nullptr;
    // timing of the last decode
    std::vector<hipEvent_t> ev;
    int timed_launches = 0;
    long long timed_frames = 0;
    const char* last_kernel = "";  // which message-passing kernel the last decode ran
    std::string err;
    std::string warn;  // last event that did not fail a call (ldpc_amd_last_warning)
};

namespace ldpc {
namespace host {

// Records the message on the context (or, without one, for ldpc_amd_last_error(NULL)) and returns the code.
int fail(ldpc_amd_ctx* ctx, int code, const std::string& msg);

#define LDPC_HIP(ctx, call)                                                                          \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(ctx, e_ == hipErrorOutOfMemory ? LDPC_AMD_ENOMEM : LDPC_AMD_EHIP,            \
                        std::string(#call) + ": " + hipGetErrorString(e_));                          \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess) ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

constexpr int kLdsMaxDegree = 128;  // 2 arrays x 128 rows x 256 B = 64 KiB per wave

enum class Src { Float, Bsc };

struct SrcArgs {
    Src kind;
    const float* llr;
    const uint8_t* r;
    double l0, l1;  // BSC magnitudes in nats, double as the reference forms them
};
// Error accounting asked of the decode launch itself (ldpc_amd_simulate): the engines that can do it set `done`
// and write no hard words, iteration counts or syndrome weights; the others ignore it.
struct CountArgs {
    const uint8_t* cw;             // transmitted words, null = the all-zero word
    unsigned long long* counters;  // device, 16 + max_iters
    bool rx;                       // channel-side counters too
    bool done;
};

// host_tables.hip
void build_graph(ldpc_amd_ctx* c, long long nnz, const int32_t* rows, const int32_t* cols);
int build_resident_tables(ldpc_amd_ctx* c, int nt, int vpt);
int build_generic_resident_tables(ldpc_amd_ctx* c, int nt);

// host_decode.hip
double llr_scale(const ldpc_amd_ctx* c);
size_t stream_tile_bytes(const ldpc_amd_ctx* c, int T);
int decode_dev(ldpc_amd_ctx* c, const SrcArgs& src, long long frames, int max_iters, int early_term, uint8_t* d_hard,
               int32_t* d_iters, int32_t* d_synd, hipStream_t s, bool noalloc = false, CountArgs* count = nullptr);
int check_nodes_barrier(ldpc_amd_ctx* c);

// host_io.hip (the only unit that contains the kernels of io_kernels.hpp)
// Channel LLRs of a round of tiles into the frame-major workspace of the streaming engine.
int load_lambda(ldpc_amd_ctx* c, const SrcArgs& src, const BpParams& p, float* ws_lam, int T, long long frames, hipStream_t s);
// Packed decisions of a round of tiles into the caller's byte-per-bit words.
int unpack_hard(ldpc_amd_ctx* c, const BpParams& p, const uint32_t* ws_bits, int T, long long frames_here, uint8_t* d_hard,
                hipStream_t s);
// Whether gen_codewords_kernel can hold a generator matrix of k rows.
bool generator_fits(int k);
int bsc_magnitudes(ldpc_amd_ctx* c, double e, int llr_mode, double* l0, double* l1);
uint32_t bsc_threshold(double e);
int count_launch(ldpc_amd_ctx* c, long long frames, const uint8_t* d_hard, const uint8_t* d_cw, const uint8_t* d_rx,
                 const float* d_rx_llr, const int32_t* d_iters, const int32_t* d_synd, int max_iters,
                 unsigned long long* d_counters, hipStream_t s);
int gen_codewords(ldpc_amd_ctx* c, uint64_t seed, long long first_frame, long long frames, uint8_t* d_cw, hipStream_t s);
int gen_bsc(ldpc_amd_ctx* c, int channel, double e, uint64_t seed, long long first_frame, long long frames,
            const uint8_t* d_cw, uint8_t* d_r, hipStream_t s);
int gen_awgn(ldpc_amd_ctx* c, double ebn0_db, uint64_t seed, long long first_frame, long long frames,
             const uint8_t* d_cw, float* d_llr, hipStream_t s);

}  // namespace host
}  // namespace ldpc
