// One decode on the device: which engine takes the batch, its workspace and its launches.
#include "host_ctx.hpp"

using namespace ldpc;

namespace ldpc {
namespace host {

// LLR unit of the selected check-node rule (see bp_kernels.hpp): bits for the default sum-product rule, nats otherwise.
double llr_scale(const ldpc_amd_ctx* c) {
    return (c->cn_rule == LDPC_AMD_CN_MINSUM || c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH) ? 1.0 : 1.4426950408889634;
}

// Bytes of streaming workspace one tile of T frames takes: messages, LLRs, packed decisions, and per frame 3 state
// words and 2 x kSyndParts syndrome accumulators.
size_t stream_tile_bytes(const ldpc_amd_ctx* c, int T) {
    return ((size_t)c->E + (size_t)c->n + (size_t)c->nw + 3 + 2 * kSyndParts) * (size_t)T * sizeof(float);
}

namespace {

// One decode as the engines below see it.
struct DecodeCall {
    const SrcArgs& src;
    long long frames;
    int max_iters;
    bool early;
    uint8_t* d_hard;
    int32_t* d_iters;
    int32_t* d_synd;
    hipStream_t s;
    CountArgs* count;
};

int kernel_rule(const ldpc_amd_ctx* c) {
    return c->cn_rule == LDPC_AMD_CN_MINSUM ? kCnMinSum : c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH ? kCnSumProduct : kCnComplement;
}

// The events that bracket the timed launches of a decode (ldpc_amd_last_decode_timing).
int ensure_events(ldpc_amd_ctx* c, size_t count) {
    while (c->ev.size() < count) {
        hipEvent_t e;
        LDPC_HIP(c, hipEventCreate(&e));
        c->ev.push_back(e);
    }
    return LDPC_AMD_OK;
}

int ensure_workspace(ldpc_amd_ctx* c, size_t need) {
    if (c->ws_bytes >= need) return LDPC_AMD_OK;
    if (c->ws) {
        LDPC_HIP(c, hipFree(c->ws));
        c->ws = nullptr;
        c->ws_bytes = 0;
    }
    LDPC_HIP(c, hipMalloc(&c->ws, need));
    c->ws_bytes = need;
    return LDPC_AMD_OK;
}

int launch_bp(ldpc_amd_ctx* c, const BpParams& p, int fpl, bool early, int wpb, size_t lds_bytes, bool pipelined,
              hipStream_t s) {
    BpLaunch l;
    l.fpl = fpl;
    l.rule = kernel_rule(c);
    l.early = early;
    l.wps = fpl == 4 ? 2 : 4;  // 4 frames per lane need ~180 VGPRs to stay out of scratch
    if (c->opt.wps) l.wps = c->opt.wps == 2 ? 2 : 4;
    l.grid = dim3((unsigned)((p.tiles + wpb - 1) / wpb));
    l.block = dim3((unsigned)(64 * wpb));
    l.lds_bytes = lds_bytes;
    l.stream = s;
    if (pipelined && c->reg_dc == 6 && c->reg_dv == 3) bp_launch_6_3(l, p);
    else if (pipelined && c->reg_dc == 8 && c->reg_dv == 2) bp_launch_8_2(l, p);
    else bp_launch_generic(l, p);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

// The runtime refused a cooperative launch of the node-parallel kernel: count it and keep the reason.
void note_nodes_refusal(ldpc_amd_ctx* c, hipError_t e, int blocks, bool retrying) {
    c->nodes_refusals++;
    c->nodes_last_error = (int)e;
    c->warn = std::string("cooperative launch of bp_nodes_kernel refused: ") + hipGetErrorName(e) + " (" +
              hipGetErrorString(e) + "), grid of " + std::to_string(blocks) + " blocks of 256 threads; " +
              (retrying ? "retrying with half the grid" : "the batch was decoded by the sweep launches");
    if (c->opt.verbose) fprintf(stderr, "ldpc_amd: %s\n", c->warn.c_str());
}

// Frame-resident engine: regular code whose messages fit at least twice into a CU's LDS.
// *taken: the batch has been launched on this engine.
int try_resident(ldpc_amd_ctx* c, const DecodeCall& d, bool* taken) {
    const SrcArgs& src = d.src;
    const long long frames = d.frames;
    const int max_iters = d.max_iters, n = c->n;
    const bool early = d.early;
    uint8_t* const d_hard = d.d_hard;
    int32_t* const d_iters = d.d_iters;
    int32_t* const d_synd = d.d_synd;
    hipStream_t s = d.s;
    CountArgs* const count = d.count;
    // a code one edge short of the (10,5) shape in places runs on the same kernels, completed with constant edges
    const bool def = c->reg_dc == 0 && c->near_dc == 10 && c->near_dv == 5 && c->cn_rule == LDPC_AMD_CN_SUMPRODUCT;
    const int dcr = def ? c->near_dc : c->reg_dc, dvr = def ? c->near_dv : c->reg_dv;
    const bool shape = (dcr == 6 && dvr == 3) || (dcr == 8 && dvr == 2) || (dcr == 8 && dvr == 4) || (dcr == 10 && dvr == 5);
    // vpt variables and vpt dv / dc checks per thread, whole wavefronts: 8 variables per thread, 4 for the (10,5)
    // kernels; the (6,3) kernels also exist with 12 (chosen below where it tiles better) and 16 (half the threads).  A code whose n is not vpt x 64 x
    // (wavefronts) is tiled as the next larger one with dummy nodes (build_resident_tables).
    int vpt = (dcr == 10) ? 4 : kLdsVpt;
    if (c->opt.lds_vpt == kLdsVptWide && dcr == 6 && n % (kLdsVptWide * 64) == 0) vpt = kLdsVptWide;
    if (dcr == 6 && c->opt.lds_vpt != kLdsVpt && c->opt.lds_vpt != kLdsVptWide) {
        // 12 variables per thread where that tiles the code in at most 4 wavefronts with less padding than 8 per
        // thread, or with the same and 8 per thread would make 3, 5, 6 or 7 wavefronts, which spread unevenly over
        // the 4 SIMDs (n = 3072: 6.59 -> 7.77 Gbit/s at fixed iterations, n = 2304: 5.67 -> 6.67, n = 1536: 7.41 -> 7.65)
        const long long w8 = ((long long)n + 511) / 512, w12 = ((long long)n + 767) / 768;
        const bool uneven8 = w8 == 3 || w8 == 5 || w8 == 6 || w8 == 7;
        if (w12 <= 4 && (w12 * 768 < w8 * 512 || (w12 * 768 == w8 * 512 && uneven8))) vpt = kLdsVptMid;
        if (c->opt.lds_vpt == kLdsVptMid && w12 <= 5) vpt = kLdsVptMid;  // measurements
    }
    const int nt = shape ? (int)(((long long)n + vpt * 64 - 1) / (vpt * 64)) * 64 : 64;
    const long long n2 = (long long)vpt * nt;
    const long long m2 = shape ? n2 * dvr / dcr : 0;
    const size_t nw2 = (size_t)(n2 / 32), mw2 = (size_t)((m2 + 31) / 32);
    // messages | packed decisions | accumulator | one syndrome bit per check
    const size_t msg_words = (size_t)(m2 / 2) * (size_t)lds_pair_stride(std::max(dcr, 2));
    // accounting inside the kernel: the default rule at up to 8 variables per thread; two sums per wavefront of the
    // frame in flight and the workgroup's counters (64-bit) follow the syndrome bits
    const bool counting = count && count->counters && c->opt.sim_fused && vpt <= kLdsVpt && max_iters <= 1024 &&
                          c->cn_rule != LDPC_AMD_CN_MINSUM && c->cn_rule != LDPC_AMD_CN_SUMPRODUCT_TANH;
    const size_t lds_bytes = (msg_words + nw2 + 4 + mw2 +
                              (counting ? 2 * (size_t)(4096 / 64 / vpt) + 2 * (size_t)(16 + max_iters) : 0)) * sizeof(float);
    bool use = shape && nt <= 4096 / vpt && lds_bytes <= 80 * 1024 && frames <= 0x7fffffffll &&
               (n2 * dvr) % dcr == 0 && m2 % 32 == 0 && n % 4 == 0 &&  // whole syndrome words, hard words stored four at a time
               (n2 == n || m2 - c->m >= dvr) &&
               m2 <= 2048;  // the convergence test reads the syndrome words with one wavefront
    if (c->opt.engine != LDPC_AMD_ENGINE_AUTO) use = false;
    if (use) {
        if (int rc = build_resident_tables(c, nt, vpt)) return rc;
        LdsParams q{};
        // m and nw of the tiled code, n of the caller's (the stride of its frames)
        q.m = (int)m2; q.n = n; q.nw = (int)nw2; q.E = n2 * dvr; q.msg_words = (int)msg_words;
        q.perm = c->d_res; q.pos_of_var = q.perm + n2; q.vn_off = q.pos_of_var + n2;
        q.llr = src.kind == Src::Float ? src.llr : nullptr;
        q.r = src.kind == Src::Bsc ? src.r : nullptr;
        q.l0 = (float)(src.l0 * llr_scale(c)); q.l1 = (float)(src.l1 * llr_scale(c)); q.g_length = n - c->m;
        q.scale = (float)llr_scale(c);
        q.frames = frames; q.max_iters = max_iters;
        q.hard = d_hard; q.out_iters = d_iters; q.out_synd = d_synd;
        if (counting) { q.cw = count->cw; q.counters = count->counters; q.count_rx = count->rx ? 1 : 0; }
        LdsLaunch l;
        l.count = counting;
        l.def = def;
        l.rule = kernel_rule(c);
        l.early = early; l.threads = nt; l.vpt = vpt; l.lds_bytes = lds_bytes; l.stream = s;
        // measured: 80 registers (six wavefronts per SIMD) pay off for the (6,3) kernels and the (10,5) kernels
        // (QC z = 108: 49.0 -> 45.0 ms per 262,144 frames); the (8,2) and (8,4) kernels lose more to register
        // spills than they gain in occupancy
        l.wpe = (dcr == 6 || dcr == 10) ? 6 : 4;
        if (c->opt.lds_wpe) l.wpe = c->opt.lds_wpe;
        // with early termination a workgroup takes several frames in turn, once there are enough of
        // them to keep every CU supplied with workgroups that finish at different times
        l.frames_per_wg = early ? (int)std::max<long long>(1, std::min<long long>(8, frames / 16384)) : 1;
        if (c->opt.lds_fpw) l.frames_per_wg = std::max(1, c->opt.lds_fpw);
        if (int rc = ensure_events(c, 2)) return rc;
        LDPC_HIP(c, hipEventRecord(c->ev[0], s));
        LDPC_HIP(c, dcr == 6 ? bp_lds_launch_6_3(l, q) : dcr == 10 ? bp_lds_launch_10_5(l, q)
                             : dvr == 4 ? bp_lds_launch_8_4(l, q) : bp_lds_launch_8_2(l, q));
        LDPC_HIP(c, hipEventRecord(c->ev[1], s));
        c->timed_launches = 1;
        c->timed_frames = frames;
        c->last_kernel = "bp_lds_kernel";
        if (counting) count->done = true;
        *taken = true;
    }
    return LDPC_AMD_OK;
}

// Generic frame-resident engine: any code whose messages and LLRs fit into a CU's LDS and whose
// degrees the register code covers.
// *taken: the batch has been launched on this engine.
int try_generic_resident(ldpc_amd_ctx* c, const DecodeCall& d, bool* taken) {
    const SrcArgs& src = d.src;
    const long long frames = d.frames;
    const int max_iters = d.max_iters, n = c->n;
    const bool early = d.early;
    uint8_t* const d_hard = d.d_hard;
    int32_t* const d_iters = d.d_iters;
    int32_t* const d_synd = d.d_synd;
    hipStream_t s = d.s;
    // at least 8 variables per thread, in 1, 2, 3, 4 or 8 whole wavefronts: workgroups of 5 to 7 wavefronts
    // spread unevenly over the 4 SIMDs and measured up to 1.5x slower (NR-BG1-shaped Z = 32, n = 2176, 65,536
    // frames x 50 iterations: 320 threads 78 ms, 384 threads 71 ms, 512 threads 53 ms, 256 threads 48 ms);
    // fewer threads than that mean more steps per phase, each a chain of dependent table reads
    int nt = 512;
    for (int cand : {64, 128, 192, 256, 512})
        if (cand >= (std::max(n, c->m) + 7) / 8) { nt = cand; break; }
    // up to 9 variables per thread rather than 512 threads, which fit only twice on a CU by their registers
    // (NR-BG1-shaped Z = 32, n = 2176, 65,536 frames x 50 iterations: 49.4 -> 45.6 ms; from n = 2720 on 512 threads win)
    if (nt == 512 && std::max(n, c->m) <= 9 * 256) nt = 256;
    if (c->opt.generic_threads) nt = c->opt.generic_threads;  // measurements
    bool use = c->m > 0 && c->max_dc <= kLdsGenMaxDc && c->max_dv <= kLdsGenMaxDv && frames <= 0x7fffffffll &&
               (size_t)c->E * sizeof(float) <= 160 * 1024;
    if (c->opt.engine == LDPC_AMD_ENGINE_STREAM) use = false;
    size_t lds_bytes = 0;
    int nwp = 0;
    if (use) {
        if (int rc = build_generic_resident_tables(c, nt)) return rc;  // also gives the padded slot count
        nwp = 2 * c->gen_vn_steps * nt / 32;
        // messages | LLRs | packed decisions | accumulator | one syndrome bit per check
        lds_bytes = ((size_t)c->gen_slots + (size_t)n + (size_t)nwp + 4 + (size_t)(c->m + 31) / 32) * sizeof(float);
        // up to the whole LDS of a CU: even at one workgroup per CU this engine measured 1.5x (fixed
        // iterations) to 3x (early exit) the streaming engine
        size_t lds_limit = 160 * 1024;
        if (c->opt.generic_lds_kb) lds_limit = c->opt.generic_lds_kb < 0 ? 0 : (size_t)c->opt.generic_lds_kb * 1024;  // measurements
        use = lds_bytes <= lds_limit && c->gen_vn_steps <= 16;  // 2 decision bits per step in one register
        if (c->opt.verbose)
            fprintf(stderr, "ldpc_amd: generic resident engine: %d threads per frame, %d + %d steps, %zu bytes of LDS\n", nt,
                    c->gen_cn_steps, c->gen_vn_steps, lds_bytes);
    }
    if (use) {
        LdsGenParams q{};
        q.m = c->m; q.n = n; q.nwp = nwp; q.E = c->E; q.slots = c->gen_slots;
        q.cn_steps = c->gen_cn_steps; q.vn_steps = c->gen_vn_steps;
        q.row_ptr = c->d_row_ptr; q.edge_col = c->d_edge_col;
        q.cn_desc = reinterpret_cast<const int4*>(c->d_genres);
        q.vn_desc = reinterpret_cast<const int4*>(c->d_genres + (size_t)4 * q.cn_steps * nt);
        q.pos_of_var = c->d_genres + (size_t)4 * q.cn_steps * nt + (size_t)8 * q.vn_steps * nt;
        q.edge_pos = q.pos_of_var + n;
        q.edge_slot = q.edge_pos + c->E;
        q.col_slot = q.edge_slot + c->E;
        q.col_check = q.col_slot + c->E;
        q.llr = src.kind == Src::Float ? src.llr : nullptr;
        q.r = src.kind == Src::Bsc ? src.r : nullptr;
        q.l0 = (float)(src.l0 * llr_scale(c)); q.l1 = (float)(src.l1 * llr_scale(c)); q.g_length = n - c->m;
        q.scale = (float)llr_scale(c);
        q.frames = frames; q.max_iters = max_iters;
        q.hard = d_hard; q.out_iters = d_iters; q.out_synd = d_synd;
        LdsGenLaunch l;
        l.rule = kernel_rule(c);
        l.early = early; l.threads = nt; l.lds_bytes = lds_bytes; l.stream = s;
        if (int rc = ensure_events(c, 2)) return rc;
        LDPC_HIP(c, hipEventRecord(c->ev[0], s));
        LDPC_HIP(c, bp_lds_gen_launch(l, q));
        LDPC_HIP(c, hipEventRecord(c->ev[1], s));
        c->timed_launches = 1;
        c->timed_frames = frames;
        c->last_kernel = "bp_lds_gen_kernel";
        *taken = true;
    }
    return LDPC_AMD_OK;
}

// Node-parallel cooperative engine: a batch of at most 32 frames of a code the resident engines do
// not take runs in one launch with one thread per node (a lane per frame would leave the chip empty).
// *taken: the batch has been launched on this engine.
int try_nodes(ldpc_amd_ctx* c, const DecodeCall& d, bool* taken) {
    const SrcArgs& src = d.src;
    const long long frames = d.frames;
    const int max_iters = d.max_iters, n = c->n;
    const bool early = d.early;
    uint8_t* const d_hard = d.d_hard;
    int32_t* const d_iters = d.d_iters;
    int32_t* const d_synd = d.d_synd;
    hipStream_t s = d.s;
    const bool can = frames <= kNodesMaxFrames && c->m > 0 && c->max_dc <= kNodesMaxDc && c->max_dv <= kNodesMaxDv &&
                     c->nodes_blocks >= 0 && c->opt.engine != LDPC_AMD_ENGINE_STREAM;
    bool use = can;
    // Per iteration this engine costs two grid barriers and the latency of two dependent phases (about
    // 14 us together) plus about 45 ns per 1000 edges of the batch (gathers of single words); the sweeps
    // cost two launches of short waves: about 60 us with the kernels specialised for a regular code, and
    // 70 us plus a round trip per variable of the largest degree otherwise (measured on an NR-BG1-sized
    // code and on a (3,6) code of n = 64800: break-even near 20 and near 5 frames).
    const bool specialised_rows = (c->reg_dc == 6 && c->reg_dv == 3) || (c->reg_dc == 8 && c->reg_dv == 2);
    const double sweeps_us = specialised_rows ? 60.0 : 70.0 + 0.9 * c->max_dv;
    use = use && 14.0 + 4.5e-5 * (double)frames * (double)c->E < sweeps_us;
    if (c->opt.nodes >= 0) use = can && c->opt.nodes != 0;  // 0: never, 1: whenever the batch and the degrees allow
    if (use && c->nodes_blocks == 0) {
        int coop = 0, blocks = 0;
        LDPC_HIP(c, hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device));
        if (coop) LDPC_HIP(c, bp_nodes_resident_blocks(&blocks));
        c->nodes_blocks = blocks > 0 ? blocks : -1;
        use = c->nodes_blocks > 0;
    }
    if (use) {
        const size_t F = (size_t)frames;
        const size_t need = (F * ((size_t)c->E + (size_t)n + (size_t)max_iters) + kNodesBarWords) * sizeof(float);
        if (int rc = ensure_workspace(c, need)) return rc;
        NodesParams q{};
        q.m = c->m; q.n = n; q.E = c->E;
        q.row_ptr = c->d_row_ptr; q.edge_col = c->d_edge_col; q.col_ptr = c->d_col_ptr; q.col_edge = c->d_col_edge;
        q.llr = src.kind == Src::Float ? src.llr : nullptr;
        q.r = src.kind == Src::Bsc ? src.r : nullptr;
        q.l0 = (float)(src.l0 * llr_scale(c)); q.l1 = (float)(src.l1 * llr_scale(c)); q.g_length = n - c->m;
        q.scale = (float)llr_scale(c);
        q.frames = (int)frames; q.max_iters = max_iters; q.early = early ? 1 : 0;
        q.flat_barrier = c->opt.nodes_flat ? 1 : 0;
        q.msg = (float*)c->ws;
        q.lam = q.msg + F * (size_t)c->E;
        q.synd_acc = (int32_t*)(q.lam + F * (size_t)n);
        q.bar = (unsigned*)(q.synd_acc + F * (size_t)max_iters);
        c->nodes_bar = q.bar;
        q.hard = d_hard; q.out_iters = d_iters; q.out_synd = d_synd;
        LDPC_HIP(c, hipMemsetAsync(q.synd_acc, 0, (F * (size_t)max_iters + kNodesBarWords) * sizeof(int32_t), s));
        // tests: the time-out flag already raised, as if a barrier of this launch had given up
        if (c->opt.test_nodes_timeout) LDPC_HIP(c, hipMemsetAsync(q.bar + 1, 1, 1, s));
        NodesLaunch l;
        l.rule = kernel_rule(c);
        // the grid-wide barrier grows with the blocks that meet in it, the phases shrink with the blocks they
        // are spread over: measured optimum at 128 blocks (flat between 128 and 256) from 1 to 32 frames
        l.max_blocks = std::min(c->nodes_blocks, 128); l.stream = s;
        if (c->opt.nodes_blocks)  // measurements: a smaller grid
            l.max_blocks = std::max(1, std::min(c->nodes_blocks, c->opt.nodes_blocks));
        // tests: a grid of four times what can be resident, which the runtime must refuse at launch
        if (c->opt.test_nodes_oversize) l.max_blocks = l.force_blocks = 4 * c->nodes_blocks;
        if (int rc = ensure_events(c, 2)) return rc;
        LDPC_HIP(c, hipEventRecord(c->ev[0], s));
        hipError_t le = bp_nodes_launch(l, q);
        if (le == hipErrorCooperativeLaunchTooLarge && !l.force_blocks && l.max_blocks > 1) {
            // the occupancy query counts an empty device: with other work resident fewer blocks fit.
            // Nothing has run; one more try with half the grid (the kernel strides over its items).
            (void)hipGetLastError();
            note_nodes_refusal(c, le, l.max_blocks, true);
            l.max_blocks /= 2;
            le = bp_nodes_launch(l, q);
        }
        if (le == hipSuccess) {
            LDPC_HIP(c, hipEventRecord(c->ev[1], s));
            c->timed_launches = 1;
            c->timed_frames = frames;
            c->last_kernel = "bp_nodes_kernel";
            c->nodes_refused_in_a_row = 0;
            *taken = true;
            return LDPC_AMD_OK;
        }
        // The runtime would not start the grid as a whole, so nothing has run: this batch goes through the
        // sweep launches below.  The refusal is counted and kept (ldpc_amd_get_option, ldpc_amd_last_warning);
        // after three in a row the context stops asking until LDPC_AMD_OPT_NODES is set again.
        (void)hipGetLastError();
        note_nodes_refusal(c, le, l.max_blocks, false);
        if (++c->nodes_refused_in_a_row >= 3) c->nodes_blocks = -1;
    }
    return LDPC_AMD_OK;
}

// Streaming engine: messages in HBM, frame-major tiles, one frame per lane (or four at fixed iterations).
// noalloc: run the sweep launches on the static scratch array and call nothing that allocates.
int run_stream(ldpc_amd_ctx* c, const DecodeCall& d, bool noalloc) {
    const SrcArgs& src = d.src;
    const long long frames = d.frames;
    const int max_iters = d.max_iters, n = c->n;
    const bool early = d.early;
    uint8_t* const d_hard = d.d_hard;
    int32_t* const d_iters = d.d_iters;
    int32_t* const d_synd = d.d_synd;
    hipStream_t s = d.s;
    const int maxdeg = std::max(c->max_dc, c->max_dv);
    if (maxdeg > kLdsMaxDegree) return fail(c, LDPC_AMD_ELIMIT, "node degree above 128 is not supported");
    int fpl = (!early && maxdeg <= kRegDegFpl4 && !noalloc) ? 4 : 1;
    // Four frames per lane move the most bytes per instruction, and make a quarter of the waves: worth it once those
    // waves fill the chip.  Below about 4,000 waves per sweep launch one frame per lane is faster, by up to 4.7x
    // (fixed 12 iterations, 4 -> 1: n = 64800, 64 frames 2.07 -> 0.44 ms, 256 frames 2.07 -> 1.58, 512 frames level;
    // (3,6) n = 4096, 4,096 frames 2.10 -> 1.63, 16,384 frames 8.05 against 8.32).
    if (fpl == 4 && c->opt.stream_kernel == 0 && !c->opt.layout) {
        const long long coarse = std::max(1, std::min(4096, std::min(c->m / 4, n / 32)));
        if ((frames + 255) / 256 * coarse < 4000) fpl = 1;
    }
    if (const int v = c->opt.fpl) {
        if (!early && maxdeg <= kRegDegFpl4 && (v == 1 || v == 2 || v == 4) && !noalloc) fpl = v;
    }
    const int T = 64 * fpl;
    // rows above the kernels' register limit are staged through LDS, one wave per block
    bool use_lds = maxdeg > kRegDegFpl1;
    int lds_maxd = use_lds ? maxdeg : 0;
    int wpb = use_lds ? 1 : 4;
    size_t lds_bytes = use_lds ? (size_t)2 * lds_maxd * 64 * sizeof(float) : 0;

    const long long tiles_total = (frames + T - 1) / T;
    // with the words of per-frame state of the row-parallel launches
    const size_t per_tile = stream_tile_bytes(c, T);
    unsigned long long limit = noalloc ? (unsigned long long)LDPC_AMD_NOALLOC_SCRATCH_BYTES : c->ws_limit;
    if (!limit) {
        size_t fr = 0, tot = 0;
        LDPC_HIP(c, hipMemGetInfo(&fr, &tot));
        limit = (unsigned long long)((double)(fr + c->ws_bytes) * 0.6);
    }
    long long max_tiles = (long long)(limit / per_tile);
    if (max_tiles < 1) max_tiles = 1;
    if (max_tiles > (1ll << 30)) max_tiles = 1ll << 30;
    // layout 0: [tile][row][T]  (a tile's rows are contiguous)
    // layout 1: [row][tile][T]  (the same row of neighbouring tiles is contiguous)
    const int layout = (c->opt.layout && !noalloc) ? 1 : 0;
    // Every sweep is its own launch over (tile, row chunk) pairs, cut as finely as the packed decision
    // words allow (32 variables per wave): short uniform waves walk through the tiles in address order,
    // a wave whose frames have all converged ends at once, and a batch of any size fills the chip.
    // LDPC_AMD_OPT_STREAM_KERNEL = 1 selects the other streaming kernel, one wave per tile for the whole decode.
    const bool by_rows = layout == 0 && fpl != 2 && (c->opt.stream_kernel == 0 || noalloc);
    if (by_rows) max_tiles = std::min<long long>(max_tiles, 65535);  // tiles are the grid's y dimension
    const long long rounds = (tiles_total + max_tiles - 1) / max_tiles;
    const long long tiles_per_round = (tiles_total + rounds - 1) / rounds;
    const size_t need = per_tile * (size_t)tiles_per_round;
    void* ws = nullptr;
    if (noalloc) {
        // the whole batch in one round inside the fixed array, or not at all
        if (rounds > 1 || need > (size_t)LDPC_AMD_NOALLOC_SCRATCH_BYTES || (long long)limit / (long long)per_tile < 1)
            return fail(c, LDPC_AMD_ELIMIT,
                        "decode_batch_dev_noalloc: " + std::to_string(frames) + " frames of this code need " +
                            std::to_string(need * (size_t)rounds) + " bytes of scratch, the static array holds " +
                            std::to_string((size_t)LDPC_AMD_NOALLOC_SCRATCH_BYTES) + " (at most " +
                            std::to_string((long long)(LDPC_AMD_NOALLOC_SCRATCH_BYTES / per_tile) * T) + " frames per call)");
        if (!c->scratch_ptr) return fail(c, LDPC_AMD_EHIP, "decode_batch_dev_noalloc: no scratch address for this device");
        ws = c->scratch_ptr;
    } else {
        if (int rc = ensure_workspace(c, need)) return rc;
        ws = c->ws;
    }
    c->last_kernel = "bp_decode_kernel";
    if (!noalloc)
        if (int rc = ensure_events(c, (size_t)(2 * std::min<long long>(rounds, 256)))) return rc;

    BpParams p{};
    p.m = c->m; p.n = n; p.nw = c->nw;
    p.row_ptr = c->d_row_ptr; p.edge_col = c->d_edge_col; p.col_ptr = c->d_col_ptr; p.col_edge = c->d_col_edge;
    if (layout == 0) {
        p.msg_row = p.lam_row = p.bits_row = T;
        p.msg_tile = c->E * T; p.lam_tile = (long long)n * T; p.bits_tile = (long long)c->nw * T;
    } else {
        p.msg_row = p.lam_row = p.bits_row = tiles_per_round * T;
        p.msg_tile = p.lam_tile = p.bits_tile = T;
    }
    float* ws_msg = (float*)ws;
    float* ws_lam = ws_msg + (size_t)tiles_per_round * c->E * T;
    uint32_t* ws_bits = (uint32_t*)(ws_lam + (size_t)tiles_per_round * n * T);
    int32_t* ws_state = (int32_t*)(ws_bits + (size_t)tiles_per_round * c->nw * T);
    p.ws_msg = ws_msg; p.ws_lam = ws_lam; p.ws_bits = ws_bits;
    p.frames_total = frames; p.max_iters = max_iters;
    p.out_iters = d_iters; p.out_synd = d_synd;
    p.lds_maxd = lds_maxd;
    if (c->d_leaf && c->opt.rows_leaves) { p.leaf_row = c->d_leaf; p.leaf_var = c->d_leaf + c->m; }  // used by the sweep launches only
    const bool pipelined = c->opt.nopipe == 0;  // A/B: per-row dispatch kernels
    int chunks = 1, rows_per_chunk = c->m, vars_per_chunk = (n + 31) / 32 * 32;
    bool fuse_syndrome = false;
    if (by_rows) {
        chunks = std::max(1, std::min(4096, std::min(c->m / 4, n / 32)));
        // A small batch has few tiles, and a wave's rows are a chain of dependent memory round trips (in the
        // any-degree kernels a node of degree above 12 is a round trip of its own): with fewer than `want`
        // waves in a launch the rows of such a code are cut finer, down to 2 checks and 8 variables per wave
        // (NR-BG1-sized code, 64 frames: 119 -> 43 us per iteration, 512 frames: 306 -> 234; from 2048 frames
        // on the coarse cut is as good or better).  Eight variables share a byte of the packed decisions,
        // which such a wave stores as a byte; cuts between 17 and 31 variables measured far slower than
        // either neighbour and are not used.  The kernels specialised for a regular code, which issue the loads
        // of several rows together, had lost by a finer cut while every wave of a launch queued its syndrome atomic on
        // one frame's accumulator; with the accumulators in parts (RowsParams::synd_parts) they gain like the others
        // ((3,6) n = 4096 through the sweeps, early termination: 64 frames 0.78 -> 0.38 ms, 1024 frames 1.10 -> 0.92;
        // n = 64800: 64 frames 1.14 -> 0.98; never slower up to 65,536 frames).
        long long want = 32768;
        if (c->opt.rows_waves) want = std::max(256, c->opt.rows_waves);  // measurements: waves per launch
        if (c->opt.rows_waves && (long long)chunks * tiles_per_round > want)
            chunks = (int)std::max<long long>(1, (want + tiles_per_round - 1) / tiles_per_round);
        // Early termination: the syndrome rides in the check-node launch of the next iteration (two launches per
        // iteration instead of four; a converged frame is recognised one check-node sweep later).  Faster than
        // launches of its own at every batch size measured, for every kind of code (NR-BG1-sized code -6 to -8 %
        // from 256 to 65,536 frames, DVB-S2-shaped code -11 %, (3,6) codes -8 %); before the sweeps formed their row
        // addresses on the scalar unit it had cost the large batches of irregular codes 4.5 %.
        fuse_syndrome = early && fpl == 1;
        if (c->opt.rows_fused >= 0) fuse_syndrome = early && fpl == 1 && c->opt.rows_fused != 0;
        if (fpl == 1 && (long long)chunks * tiles_per_round < want) {
            const long long finest = std::max(1, std::min(c->m / 2, n / 8));
            const long long finer = std::min<long long>((want + tiles_per_round - 1) / tiles_per_round, finest);
            if (finer >= (n + 15) / 16) chunks = (int)std::max<long long>(chunks, finer);
        }
        rows_per_chunk = (c->m + chunks - 1) / chunks;
        const int unit = (fpl == 1 && chunks > n / 32) ? 8 : 32;  // variables that share a stored byte / word
        vars_per_chunk = ((n + chunks - 1) / chunks + unit - 1) / unit * unit;
        p.bits_bytes = unit == 8 ? 1 : 0;
        c->last_kernel = "rows_cn_kernel";
        use_lds = c->max_dc > kRowsRegDegCn || c->max_dv > kRowsRegDegVn;
        lds_maxd = use_lds ? maxdeg : 0;
        wpb = use_lds ? 1 : 4;
        lds_bytes = use_lds ? (size_t)2 * lds_maxd * 64 * sizeof(float) : 0;
        p.lds_maxd = lds_maxd;
    }

    for (long long r = 0; r < rounds; ++r) {
        const long long tile0 = r * tiles_per_round;
        const int tiles = (int)std::min<long long>(tiles_per_round, tiles_total - tile0);
        const long long frame_base = tile0 * T;
        const long long frames_here = std::min<long long>((long long)tiles * T, frames - frame_base);
        p.tiles = tiles;
        p.frame_base = frame_base;
        if (int rc = load_lambda(c, src, p, ws_lam, T, frames, s)) return rc;
        const bool timed = r < 256 && !noalloc;  // the no-allocation path records nothing on the caller's stream but its work
        if (timed) LDPC_HIP(c, hipEventRecord(c->ev[(size_t)(2 * r)], s));
        if (by_rows) {
            RowsParams q{};
            q.b = p;
            q.rows_per_chunk = rows_per_chunk;
            q.vars_per_chunk = vars_per_chunk;
            q.early = early ? 1 : 0;
            q.fused = fuse_syndrome ? 1 : 0;
            const size_t slots = (size_t)tiles * T;
            q.st_done = ws_state; q.st_iters = ws_state + slots; q.st_synd = ws_state + 2 * slots; q.st_acc = ws_state + 3 * slots;
            // few tiles: every wave of a launch adds to the same few accumulator lines (NR-BG1-sized code, 64 frames,
            // early termination: check-node launch 72 -> 26 us; (3,6) n = 64800, 64 frames: 1.57 -> 1.14 ms per decode)
            q.synd_parts = tiles <= 32 ? kSyndParts : 1;
            LDPC_HIP(c, hipMemsetAsync(ws_state, 0, (3 + 2 * (size_t)q.synd_parts) * slots * sizeof(int32_t), s));
            RowsLaunch l;
            l.fpl = fpl == 4 ? 4 : 1;
            l.dc = c->reg_dc; l.dv = c->reg_dv;  // 0 for an irregular code
            l.rule = kernel_rule(c);
            l.chunks = chunks; l.wpb = wpb; l.lds_bytes = lds_bytes; l.stream = s;
            LDPC_HIP(c, bp_rows_run(l, q));
        } else if (int rc = launch_bp(c, p, fpl, early, wpb, lds_bytes, pipelined, s)) {
            return rc;
        }
        if (timed) {
            LDPC_HIP(c, hipEventRecord(c->ev[(size_t)(2 * r + 1)], s));
            c->timed_launches++;
            c->timed_frames += frames_here;
        }
        if (int rc = unpack_hard(c, p, ws_bits, T, frames_here, d_hard, s)) return rc;
    }
    return LDPC_AMD_OK;
}

}  // namespace

int decode_dev(ldpc_amd_ctx* c, const SrcArgs& src, long long frames, int max_iters, int early_term, uint8_t* d_hard,
               int32_t* d_iters, int32_t* d_synd, hipStream_t s, bool noalloc, CountArgs* count) {
    if (frames < 0 || max_iters < 0 || !d_hard) return fail(c, LDPC_AMD_EINVAL, "decode: bad argument");
    c->timed_launches = 0;
    c->timed_frames = 0;
    if (frames == 0) return LDPC_AMD_OK;
    if (max_iters == 0) {  // maxL = 0: all-zero word, 0 iterations, syndrome 0 (LogDecoder.java:61, :122)
        LDPC_HIP(c, hipMemsetAsync(d_hard, 0, (size_t)frames * c->n, s));
        if (d_iters) LDPC_HIP(c, hipMemsetAsync(d_iters, 0, (size_t)frames * 4, s));
        if (d_synd) LDPC_HIP(c, hipMemsetAsync(d_synd, 0, (size_t)frames * 4, s));
        return LDPC_AMD_OK;
    }
    const DecodeCall d{src, frames, max_iters, early_term != 0, d_hard, d_iters, d_synd, s, count};
    if (!noalloc) {  // the engines that allocate tables, events or workspace, in the order of preference
        bool taken = false;
        for (auto engine : {try_resident, try_generic_resident, try_nodes}) {
            if (int rc = engine(c, d, &taken)) return rc;
            if (taken) return LDPC_AMD_OK;
        }
    }
    return run_stream(c, d, noalloc);
}

// After the stream has been synchronised: did a grid barrier of the last node-parallel launch time out?
int check_nodes_barrier(ldpc_amd_ctx* c) {
    if (!c->nodes_bar || !c->timed_launches || strcmp(c->last_kernel, "bp_nodes_kernel")) return LDPC_AMD_OK;
    unsigned flag = 0;
    LDPC_HIP(c, hipMemcpy(&flag, c->nodes_bar + 1, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) return fail(c, LDPC_AMD_EHIP, "bp_nodes_kernel: a grid barrier timed out; the results of that decode are invalid");
    return LDPC_AMD_OK;
}

}  // namespace host
}  // namespace ldpc
