// C ABI of the gfx950 belief-propagation decoder (declared in include/ldpc_amd.h).
// Host side: Tanner-graph construction, workspace and launch sequencing.  No CPU decode
// path exists in this library: without a HIP device every compute entry point fails.
#include "host_ctx.hpp"

using namespace ldpc;
using namespace ldpc::host;

// Scratch memory of ldpc_amd_decode_batch_dev_noalloc: the frame-major message, LLR, decision and
// per-frame state arrays of the sweep kernels, which every other entry point keeps in a hipMalloc'd
// workspace.  One copy per GPU the library runs on, held until the process ends; host code reaches it only
// through the address hipGetSymbolAddress returns for the current device (g_scratch_ptr), never by name.
static __device__ __align__(256) unsigned char scratch[LDPC_AMD_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices];  // [device] address of that device's copy of `scratch`, set by ldpc_amd_create

std::mutex g_err_mutex;
std::string g_create_err;

}  // namespace

int ldpc::host::fail(ldpc_amd_ctx* ctx, int code, const std::string& msg) {
    if (ctx) {
        ctx->err = msg;
    } else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_create_err = msg;
    }
    return code;
}

namespace {

int ensure_stage(ldpc_amd_ctx* c, int slot, size_t bytes) {
    if (c->stage_bytes[slot] >= bytes) return LDPC_AMD_OK;
    if (c->stage[slot]) {
        LDPC_HIP(c, hipFree(c->stage[slot]));
        c->stage[slot] = nullptr;
        c->stage_bytes[slot] = 0;
    }
    LDPC_HIP(c, hipMalloc(&c->stage[slot], bytes));
    c->stage_bytes[slot] = bytes;
    return LDPC_AMD_OK;
}

}  // namespace

extern "C" {

int ldpc_amd_version(void) { return LDPC_AMD_VERSION; }

const char* ldpc_amd_last_error(const ldpc_amd_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mutex);
    static thread_local std::string copy;
    copy = g_create_err;
    return copy.c_str();
}

int ldpc_amd_create(int32_t m, int32_t n, int64_t nnz, const int32_t* row_idx, const int32_t* col_idx,
                    int32_t device_id, ldpc_amd_ctx** out) {
    if (!out) return fail(nullptr, LDPC_AMD_EINVAL, "create: out is null");
    *out = nullptr;
    if (m < 0 || n <= 0 || nnz < 0 || (nnz > 0 && (!row_idx || !col_idx)))
        return fail(nullptr, LDPC_AMD_EINVAL, "create: bad matrix shape or null index arrays");
    // every kernel object must have been compiled with the parameter structs this file uses
    if (bp_generic_params_bytes() != sizeof(BpParams) || bp_6_3_params_bytes() != sizeof(BpParams) ||
        bp_8_2_params_bytes() != sizeof(BpParams) || bp_lds_6_3_params_bytes() != sizeof(LdsParams) ||
        bp_lds_8_2_params_bytes() != sizeof(LdsParams) || bp_lds_8_4_params_bytes() != sizeof(LdsParams) ||
        bp_lds_10_5_params_bytes() != sizeof(LdsParams) || bp_lds_gen_params_bytes() != sizeof(LdsGenParams) ||
        bp_rows_params_bytes() != sizeof(RowsParams) || bp_nodes_params_bytes() != sizeof(NodesParams))
        return fail(nullptr, LDPC_AMD_EHIP, "create: the library was linked from objects built against different headers");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LDPC_AMD_ENODEV, "create: no HIP device (this library has no CPU path)");
    int dev = device_id;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(nullptr, LDPC_AMD_ENODEV, "create: hipGetDevice failed");
    if (dev >= ndev) return fail(nullptr, LDPC_AMD_EINVAL, "create: device_id out of range");
    ldpc_amd_ctx* c = new (std::nothrow) ldpc_amd_ctx();
    if (!c) return fail(nullptr, LDPC_AMD_ENOMEM, "create: out of host memory");
    c->device = dev;
    if (const char* env = getenv("LDPC_AMD_VERBOSE")) c->opt.verbose = atoi(env) != 0 || env[0] == '\0' ? 1 : 0;
    c->m = m; c->n = n; c->nw = (n + 31) / 32;
    build_graph(c, nnz, row_idx, col_idx);
    DeviceGuard guard(dev);
    auto bail = [&](int code, const std::string& msg) {
        fail(nullptr, code, msg);
        ldpc_amd_destroy(c);
        return code;
    };
    if (!guard.ok) return bail(LDPC_AMD_EHIP, "create: hipSetDevice failed");
    // Address of this device's copy of the static scratch array, asked for once per device with that device
    // current; and the two timing events of a one-round decode, so that ldpc_amd_decode_batch_dev_noalloc
    // finds everything it needs in place.
    if (dev >= kMaxDevices) return bail(LDPC_AMD_EINVAL, "create: device_id above the library's table of devices");
    {
        std::lock_guard<std::mutex> lk(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void* ptr = nullptr;
            const hipError_t se = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (se != hipSuccess || !ptr)
                return bail(LDPC_AMD_EHIP, std::string("create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(se));
            g_scratch_ptr[dev] = ptr;
        }
        c->scratch_ptr = g_scratch_ptr[dev];
    }
    for (int i = 0; i < 2; ++i) {
        hipEvent_t ev;
        const hipError_t ee = hipEventCreate(&ev);
        if (ee != hipSuccess) return bail(LDPC_AMD_EHIP, std::string("create: hipEventCreate: ") + hipGetErrorString(ee));
        c->ev.push_back(ev);
    }
    const size_t eb = (size_t)std::max<long long>(c->E, 1) * sizeof(int);
    hipError_t e;
    if ((e = hipMalloc((void**)&c->d_row_ptr, ((size_t)m + 1) * sizeof(int))) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_col_ptr, ((size_t)n + 1) * sizeof(int))) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_edge_col, eb)) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_col_edge, eb)) != hipSuccess)
        return bail(LDPC_AMD_ENOMEM, std::string("create: hipMalloc: ") + hipGetErrorString(e));
    if ((e = hipMemcpy(c->d_row_ptr, c->h_row_ptr.data(), ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(c->d_col_ptr, c->h_col_ptr.data(), ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (c->E && (e = hipMemcpy(c->d_edge_col, c->h_edge_col.data(), (size_t)c->E * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) ||
        (c->E && (e = hipMemcpy(c->d_col_edge, c->h_col_edge.data(), (size_t)c->E * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess))
        return bail(LDPC_AMD_EHIP, std::string("create: hipMemcpy: ") + hipGetErrorString(e));
    {
        // leaves the sweeps may leave alone (BpParams::leaf_row): variables of degree 1 whose edge is the last of a
        // check that every kernel keeps in registers
        std::vector<uint8_t> leaf((size_t)m + (size_t)n, 0);
        size_t marked = 0;
        for (int i = 0; i < m; ++i) {
            const int e0 = c->h_row_ptr[(size_t)i], e1 = c->h_row_ptr[(size_t)i + 1];
            if (e1 - e0 < 2 || e1 - e0 > 16) continue;
            const int j = c->h_edge_col[(size_t)e1 - 1];
            if (c->h_col_ptr[(size_t)j + 1] - c->h_col_ptr[(size_t)j] != 1) continue;
            leaf[(size_t)i] = 1;
            leaf[(size_t)m + (size_t)j] = 1;
            ++marked;
        }
        if (marked) {
            if ((e = hipMalloc((void**)&c->d_leaf, leaf.size())) != hipSuccess)
                return bail(LDPC_AMD_ENOMEM, std::string("create: hipMalloc: ") + hipGetErrorString(e));
            if ((e = hipMemcpy(c->d_leaf, leaf.data(), leaf.size(), hipMemcpyHostToDevice)) != hipSuccess)
                return bail(LDPC_AMD_EHIP, std::string("create: hipMemcpy: ") + hipGetErrorString(e));
        }
        if (c->opt.verbose) fprintf(stderr, "ldpc_amd: %zu leaves marked\n", marked);
    }
    *out = c;
    return LDPC_AMD_OK;
}

int ldpc_amd_destroy(ldpc_amd_ctx* c) {
    if (!c) return LDPC_AMD_OK;
    {
        DeviceGuard guard(c->device);
        (void)hipDeviceSynchronize();
        for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
        void* ptrs[] = {c->d_row_ptr, c->d_edge_col, c->d_col_ptr, c->d_col_edge, c->ws, c->d_res,
                        c->stage[0],  c->stage[1],   c->stage[2],  c->stage[3],   c->d_counters, c->d_gen,      c->stage_cw,
                        c->stage[4],  c->stage[5],   c->stage[6],  c->stage[7],   c->d_genres,   c->d_leaf};
        for (int i = 0; i < 2; ++i) {
            if (c->io_stream[i]) (void)hipStreamDestroy(c->io_stream[i]);
            if (c->io_decoded[i]) (void)hipEventDestroy(c->io_decoded[i]);
        }
        if (c->sim_enter) (void)hipEventDestroy(c->sim_enter);
        for (int i = 0; i < 2; ++i) {
            if (c->sim_generated[i]) (void)hipEventDestroy(c->sim_generated[i]);
            if (c->sim_decoded[i]) (void)hipEventDestroy(c->sim_decoded[i]);
        }
        for (int i = 0; i < 3; ++i) {
            if (c->sim_cw_done[i]) (void)hipEventDestroy(c->sim_cw_done[i]);
            if (c->sim_cw_free[i]) (void)hipEventDestroy(c->sim_cw_free[i]);
        }
        if (c->sim_cw_stream) (void)hipStreamDestroy(c->sim_cw_stream);
        if (c->sim_begin) (void)hipEventDestroy(c->sim_begin);
        if (c->small_host) (void)hipHostFree(c->small_host);
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
    delete c;
    return LDPC_AMD_OK;
}

int ldpc_amd_info(const ldpc_amd_ctx* c, int32_t* m, int32_t* n, int64_t* edges, int32_t* max_dc, int32_t* max_dv) {
    if (!c) return LDPC_AMD_EINVAL;
    if (m) *m = c->m;
    if (n) *n = c->n;
    if (edges) *edges = c->E;
    if (max_dc) *max_dc = c->max_dc;
    if (max_dv) *max_dv = c->max_dv;
    return LDPC_AMD_OK;
}

int ldpc_amd_graph(const ldpc_amd_ctx* c, int32_t* row_ptr, int32_t* edge_col, int32_t* col_ptr, int32_t* col_edge) {
    if (!c) return LDPC_AMD_EINVAL;
    if (row_ptr) memcpy(row_ptr, c->h_row_ptr.data(), c->h_row_ptr.size() * sizeof(int));
    if (edge_col) memcpy(edge_col, c->h_edge_col.data(), c->h_edge_col.size() * sizeof(int));
    if (col_ptr) memcpy(col_ptr, c->h_col_ptr.data(), c->h_col_ptr.size() * sizeof(int));
    if (col_edge) memcpy(col_edge, c->h_col_edge.data(), c->h_col_edge.size() * sizeof(int));
    return LDPC_AMD_OK;
}

int ldpc_amd_set_cn_rule(ldpc_amd_ctx* c, int32_t cn_rule) {
    if (!c) return LDPC_AMD_EINVAL;
    if (cn_rule != LDPC_AMD_CN_SUMPRODUCT && cn_rule != LDPC_AMD_CN_MINSUM && cn_rule != LDPC_AMD_CN_SUMPRODUCT_TANH)
        return fail(c, LDPC_AMD_EINVAL, "unknown check-node rule");
    c->cn_rule = cn_rule;
    return LDPC_AMD_OK;
}

int ldpc_amd_set_workspace_limit(ldpc_amd_ctx* c, uint64_t bytes) {
    if (!c) return LDPC_AMD_EINVAL;
    c->ws_limit = bytes;
    return LDPC_AMD_OK;
}

int ldpc_amd_set_option(ldpc_amd_ctx* c, int32_t key, int64_t value) {
    if (!c) return LDPC_AMD_EINVAL;
    auto range = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
    Options& o = c->opt;
    const int v = (int)value;
    switch (key) {
        case LDPC_AMD_OPT_ENGINE: if (!range(0, 2)) break; o.engine = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_STREAM_KERNEL: if (!range(0, 1)) break; o.stream_kernel = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_NODES:
            if (!range(-1, 1)) break;
            o.nodes = v;
            if (c->nodes_blocks < 0) c->nodes_blocks = 0;  // ask the runtime again
            c->nodes_refused_in_a_row = 0;
            return LDPC_AMD_OK;
        case LDPC_AMD_OPT_SIM_OVERLAP: if (!range(0, 1)) break; o.sim_overlap = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_SIM_FUSED: if (!range(0, 1)) break; o.sim_fused = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_VERBOSE: if (!range(0, 1)) break; o.verbose = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_LDS_WPE: if (!(value == 0 || value == 4 || value == 6)) break; o.lds_wpe = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_LDS_FPW: if (!range(0, 1 << 20)) break; o.lds_fpw = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_GENERIC_LDS_KB: if (!range(-1, 160)) break; o.generic_lds_kb = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_NODES_BLOCKS: if (!range(0, 1 << 20)) break; o.nodes_blocks = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_FPL: if (!(value == 0 || value == 1 || value == 2 || value == 4)) break; o.fpl = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_LAYOUT: if (!range(0, 1)) break; o.layout = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_ROWS_WAVES: if (!range(0, 1 << 30)) break; o.rows_waves = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_WPS: if (!(value == 0 || value == 2 || value == 4)) break; o.wps = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_NOPIPE: if (!range(0, 1)) break; o.nopipe = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_ROWS_FUSED: if (!range(-1, 1)) break; o.rows_fused = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_ROWS_LEAVES: if (!range(0, 1)) break; o.rows_leaves = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_SIM_CHUNK_MB: if (value != 0 && !range(16, 65536)) break; o.sim_chunk_mb = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_LDS_VPT: if (!(value == 0 || value == 8 || value == 12 || value == 16)) break; o.lds_vpt = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_GENERIC_THREADS: if (!(value == 0 || (value >= 64 && value <= 512 && value % 64 == 0))) break; o.generic_threads = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TEST_NODES_OVERSIZE: if (!range(0, 1)) break; o.test_nodes_oversize = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TEST_NODES_TIMEOUT: if (!range(0, 1)) break; o.test_nodes_timeout = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_NODES_FLAT_BARRIER: if (!range(0, 1)) break; o.nodes_flat = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_SMALL_CALL_KB: if (!range(-1, 65536)) break; o.small_call_kb = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_NODES_REFUSALS:
        case LDPC_AMD_OPT_NODES_LAST_HIP_ERROR:
        case LDPC_AMD_OPT_NOALLOC_MAX_FRAMES:
        case LDPC_AMD_OPT_NODES_BARRIER_TIMED_OUT:
            return fail(c, LDPC_AMD_EINVAL, "set_option: key " + std::to_string(key) + " is read-only");
        default: return fail(c, LDPC_AMD_EINVAL, "set_option: unknown key " + std::to_string(key));
    }
    return fail(c, LDPC_AMD_EINVAL, "set_option: value " + std::to_string(value) + " is out of range for key " + std::to_string(key));
}

int ldpc_amd_get_option(const ldpc_amd_ctx* c, int32_t key, int64_t* value) {
    if (!c || !value) return LDPC_AMD_EINVAL;
    const Options& o = c->opt;
    switch (key) {
        case LDPC_AMD_OPT_ENGINE: *value = o.engine; break;
        case LDPC_AMD_OPT_STREAM_KERNEL: *value = o.stream_kernel; break;
        case LDPC_AMD_OPT_NODES: *value = o.nodes; break;
        case LDPC_AMD_OPT_SIM_OVERLAP: *value = o.sim_overlap; break;
        case LDPC_AMD_OPT_SIM_FUSED: *value = o.sim_fused; break;
        case LDPC_AMD_OPT_VERBOSE: *value = o.verbose; break;
        case LDPC_AMD_OPT_TUNE_LDS_WPE: *value = o.lds_wpe; break;
        case LDPC_AMD_OPT_TUNE_LDS_FPW: *value = o.lds_fpw; break;
        case LDPC_AMD_OPT_TUNE_GENERIC_LDS_KB: *value = o.generic_lds_kb; break;
        case LDPC_AMD_OPT_TUNE_NODES_BLOCKS: *value = o.nodes_blocks; break;
        case LDPC_AMD_OPT_TUNE_FPL: *value = o.fpl; break;
        case LDPC_AMD_OPT_TUNE_LAYOUT: *value = o.layout; break;
        case LDPC_AMD_OPT_TUNE_ROWS_WAVES: *value = o.rows_waves; break;
        case LDPC_AMD_OPT_TUNE_WPS: *value = o.wps; break;
        case LDPC_AMD_OPT_TUNE_NOPIPE: *value = o.nopipe; break;
        case LDPC_AMD_OPT_TUNE_ROWS_FUSED: *value = o.rows_fused; break;
        case LDPC_AMD_OPT_TUNE_ROWS_LEAVES: *value = o.rows_leaves; break;
        case LDPC_AMD_OPT_TUNE_SIM_CHUNK_MB: *value = o.sim_chunk_mb; break;
        case LDPC_AMD_OPT_TUNE_GENERIC_THREADS: *value = o.generic_threads; break;
        case LDPC_AMD_OPT_TUNE_LDS_VPT: *value = o.lds_vpt; break;
        case LDPC_AMD_OPT_TEST_NODES_OVERSIZE: *value = o.test_nodes_oversize; break;
        case LDPC_AMD_OPT_TEST_NODES_TIMEOUT: *value = o.test_nodes_timeout; break;
        case LDPC_AMD_OPT_TUNE_NODES_FLAT_BARRIER: *value = o.nodes_flat; break;
        case LDPC_AMD_OPT_TUNE_SMALL_CALL_KB: *value = o.small_call_kb; break;
        case LDPC_AMD_OPT_NODES_BARRIER_TIMED_OUT: {
            // the flag of the most recent node-parallel launch; the caller has synchronised the stream it ran on
            *value = 0;
            if (c->nodes_bar && c->timed_launches && !strcmp(c->last_kernel, "bp_nodes_kernel")) {
                DeviceGuard guard(c->device);
                unsigned flag = 0;
                if (!guard.ok || hipMemcpy(&flag, c->nodes_bar + 1, sizeof(flag), hipMemcpyDeviceToHost) != hipSuccess)
                    return LDPC_AMD_EHIP;
                *value = flag ? 1 : 0;
            }
            break;
        }
        case LDPC_AMD_OPT_NODES_REFUSALS: *value = c->nodes_refusals; break;
        case LDPC_AMD_OPT_NODES_LAST_HIP_ERROR: *value = c->nodes_last_error; break;
        case LDPC_AMD_OPT_NOALLOC_MAX_FRAMES:
            *value = (int64_t)(LDPC_AMD_NOALLOC_SCRATCH_BYTES / stream_tile_bytes(c, 64)) * 64;
            break;
        default: return LDPC_AMD_EINVAL;
    }
    return LDPC_AMD_OK;
}

const char* ldpc_amd_last_warning(const ldpc_amd_ctx* c) { return c ? c->warn.c_str() : ""; }

int ldpc_amd_decode_batch_dev(ldpc_amd_ctx* c, const float* d_llr, int64_t frames, int32_t max_iters,
                              int32_t early_term, uint8_t* d_hard, int32_t* d_iters, int32_t* d_synd, void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (!d_llr && frames > 0) return fail(c, LDPC_AMD_EINVAL, "decode_batch: llr is null");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    SrcArgs src{Src::Float, d_llr, nullptr, 0.0, 0.0};
    return decode_dev(c, src, frames, max_iters, early_term, d_hard, d_iters, d_synd, (hipStream_t)stream);
}

int ldpc_amd_decode_batch_dev_noalloc(ldpc_amd_ctx* c, const float* d_llr, int64_t frames, int32_t max_iters,
                                      int32_t early_term, uint8_t* d_hard, int32_t* d_iters, int32_t* d_synd,
                                      void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (!d_llr && frames > 0) return fail(c, LDPC_AMD_EINVAL, "decode_batch_dev_noalloc: llr is null");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    SrcArgs src{Src::Float, d_llr, nullptr, 0.0, 0.0};
    return decode_dev(c, src, frames, max_iters, early_term, d_hard, d_iters, d_synd, (hipStream_t)stream, true);
}

int ldpc_amd_decode_bsc_batch_dev(ldpc_amd_ctx* c, const uint8_t* d_r, int64_t frames, double e, int32_t llr_mode,
                                  int32_t max_iters, int32_t early_term, uint8_t* d_hard, int32_t* d_iters,
                                  int32_t* d_synd, void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (!d_r && frames > 0) return fail(c, LDPC_AMD_EINVAL, "decode_bsc_batch: r is null");
    double l0, l1;
    if (int rc = bsc_magnitudes(c, e, llr_mode, &l0, &l1)) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    SrcArgs src{Src::Bsc, nullptr, d_r, l0, l1};
    return decode_dev(c, src, frames, max_iters, early_term, d_hard, d_iters, d_synd, (hipStream_t)stream);
}

// Host-pointer variants: chunks of frames are staged through two sets of device buffers on two
// streams, so that the copy in of chunk i + 1 and the copy out of chunk i - 1 run while chunk i is
// decoded.  The decodes themselves stay in order (they share the decode workspace).
static int decode_host(ldpc_amd_ctx* c, const void* in, size_t in_elem, bool bsc, double e, int llr_mode,
                       int64_t frames, int32_t max_iters, int32_t early_term, uint8_t* hard, int32_t* iters,
                       int32_t* synd) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || (frames > 0 && (!in || !hard))) return fail(c, LDPC_AMD_EINVAL, "decode: null buffer");
    double l0 = 0.0, l1 = 0.0;
    if (bsc)
        if (int rc = bsc_magnitudes(c, e, llr_mode, &l0, &l1)) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    if (frames == 0) return LDPC_AMD_OK;
    for (int i = 0; i < 2; ++i) {
        if (!c->io_stream[i]) LDPC_HIP(c, hipStreamCreateWithFlags(&c->io_stream[i], hipStreamNonBlocking));
        if (!c->io_decoded[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->io_decoded[i], hipEventDisableTiming));
    }
    const size_t n = (size_t)c->n;
    // A small call (the reference's drivers decode one received word per call, LDPC_GallagerIntStreamSimu.java:122):
    // four copies and their waits would cost more than the decode.  The words go through page-locked host memory
    // that the kernels read and write in place: a memcpy in, the launches, one wait, a memcpy out.
    {
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t in_bytes = (size_t)frames * n * in_elem, o_hard = up(in_bytes), o_it = o_hard + up((size_t)frames * n),
                     o_sw = o_it + up((size_t)frames * 4), total = o_sw + up((size_t)frames * 4);
        // Measured on the reference's default code (tools/perf_per_call.py): from page-locked caller memory this route
        // wins up to about 1 MiB (512 words: 141 -> 120 us; 2048 words: 243 -> 287 us), from pageable memory, whose
        // staged copies the runtime bounces through buffers of its own, at every size tried (33 MiB: 1.91 -> 1.43 ms).
        size_t limit = c->opt.small_call_kb < 0 ? 0 : (size_t)c->opt.small_call_kb << 10;
        if (c->opt.small_call_kb == 0) {
            limit = (size_t)1 << 20;
            if (total > limit && total <= ((size_t)32 << 20)) {
                hipPointerAttribute_t attr{};
                const hipError_t pe = hipPointerGetAttributes(&attr, in);
                if (pe != hipSuccess) (void)hipGetLastError();  // memory the runtime has never seen: pageable
                if (pe != hipSuccess || attr.type == hipMemoryTypeUnregistered) limit = (size_t)32 << 20;
            }
        }
        if (total <= limit) {
            if (c->small_bytes < total) {
                if (c->small_host) {
                    LDPC_HIP(c, hipStreamSynchronize(c->io_stream[0]));
                    LDPC_HIP(c, hipHostFree(c->small_host));
                    c->small_host = c->small_dev = nullptr;
                    c->small_bytes = 0;
                }
                const size_t want = std::max(total, (size_t)64 << 10);
                LDPC_HIP(c, hipHostMalloc(&c->small_host, want, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
                LDPC_HIP(c, hipHostGetDevicePointer(&c->small_dev, c->small_host, 0));
                c->small_bytes = want;
            }
            char* hs = (char*)c->small_host;
            char* ds = (char*)c->small_dev;
            memcpy(hs, in, in_bytes);
            hipStream_t s = c->io_stream[0];
            SrcArgs src{bsc ? Src::Bsc : Src::Float, (const float*)ds, (const uint8_t*)ds, l0, l1};
            if (int rc = decode_dev(c, src, frames, max_iters, early_term, (uint8_t*)(ds + o_hard), (int32_t*)(ds + o_it),
                                    (int32_t*)(ds + o_sw), s))
                return rc;
            LDPC_HIP(c, hipStreamSynchronize(s));
            if (int rc = check_nodes_barrier(c)) return rc;
            memcpy(hard, hs + o_hard, (size_t)frames * n);
            if (iters) memcpy(iters, hs + o_it, (size_t)frames * 4);
            if (synd) memcpy(synd, hs + o_sw, (size_t)frames * 4);
            return LDPC_AMD_OK;
        }
    }
    // about 64 MiB of input per chunk: long enough to keep the link busy, short enough to overlap
    const long long chunk = std::max<long long>(256, (long long)((64ull << 20) / (n * in_elem)));
    const long long chunks = (frames + chunk - 1) / chunk;
    auto copy_out = [&](long long i) -> int {  // results of chunk i, on its stream
        const int set = (int)(i & 1);
        const long long f0 = i * chunk, nf = std::min<long long>(chunk, frames - f0);
        void** st = c->stage + 4 * set;
        hipStream_t s = c->io_stream[set];
        LDPC_HIP(c, hipMemcpyAsync(hard + (size_t)f0 * n, st[1], (size_t)nf * n, hipMemcpyDeviceToHost, s));
        if (iters) LDPC_HIP(c, hipMemcpyAsync(iters + f0, st[2], (size_t)nf * 4, hipMemcpyDeviceToHost, s));
        if (synd) LDPC_HIP(c, hipMemcpyAsync(synd + f0, st[3], (size_t)nf * 4, hipMemcpyDeviceToHost, s));
        return LDPC_AMD_OK;
    };
    for (long long i = 0; i < chunks; ++i) {
        const int set = (int)(i & 1);
        const long long f0 = i * chunk, nf = std::min<long long>(chunk, frames - f0);
        hipStream_t s = c->io_stream[set];
        LDPC_HIP(c, hipStreamSynchronize(s));  // chunk i - 2 has left this set's buffers
        if (int rc = ensure_stage(c, 4 * set + 0, (size_t)nf * n * in_elem)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 1, (size_t)nf * n)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 2, (size_t)nf * 4)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 3, (size_t)nf * 4)) return rc;
        void** st = c->stage + 4 * set;
        LDPC_HIP(c, hipMemcpyAsync(st[0], (const char*)in + (size_t)f0 * n * in_elem, (size_t)nf * n * in_elem,
                                   hipMemcpyHostToDevice, s));
        if (i > 0) LDPC_HIP(c, hipStreamWaitEvent(s, c->io_decoded[set ^ 1], 0));
        SrcArgs src{bsc ? Src::Bsc : Src::Float, (const float*)st[0], (const uint8_t*)st[0], l0, l1};
        if (int rc = decode_dev(c, src, nf, max_iters, early_term, (uint8_t*)st[1], (int32_t*)st[2], (int32_t*)st[3], s))
            return rc;
        LDPC_HIP(c, hipEventRecord(c->io_decoded[set], s));
        // the previous chunk's results travel while this chunk is decoded
        if (i > 0)
            if (int rc = copy_out(i - 1)) return rc;
    }
    if (int rc = copy_out(chunks - 1)) return rc;
    LDPC_HIP(c, hipStreamSynchronize(c->io_stream[0]));
    LDPC_HIP(c, hipStreamSynchronize(c->io_stream[1]));
    return check_nodes_barrier(c);
}

int ldpc_amd_decode_batch(ldpc_amd_ctx* c, const float* llr, int64_t frames, int32_t max_iters, int32_t early_term,
                          uint8_t* hard, int32_t* iters, int32_t* synd) {
    return decode_host(c, llr, sizeof(float), false, 0.0, 0, frames, max_iters, early_term, hard, iters, synd);
}

int ldpc_amd_decode_bsc_batch(ldpc_amd_ctx* c, const uint8_t* r, int64_t frames, double e, int32_t llr_mode,
                              int32_t max_iters, int32_t early_term, uint8_t* hard, int32_t* iters, int32_t* synd) {
    return decode_host(c, r, 1, true, e, llr_mode, frames, max_iters, early_term, hard, iters, synd);
}

int ldpc_amd_set_generator(ldpc_amd_ctx* c, int32_t k, const uint8_t* G) {
    if (!c) return LDPC_AMD_EINVAL;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    if (c->d_gen) {
        LDPC_HIP(c, hipFree(c->d_gen));
        c->d_gen = nullptr;
        c->gen_k = 0;
    }
    if (k == 0 || !G) return LDPC_AMD_OK;
    if (k < 0) return fail(c, LDPC_AMD_EINVAL, "set_generator: k must be >= 0");
    if (!generator_fits(k))
        return fail(c, LDPC_AMD_ELIMIT, "set_generator: generator matrices of more than " + std::to_string(LDPC_AMD_MAX_GENERATOR_ROWS) +
                                            " rows are not supported (the codeword kernel keeps the message bits of 64 frames in LDS)");
    const size_t kw64 = ((size_t)k + 63) / 64;
    std::vector<unsigned long long> packed((size_t)c->n * kw64, 0ull);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < c->n; ++j)
            if (G[(size_t)i * c->n + j] & 1) packed[(size_t)(i >> 6) * c->n + j] |= 1ull << (i & 63);  // [kw64][n]
    LDPC_HIP(c, hipMalloc((void**)&c->d_gen, packed.size() * sizeof(unsigned long long)));
    LDPC_HIP(c, hipMemcpy(c->d_gen, packed.data(), packed.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    c->gen_k = k;
    return LDPC_AMD_OK;
}

int ldpc_amd_gen_codewords_dev(ldpc_amd_ctx* c, uint64_t seed, int64_t first_frame, int64_t frames, uint8_t* d_cw,
                               void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || (frames > 0 && !d_cw)) return fail(c, LDPC_AMD_EINVAL, "gen_codewords: bad argument");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    return gen_codewords(c, seed, first_frame, frames, d_cw, (hipStream_t)stream);
}

int ldpc_amd_gen_bsc_dev(ldpc_amd_ctx* c, int32_t channel, double e, uint64_t seed, int64_t first_frame,
                         int64_t frames, const uint8_t* d_cw, uint8_t* d_r, void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || (frames > 0 && !d_r)) return fail(c, LDPC_AMD_EINVAL, "gen_bsc: bad argument");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    return gen_bsc(c, channel, e, seed, first_frame, frames, d_cw, d_r, (hipStream_t)stream);
}

int ldpc_amd_gen_awgn_llr_dev(ldpc_amd_ctx* c, double ebn0_db, uint64_t seed, int64_t first_frame, int64_t frames,
                              const uint8_t* d_cw, float* d_llr, void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || (frames > 0 && !d_llr)) return fail(c, LDPC_AMD_EINVAL, "gen_awgn: bad argument");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    return gen_awgn(c, ebn0_db, seed, first_frame, frames, d_cw, d_llr, (hipStream_t)stream);
}

int ldpc_amd_count_errors_dev(ldpc_amd_ctx* c, int64_t frames, const uint8_t* d_hard, const uint8_t* d_cw,
                              const uint8_t* d_rx_hard, const int32_t* d_iters, const int32_t* d_synd, int32_t max_iters, uint64_t* d_counters,
                              void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || max_iters < 0 || (frames > 0 && (!d_hard || !d_iters || !d_synd || !d_counters)))
        return fail(c, LDPC_AMD_EINVAL, "count_errors: bad argument");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    return count_launch(c, frames, d_hard, d_cw, d_rx_hard, nullptr, d_iters, d_synd, max_iters,
                        (unsigned long long*)d_counters, (hipStream_t)stream);
}

// One leg: generate, decode, count, in chunks.  The counters of the chunks are added into `d_cnt` (device) by the
// kernels themselves.  With `user` set (ldpc_amd_simulate_dev) the leg is ordered after the work already on
// that stream and the stream waits for the leg; otherwise the caller synchronises.
static int simulate_leg(ldpc_amd_ctx* c, int32_t channel, double param, uint64_t seed, int64_t first_frame, int64_t frames,
                        int32_t max_iters, int32_t early_term, unsigned long long* d_cnt, bool ordered, hipStream_t user) {
    const size_t n = (size_t)c->n;
    double l0 = 0.0, l1 = 0.0;
    const bool awgn = channel == LDPC_AMD_CH_AWGN;
    if (!awgn) {
        // the parity LLR constant belongs to the decoder class: LogDecoder.java:47 vs MinSumDecoder.java:41
        const int parity_mode = (c->cn_rule == LDPC_AMD_CN_MINSUM) ? LDPC_AMD_LLR_PARITY_1EM7 : LDPC_AMD_LLR_PARITY_LN1E20;
        const int mode = (channel == LDPC_AMD_CH_BSC_INFO) ? parity_mode : LDPC_AMD_LLR_UNIFORM;
        if (int rc = bsc_magnitudes(c, param, mode, &l0, &l1)) return rc;
    }
    const size_t in_elem = awgn ? sizeof(float) : 1;
    unsigned long long chunk_mb = (unsigned long long)c->opt.sim_chunk_mb;
    if (!chunk_mb) {
        // larger chunks leave fewer half-empty last rounds of workgroups (one leg of the (3,6) n = 4096 code at 4 dB:
        // 2048 -> 8192 MiB + 3 to 8 %); what the staging buffers already hold counts as free
        size_t fr = 0, tot = 0;
        LDPC_HIP(c, hipMemGetInfo(&fr, &tot));
        size_t held = c->stage_cw_bytes;
        for (int i = 0; i < 8; ++i) held += c->stage_bytes[i];
        chunk_mb = std::min<unsigned long long>(8192, std::max<unsigned long long>(256, ((fr + held) >> 20) / 16));
    }
    long long chunk = (long long)((chunk_mb << 20) / (n * (in_elem + 1) + 8));
    chunk = std::max<long long>(256, std::min<long long>(chunk, 1ll << 20));
    chunk = std::min<long long>(chunk, std::max<long long>(frames, 1));
    // The decodes follow each other on one stream; the channel of chunk i + 1 is generated on a second stream and the
    // random codewords of chunk i + 2 on a third while chunk i is decoded: the channel kernel (HBM-bound, no LDS) runs in
    // the wave slots the resident decode kernel (VALU-bound, all of the LDS) leaves free, the codeword kernel needs LDS and
    // gets its blocks in as decode workgroups leave, and no two decodes ever share the GPU.  (Two streams that each
    // generated and decoded their own chunks ran in step: both generating, then both decoding.)  LLRs and results have
    // two sets of buffers, codewords three; a set is written once the decode that last read it is over, and read once
    // it is full.  Engines whose sweeps are HBM-bound themselves gain nothing from the overlap: one stream, one set.
    for (int i = 0; i < 2; ++i) {
        if (!c->io_stream[i]) LDPC_HIP(c, hipStreamCreateWithFlags(&c->io_stream[i], hipStreamNonBlocking));
        if (!c->io_decoded[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->io_decoded[i], hipEventDisableTiming));
        if (!c->sim_generated[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_generated[i], hipEventDisableTiming));
        if (!c->sim_decoded[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_decoded[i], hipEventDisableTiming));
    }
    if (!c->sim_cw_stream) LDPC_HIP(c, hipStreamCreateWithFlags(&c->sim_cw_stream, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) {
        if (!c->sim_cw_done[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_cw_done[i], hipEventDisableTiming));
        if (!c->sim_cw_free[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_cw_free[i], hipEventDisableTiming));
    }
    hipStream_t sd = c->io_stream[0];  // decodes
    if (ordered) {
        if (!c->sim_enter) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_enter, hipEventDisableTiming));
        LDPC_HIP(c, hipEventRecord(c->sim_enter, user));
        for (int i = 0; i < 2; ++i) LDPC_HIP(c, hipStreamWaitEvent(c->io_stream[i], c->sim_enter, 0));
        LDPC_HIP(c, hipStreamWaitEvent(c->sim_cw_stream, c->sim_enter, 0));
    }
    // the generators of this leg write the buffers the decodes of the leg before it may still be reading
    if (!c->sim_begin) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_begin, hipEventDisableTiming));
    LDPC_HIP(c, hipEventRecord(c->sim_begin, sd));
    LDPC_HIP(c, hipStreamWaitEvent(c->io_stream[1], c->sim_begin, 0));
    LDPC_HIP(c, hipStreamWaitEvent(c->sim_cw_stream, c->sim_begin, 0));
    bool overlap = false;
    long long index = 0;
    for (long long f0 = 0; f0 < frames; f0 += chunk, ++index) {
        const long long nf = std::min<long long>(chunk, frames - f0);
        const int set = overlap ? (int)(index & 1) : 0, cw_set = overlap ? (int)(index % 3) : 0;
        hipStream_t sg = overlap ? c->io_stream[1] : sd;     // channel
        hipStream_t sc = overlap ? c->sim_cw_stream : sd;    // codewords
        void** st = c->stage + 4 * set;
        if (int rc = ensure_stage(c, 4 * set + 0, (size_t)nf * n * in_elem)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 1, (size_t)nf * n)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 2, (size_t)nf * 4)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 3, (size_t)nf * 4)) return rc;
        int rc;
        const uint8_t* cw = nullptr;  // transmitted words of this chunk; null = the all-zero word
        if (c->d_gen) {
            if (c->stage_cw_bytes < (size_t)3 * chunk * n) {
                LDPC_HIP(c, hipStreamSynchronize(c->io_stream[0]));
                LDPC_HIP(c, hipStreamSynchronize(c->io_stream[1]));
                LDPC_HIP(c, hipStreamSynchronize(c->sim_cw_stream));
                if (c->stage_cw) LDPC_HIP(c, hipFree(c->stage_cw));
                c->stage_cw = nullptr;
                c->stage_cw_bytes = 0;
                LDPC_HIP(c, hipMalloc(&c->stage_cw, (size_t)3 * chunk * n));
                c->stage_cw_bytes = (size_t)3 * chunk * n;
            }
            uint8_t* mine = (uint8_t*)c->stage_cw + (size_t)cw_set * chunk * n;
            // the words of chunk index - 3 were last read by its decode
            if (overlap && index >= 3) LDPC_HIP(c, hipStreamWaitEvent(sc, c->sim_cw_free[cw_set], 0));
            rc = gen_codewords(c, seed, first_frame + f0, nf, mine, sc);
            if (rc) return rc;
            cw = mine;
            if (overlap) {
                LDPC_HIP(c, hipEventRecord(c->sim_cw_done[cw_set], sc));
                LDPC_HIP(c, hipStreamWaitEvent(sg, c->sim_cw_done[cw_set], 0));
            }
        }
        // the set's LLRs and results are free once the decode (and accounting) of chunk index - 2 is over
        if (overlap && index >= 2) LDPC_HIP(c, hipStreamWaitEvent(sg, c->sim_decoded[set], 0));
        SrcArgs src{};
        if (awgn) {
            rc = gen_awgn(c, param, seed, first_frame + f0, nf, cw, (float*)st[0], sg);
            src = SrcArgs{Src::Float, (const float*)st[0], nullptr, 0.0, 0.0};
        } else {
            rc = gen_bsc(c, channel, param, seed, first_frame + f0, nf, cw, (uint8_t*)st[0], sg);
            src = SrcArgs{Src::Bsc, nullptr, (const uint8_t*)st[0], l0, l1};
        }
        if (rc) return rc;
        if (overlap) {
            LDPC_HIP(c, hipEventRecord(c->sim_generated[set], sg));
            LDPC_HIP(c, hipStreamWaitEvent(sd, c->sim_generated[set], 0));
        }
        // the frame-resident kernels of the (6,3) and (8,2) codes count the errors themselves; for every other
        // engine the accounting is a pass of its own over the decoder's outputs
        CountArgs count{cw, d_cnt, true, false};
        rc = decode_dev(c, src, nf, max_iters, early_term, (uint8_t*)st[1], (int32_t*)st[2], (int32_t*)st[3], sd, false, &count);
        if (rc) return rc;
        if (!strcmp(c->last_kernel, "bp_nodes_kernel")) {
            // a chunk of a few frames went through the cooperative kernel: its barriers are bounded waits, and a
            // wait that gave up must not be counted as a decode
            LDPC_HIP(c, hipStreamSynchronize(sd));
            if (int brc = check_nodes_barrier(c)) return brc;
        }
        if (!count.done)
            rc = count_launch(c, nf, (const uint8_t*)st[1], cw, awgn ? nullptr : (const uint8_t*)st[0],
                              awgn ? (const float*)st[0] : nullptr, (const int32_t*)st[2], (const int32_t*)st[3], max_iters,
                              d_cnt, sd);
        if (rc) return rc;
        if (index == 0) {
            const char* k = c->last_kernel;
            overlap = k && (!strcmp(k, "bp_lds_kernel") || !strcmp(k, "bp_lds_gen_kernel")) && c->opt.sim_overlap != 0;
        }
        if (overlap) {
            LDPC_HIP(c, hipEventRecord(c->sim_decoded[index & 1], sd));
            LDPC_HIP(c, hipEventRecord(c->sim_cw_free[index % 3], sd));
        }
    }
    if (ordered) {
        if (!c->sim_cw_done[0]) return fail(c, LDPC_AMD_EHIP, "simulate: no event");
        LDPC_HIP(c, hipEventRecord(c->sim_cw_done[0], c->sim_cw_stream));
        LDPC_HIP(c, hipStreamWaitEvent(user, c->sim_cw_done[0], 0));
    }
    if (ordered)
        for (int i = 0; i < 2; ++i) {
            LDPC_HIP(c, hipEventRecord(c->io_decoded[i], c->io_stream[i]));
            LDPC_HIP(c, hipStreamWaitEvent(user, c->io_decoded[i], 0));
        }
    return LDPC_AMD_OK;
}

int ldpc_amd_simulate(ldpc_amd_ctx* c, int32_t channel, double param, uint64_t seed, int64_t first_frame,
                      int64_t frames, int32_t max_iters, int32_t early_term, uint64_t* counters) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || max_iters < 0 || !counters) return fail(c, LDPC_AMD_EINVAL, "simulate: bad argument");
    if (channel != LDPC_AMD_CH_BSC_INFO && channel != LDPC_AMD_CH_BSC_ALL && channel != LDPC_AMD_CH_AWGN)
        return fail(c, LDPC_AMD_EINVAL, "simulate: unknown channel");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    const size_t nc = (size_t)LDPC_AMD_CNT_FIXED + (size_t)max_iters;
    if (c->counters_len < nc) {
        if (c->d_counters) LDPC_HIP(c, hipFree(c->d_counters));
        c->d_counters = nullptr;
        LDPC_HIP(c, hipMalloc((void**)&c->d_counters, nc * sizeof(unsigned long long)));
        c->counters_len = nc;
    }
    LDPC_HIP(c, hipMemset(c->d_counters, 0, nc * sizeof(unsigned long long)));
    LDPC_HIP(c, hipStreamSynchronize(nullptr));  // the counters are zero before the other streams add to them
    if (int rc = simulate_leg(c, channel, param, seed, first_frame, frames, max_iters, early_term, c->d_counters, false, nullptr))
        return rc;
    LDPC_HIP(c, hipStreamSynchronize(c->io_stream[0]));
    LDPC_HIP(c, hipStreamSynchronize(c->io_stream[1]));
    LDPC_HIP(c, hipStreamSynchronize(c->sim_cw_stream));
    std::vector<unsigned long long> h(nc);
    LDPC_HIP(c, hipMemcpy(h.data(), c->d_counters, nc * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nc; ++i) counters[i] += h[i];
    return LDPC_AMD_OK;
}

int ldpc_amd_simulate_dev(ldpc_amd_ctx* c, int32_t channel, double param, uint64_t seed, int64_t first_frame,
                          int64_t frames, int32_t max_iters, int32_t early_term, uint64_t* d_counters, void* stream) {
    if (!c) return LDPC_AMD_EINVAL;
    if (frames < 0 || max_iters < 0 || !d_counters) return fail(c, LDPC_AMD_EINVAL, "simulate_dev: bad argument");
    if (channel != LDPC_AMD_CH_BSC_INFO && channel != LDPC_AMD_CH_BSC_ALL && channel != LDPC_AMD_CH_AWGN)
        return fail(c, LDPC_AMD_EINVAL, "simulate_dev: unknown channel");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    return simulate_leg(c, channel, param, seed, first_frame, frames, max_iters, early_term,
                        (unsigned long long*)d_counters, true, (hipStream_t)stream);
}

int ldpc_amd_host_alloc(uint64_t bytes, void** out) {
    if (!out || !bytes) return fail(nullptr, LDPC_AMD_EINVAL, "host_alloc: null result pointer or empty size");
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(nullptr, e == hipErrorOutOfMemory ? LDPC_AMD_ENOMEM : LDPC_AMD_EHIP,
                    std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    return LDPC_AMD_OK;
}

int ldpc_amd_host_free(void* ptr) {
    if (!ptr) return LDPC_AMD_OK;
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, LDPC_AMD_EHIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    }
    return LDPC_AMD_OK;
}

const char* ldpc_amd_last_decode_kernel(const ldpc_amd_ctx* c) { return c ? c->last_kernel : ""; }

int ldpc_amd_last_decode_timing(ldpc_amd_ctx* c, int32_t* launches, double* kernel_ms, int64_t* frames) {
    if (!c) return LDPC_AMD_EINVAL;
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(c, LDPC_AMD_EHIP, "hipSetDevice failed");
    double total = 0.0;
    for (int i = 0; i < c->timed_launches; ++i) {
        LDPC_HIP(c, hipEventSynchronize(c->ev[(size_t)(2 * i + 1)]));
        float ms = 0.f;
        LDPC_HIP(c, hipEventElapsedTime(&ms, c->ev[(size_t)(2 * i)], c->ev[(size_t)(2 * i + 1)]));
        total += ms;
    }
    if (int rc = check_nodes_barrier(c)) return rc;
    if (launches) *launches = c->timed_launches;
    if (kernel_ms) *kernel_ms = total;
    if (frames) *frames = c->timed_frames;
    return LDPC_AMD_OK;
}

}  // extern "C"
