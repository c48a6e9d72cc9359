// C ABI of the gfx950 belief-propagation decoder (declared in include/ldpc_amd.h).
// Host side: Tanner-graph construction, workspace and launch sequencing.  No CPU decode
// path exists in this library: without a HIP device every compute entry point fails.
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
#include "io_kernels.hpp"

using namespace ldpc;

// Per-context switches (ldpc_amd_set_option); 0 in a tuning field = the library's own choice.
struct Options {
    int engine = LDPC_AMD_ENGINE_AUTO;
    int stream_kernel = 0;
    int nodes = -1;
    int sim_overlap = 1;
    int sim_fused = 1;
    int rows_leaves = 1;
    int sim_chunk_mb = 2048;
    int verbose = 0;
    int lds_wpe = 0, lds_fpw = 0, generic_lds_kb = 0, nodes_blocks = 0, fpl = 0, layout = 0, rows_waves = 0, wps = 0,
        nopipe = 0, rows_fused = -1, generic_threads = 0, lds_vpt = 0;
    int test_nodes_oversize = 0, test_nodes_timeout = 0, nodes_flat = 0;
};

struct ldpc_amd_ctx {
    int device = 0;
    Options opt;
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
    hipStream_t io_stream[2] = {nullptr, nullptr};
    hipEvent_t io_decoded[2] = {nullptr, nullptr};
    hipEvent_t sim_enter = nullptr;  // ldpc_amd_simulate_dev: the caller's stream at the time of the call
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

int fail(ldpc_amd_ctx* ctx, int code, const std::string& msg) {
    if (ctx) {
        ctx->err = msg;
    } else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_create_err = msg;
    }
    return code;
}

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

// A(i) / B(j) as the row-major scan of LogDecoder.java:35-42 produces them: ascending.
// Out-of-range pairs are dropped and repeats collapse (CheckMatrixIO.java:37-45).
void build_graph(ldpc_amd_ctx* c, long long nnz, const int32_t* rows, const int32_t* cols) {
    const int m = c->m, n = c->n;
    std::vector<long long> key;
    key.reserve((size_t)nnz);
    for (long long t = 0; t < nnz; ++t) {
        const long long r = rows[t], q = cols[t];
        if (r < 0 || r >= m || q < 0 || q >= n) continue;
        key.push_back(r * (long long)n + q);
    }
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    const long long E = (long long)key.size();
    c->E = E;
    c->h_row_ptr.assign((size_t)m + 1, 0);
    c->h_col_ptr.assign((size_t)n + 1, 0);
    c->h_edge_col.resize((size_t)E);
    c->h_col_edge.resize((size_t)E);
    for (long long e = 0; e < E; ++e) {
        const int r = (int)(key[(size_t)e] / n), q = (int)(key[(size_t)e] % n);
        c->h_edge_col[(size_t)e] = q;
        c->h_row_ptr[(size_t)r + 1]++;
        c->h_col_ptr[(size_t)q + 1]++;
    }
    for (int i = 0; i < m; ++i) c->h_row_ptr[(size_t)i + 1] += c->h_row_ptr[(size_t)i];
    for (int j = 0; j < n; ++j) c->h_col_ptr[(size_t)j + 1] += c->h_col_ptr[(size_t)j];
    std::vector<int> fill(c->h_col_ptr.begin(), c->h_col_ptr.end() - 1);
    for (long long e = 0; e < E; ++e) c->h_col_edge[(size_t)fill[(size_t)c->h_edge_col[(size_t)e]]++] = (int)e;
    c->max_dc = c->max_dv = 0;
    int min_dc = 1 << 30, min_dv = 1 << 30;
    for (int i = 0; i < m; ++i) {
        const int d = c->h_row_ptr[(size_t)i + 1] - c->h_row_ptr[(size_t)i];
        c->max_dc = std::max(c->max_dc, d);
        min_dc = std::min(min_dc, d);
    }
    for (int j = 0; j < n; ++j) {
        const int d = c->h_col_ptr[(size_t)j + 1] - c->h_col_ptr[(size_t)j];
        c->max_dv = std::max(c->max_dv, d);
        min_dv = std::min(min_dv, d);
    }
    c->reg_dc = (m > 0 && min_dc == c->max_dc) ? c->max_dc : 0;
    c->reg_dv = (n > 0 && min_dv == c->max_dv) ? c->max_dv : 0;
    c->near_dc = c->near_dv = 0;
    if (m > 0 && n > 0 && min_dc >= c->max_dc - 1 && min_dv >= c->max_dv - 1 && min_dv >= 1 && (min_dc < c->max_dc || min_dv < c->max_dv) &&
        (long long)m * c->max_dc == (long long)n * c->max_dv) {
        // the completed graph has m max_dc = n max_dv edges, so the short checks and the short variables are equally many
        c->near_dc = c->max_dc;
        c->near_dv = c->max_dv;
    }
}

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

// Deterministic generator for the placement searches below: the tables must not depend on the C library.
struct PlaceRng {
    uint64_t s;
    uint32_t next() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)(s >> 33);  // 31 bits
    }
    int below(int n) { return (int)(((uint64_t)next() * (uint64_t)n) >> 31); }
};

// Tables of the frame-resident engine.  In the variable-node phase the 32 lanes of a half wavefront, which
// the LDS serves together for 4-byte accesses, read (and later write) the k-th edge slots of their variables:
// one LDS cycle if the 32 slots lie in 32 different banks (bank = dword address mod 32), N cycles if some bank
// is hit N times.  Two things are free and are chosen here so that most half wavefronts get their 32 banks:
//   1. which row of the message array a check owns (any one-to-one placement serves: a row is processed by
//      whatever thread it belongs to, and the syndrome bits are numbered by rows).  A local search over
//      swaps of two checks' rows flattens, for every k, the histogram of the banks of all k-th edge slots -
//      without that no group of 32 variables can cover the 32 banks (natural placement of the (3,6) n = 4096
//      code: 70 to 184 slots per bank where 128 are needed);
//   2. which lane owns which variable (position p = thread p % nt, step p / nt): groups of 32 variables that
//      are transversals of the banks in every k at once are built one after the other, always serving the
//      scarcest (k, bank) among the variables still compatible, with random restarts; what is left when no
//      such group is found any more is filled greedily, level by level.
// (3,6) n = 4096: 103 of 128 groups conflict-free, mean worst multiplicity 2.10 -> 1.21.  The assignment only
// changes which lane owns which variable and where a check's row lies; every variable still sums its own
// edges in ascending check order and every check its own in ascending variable order.
//
// Codes whose n is not (variables per thread) x (whole wavefronts) are tiled as a larger code: n2 = vpt * nt
// variables and m2 = n2 * dv / dc checks, of which the last n2 - n and m2 - m form a regular graph of their own
// (socket s = j' dv + k of dummy variable j' belongs to dummy check s mod (m2 - m), position s / (m2 - m) of its
// row).  The kernels give the dummy variables a positive LLR: they decide 0 at once, their checks are satisfied
// from the first iteration on, no message passes between them and the code, and their decisions are not
// written.  Every table below is built for that larger graph.
int build_resident_tables(ldpc_amd_ctx* c, int nt, int vpt) {
    if (c->d_res && c->res_nt == nt && c->res_vpt == vpt) return LDPC_AMD_OK;
    const bool completed = c->reg_dc == 0 && c->near_dc > 0;  // one edge short of regular in places: see bp_lds_kernel, DEF
    const int dc = completed ? c->near_dc : c->reg_dc, dv = completed ? c->near_dv : c->reg_dv;
    const int n = vpt * nt, m = (int)((long long)n * dv / dc);
    const long long E = (long long)n * dv;
    const int pad_m = m - c->m;
    if (n < c->n || m < c->m || (long long)m * dc != E || (n > c->n && pad_m < dv))
        return fail(c, LDPC_AMD_EINVAL, "resident tables: the code does not tile");
    std::vector<int> col_edge((size_t)E);  // edge ids (row-major positions in the larger graph) of every variable
    std::vector<char> extra_edge((size_t)n, 0);  // the variable's last edge is the constant one that completes a short node
    if (!completed) {
        std::copy(c->h_col_edge.begin(), c->h_col_edge.end(), col_edge.begin());
    } else {
        // Edge ids of the completed graph: edge e of check i keeps its place in the row, i dc + (e - row_ptr[i]); a
        // short check's last place, i dc + dc - 1, goes to a short variable as its last edge (any one-to-one
        // pairing serves: the edge carries constants).
        std::vector<int> edge_row((size_t)c->E), short_checks;
        for (int i = 0; i < c->m; ++i) {
            for (int e = c->h_row_ptr[(size_t)i]; e < c->h_row_ptr[(size_t)i + 1]; ++e) edge_row[(size_t)e] = i;
            if (c->h_row_ptr[(size_t)i + 1] - c->h_row_ptr[(size_t)i] == dc - 1) short_checks.push_back(i);
        }
        size_t next_short = 0;
        for (int j = 0; j < c->n; ++j) {
            const int c0 = c->h_col_ptr[(size_t)j], d = c->h_col_ptr[(size_t)j + 1] - c0;
            for (int k = 0; k < d; ++k) {
                const int e = c->h_col_edge[(size_t)c0 + k], i = edge_row[(size_t)e];
                col_edge[(size_t)j * dv + k] = i * dc + (e - c->h_row_ptr[(size_t)i]);
            }
            if (d == dv - 1) {
                if (next_short >= short_checks.size()) return fail(c, LDPC_AMD_EINVAL, "resident tables: short checks and variables do not pair up");
                col_edge[(size_t)j * dv + dv - 1] = short_checks[next_short++] * dc + dc - 1;
                extra_edge[(size_t)j] = 1;
            }
        }
        if (next_short != short_checks.size()) return fail(c, LDPC_AMD_EINVAL, "resident tables: short checks and variables do not pair up");
    }
    for (int j = c->n; j < n; ++j)
        for (int k = 0; k < dv; ++k) {
            const long long sock = (long long)(j - c->n) * dv + k;
            col_edge[(size_t)j * dv + k] = (int)(((long long)c->m + sock % pad_m) * dc + sock / pad_m);
        }
    const int target = n / 32;
    // coordinate of an edge: its rank among the edges of its variable
    std::vector<unsigned char> coord((size_t)E);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < dv; ++k) coord[(size_t)col_edge[(size_t)j * dv + k]] = (unsigned char)k;
    // 1. rows of the checks
    std::vector<int> where((size_t)m);  // check i owns the row that check index where[i] has in lds_slot()
    for (int i = 0; i < m; ++i) where[(size_t)i] = i;
    auto bank_of = [&](int i, int kk) { return lds_slot(where[(size_t)i] * dc + kk, dc, nt) & 31; };
    std::vector<int> hist((size_t)dv * 32, 0);
    for (int i = 0; i < m; ++i)
        for (int kk = 0; kk < dc; ++kk) ++hist[(size_t)coord[(size_t)i * dc + kk] * 32 + bank_of(i, kk)];
    {
        PlaceRng rng{0x9E3779B97F4A7C15ull};
        auto deviation = [&]() {
            long long d = 0;
            for (int v : hist) d += (long long)(v - target) * (v - target);
            return d;
        };
        long long dev = deviation();
        struct Change { int bin, by; };
        std::vector<Change> done;
        for (long long it = 0; it < 64ll * m && dev > 0; ++it) {
            const int a = rng.below(m), b2 = rng.below(m);
            if (a == b2) continue;
            done.clear();
            long long delta = 0;
            auto apply = [&](int bin, int by) {
                const int h = hist[(size_t)bin] - target;
                delta += by > 0 ? 2 * h + 1 : -2 * h + 1;
                hist[(size_t)bin] += by;
                done.push_back(Change{bin, by});
            };
            for (int kk = 0; kk < dc; ++kk) {
                apply(coord[(size_t)a * dc + kk] * 32 + bank_of(a, kk), -1);
                apply(coord[(size_t)b2 * dc + kk] * 32 + bank_of(b2, kk), -1);
            }
            std::swap(where[(size_t)a], where[(size_t)b2]);
            for (int kk = 0; kk < dc; ++kk) {
                apply(coord[(size_t)a * dc + kk] * 32 + bank_of(a, kk), +1);
                apply(coord[(size_t)b2 * dc + kk] * 32 + bank_of(b2, kk), +1);
            }
            if (delta < 0) {
                dev += delta;
            } else {  // no gain: undo
                std::swap(where[(size_t)a], where[(size_t)b2]);
                for (const Change& ch : done) hist[(size_t)ch.bin] -= ch.by;
            }
        }
    }
    std::vector<int> slot((size_t)E);
    for (long long e = 0; e < E; ++e) {
        const int i = (int)(e / dc), kk = (int)(e - (long long)i * dc);
        slot[(size_t)e] = lds_slot(where[(size_t)i] * dc + kk, dc, nt);
    }
    // 2. lanes of the variables
    std::vector<unsigned char> bank((size_t)n * dv);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < dv; ++k) bank[(size_t)j * dv + k] = (unsigned char)(slot[(size_t)col_edge[(size_t)j * dv + k]] & 31);
    std::vector<int> perm((size_t)n), pos((size_t)n);
    std::vector<char> taken((size_t)n, 0);
    int next_group = 0;
    const int groups = n / 32;
    {
        PlaceRng rng{0xD1B54A32D192ED03ull};
        std::vector<int> pool((size_t)n), cand, keep;
        for (int j = 0; j < n; ++j) pool[(size_t)j] = j;
        int members[32];
        bool used[8][32];
        int cnt[8][32];
        int failures = 0;
        while (next_group < groups && (int)pool.size() >= 32 && failures < 3) {
            bool found = false;
            for (int attempt = 0; attempt < 40 && !found; ++attempt) {
                memset(used, 0, sizeof(used));
                cand = pool;
                int got = 0;
                bool ok = true;
                for (int s2 = 0; s2 < 32 && ok; ++s2) {
                    // the variables still compatible with the banks taken so far, and how many serve each bank
                    keep.clear();
                    memset(cnt, 0, sizeof(cnt));
                    for (int j : cand) {
                        bool fits = true;
                        for (int k = 0; k < dv && fits; ++k) fits = !used[k][bank[(size_t)j * dv + k]];
                        if (!fits) continue;
                        keep.push_back(j);
                        for (int k = 0; k < dv; ++k) ++cnt[k][bank[(size_t)j * dv + k]];
                    }
                    cand.swap(keep);
                    // the scarcest (k, bank) still to be served; none may have run dry
                    int best_k = -1, best_b = -1, best_cnt = 1 << 30;
                    for (int k = 0; k < dv && ok; ++k)
                        for (int bb = 0; bb < 32; ++bb) {
                            if (used[k][bb]) continue;
                            if (cnt[k][bb] == 0) { ok = false; break; }
                            // ties broken at random, so that attempts differ
                            const int key = cnt[k][bb] * 4 + (int)(rng.next() & 3);
                            if (key < best_cnt) { best_cnt = key; best_k = k; best_b = bb; }
                        }
                    if (!ok) break;
                    int pick = rng.below(cnt[best_k][best_b]), chosen = -1;
                    for (int j : cand)
                        if (bank[(size_t)j * dv + best_k] == best_b && pick-- == 0) { chosen = j; break; }
                    members[got++] = chosen;
                    for (int k = 0; k < dv; ++k) used[k][bank[(size_t)chosen * dv + k]] = true;
                }
                found = ok && got == 32;
            }
            if (!found) {
                ++failures;
                continue;
            }
            failures = 0;
            const int v = next_group / (nt / 32), h = next_group % (nt / 32);
            for (int l = 0; l < 32; ++l) {
                const int p = (32 * h + l) + v * nt;
                perm[(size_t)p] = members[l];
                pos[(size_t)members[l]] = p;
                taken[(size_t)members[l]] = 1;
            }
            keep.clear();
            for (int j : pool)
                if (!taken[(size_t)j]) keep.push_back(j);
            pool.swap(keep);
            ++next_group;
        }
    }
    int first_free = 0;
    for (int g = next_group; g < groups; ++g) {
        // group g = positions with step v = g / (nt/32) and lanes [32 h, 32 h + 32), h = g % (nt/32)
        const int v = g / (nt / 32), h = g % (nt / 32);
        // pass L admits a variable only if none of its banks would be used more than L times
        unsigned char cnt[8][32];
        memset(cnt, 0, sizeof(cnt));
        int got = 0;
        int members[32];
        while (first_free < n && taken[(size_t)first_free]) ++first_free;
        for (int level = 1; level <= 32 && got < 32; ++level) {
            for (int j = first_free; j < n && got < 32; ++j) {
                if (taken[(size_t)j]) continue;
                bool ok = true;
                for (int k = 0; k < dv && ok; ++k) ok = cnt[k][bank[(size_t)j * dv + k]] < level;
                if (!ok) continue;
                for (int k = 0; k < dv; ++k) ++cnt[k][bank[(size_t)j * dv + k]];
                taken[(size_t)j] = 1;
                members[got++] = j;
            }
        }
        for (int l = 0; l < 32; ++l) {
            const int p = (32 * h + l) + v * nt;
            perm[(size_t)p] = members[l];
            pos[(size_t)members[l]] = p;
        }
    }
    // statistics: worst bank multiplicity of every group's k-th gather
    long long mult_sum = 0, mult_cnt = 0;
    int clean = 0;
    for (int g = 0; g < groups; ++g) {
        const int v = g / (nt / 32), h = g % (nt / 32);
        int group_worst = 0;
        for (int k = 0; k < dv; ++k) {
            int cnt[32] = {0}, worst = 0;
            for (int l = 0; l < 32; ++l) worst = std::max(worst, ++cnt[bank[(size_t)perm[(size_t)((32 * h + l) + v * nt)] * dv + k]]);
            mult_sum += worst;
            ++mult_cnt;
            group_worst = std::max(group_worst, worst);
        }
        clean += group_worst == 1;
    }
    c->res_clean_groups = clean;
    c->res_conflict = mult_cnt ? (double)mult_sum / (double)mult_cnt : 1.0;
    // layout: perm[n] (variable at each position) | pos[n] (position of each variable) | vn_off[E] (byte
    // offsets of the edge slots of the variable at each position)
    std::vector<int> tab((size_t)(2 * n) + (size_t)E);
    int* t_perm = tab.data();
    int* t_pos = t_perm + n;
    int* t_vn = t_pos + n;
    for (int p = 0; p < n; ++p) {
        t_perm[p] = perm[(size_t)p];
        t_pos[p] = pos[(size_t)p];
        for (int k = 0; k < dv; ++k) t_vn[(size_t)p * dv + k] = 4 * slot[(size_t)col_edge[(size_t)perm[(size_t)p] * dv + k]];
        if (extra_edge[(size_t)perm[(size_t)p]]) t_vn[(size_t)p * dv + dv - 1] |= 1;  // offsets are multiples of 4: bit 0 marks the extra edge
    }
    if (c->d_res) {
        LDPC_HIP(c, hipFree(c->d_res));
        c->d_res = nullptr;
    }
    LDPC_HIP(c, hipMalloc((void**)&c->d_res, tab.size() * sizeof(int)));
    LDPC_HIP(c, hipMemcpy(c->d_res, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    c->res_nt = nt;
    c->res_vpt = vpt;
    if (c->opt.verbose)
        fprintf(stderr, "ldpc_amd: resident tables nt=%d, %d variables per thread (%d dummy variables), mean worst bank multiplicity of VN gathers %.3f, %d of %d half wavefronts conflict-free\n",
                nt, vpt, n - c->n, c->res_conflict, c->res_clean_groups, n / 32);
    return LDPC_AMD_OK;
}

// Tables of the generic frame-resident engine: checks and variables sorted by degree (ties in index
// order); neighbours 2q, 2q+1 of the sorted list form work item q, and the items are dealt out to the
// slots t + u * nt in that order, so that a wave meets one degree at a time and the two nodes of an
// item nearly always share a degree.  The decision bit of the variable that is node h of item q sits at
// position 128 * (q / 64) + 64 * h + q % 64 (one 64-bit ballot per half).
int build_generic_resident_tables(ldpc_amd_ctx* c, int nt) {
    if (c->d_genres && c->gen_nt == nt) return LDPC_AMD_OK;
    const int m = c->m, n = c->n;
    const long long E = c->E;
    const int cn_items = (m + 1) / 2, vn_items = (n + 1) / 2;
    const int cn_steps = (cn_items + nt - 1) / nt, vn_steps = (vn_items + nt - 1) / nt;
    // layout: cn_desc int4[cn_steps*nt] | vn_desc int4[vn_steps*nt][2] | pos_of_var[n] | edge_pos[E] | edge_slot[E] |
    //         col_slot[E] | col_check[E]
    std::vector<int> tab((size_t)4 * cn_steps * nt + (size_t)8 * vn_steps * nt + (size_t)n + (size_t)4 * E, 0);
    int* cn_desc = tab.data();
    int* vn_desc = cn_desc + (size_t)4 * cn_steps * nt;
    int* pos_of_var = vn_desc + (size_t)8 * vn_steps * nt;
    int* edge_pos = pos_of_var + n;
    int* edge_slot = edge_pos + E;
    int* col_slot = edge_slot + E;
    int* col_check = col_slot + E;
    for (int q = 0; q < 2 * vn_steps * nt; ++q) vn_desc[4 * q] = -1;
    std::vector<int> idx((size_t)std::max(m, n));
    for (int i = 0; i < m; ++i) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.begin() + m, [&](int a, int b) {
        return c->h_row_ptr[(size_t)a + 1] - c->h_row_ptr[(size_t)a] > c->h_row_ptr[(size_t)b + 1] - c->h_row_ptr[(size_t)b];
    });
    // Message slots: block (step u, wave w) holds the 128 checks of that wave's lanes in that step as
    // [position in the row][check 0/1 of the lane][lane]; sorted descending, the first check of a block
    // has its largest degree, which is the block's depth.
    int slots = 0;
    for (int item0 = 0; item0 < cn_items; item0 += 64) {  // items t = 64 w .. 64 w + 63 of step u, in order
        const int first = idx[(size_t)(2 * item0)];
        const int depth = c->h_row_ptr[(size_t)first + 1] - c->h_row_ptr[(size_t)first];
        for (int q = 2 * item0; q < std::min(m, 2 * (item0 + 64)); ++q) {  // node q & 1 of item q >> 1
            const int i = idx[(size_t)q], lane = (q >> 1) - item0, h = q & 1;
            const int e0 = c->h_row_ptr[(size_t)i], d = c->h_row_ptr[(size_t)i + 1] - e0;
            const int s0 = slots + h * 64 + lane;
            cn_desc[2 * q] = s0;
            cn_desc[2 * q + 1] = d;
            for (int k = 0; k < d; ++k) edge_slot[e0 + k] = s0 + k * kGenRowStride;
        }
        slots += depth * kGenRowStride;
    }
    for (int j = 0; j < n; ++j) idx[(size_t)j] = j;
    std::stable_sort(idx.begin(), idx.begin() + n, [&](int a, int b) {
        return c->h_col_ptr[(size_t)a + 1] - c->h_col_ptr[(size_t)a] > c->h_col_ptr[(size_t)b + 1] - c->h_col_ptr[(size_t)b];
    });
    for (int q = 0; q < n; ++q) {
        const int j = idx[(size_t)q];
        vn_desc[4 * q] = j;
        vn_desc[4 * q + 1] = c->h_col_ptr[(size_t)j];
        vn_desc[4 * q + 2] = c->h_col_ptr[(size_t)j + 1] - c->h_col_ptr[(size_t)j];
        const int item = q >> 1, half = q & 1;
        pos_of_var[j] = 128 * (item / 64) + 64 * half + item % 64;
    }
    for (long long e = 0; e < E; ++e) edge_pos[e] = pos_of_var[c->h_edge_col[(size_t)e]];
    for (long long t = 0; t < E; ++t) col_slot[t] = edge_slot[c->h_col_edge[(size_t)t]];
    {
        std::vector<int> edge_row((size_t)E);
        for (int i = 0; i < m; ++i)
            for (int e = c->h_row_ptr[(size_t)i]; e < c->h_row_ptr[(size_t)i + 1]; ++e) edge_row[(size_t)e] = i;
        for (long long t = 0; t < E; ++t) col_check[t] = edge_row[(size_t)c->h_col_edge[(size_t)t]];
    }
    if (c->d_genres) {
        LDPC_HIP(c, hipFree(c->d_genres));
        c->d_genres = nullptr;
    }
    LDPC_HIP(c, hipMalloc((void**)&c->d_genres, tab.size() * sizeof(int)));
    LDPC_HIP(c, hipMemcpy(c->d_genres, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    c->gen_nt = nt;
    c->gen_cn_steps = cn_steps;
    c->gen_vn_steps = vn_steps;
    c->gen_slots = slots;
    return LDPC_AMD_OK;
}

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

// LLR unit of the selected check-node rule (see bp_kernels.hpp): bits for the default sum-product rule, nats otherwise.
double llr_scale(const ldpc_amd_ctx* c) {
    return (c->cn_rule == LDPC_AMD_CN_MINSUM || c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH) ? 1.0 : 1.4426950408889634;
}

int launch_bp(ldpc_amd_ctx* c, const BpParams& p, int fpl, bool early, int wpb, size_t lds_bytes, bool pipelined,
              hipStream_t s) {
    BpLaunch l;
    l.fpl = fpl;
    l.rule = c->cn_rule == LDPC_AMD_CN_MINSUM ? kCnMinSum
             : c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH ? kCnSumProduct : kCnComplement;
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

// Bytes of streaming workspace one tile of T frames takes: messages, LLRs, packed decisions, 4 state words.
size_t stream_tile_bytes(const ldpc_amd_ctx* c, int T) {
    return ((size_t)c->E + (size_t)c->n + (size_t)c->nw + 5) * (size_t)T * sizeof(float);
}

// noalloc: run the sweep launches on the static scratch array and call nothing that allocates.
int decode_dev(ldpc_amd_ctx* c, const SrcArgs& src, long long frames, int max_iters, int early_term, uint8_t* d_hard,
               int32_t* d_iters, int32_t* d_synd, hipStream_t s, bool noalloc = false, CountArgs* count = nullptr) {
    if (frames < 0 || max_iters < 0 || !d_hard) return fail(c, LDPC_AMD_EINVAL, "decode: bad argument");
    c->timed_launches = 0;
    c->timed_frames = 0;
    if (frames == 0) return LDPC_AMD_OK;
    const int n = c->n;
    if (max_iters == 0) {  // maxL = 0: all-zero word, 0 iterations, syndrome 0 (LogDecoder.java:61, :122)
        LDPC_HIP(c, hipMemsetAsync(d_hard, 0, (size_t)frames * n, s));
        if (d_iters) LDPC_HIP(c, hipMemsetAsync(d_iters, 0, (size_t)frames * 4, s));
        if (d_synd) LDPC_HIP(c, hipMemsetAsync(d_synd, 0, (size_t)frames * 4, s));
        return LDPC_AMD_OK;
    }
    const bool early = early_term != 0;
    // Frame-resident engine: regular code whose messages fit at least twice into a CU's LDS.
    if (!noalloc) {
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
            l.rule = c->cn_rule == LDPC_AMD_CN_MINSUM ? kCnMinSum
                     : c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH ? kCnSumProduct : kCnComplement;
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
            while (c->ev.size() < 2) {
                hipEvent_t e;
                LDPC_HIP(c, hipEventCreate(&e));
                c->ev.push_back(e);
            }
            LDPC_HIP(c, hipEventRecord(c->ev[0], s));
            LDPC_HIP(c, dcr == 6 ? bp_lds_launch_6_3(l, q) : dcr == 10 ? bp_lds_launch_10_5(l, q)
                                 : dvr == 4 ? bp_lds_launch_8_4(l, q) : bp_lds_launch_8_2(l, q));
            LDPC_HIP(c, hipEventRecord(c->ev[1], s));
            c->timed_launches = 1;
            c->timed_frames = frames;
            c->last_kernel = "bp_lds_kernel";
            if (counting) count->done = true;
            return LDPC_AMD_OK;
        }
    }
    // Generic frame-resident engine: any code whose messages and LLRs fit into a CU's LDS and whose
    // degrees the register code covers.
    if (!noalloc) {
        // at least 8 variables per thread, in 1, 2, 3, 4 or 8 whole wavefronts: workgroups of 5 to 7 wavefronts
        // spread unevenly over the 4 SIMDs and measured up to 1.5x slower (NR-BG1-shaped Z = 32, n = 2176, 65,536
        // frames x 50 iterations: 320 threads 78 ms, 384 threads 71 ms, 512 threads 53 ms, 256 threads 48 ms);
        // fewer threads than that mean more steps per phase, each a chain of dependent table reads
        int nt = 512;
        for (int cand : {64, 128, 192, 256, 512})
            if (cand >= (std::max(n, c->m) + 7) / 8) { nt = cand; break; }
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
            l.rule = c->cn_rule == LDPC_AMD_CN_MINSUM ? kCnMinSum
                     : c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH ? kCnSumProduct : kCnComplement;
            l.early = early; l.threads = nt; l.lds_bytes = lds_bytes; l.stream = s;
            while (c->ev.size() < 2) {
                hipEvent_t e;
                LDPC_HIP(c, hipEventCreate(&e));
                c->ev.push_back(e);
            }
            LDPC_HIP(c, hipEventRecord(c->ev[0], s));
            LDPC_HIP(c, bp_lds_gen_launch(l, q));
            LDPC_HIP(c, hipEventRecord(c->ev[1], s));
            c->timed_launches = 1;
            c->timed_frames = frames;
            c->last_kernel = "bp_lds_gen_kernel";
            return LDPC_AMD_OK;
        }
    }
    // Node-parallel cooperative engine: a batch of at most 32 frames of a code the resident engines do
    // not take runs in one launch with one thread per node (a lane per frame would leave the chip empty).
    if (!noalloc) {
        const bool can = frames <= kNodesMaxFrames && c->m > 0 && c->max_dc <= kNodesMaxDc && c->max_dv <= kNodesMaxDv &&
                         c->nodes_blocks >= 0 && c->opt.engine != LDPC_AMD_ENGINE_STREAM;
        bool use = can;
        // Per iteration this engine costs two grid barriers and the latency of two dependent phases (about
        // 14 us together) plus about 45 ns per 1000 edges of the batch (gathers of single words); the sweeps
        // cost two launches of short waves: about 80 us with the kernels specialised for a regular code, and
        // 70 us plus a round trip per variable of the largest degree otherwise (measured on an NR-BG1-sized
        // code and on a (3,6) code of n = 64800: break-even near 20 and near 9 frames).
        const bool specialised_rows = (c->reg_dc == 6 && c->reg_dv == 3) || (c->reg_dc == 8 && c->reg_dv == 2);
        const double sweeps_us = specialised_rows ? 80.0 : 70.0 + 0.9 * c->max_dv;
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
            if (c->ws_bytes < need) {
                if (c->ws) {
                    LDPC_HIP(c, hipFree(c->ws));
                    c->ws = nullptr;
                    c->ws_bytes = 0;
                }
                LDPC_HIP(c, hipMalloc(&c->ws, need));
                c->ws_bytes = need;
            }
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
            l.rule = c->cn_rule == LDPC_AMD_CN_MINSUM ? kCnMinSum
                     : c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH ? kCnSumProduct : kCnComplement;
            // the grid-wide barrier grows with the blocks that meet in it, the phases shrink with the blocks they
            // are spread over: measured optimum at 128 blocks (flat between 128 and 256) from 1 to 32 frames
            l.max_blocks = std::min(c->nodes_blocks, 128); l.stream = s;
            if (c->opt.nodes_blocks)  // measurements: a smaller grid
                l.max_blocks = std::max(1, std::min(c->nodes_blocks, c->opt.nodes_blocks));
            // tests: a grid of four times what can be resident, which the runtime must refuse at launch
            if (c->opt.test_nodes_oversize) l.max_blocks = l.force_blocks = 4 * c->nodes_blocks;
            while (c->ev.size() < 2) {
                hipEvent_t e;
                LDPC_HIP(c, hipEventCreate(&e));
                c->ev.push_back(e);
            }
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
                return LDPC_AMD_OK;
            }
            // The runtime would not start the grid as a whole, so nothing has run: this batch goes through the
            // sweep launches below.  The refusal is counted and kept (ldpc_amd_get_option, ldpc_amd_last_warning);
            // after three in a row the context stops asking until LDPC_AMD_OPT_NODES is set again.
            (void)hipGetLastError();
            note_nodes_refusal(c, le, l.max_blocks, false);
            if (++c->nodes_refused_in_a_row >= 3) c->nodes_blocks = -1;
        }
    }
    const int maxdeg = std::max(c->max_dc, c->max_dv);
    if (maxdeg > kLdsMaxDegree) return fail(c, LDPC_AMD_ELIMIT, "node degree above 128 is not supported");
    int fpl = (!early && maxdeg <= kRegDegFpl4 && !noalloc) ? 4 : 1;
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
    // + 5 words of per-frame state for the row-parallel launches
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
        if (c->ws_bytes < need) {
            if (c->ws) {
                LDPC_HIP(c, hipFree(c->ws));
                c->ws = nullptr;
                c->ws_bytes = 0;
            }
            LDPC_HIP(c, hipMalloc(&c->ws, need));
            c->ws_bytes = need;
        }
        ws = c->ws;
    }
    c->last_kernel = "bp_decode_kernel";
    while (!noalloc && (long long)c->ev.size() < 2 * std::min<long long>(rounds, 256)) {
        hipEvent_t e;
        LDPC_HIP(c, hipEventCreate(&e));
        c->ev.push_back(e);
    }

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
        // either neighbour and are not used.  The kernels specialised for a regular code issue the loads of
        // several rows together and lose by a finer cut.
        const bool specialised = (c->reg_dc == 6 && c->reg_dv == 3) || (c->reg_dc == 8 && c->reg_dv == 2);
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
        if (fpl == 1 && !specialised && (long long)chunks * tiles_per_round < want) {
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
        {
            const dim3 grid((unsigned)(tiles * (T / 64)), (unsigned)((n + 63) / 64)), block(256);
            if (src.kind == Src::Float) {
                hipLaunchKernelGGL((load_lambda_kernel<FloatLlr>), grid, block, 0, s, FloatLlr{src.llr, (float)llr_scale(c)}, ws_lam,
                                   p.lam_tile, p.lam_row, n, T, frame_base, frames);
            } else {
                hipLaunchKernelGGL((load_lambda_kernel<BscLlr>), grid, block, 0, s,
                                   BscLlr{src.r, (float)(src.l0 * llr_scale(c)), (float)(src.l1 * llr_scale(c)), n - c->m}, ws_lam, p.lam_tile, p.lam_row, n, T,
                                   frame_base, frames);
            }
            LDPC_HIP(c, hipGetLastError());
        }
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
            LDPC_HIP(c, hipMemsetAsync(ws_state, 0, 5 * slots * sizeof(int32_t), s));
            RowsLaunch l;
            l.fpl = fpl == 4 ? 4 : 1;
            l.dc = c->reg_dc; l.dv = c->reg_dv;  // 0 for an irregular code
            l.rule = c->cn_rule == LDPC_AMD_CN_MINSUM ? kCnMinSum
                     : c->cn_rule == LDPC_AMD_CN_SUMPRODUCT_TANH ? kCnSumProduct : kCnComplement;
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
        {
            const long long n4 = (n + 3) / 4;
            const long long threads = frames_here * n4;
            const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
            hipLaunchKernelGGL(unpack_hard_kernel, grid, block, 0, s, ws_bits, p.bits_tile, p.bits_row, n, T,
                               frame_base, frames_here, d_hard);
            LDPC_HIP(c, hipGetLastError());
        }
    }
    return LDPC_AMD_OK;
}

// After the stream has been synchronised: did a grid barrier of the last node-parallel launch time out?
int check_nodes_barrier(ldpc_amd_ctx* c) {
    if (!c->nodes_bar || !c->timed_launches || strcmp(c->last_kernel, "bp_nodes_kernel")) return LDPC_AMD_OK;
    unsigned flag = 0;
    LDPC_HIP(c, hipMemcpy(&flag, c->nodes_bar + 1, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) return fail(c, LDPC_AMD_EHIP, "bp_nodes_kernel: a grid barrier timed out; the results of that decode are invalid");
    return LDPC_AMD_OK;
}

int bsc_magnitudes(ldpc_amd_ctx* c, double e, int llr_mode, double* l0, double* l1) {
    if (!(e > 0.0) || !(e < 1.0)) return fail(c, LDPC_AMD_EINVAL, "crossover probability must be in (0,1)");
    const double d0 = std::log((1 - e) / e);  // LogDecoder.java:46
    double d1 = d0;
    if (llr_mode == LDPC_AMD_LLR_PARITY_LN1E20) d1 = std::log((1 - 0.00000000000000000001) / 0.00000000000000000001);
    else if (llr_mode == LDPC_AMD_LLR_PARITY_1EM7) d1 = std::log((1 - 0.0000001) / 0.0000001);
    else if (llr_mode != LDPC_AMD_LLR_UNIFORM) return fail(c, LDPC_AMD_EINVAL, "unknown llr_mode");
    *l0 = d0;
    *l1 = d1;
    return LDPC_AMD_OK;
}

uint32_t bsc_threshold(double e) {
    if (!(e > 0.0)) return 0u;
    const double t = std::floor(e * 4294967296.0);
    return (t >= 4294967295.0) ? 4294967295u : (uint32_t)t;
}

int count_launch(ldpc_amd_ctx* c, long long frames, const uint8_t* d_hard, const uint8_t* d_cw, const uint8_t* d_rx,
                 const float* d_rx_llr,
                 const int32_t* d_iters, const int32_t* d_synd, int max_iters, unsigned long long* d_counters,
                 hipStream_t s) {
    if (frames <= 0) return LDPC_AMD_OK;
    const dim3 grid((unsigned)((frames + 63) / 64)), block(256);
    const size_t sh = (size_t)(16 + max_iters) * sizeof(unsigned long long);
    const int k = c->n - c->m;
    if (d_rx) {
        hipLaunchKernelGGL((count_errors_kernel<RxBytes, true>), grid, block, sh, s, c->n, k, frames, d_hard,
                           d_cw, RxBytes{d_rx}, d_iters, d_synd, max_iters, d_counters);
    } else if (d_rx_llr) {
        hipLaunchKernelGGL((count_errors_kernel<RxLlrSign, true>), grid, block, sh, s, c->n, k, frames, d_hard,
                           d_cw, RxLlrSign{d_rx_llr}, d_iters, d_synd, max_iters, d_counters);
    } else {
        hipLaunchKernelGGL((count_errors_kernel<RxNone, false>), grid, block, sh, s, c->n, k, frames, d_hard, d_cw,
                           RxNone{}, d_iters, d_synd, max_iters, d_counters);
    }
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int gen_codewords(ldpc_amd_ctx* c, uint64_t seed, long long first_frame, long long frames, uint8_t* d_cw,
                  hipStream_t s) {
    if (frames <= 0) return LDPC_AMD_OK;
    if (!c->d_gen) {  // no generator matrix: the all-zero codeword
        LDPC_HIP(c, hipMemsetAsync(d_cw, 0, (size_t)frames * c->n, s));
        return LDPC_AMD_OK;
    }
    const size_t lds = gen_codewords_lds_bytes(c->gen_k);
    if (lds > 160 * 1024) return fail(c, LDPC_AMD_ELIMIT, "gen_codewords: generator matrices of more than 10,208 rows are not supported");
    auto kern = gen_codewords_kernel;
    LDPC_HIP(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((frames + kCwFrames - 1) / kCwFrames)), dim3(kCwThreads), lds, s, c->n, c->gen_k,
                       c->d_gen, seed, first_frame, frames, d_cw);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int gen_bsc(ldpc_amd_ctx* c, int channel, double e, uint64_t seed, long long first_frame, long long frames,
            const uint8_t* d_cw, uint8_t* d_r, hipStream_t s) {
    if (channel != LDPC_AMD_CH_BSC_INFO && channel != LDPC_AMD_CH_BSC_ALL)
        return fail(c, LDPC_AMD_EINVAL, "gen_bsc: channel must be a BSC channel");
    if (frames <= 0) return LDPC_AMD_OK;
    const int flip_end = (channel == LDPC_AMD_CH_BSC_INFO) ? (c->n - c->m) : c->n;
    const long long threads = frames * ((c->n + 3) / 4);
    hipLaunchKernelGGL(gen_bsc_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, c->n, flip_end,
                       bsc_threshold(e), seed, first_frame, frames, d_cw, d_r);
    LDPC_HIP(c, hipGetLastError());
    return LDPC_AMD_OK;
}

int gen_awgn(ldpc_amd_ctx* c, double ebn0_db, uint64_t seed, long long first_frame, long long frames,
             const uint8_t* d_cw, float* d_llr, hipStream_t s) {
    if (frames <= 0) return LDPC_AMD_OK;
    if (c->n <= c->m) return fail(c, LDPC_AMD_EINVAL, "gen_awgn: code rate (n-m)/n must be positive");
    const double rate = (double)(c->n - c->m) / (double)c->n;
    const double sigma2 = 1.0 / (2.0 * rate * std::pow(10.0, ebn0_db / 10.0));
    const float a = (float)(2.0 / sigma2), b = (float)(2.0 / std::sqrt(sigma2));
    const long long threads = frames * ((c->n + 3) / 4);
    hipLaunchKernelGGL(gen_awgn_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, c->n, a, b, seed,
                       first_frame, frames, d_cw, d_llr);
    LDPC_HIP(c, hipGetLastError());
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
        case LDPC_AMD_OPT_TUNE_SIM_CHUNK_MB: if (!range(16, 65536)) break; o.sim_chunk_mb = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_LDS_VPT: if (!(value == 0 || value == 8 || value == 12 || value == 16)) break; o.lds_vpt = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_GENERIC_THREADS: if (!(value == 0 || (value >= 64 && value <= 512 && value % 64 == 0))) break; o.generic_threads = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TEST_NODES_OVERSIZE: if (!range(0, 1)) break; o.test_nodes_oversize = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TEST_NODES_TIMEOUT: if (!range(0, 1)) break; o.test_nodes_timeout = v; return LDPC_AMD_OK;
        case LDPC_AMD_OPT_TUNE_NODES_FLAT_BARRIER: if (!range(0, 1)) break; o.nodes_flat = v; return LDPC_AMD_OK;
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
    if (gen_codewords_lds_bytes(k) > 160 * 1024)
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
    long long chunk = (long long)(((unsigned long long)c->opt.sim_chunk_mb << 20) / (n * (in_elem + 1) + 8));
    chunk = std::max<long long>(256, std::min<long long>(chunk, 1ll << 20));
    chunk = std::min<long long>(chunk, std::max<long long>(frames, 1));
    // Chunks alternate between two sets of buffers on two streams: the channel generation of chunk i + 1
    // and the error accounting of chunk i - 1 (HBM-bound, no LDS) run in the wave slots the resident
    // decode kernel of chunk i (VALU-bound, all of the LDS) leaves free.  Engines that keep their
    // messages in the context's one workspace must not overlap: they stay on one stream and one set.
    for (int i = 0; i < 2; ++i) {
        if (!c->io_stream[i]) LDPC_HIP(c, hipStreamCreateWithFlags(&c->io_stream[i], hipStreamNonBlocking));
        if (!c->io_decoded[i]) LDPC_HIP(c, hipEventCreateWithFlags(&c->io_decoded[i], hipEventDisableTiming));
    }
    if (ordered) {
        if (!c->sim_enter) LDPC_HIP(c, hipEventCreateWithFlags(&c->sim_enter, hipEventDisableTiming));
        LDPC_HIP(c, hipEventRecord(c->sim_enter, user));
        for (int i = 0; i < 2; ++i) LDPC_HIP(c, hipStreamWaitEvent(c->io_stream[i], c->sim_enter, 0));
    }
    bool overlap = false;
    long long index = 0;
    for (long long f0 = 0; f0 < frames; f0 += chunk, ++index) {
        const long long nf = std::min<long long>(chunk, frames - f0);
        const int set = overlap ? (int)(index & 1) : 0;
        hipStream_t s = c->io_stream[set];
        void** st = c->stage + 4 * set;
        if (int rc = ensure_stage(c, 4 * set + 0, (size_t)nf * n * in_elem)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 1, (size_t)nf * n)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 2, (size_t)nf * 4)) return rc;
        if (int rc = ensure_stage(c, 4 * set + 3, (size_t)nf * 4)) return rc;
        int rc;
        const uint8_t* cw = nullptr;  // transmitted words of this chunk; null = the all-zero word
        if (c->d_gen) {
            if (c->stage_cw_bytes < (size_t)2 * chunk * n) {
                LDPC_HIP(c, hipStreamSynchronize(c->io_stream[0]));
                LDPC_HIP(c, hipStreamSynchronize(c->io_stream[1]));
                if (c->stage_cw) LDPC_HIP(c, hipFree(c->stage_cw));
                c->stage_cw = nullptr;
                c->stage_cw_bytes = 0;
                LDPC_HIP(c, hipMalloc(&c->stage_cw, (size_t)2 * chunk * n));
                c->stage_cw_bytes = (size_t)2 * chunk * n;
            }
            uint8_t* mine = (uint8_t*)c->stage_cw + (size_t)set * chunk * n;
            rc = gen_codewords(c, seed, first_frame + f0, nf, mine, s);
            if (rc) return rc;
            cw = mine;
        }
        SrcArgs src{};
        if (awgn) {
            rc = gen_awgn(c, param, seed, first_frame + f0, nf, cw, (float*)st[0], s);
            src = SrcArgs{Src::Float, (const float*)st[0], nullptr, 0.0, 0.0};
        } else {
            rc = gen_bsc(c, channel, param, seed, first_frame + f0, nf, cw, (uint8_t*)st[0], s);
            src = SrcArgs{Src::Bsc, nullptr, (const uint8_t*)st[0], l0, l1};
        }
        if (rc) return rc;
        // the frame-resident kernels of the (6,3) and (8,2) codes count the errors themselves; for every other
        // engine the accounting is a pass of its own over the decoder's outputs
        CountArgs count{cw, d_cnt, true, false};
        rc = decode_dev(c, src, nf, max_iters, early_term, (uint8_t*)st[1], (int32_t*)st[2], (int32_t*)st[3], s, false, &count);
        if (rc) return rc;
        if (!strcmp(c->last_kernel, "bp_nodes_kernel")) {
            // a chunk of a few frames went through the cooperative kernel: its barriers are bounded waits, and a
            // wait that gave up must not be counted as a decode
            LDPC_HIP(c, hipStreamSynchronize(s));
            if (int brc = check_nodes_barrier(c)) return brc;
        }
        if (!count.done)
            rc = count_launch(c, nf, (const uint8_t*)st[1], cw, awgn ? nullptr : (const uint8_t*)st[0],
                              awgn ? (const float*)st[0] : nullptr, (const int32_t*)st[2], (const int32_t*)st[3], max_iters,
                              d_cnt, s);
        if (rc) return rc;
        if (index == 0) {
            const char* k = c->last_kernel;
            overlap = k && (!strcmp(k, "bp_lds_kernel") || !strcmp(k, "bp_lds_gen_kernel")) && c->opt.sim_overlap != 0;
        }
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
