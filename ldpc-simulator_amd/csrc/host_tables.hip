// Tanner graph of a context and the tables of the two frame-resident engines (host code only).
#include "host_ctx.hpp"

using namespace ldpc;

namespace ldpc {
namespace host {

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

// Deals the work items of one phase of the generic frame-resident engine out to the lanes of its workgroup.
// deg[q] = degrees of the two nodes of item q.  A wavefront runs the code of every distinct pair of degrees
// among its 64 lanes of a step one after the other, so the time of a (step, wavefront) block is about the sum
// over its distinct pairs of (da + db + overhead), whatever the number of lanes on each, and a phase lasts as
// long as its slowest wavefront.  Items are grouped by their pair of degrees, cut into chunks of at most 64 and
// placed, dearest first, on the wavefront with the least time so far (in its emptiest block); a chunk whose pair
// a block already holds joins it for free.  Returns the slot t + u * nt of every item; *slowest = the modelled
// time of the slowest wavefront.
static std::vector<int> deal_items(const std::vector<std::pair<int, int>>& deg, int nt, int steps, int overhead, int* slowest) {
    const int waves = nt / 64, items = (int)deg.size();
    struct Chunk { int cost, da, db; std::vector<int> q; };
    std::vector<Chunk> chunks;
    {
        std::vector<int> order((size_t)items);
        for (int q = 0; q < items; ++q) order[(size_t)q] = q;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            const int ca = deg[(size_t)a].first + deg[(size_t)a].second, cb = deg[(size_t)b].first + deg[(size_t)b].second;
            if (ca != cb) return ca > cb;
            return deg[(size_t)a].first > deg[(size_t)b].first;
        });
        for (int k = 0; k < items; ++k) {
            const int q = order[(size_t)k];
            const int da = deg[(size_t)q].first, db = deg[(size_t)q].second;
            if (chunks.empty() || chunks.back().da != da || chunks.back().db != db || chunks.back().q.size() == 64)
                chunks.push_back(Chunk{da + db + overhead, da, db, {}});
            chunks.back().q.push_back(q);
        }
        std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& a, const Chunk& b) {
            if (a.cost != b.cost) return a.cost > b.cost;
            return a.q.size() > b.q.size();
        });
    }
    std::vector<int> load((size_t)waves, 0), used((size_t)steps * waves, 0), slot((size_t)items, -1);
    std::vector<std::vector<std::pair<int, int>>> held((size_t)steps * waves);
    for (Chunk& ch : chunks) {
        size_t done = 0;
        while (done < ch.q.size()) {
            int blk = -1, add = 0;
            for (int b = 0; b < steps * waves && blk < 0; ++b)
                if (used[(size_t)b] < 64)
                    for (const auto& pr : held[(size_t)b])
                        if (pr.first == ch.da && pr.second == ch.db) { blk = b; break; }
            if (blk < 0) {
                int best_w = -1;
                for (int w = 0; w < waves; ++w) {
                    bool room = false;
                    for (int u = 0; u < steps; ++u) room = room || used[(size_t)(u * waves + w)] < 64;
                    if (room && (best_w < 0 || load[(size_t)w] < load[(size_t)best_w])) best_w = w;
                }
                for (int u = 0; u < steps; ++u) {
                    const int b = u * waves + best_w;
                    if (used[(size_t)b] < 64 && (blk < 0 || used[(size_t)b] < used[(size_t)blk])) blk = b;
                }
                add = ch.cost;
                held[(size_t)blk].push_back({ch.da, ch.db});
            }
            const int u = blk / waves, w = blk % waves;
            load[(size_t)w] += add;
            while (done < ch.q.size() && used[(size_t)blk] < 64) {
                slot[(size_t)ch.q[done++]] = u * nt + 64 * w + used[(size_t)blk];
                ++used[(size_t)blk];
            }
        }
    }
    *slowest = *std::max_element(load.begin(), load.end());
    return slot;
}

// Work items of one phase: nodes in descending order of degree (`sorted`, degrees `d`), two nodes of one degree
// per item.  Packed arithmetic takes as long as the same operations one element at a time, so an item of two
// nodes costs a wavefront twice the time of an item of one, whatever the number of lanes that hold such items:
// with `singles`, what is left of a degree after whole wavefronts of pairs (128 nodes) goes one node to a lane
// when that fits in one wavefront (at most 64 nodes), which halves its time.  Second node -1 = none.
static std::vector<std::pair<int, int>> form_items(const std::vector<int>& sorted, const std::vector<int>& d, bool singles) {
    std::vector<std::pair<int, int>> items;
    const size_t total = sorted.size();
    for (size_t g0 = 0; g0 < total;) {
        size_t g1 = g0;
        while (g1 < total && d[(size_t)sorted[g1]] == d[(size_t)sorted[g0]]) ++g1;
        const size_t count = g1 - g0, rest = count % 128, paired = (singles && rest <= 64) ? count - rest : count;
        size_t k = g0;
        for (; k + 1 < g0 + paired; k += 2) items.push_back({sorted[k], sorted[k + 1]});
        for (; k < g1; ++k) items.push_back({sorted[k], -1});
        g0 = g1;
    }
    return items;
}

// One phase's items and their places: both ways of forming the items are dealt out and the one whose slowest
// wavefront is faster by the model is kept (every step costs each wavefront `step_cost` more: its descriptor
// reads and branches).  max_steps bounds the steps per thread.
struct PhasePlan {
    std::vector<std::pair<int, int>> items;
    std::vector<int> slot;
    int steps = 0;
};
static PhasePlan plan_phase(const std::vector<int>& sorted, const std::vector<int>& d, int nt, int overhead, int step_cost,
                            int max_steps) {
    PhasePlan best;
    int best_time = 0;
    for (int singles = 0; singles < 2; ++singles) {
        PhasePlan p;
        p.items = form_items(sorted, d, singles != 0);
        p.steps = ((int)p.items.size() + nt - 1) / nt;
        if (singles && p.steps > max_steps && !best.items.empty()) continue;
        std::vector<std::pair<int, int>> deg(p.items.size());
        for (size_t q = 0; q < p.items.size(); ++q)
            deg[q] = {d[(size_t)p.items[q].first], p.items[q].second >= 0 ? d[(size_t)p.items[q].second] : 0};
        int slowest = 0;
        p.slot = deal_items(deg, nt, p.steps, overhead, &slowest);
        const int time = slowest + p.steps * step_cost;
        if (best.items.empty() || time < best_time) {
            best = std::move(p);
            best_time = time;
        }
    }
    return best;
}

// Tables of the generic frame-resident engine.  Checks and variables are sorted by degree (ties in index
// order) and formed into work items of two nodes of one degree, or of one node (form_items); deal_items places
// the items on the slots t + u * nt (thread t, step u) so that the wavefronts of a workgroup have equal work.
// The decision bit of the variable that is node h of the item in slot s sits at position
// 128 * (s / 64) + 64 * h + s % 64 (one 64-bit ballot per half).
// Message slots: a check's row is contiguous; the rows of the first checks of a run of lanes with one pair of
// degrees follow each other, then those of the second checks, with a word of skew wherever the banks would
// repeat, so the lanes that run one degree's code together meet no bank conflict.
int build_generic_resident_tables(ldpc_amd_ctx* c, int nt) {
    if (c->d_genres && c->gen_nt == nt) return LDPC_AMD_OK;
    const int m = c->m, n = c->n;
    const long long E = c->E;
    // model constants, in units of one edge of a node: what a wavefront pays per distinct pair of degrees in a step
    // and per step (measured flat between 0 and 12 / 24 on NR-BG1-shaped and quasi-cyclic codes)
    const int overhead_cn = 2, overhead_vn = 12, step_cn = 4, step_vn = 8;
    std::vector<int> dc((size_t)m), dv((size_t)n), sorted((size_t)m);
    for (int i = 0; i < m; ++i) {
        dc[(size_t)i] = c->h_row_ptr[(size_t)i + 1] - c->h_row_ptr[(size_t)i];
        sorted[(size_t)i] = i;
    }
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return dc[(size_t)a] > dc[(size_t)b]; });
    const PhasePlan cn = plan_phase(sorted, dc, nt, overhead_cn, step_cn, 1 << 20);
    sorted.resize((size_t)n);
    for (int j = 0; j < n; ++j) {
        dv[(size_t)j] = c->h_col_ptr[(size_t)j + 1] - c->h_col_ptr[(size_t)j];
        sorted[(size_t)j] = j;
    }
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return dv[(size_t)a] > dv[(size_t)b]; });
    const PhasePlan vn = plan_phase(sorted, dv, nt, overhead_vn, step_vn, 16);  // 2 decision bits per step in one register
    const int cn_steps = cn.steps, vn_steps = vn.steps;
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
    int slots = 0;
    {
        std::vector<int> item_at((size_t)cn_steps * nt, -1);
        for (size_t q = 0; q < cn.items.size(); ++q) item_at[(size_t)cn.slot[q]] = (int)q;
        auto degrees = [&](int q) {
            const auto& it = cn.items[(size_t)q];
            return std::make_pair(dc[(size_t)it.first], it.second >= 0 ? dc[(size_t)it.second] : 0);
        };
        for (int s0 = 0; s0 < cn_steps * nt;) {
            const int q0 = item_at[(size_t)s0];
            if (q0 < 0) { ++s0; continue; }
            int s1 = s0 + 1;  // run of lanes of one block with one pair of degrees
            while (s1 < cn_steps * nt && (s1 & 63) != 0 && item_at[(size_t)s1] >= 0 && degrees(item_at[(size_t)s1]) == degrees(q0)) ++s1;
            const int run = s1 - s0, da = degrees(q0).first, db = degrees(q0).second;
            // row r of a region of rows of degree d starts r * d + r / (32 / gcd(d, 32)) words into it: a 4-byte
            // access of a wavefront is served 32 lanes at a time from 32 banks, rows d words apart repeat their
            // banks every 32 / gcd rows, and one word of skew there gives 32 lanes 32 different banks
            auto skew_rows = [](int d) { int g = 1; while (g < 32 && d % (2 * g) == 0) g *= 2; return 32 / g; };
            auto row_at = [&](int r, int d) { return r * d + r / skew_rows(d); };
            const int size_a = da ? row_at(run - 1, da) + da : 0, size_b = db ? row_at(run - 1, db) + db : 0;
            for (int s = s0; s < s1; ++s) {
                const auto& it = cn.items[(size_t)item_at[(size_t)s]];
                for (int h = 0; h < 2; ++h) {
                    const int i = h ? it.second : it.first;
                    if (i < 0) continue;
                    const int e0 = c->h_row_ptr[(size_t)i], d = h ? db : da;
                    const int first = h ? slots + size_a + row_at(s - s0, db) : slots + row_at(s - s0, da);
                    cn_desc[4 * s + 2 * h] = first;
                    cn_desc[4 * s + 2 * h + 1] = d;
                    for (int k = 0; k < d; ++k) edge_slot[e0 + k] = first + k * kGenRowStride;
                }
            }
            slots += size_a + size_b;
            s0 = s1;
        }
    }
    for (size_t q = 0; q < vn.items.size(); ++q)
        for (int half = 0; half < 2; ++half) {
            const int j = half ? vn.items[q].second : vn.items[q].first;
            if (j < 0) continue;
            const int s = vn.slot[q];
            int* vd = vn_desc + 4 * (2 * (size_t)s + half);
            vd[0] = j;
            vd[1] = c->h_col_ptr[(size_t)j];
            vd[2] = dv[(size_t)j];
            pos_of_var[j] = 128 * (s / 64) + 64 * half + s % 64;
        }
    for (long long e = 0; e < E; ++e) edge_pos[e] = pos_of_var[c->h_edge_col[(size_t)e]];
    for (long long t = 0; t < E; ++t) col_slot[t] = edge_slot[c->h_col_edge[(size_t)t]];
    {
        // the kernels index LDS and the decision words with these tables: every edge its own slot inside the
        // message array, every variable its own bit inside the decision words
        std::vector<char> seen((size_t)slots, 0), seen_pos((size_t)2 * vn_steps * nt, 0);
        bool ok = true;
        for (long long e = 0; e < E && ok; ++e) {
            const int sl = edge_slot[e];
            ok = sl >= 0 && sl < slots && !seen[(size_t)sl];
            if (ok) seen[(size_t)sl] = 1;
        }
        for (int j = 0; j < n && ok; ++j) {
            const int q = pos_of_var[j];
            ok = q >= 0 && q < 2 * vn_steps * nt && !seen_pos[(size_t)q];
            if (ok) seen_pos[(size_t)q] = 1;
        }
        if (!ok) return fail(c, LDPC_AMD_EHIP, "internal error: the tables of the generic resident engine are inconsistent");
    }
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

}  // namespace host
}  // namespace ldpc
