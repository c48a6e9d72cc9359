/*
 * TEST INFRASTRUCTURE ONLY -- see ldpc_oracle.h.  PARITY UNPINNED by the reference
 * (no tests, golden vectors or seeded runs exist there; no JVM here to run it).
 *
 * Build without -ffast-math and without FMA contraction: the reference is Java,
 * which evaluates every double operation with IEEE-754 round-to-nearest and never
 * fuses a multiply with an add.
 */
#include "ldpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct oracle_graph {
    int m, n;
    int64_t E;
    int32_t* row_ptr;  /* [m+1]  A(i) = edge_col[row_ptr[i] .. row_ptr[i+1])  ascending column */
    int32_t* edge_col; /* [E] */
    int32_t* col_ptr;  /* [n+1]  B(j) = col_edge[col_ptr[j] .. col_ptr[j+1])  ascending row    */
    int32_t* col_edge; /* [E]   CSR edge id of each entry of B(j) */
    int32_t* edge_row; /* [E] */
};

/* ---- adjacency: LogDecoder.java:27-42 ------------------------------------
 * The reference scans the dense matrix row-major, so A(i) is ascending in j and
 * B(j) is ascending in i.  The same lists are produced here from COO pairs by a
 * counting sort on (row, col); a pair outside [0,m)x[0,n) is ignored and a
 * repeated pair sets the same matrix entry once (CheckMatrixIO.java:41-43). */
int oracle_graph_create(int m, int n, int64_t nnz, const int32_t* rows, const int32_t* cols,
                        oracle_graph** out) {
    if (!out || m < 0 || n < 0 || nnz < 0) return -1;
    oracle_graph* g = (oracle_graph*)calloc(1, sizeof(*g));
    if (!g) return -2;
    g->m = m; g->n = n;
    g->row_ptr = (int32_t*)calloc((size_t)m + 1, sizeof(int32_t));
    g->col_ptr = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    /* bucket by row, then sort columns inside each row and drop repeats */
    int32_t* cnt = (int32_t*)calloc((size_t)m + 1, sizeof(int32_t));
    for (int64_t t = 0; t < nnz; t++) {
        int r = rows[t], c = cols[t];
        if (r < 0 || r >= m || c < 0 || c >= n) continue;
        cnt[r + 1]++;
    }
    for (int i = 0; i < m; i++) cnt[i + 1] += cnt[i];
    int64_t kept = cnt[m];
    int32_t* tmp = (int32_t*)malloc((size_t)(kept ? kept : 1) * sizeof(int32_t));
    int32_t* fill = (int32_t*)malloc((size_t)(m + 1) * sizeof(int32_t));
    memcpy(fill, cnt, (size_t)(m + 1) * sizeof(int32_t));
    for (int64_t t = 0; t < nnz; t++) {
        int r = rows[t], c = cols[t];
        if (r < 0 || r >= m || c < 0 || c >= n) continue;
        tmp[fill[r]++] = c;
    }
    g->edge_col = (int32_t*)malloc((size_t)(kept ? kept : 1) * sizeof(int32_t));
    int64_t E = 0;
    for (int i = 0; i < m; i++) {
        int32_t a = cnt[i], b = cnt[i + 1];
        /* insertion sort: rows are short */
        for (int32_t p = a + 1; p < b; p++) {
            int32_t v = tmp[p], q = p - 1;
            while (q >= a && tmp[q] > v) { tmp[q + 1] = tmp[q]; q--; }
            tmp[q + 1] = v;
        }
        g->row_ptr[i] = (int32_t)E;
        for (int32_t p = a; p < b; p++) {
            if (p > a && tmp[p] == tmp[p - 1]) continue;
            g->edge_col[E++] = tmp[p];
        }
    }
    g->row_ptr[m] = (int32_t)E;
    g->E = E;
    free(tmp); free(fill); free(cnt);
    g->edge_row = (int32_t*)malloc((size_t)(E ? E : 1) * sizeof(int32_t));
    g->col_edge = (int32_t*)malloc((size_t)(E ? E : 1) * sizeof(int32_t));
    for (int i = 0; i < m; i++)
        for (int32_t p = g->row_ptr[i]; p < g->row_ptr[i + 1]; p++) {
            g->edge_row[p] = i;
            g->col_ptr[g->edge_col[p] + 1]++;
        }
    for (int j = 0; j < n; j++) g->col_ptr[j + 1] += g->col_ptr[j];
    int32_t* cf = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    memcpy(cf, g->col_ptr, (size_t)(n + 1) * sizeof(int32_t));
    for (int64_t p = 0; p < E; p++) g->col_edge[cf[g->edge_col[p]]++] = (int32_t)p; /* ascending row */
    free(cf);
    *out = g;
    return 0;
}

void oracle_graph_destroy(oracle_graph* g) {
    if (!g) return;
    free(g->row_ptr); free(g->edge_col); free(g->col_ptr); free(g->col_edge); free(g->edge_row);
    free(g);
}

int64_t oracle_graph_edges(const oracle_graph* g) { return g->E; }

void oracle_graph_csr(const oracle_graph* g, int32_t* row_ptr, int32_t* edge_col) {
    memcpy(row_ptr, g->row_ptr, (size_t)(g->m + 1) * sizeof(int32_t));
    memcpy(edge_col, g->edge_col, (size_t)g->E * sizeof(int32_t));
}
void oracle_graph_csc(const oracle_graph* g, int32_t* col_ptr, int32_t* col_edge) {
    memcpy(col_ptr, g->col_ptr, (size_t)(g->n + 1) * sizeof(int32_t));
    memcpy(col_edge, g->col_edge, (size_t)g->E * sizeof(int32_t));
}

/* ---- sensitivity to the library functions' unspecified last bits ------------------------------
 * The reference computes through java.lang.Math.tanh and Math.log (LogDecoder.java:10, :90), which
 * the JDK specifies only to within 2.5 ulp and 1 ulp of the exact result; this file computes
 * through the C library's.  To see whether any verdict depends on those bits, a decode can be
 * run with every tanh and log result moved by a random whole number of ulps inside those bounds
 * (|k| <= tanh_ulps, |k| <= log_ulps; 0 = off).  What the JDK does specify is kept: NaN, +-Inf and
 * +-0 results are returned as they are, |tanh| never exceeds 1 and a tanh of exactly +-1 stays. */
typedef struct {
    int tanh_ulps, log_ulps;
    uint64_t state;
} ulp_noise;

static inline uint64_t noise_next(ulp_noise* z) { /* xorshift64* */
    uint64_t x = z->state;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    z->state = x;
    return x * 0x2545F4914F6CDD1DULL;
}

static inline double noise_move(ulp_noise* z, double v, int max_ulps) {
    if (max_ulps <= 0 || v == 0.0 || isnan(v) || isinf(v)) return v;
    /* k ulps up or down: neighbouring doubles of one sign are neighbouring integers in their bit patterns */
    const int64_t k = (int64_t)(noise_next(z) % (uint64_t)(2 * (int64_t)max_ulps + 1)) - max_ulps;
    int64_t bits;
    memcpy(&bits, &v, sizeof bits);
    const int64_t mag = (bits & INT64_MAX) + ((bits < 0) ? -k : k);
    if (mag <= 0 || mag >= 0x7FF0000000000000LL) return v; /* would cross zero or leave the finite range */
    bits = (bits & INT64_MIN) | mag;
    memcpy(&v, &bits, sizeof bits);
    return v;
}

static inline double ref_tanh(double x, ulp_noise* z) {
    double t = tanh(x);
    if (!z || z->tanh_ulps <= 0 || t == 1.0 || t == -1.0) return t;
    t = noise_move(z, t, z->tanh_ulps);
    return (t > 1.0) ? 1.0 : (t < -1.0) ? -1.0 : t;
}

static inline double ref_log(double x, ulp_noise* z) {
    const double v = log(x);
    return z ? noise_move(z, v, z->log_ulps) : v;
}

/* LogDecoder.java:9-11 */
static inline double ref_atanh(double x, ulp_noise* z) { return 0.5 * ref_log((1.0 + x) / (1.0 - x), z); }

/* Math.signum: 0 for +-0, NaN for NaN (MinSumDecoder.java:84) */
static inline double ref_signum(double x) { return (x > 0.0) ? 1.0 : (x < 0.0) ? -1.0 : x; }

/* ---- one frame on sparse state -------------------------------------------
 * alpha/beta are indexed by CSR edge id instead of [m][n]/[n][m]; every sum and
 * product visits the same neighbours in the same ascending order and skips the
 * target edge instead of subtracting/dividing it out (LogDecoder.java:67-95). */
/* What a message array of 4-byte floats keeps of a double: `storage` 0 = the double itself (the
 * reference, LogDecoder.java:57-58), 1 = rounded to float, 2 = rounded to float after scaling to bits
 * (x / ln 2), the unit an implementation built on exp2/log2 stores.  All arithmetic stays double, so a
 * run with storage 1 or 2 shows what float STORAGE alone does to a decode.  Two more bits switch single
 * parts of the arithmetic to float, in the association a prefix / suffix implementation uses (they need
 * storage 2): +4 = the variable-node sums (prefix from lambda upwards, suffix from the last neighbour
 * downwards, beta = prefix + suffix), +8 = the check-node rule on complements, q = 1 - tanh(|beta|/2) =
 * 1 / (0.5 * 2^|beta| + 0.5), Q <- Q + q (1 - Q) for prefix and suffix, |alpha| = log2(2/Q - 1), with
 * correctly rounded float exp2 / log2 / division.  A model of a float implementation, not the reference. */
#define MODEL_FLOAT_VN 4
#define MODEL_FLOAT_CN 8
#define MODEL_MAX_DEG 256
static inline double stored(double x, int storage) {
    storage &= 3;
    if (storage == 1) return (double)(float)x;
    if (storage == 2) return (double)(float)(x * 1.4426950408889634) * 0.6931471805599453;
    return x;
}

static int decode_sparse_stored(const oracle_graph* g, const double* lambda, int max_iters, int early_term,
                                int cn_rule, uint8_t* hard, int32_t* iters_out, int32_t* synd_out,
                                double* alpha, double* beta, double* th, ulp_noise* nz, int storage) {
    const int m = g->m, n = g->n;
    const int64_t E = g->E;
    for (int64_t p = 0; p < E; p++) alpha[p] = 0.0; /* new double[numC][numV], :57 */
    memset(hard, 0, (size_t)n);                     /* new int[numV], :61 */
    int sum_syndro = 0;
    for (int l = 0; l < max_iters; l++) {
        /* variable nodes, :67-79 */
        for (int j = 0; j < n; j++) {
            const int32_t a = g->col_ptr[j], b = g->col_ptr[j + 1];
            if ((storage & MODEL_FLOAT_VN) && b - a <= MODEL_MAX_DEG && b > a) {
                const float kb = 1.4426950408889634f, kn = 0.6931471805599453f;
                float pre[MODEL_MAX_DEG + 1], al[MODEL_MAX_DEG];
                const int d = b - a;
                pre[0] = (float)(lambda[j] * (double)kb);
                for (int k = 0; k < d; k++) {
                    al[k] = (float)(alpha[g->col_edge[a + k]] * (double)kb);
                    pre[k + 1] = pre[k] + al[k];
                }
                float suf = al[d - 1];
                beta[g->col_edge[a + d - 1]] = (double)pre[d - 1] * (double)kn;
                for (int k = d - 2; k >= 0; k--) {
                    beta[g->col_edge[a + k]] = (double)(pre[k] + suf) * (double)kn;
                    suf = al[k] + suf;
                }
                continue;
            }
            for (int32_t x = a; x < b; x++) {
                double s = lambda[j];
                for (int32_t y = a; y < b; y++) {
                    if (y == x) continue;
                    s += alpha[g->col_edge[y]];
                }
                beta[g->col_edge[x]] = stored(s, storage);
            }
        }
        /* check nodes, :82-95 (sum-product) or MinSumDecoder.java:76-94 */
        if (cn_rule == ORACLE_CN_SUMPRODUCT && (storage & MODEL_FLOAT_CN)) {
            const double kb = 1.4426950408889634, kn = 0.6931471805599453;
            for (int i = 0; i < m; i++) {
                const int32_t a = g->row_ptr[i];
                const int d = g->row_ptr[i + 1] - a;
                if (d > MODEL_MAX_DEG) return -1;
                float q[MODEL_MAX_DEG], pq[MODEL_MAX_DEG];
                int sg[MODEL_MAX_DEG], all = 0;
                float Q = 0.0f;
                for (int k = 0; k < d; k++) {
                    const float bb = (float)(beta[a + k] * kb);
                    sg[k] = signbit(bb) ? 1 : 0;
                    q[k] = (fabsf(bb) >= 55.0f) ? 0.0f : 1.0f / fmaf(exp2f(fabsf(bb)), 0.5f, 0.5f);
                    if (isnan(bb)) q[k] = bb;
                    pq[k] = Q;
                    Q = (k == 0) ? q[k] : fmaf(q[k], 1.0f - Q, Q);
                    all ^= sg[k];
                }
                float sQ = 0.0f;
                for (int k = d - 1; k >= 0; k--) {
                    const float QQ = (k == d - 1) ? pq[k] : (k == 0) ? sQ : fmaf(sQ, 1.0f - pq[k], pq[k]);
                    const float mag = log2f(fmaf(1.0f / QQ, 2.0f, -1.0f));
                    alpha[a + k] = (double)((all ^ sg[k]) ? -mag : mag) * kn;
                    sQ = (k == d - 1) ? q[k] : fmaf(q[k], 1.0f - sQ, sQ);
                }
            }
        } else if (cn_rule == ORACLE_CN_SUMPRODUCT) {
            for (int64_t p = 0; p < E; p++) th[p] = ref_tanh(0.5 * beta[p], nz);
            for (int i = 0; i < m; i++) {
                const int32_t a = g->row_ptr[i], b = g->row_ptr[i + 1];
                for (int32_t x = a; x < b; x++) {
                    double prod = 1.0;
                    for (int32_t y = a; y < b; y++) {
                        if (y == x) continue;
                        prod *= th[y];
                    }
                    alpha[x] = stored(2 * ref_atanh(prod, nz), storage);
                }
            }
        } else {
            for (int i = 0; i < m; i++) {
                const int32_t a = g->row_ptr[i], b = g->row_ptr[i + 1];
                for (int32_t x = a; x < b; x++) {
                    double prod = 1.0;
                    for (int32_t y = a; y < b; y++) {
                        if (y == x) continue;
                        prod *= ref_signum(beta[y]);
                    }
                    double mn = 1.7976931348623157e308; /* Double.MAX_VALUE */
                    for (int32_t y = a; y < b; y++)
                        if (y != x && mn > fabs(beta[y])) mn = fabs(beta[y]);
                    alpha[x] = stored(prod * mn * 0.7, storage);
                }
            }
        }
        /* tentative decision, :98-105 */
        for (int j = 0; j < n; j++) {
            double gsum = lambda[j];
            if (storage & MODEL_FLOAT_VN) {
                float gf = (float)(lambda[j] * 1.4426950408889634);
                for (int32_t y = g->col_ptr[j]; y < g->col_ptr[j + 1]; y++)
                    gf += (float)(alpha[g->col_edge[y]] * 1.4426950408889634);
                gsum = (double)gf;
            } else {
                for (int32_t y = g->col_ptr[j]; y < g->col_ptr[j + 1]; y++) gsum += alpha[g->col_edge[y]];
            }
            hard[j] = (gsum < 0) ? 1 : 0;
        }
        /* parity check, :108-116, on the sparse rows */
        sum_syndro = 0;
        for (int i = 0; i < m; i++) {
            int s = 0;
            for (int32_t y = g->row_ptr[i]; y < g->row_ptr[i + 1]; y++) s += hard[g->edge_col[y]];
            sum_syndro += s % 2;
        }
        if (early_term && sum_syndro == 0) { /* :117-119 */
            *iters_out = l + 1;
            *synd_out = 0;
            return 0;
        }
    }
    *iters_out = max_iters; /* :122 */
    *synd_out = sum_syndro;
    return 0;
}

static int decode_sparse(const oracle_graph* g, const double* lambda, int max_iters, int early_term,
                         int cn_rule, uint8_t* hard, int32_t* iters_out, int32_t* synd_out,
                         double* alpha, double* beta, double* th, ulp_noise* nz) {
    return decode_sparse_stored(g, lambda, max_iters, early_term, cn_rule, hard, iters_out, synd_out,
                                alpha, beta, th, nz, 0);
}

int oracle_decode_llr(const oracle_graph* g, const double* llr, int max_iters, int early_term,
                      int cn_rule, uint8_t* hard, int32_t* iters, int32_t* syndrome_weight,
                      double* alpha_out, double* beta_out) {
    const int64_t E = g->E ? g->E : 1;
    double* alpha = (double*)malloc((size_t)E * sizeof(double));
    double* beta = (double*)calloc((size_t)E, sizeof(double));
    double* th = (double*)malloc((size_t)E * sizeof(double));
    if (!alpha || !beta || !th) { free(alpha); free(beta); free(th); return -2; }
    int rc = decode_sparse(g, llr, max_iters, early_term, cn_rule, hard, iters, syndrome_weight,
                           alpha, beta, th, NULL);
    if (alpha_out) memcpy(alpha_out, alpha, (size_t)g->E * sizeof(double));
    if (beta_out) memcpy(beta_out, beta, (size_t)g->E * sizeof(double));
    free(alpha); free(beta); free(th);
    return rc;
}

int oracle_decode_llr_noisy(const oracle_graph* g, const double* llr, int max_iters, int early_term,
                            int cn_rule, uint8_t* hard, int32_t* iters, int32_t* syndrome_weight,
                            double* alpha_out, double* beta_out, int tanh_ulps, int log_ulps,
                            uint64_t noise_seed) {
    const int64_t E = g->E ? g->E : 1;
    double* alpha = (double*)malloc((size_t)E * sizeof(double));
    double* beta = (double*)calloc((size_t)E, sizeof(double));
    double* th = (double*)malloc((size_t)E * sizeof(double));
    if (!alpha || !beta || !th) { free(alpha); free(beta); free(th); return -2; }
    ulp_noise nz = {tanh_ulps, log_ulps, noise_seed | 1ULL};
    int rc = decode_sparse(g, llr, max_iters, early_term, cn_rule, hard, iters, syndrome_weight,
                           alpha, beta, th, (tanh_ulps > 0 || log_ulps > 0) ? &nz : NULL);
    if (alpha_out) memcpy(alpha_out, alpha, (size_t)g->E * sizeof(double));
    if (beta_out) memcpy(beta_out, beta, (size_t)g->E * sizeof(double));
    free(alpha); free(beta); free(th);
    return rc;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_decode_llr_batch_f32(const oracle_graph* g, const float* llr, int64_t frames,
                                int max_iters, int early_term, int cn_rule, uint8_t* hard,
                                int32_t* iters, int32_t* syndrome_weight, int threads) {
    const int n = g->n;
    const int64_t E = g->E ? g->E : 1;
    int fail = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double* alpha = (double*)malloc((size_t)E * sizeof(double));
        double* beta = (double*)calloc((size_t)E, sizeof(double));
        double* th = (double*)malloc((size_t)E * sizeof(double));
        double* lam = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
        if (!alpha || !beta || !th || !lam) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t f = 0; f < frames; f++) {
                for (int j = 0; j < n; j++) lam[j] = (double)llr[f * n + j];
                decode_sparse(g, lam, max_iters, early_term, cn_rule, hard + f * n, iters + f,
                              syndrome_weight + f, alpha, beta, th, NULL);
            }
        }
        free(alpha); free(beta); free(th); free(lam);
    }
    return fail ? -2 : 0;
}

int oracle_decode_llr_batch_f32_stored(const oracle_graph* g, const float* llr, int64_t frames,
                                       int max_iters, int early_term, int cn_rule, uint8_t* hard,
                                       int32_t* iters, int32_t* syndrome_weight, int threads, int storage) {
    const int n = g->n;
    const int64_t E = g->E ? g->E : 1;
    int fail = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double* alpha = (double*)malloc((size_t)E * sizeof(double));
        double* beta = (double*)calloc((size_t)E, sizeof(double));
        double* th = (double*)malloc((size_t)E * sizeof(double));
        double* lam = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
        if (!alpha || !beta || !th || !lam) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t f = 0; f < frames; f++) {
                for (int j = 0; j < n; j++) lam[j] = stored((double)llr[f * n + j], storage);
                decode_sparse_stored(g, lam, max_iters, early_term, cn_rule, hard + f * n, iters + f,
                                     syndrome_weight + f, alpha, beta, th, NULL, storage);
            }
        }
        free(alpha); free(beta); free(th); free(lam);
    }
    return fail ? -2 : 0;
}

int oracle_decode_llr_batch_f64(const oracle_graph* g, const double* llr, int64_t frames,
                                int max_iters, int early_term, int cn_rule, uint8_t* hard,
                                int32_t* iters, int32_t* syndrome_weight, int threads,
                                int tanh_ulps, int log_ulps, uint64_t noise_seed) {
    const int n = g->n;
    const int64_t E = g->E ? g->E : 1;
    int fail = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double* alpha = (double*)malloc((size_t)E * sizeof(double));
        double* beta = (double*)calloc((size_t)E, sizeof(double));
        double* th = (double*)malloc((size_t)E * sizeof(double));
        if (!alpha || !beta || !th) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t f = 0; f < frames; f++) {
                /* the noise of a frame depends on (seed, frame) only, not on the thread that decodes it */
                ulp_noise nz = {tanh_ulps, log_ulps,
                                (noise_seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(f + 1)) | 1ULL};
                const int on = tanh_ulps > 0 || log_ulps > 0;
                decode_sparse(g, llr + f * n, max_iters, early_term, cn_rule, hard + f * n, iters + f,
                              syndrome_weight + f, alpha, beta, th, on ? &nz : NULL);
            }
        }
        free(alpha); free(beta); free(th);
    }
    return fail ? -2 : 0;
}

/* LogDecoder.java:45-54, LogDecoderBefore.java:45-50, MinSumDecoder.java:39-48 */
void oracle_bsc_llr(int m, int n, const uint8_t* r, double e, int llr_mode, double* llr) {
    const int g_length = n - m; /* LogDecoder.java:24 */
    const double l0 = log((1 - e) / e);
    double l1 = l0;
    if (llr_mode == ORACLE_LLR_PARITY_LN1E20) l1 = log((1 - 0.00000000000000000001) / 0.00000000000000000001);
    if (llr_mode == ORACLE_LLR_PARITY_1EM7) l1 = log((1 - 0.0000001) / 0.0000001);
    for (int j = 0; j < n; j++) {
        const double mag = (llr_mode == ORACLE_LLR_UNIFORM || j < g_length) ? l0 : l1;
        llr[j] = (r[j] == 0) ? mag : -mag;
    }
}

/* ---- dense, line-by-line mode: LogDecoder.java:17-123 ---------------------- */
int oracle_logdecoder_dense(const int32_t* H, int m, int n, const int32_t* r, double e,
                            int max_iters, int llr_mode, int cn_rule, int32_t* decoded,
                            int32_t* iters, int32_t* syndrome) {
    const int numC = m, numV = n;
    int sumSyndro = 0;
    const int gLength = n - m;
    /* A(i), B(j): :27-42 */
    int32_t* Acnt = (int32_t*)calloc((size_t)numC + 1, sizeof(int32_t));
    int32_t* Bcnt = (int32_t*)calloc((size_t)numV + 1, sizeof(int32_t));
    int64_t nnz = 0;
    for (int i = 0; i < numC; i++)
        for (int j = 0; j < numV; j++)
            if (H[(size_t)i * n + j] == 1) { Acnt[i + 1]++; Bcnt[j + 1]++; nnz++; }
    for (int i = 0; i < numC; i++) Acnt[i + 1] += Acnt[i];
    for (int j = 0; j < numV; j++) Bcnt[j + 1] += Bcnt[j];
    int32_t* A = (int32_t*)malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    int32_t* B = (int32_t*)malloc((size_t)(nnz ? nnz : 1) * sizeof(int32_t));
    int32_t* af = (int32_t*)malloc((size_t)(numC + 1) * sizeof(int32_t));
    int32_t* bf = (int32_t*)malloc((size_t)(numV + 1) * sizeof(int32_t));
    memcpy(af, Acnt, (size_t)(numC + 1) * sizeof(int32_t));
    memcpy(bf, Bcnt, (size_t)(numV + 1) * sizeof(int32_t));
    for (int i = 0; i < numC; i++)
        for (int j = 0; j < numV; j++)
            if (H[(size_t)i * n + j] == 1) { B[bf[j]++] = i; A[af[i]++] = j; }
    free(af); free(bf);
    /* lambda: :45-54 */
    double* lambda = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
    const double l0 = log((1 - e) / e);
    double l1 = l0;
    if (llr_mode == ORACLE_LLR_PARITY_LN1E20) l1 = log((1 - 0.00000000000000000001) / 0.00000000000000000001);
    if (llr_mode == ORACLE_LLR_PARITY_1EM7) l1 = log((1 - 0.0000001) / 0.0000001);
    for (int j = 0; j < gLength && j < n; j++) lambda[j] = (r[j] == 0) ? l0 : -l0;
    for (int j = (gLength > 0 ? gLength : 0); j < n; j++) lambda[j] = (r[j] == 0) ? l1 : -l1;
    /* dense messages: :57-58 */
    double* alpha = (double*)calloc((size_t)numC * numV + 1, sizeof(double));
    double* beta = (double*)calloc((size_t)numV * numC + 1, sizeof(double));
    double* gamma = (double*)malloc((size_t)(n ? n : 1) * sizeof(double));
    int32_t* synd = (int32_t*)malloc((size_t)(m ? m : 1) * sizeof(int32_t));
    if (!alpha || !beta || !gamma || !synd || !lambda) return -2;
    for (int j = 0; j < n; j++) decoded[j] = 0;
    int ret_iters = max_iters;
    for (int l = 0; l < max_iters; l++) {
        for (int j = 0; j < numV; j++) { /* :67-79 */
            for (int32_t x = Bcnt[j]; x < Bcnt[j + 1]; x++) {
                const int i = B[x];
                double product = lambda[j];
                for (int32_t y = Bcnt[j]; y < Bcnt[j + 1]; y++) {
                    const int k = B[y];
                    if (k == i) continue;
                    product += alpha[(size_t)k * numV + j];
                }
                beta[(size_t)j * numC + i] = product;
            }
        }
        for (int i = 0; i < numC; i++) { /* :82-95 */
            for (int32_t x = Acnt[i]; x < Acnt[i + 1]; x++) {
                const int j = A[x];
                double product = 1.0;
                if (cn_rule == ORACLE_CN_SUMPRODUCT) {
                    for (int32_t y = Acnt[i]; y < Acnt[i + 1]; y++) {
                        const int k = A[y];
                        if (k == j) continue;
                        product *= ref_tanh(0.5 * beta[(size_t)k * numC + i], NULL);
                    }
                    alpha[(size_t)i * numV + j] = 2 * ref_atanh(product, NULL);
                } else {
                    for (int32_t y = Acnt[i]; y < Acnt[i + 1]; y++) {
                        const int k = A[y];
                        if (k == j) continue;
                        product *= ref_signum(beta[(size_t)k * numC + i]);
                    }
                    double minbata = 1.7976931348623157e308;
                    for (int32_t y = Acnt[i]; y < Acnt[i + 1]; y++) {
                        const int k = A[y];
                        if (k != j && minbata > fabs(beta[(size_t)k * numC + i]))
                            minbata = fabs(beta[(size_t)k * numC + i]);
                    }
                    alpha[(size_t)i * numV + j] = product * minbata * 0.7;
                }
            }
        }
        for (int j = 0; j < numV; j++) { /* :98-105 */
            gamma[j] = lambda[j];
            for (int32_t y = Bcnt[j]; y < Bcnt[j + 1]; y++) gamma[j] += alpha[(size_t)B[y] * numV + j];
            decoded[j] = (gamma[j] < 0) ? 1 : 0;
        }
        sumSyndro = 0; /* :108-116 */
        for (int mm = 0; mm < m; mm++) {
            synd[mm] = 0;
            for (int nn = 0; nn < n; nn++) synd[mm] += H[(size_t)mm * n + nn] * decoded[nn];
            synd[mm] %= 2;
            sumSyndro += synd[mm];
        }
        if (sumSyndro == 0) { ret_iters = l + 1; break; } /* :117-119 */
    }
    *iters = ret_iters;
    *syndrome = sumSyndro;
    free(Acnt); free(Bcnt); free(A); free(B); free(lambda); free(alpha); free(beta); free(gamma); free(synd);
    return 0;
}

int oracle_logdecoder_dense_batch(const int32_t* H, int m, int n, const uint8_t* r, int64_t frames,
                                  double e, int max_iters, int llr_mode, uint8_t* hard,
                                  int32_t* iters, int32_t* syndrome, int threads) {
    int fail = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int32_t* ri = (int32_t*)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
        int32_t* di = (int32_t*)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
#pragma omp for schedule(dynamic, 1)
        for (int64_t f = 0; f < frames; f++) {
            for (int j = 0; j < n; j++) ri[j] = r[f * n + j];
            int rc = oracle_logdecoder_dense(H, m, n, ri, e, max_iters, llr_mode, ORACLE_CN_SUMPRODUCT,
                                             di, iters + f, syndrome + f);
            if (rc) {
#pragma omp atomic write
                fail = 1;
            }
            for (int j = 0; j < n; j++) hard[f * n + j] = (uint8_t)di[j];
        }
        free(ri); free(di);
    }
    return fail ? -2 : 0;
}

/* ---- Philox4x32-10 ------------------------------------------------------- */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void frame_block(uint64_t seed, uint64_t frame, uint32_t group, uint32_t stream,
                               uint32_t x[4]) {
    const uint32_t ctr[4] = {(uint32_t)frame, (uint32_t)(frame >> 32), group, stream};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    oracle_philox4x32_10(ctr, key, x);
}

static inline uint32_t bsc_threshold(double e) {
    if (!(e > 0.0)) return 0u;
    const double t = floor(e * 4294967296.0);
    return (t >= 4294967295.0) ? 4294967295u : (uint32_t)t;
}

/* GenerateC.java:5-27: c = u * G over GF(2).  The reference draws u from Math.random();
 * here message bit i of frame f is bit (i mod 128) of the Philox block (f, i / 128, stream 2).
 * G is [k][n] bytes; G == NULL gives the all-zero word. */
void oracle_gen_codewords(int n, int k, const uint8_t* G, uint64_t seed, int64_t first_frame,
                          int64_t frames, uint8_t* cw) {
    memset(cw, 0, (size_t)frames * (size_t)n);
    if (!G) return;
    for (int64_t f = 0; f < frames; f++) {
        uint32_t x[4] = {0, 0, 0, 0};
        for (int i = 0; i < k; i++) {
            if ((i & 127) == 0) frame_block(seed, (uint64_t)(first_frame + f), (uint32_t)(i >> 7), 2u, x);
            if ((x[(i & 127) >> 5] >> (i & 31)) & 1u)
                for (int j = 0; j < n; j++) cw[f * n + j] ^= (uint8_t)(G[(size_t)i * n + j] & 1);
        }
    }
}

/* Channel.java:5-17 (info positions only) / :28-39 (all positions); cw == NULL is the all-zero
 * word.  nextDouble() < e becomes u32 < floor(e * 2^32) on one Philox word per position. */
void oracle_gen_bsc(int m, int n, int channel, double e, uint64_t seed, int64_t first_frame,
                    int64_t frames, const uint8_t* cw, uint8_t* r) {
    const uint32_t thr = bsc_threshold(e);
    const int flip_end = (channel == ORACLE_CH_BSC_INFO) ? (n - m) : n;
    for (int64_t f = 0; f < frames; f++) {
        uint32_t x[4];
        for (int j = 0; j < n; j++) {
            if ((j & 3) == 0) frame_block(seed, (uint64_t)(first_frame + f), (uint32_t)(j >> 2), 0u, x);
            const uint8_t flip = (j < flip_end && x[j & 3] < thr) ? 1 : 0;
            r[f * n + j] = (uint8_t)((cw ? (cw[f * n + j] & 1) : 0) ^ flip);
        }
    }
}

void oracle_gen_awgn_llr(int m, int n, double ebn0_db, uint64_t seed, int64_t first_frame,
                         int64_t frames, const uint8_t* cw, float* llr) {
    const double rate = (double)(n - m) / (double)n;
    const double sigma2 = 1.0 / (2.0 * rate * pow(10.0, ebn0_db / 10.0));
    const float a = (float)(2.0 / sigma2);
    const float b = (float)(2.0 / sqrt(sigma2));
    for (int64_t f = 0; f < frames; f++) {
        for (int j0 = 0; j0 < n; j0 += 4) {
            uint32_t x[4];
            frame_block(seed, (uint64_t)(first_frame + f), (uint32_t)(j0 >> 2), 1u, x);
            float z[4];
            for (int h = 0; h < 2; h++) {
                const float u1 = ((float)x[2 * h] + 0.5f) * 2.3283064365386963e-10f;
                const float u2 = (float)x[2 * h + 1] * 2.3283064365386963e-10f;
                const float rad = sqrtf(-2.0f * logf(u1));
                const float th = 6.283185307179586f * u2;
                z[2 * h] = rad * cosf(th);
                z[2 * h + 1] = rad * sinf(th);
            }
            for (int t = 0; t < 4 && j0 + t < n; t++) {
                const float sa = (cw && cw[f * n + j0 + t]) ? -a : a; /* bit 1 is sent as -1 */
                llr[f * n + j0 + t] = sa + b * z[t];
            }
        }
    }
}

/* LDPC_GallagerIntStreamSimu.java:136-183; cw == NULL is the all-zero codeword. */
void oracle_count_errors(int m, int n, int64_t frames, const uint8_t* hard, const uint8_t* cw,
                         const uint8_t* rx_hard, const int32_t* iters, const int32_t* syndrome_weight,
                         int max_iters, uint64_t* counters) {
    const int k = n - m;
    for (int64_t f = 0; f < frames; f++) {
        const uint8_t* d = hard + f * n;
        const uint8_t* r = rx_hard + f * n;
        uint64_t info = 0, chk = 0, mis = 0, ch = 0;
        for (int j = 0; j < n; j++) {
            const int cj = cw ? (cw[f * n + j] != 0) : 0;
            const int dj = (d[j] != 0) != cj; /* decodedC[j] != c[j] */
            const int rj = (r[j] != 0) != cj; /* r[j] != c[j] */
            if (j < k) {
                info += (uint64_t)dj;       /* :156-161 */
                if (!rj && dj) mis++;       /* :162-164 */
                ch += (uint64_t)rj;         /* Channel.java:18-27 */
            } else {
                chk += (uint64_t)dj;        /* :167-172 */
            }
        }
        const int correct = (info + chk) == 0; /* Arrays.equals(c, decodedC), :136 */
        counters[0] += 1;
        if (!correct) counters[1] += 1;
        counters[2] += info;
        counters[3] += chk;
        if (info) counters[4] += 1;
        if (chk) counters[5] += 1;
        if (syndrome_weight[f] == 0 && !correct) counters[6] += 1; /* :182 */
        counters[7] += mis;
        if (correct) counters[8] += (uint64_t)iters[f]; else counters[9] += (uint64_t)iters[f];
        counters[10] += ch;
        counters[11] += ch * ch;
        if (iters[f] >= 1 && iters[f] <= max_iters)
            counters[ORACLE_NUM_FIXED_COUNTERS + iters[f] - 1] += 1; /* :144 */
    }
}

int oracle_simulate(const oracle_graph* g, int gen_k, const uint8_t* G, int channel, double param,
                    uint64_t seed, int64_t first_frame, int64_t frames, int max_iters, int early_term,
                    int cn_rule, uint64_t* counters, int threads) {
    const int m = g->m, n = g->n;
    const int64_t chunk = 256;
    uint8_t* r = (uint8_t*)malloc((size_t)chunk * n);
    uint8_t* hard = (uint8_t*)malloc((size_t)chunk * n);
    uint8_t* cw = G ? (uint8_t*)malloc((size_t)chunk * n) : NULL;
    float* llr = (float*)malloc((size_t)chunk * n * sizeof(float));
    double* tmp = (double*)malloc((size_t)chunk * (n ? n : 1) * sizeof(double));
    int32_t* it = (int32_t*)malloc((size_t)chunk * sizeof(int32_t));
    int32_t* sw = (int32_t*)malloc((size_t)chunk * sizeof(int32_t));
    if (!r || !hard || (G && !cw) || !llr || !tmp || !it || !sw) return -2;
    int rc = 0;
    for (int64_t f0 = 0; f0 < frames && rc == 0; f0 += chunk) {
        const int64_t nf = (frames - f0 < chunk) ? (frames - f0) : chunk;
        if (G) oracle_gen_codewords(n, gen_k, G, seed, first_frame + f0, nf, cw);
        if (channel == ORACLE_CH_AWGN) {
            oracle_gen_awgn_llr(m, n, param, seed, first_frame + f0, nf, cw, llr);
            for (int64_t t = 0; t < nf * n; t++) r[t] = (llr[t] < 0.0f) ? 1 : 0;
        } else {
            oracle_gen_bsc(m, n, channel, param, seed, first_frame + f0, nf, cw, r);
            /* LogDecoder.java:47 / MinSumDecoder.java:41: the parity constant goes with the decoder class */
            const int pmode = (cn_rule == ORACLE_CN_MINSUM) ? ORACLE_LLR_PARITY_1EM7 : ORACLE_LLR_PARITY_LN1E20;
            const int mode = (channel == ORACLE_CH_BSC_INFO) ? pmode : ORACLE_LLR_UNIFORM;
            /* lambda stays double, as in the reference (LogDecoder.java:45-54) */
            for (int64_t f = 0; f < nf; f++) oracle_bsc_llr(m, n, r + f * n, param, mode, tmp + f * n);
        }
        if (channel == ORACLE_CH_AWGN)
            rc = oracle_decode_llr_batch_f32(g, llr, nf, max_iters, early_term, cn_rule, hard, it, sw, threads);
        else
            rc = oracle_decode_llr_batch_f64(g, tmp, nf, max_iters, early_term, cn_rule, hard, it, sw, threads, 0, 0, 0);
        if (rc == 0) oracle_count_errors(m, n, nf, hard, cw, r, it, sw, max_iters, counters);
    }
    free(r); free(hard); free(cw); free(llr); free(tmp); free(it); free(sw);
    return rc;
}
