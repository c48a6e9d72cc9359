/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the sum-product decoder path.
 *
 * Double-precision restatement of the reference's flooding belief-propagation
 * decoder (src/LDPC/LogDecoder.java, src/LDPC/LogDecoderBefore.java,
 * src/LDPC/MinSumDecoder.java), its channel (src/LDPC/Channel.java) and its
 * per-frame error accounting (src/LDPC/LDPC_GallagerIntStreamSimu.java:136-183).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path never does.
 *
 * PARITY UNPINNED by the reference: it ships no tests, no golden vectors and no
 * seeded run, and no JVM exists in the build container to execute it.  The pins
 * are (1) closed-form known answers, (2) a second, independent NumPy
 * restatement (oracle/np_restatement.py) that must agree with this file, and
 * (3) the dense, line-by-line mode below agreeing with the sparse mode.
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_graph oracle_graph;

/* check-node rules */
enum { ORACLE_CN_SUMPRODUCT = 0, ORACLE_CN_MINSUM = 1 };
/* BSC LLR modes (LogDecoder.java:45-54 / LogDecoderBefore.java:45-50 / MinSumDecoder.java:40-48) */
enum { ORACLE_LLR_UNIFORM = 0, ORACLE_LLR_PARITY_LN1E20 = 1, ORACLE_LLR_PARITY_1EM7 = 2 };
/* channels (Channel.java:5-17, :28-39; AWGN is not in the reference) */
enum { ORACLE_CH_BSC_INFO = 0, ORACLE_CH_BSC_ALL = 1, ORACLE_CH_AWGN = 2 };

#define ORACLE_NUM_FIXED_COUNTERS 16

/* COO (row, col) pairs in any order; duplicates and out-of-range pairs are
 * dropped as CheckMatrixIO.loadCheckMatrix does (CheckMatrixIO.java:37-45). */
int  oracle_graph_create(int m, int n, int64_t nnz, const int32_t* rows, const int32_t* cols,
                         oracle_graph** out);
void oracle_graph_destroy(oracle_graph* g);
int64_t oracle_graph_edges(const oracle_graph* g);
/* copies of the adjacency, for indexing tests */
void oracle_graph_csr(const oracle_graph* g, int32_t* row_ptr, int32_t* edge_col);
void oracle_graph_csc(const oracle_graph* g, int32_t* col_ptr, int32_t* col_edge);

/* One frame, channel LLRs in (double), sparse state, reference arithmetic order.
 * early_term=1 is the reference's behaviour (LogDecoder.java:117-119).
 * alpha_out/beta_out (length E, CSR edge order) may be NULL. */
int oracle_decode_llr(const oracle_graph* g, const double* llr, int max_iters, int early_term,
                      int cn_rule, uint8_t* hard, int32_t* iters, int32_t* syndrome_weight,
                      double* alpha_out, double* beta_out);

/* Frame batch from float LLRs (what the GPU path is fed), frame-parallel over
 * `threads` OpenMP threads.  llr is [frames][n]; hard is [frames][n]. */
int oracle_decode_llr_batch_f32(const oracle_graph* g, const float* llr, int64_t frames,
                                int max_iters, int early_term, int cn_rule, uint8_t* hard,
                                int32_t* iters, int32_t* syndrome_weight, int threads);

/* The same batch with the messages (and LLRs) STORED as 4-byte floats while every operation stays
 * double: storage 0 = double (the call above), 1 = each alpha / beta rounded to float when it is
 * written, 2 = rounded to float in units of bits (x / ln 2).  Not the reference's arithmetic: it
 * separates what float storage alone does to a decode from what float arithmetic does. */
int oracle_decode_llr_batch_f32_stored(const oracle_graph* g, const float* llr, int64_t frames,
                                       int max_iters, int early_term, int cn_rule, uint8_t* hard,
                                       int32_t* iters, int32_t* syndrome_weight, int threads, int storage);

/* Frame batch from double LLRs: what the reference itself decodes from in its BSC modes (lambda is a
 * double[] there, LogDecoder.java:45-54).  tanh_ulps / log_ulps > 0 move every tanh / log result by a
 * random whole number of ulps of at most that size (seeded per frame by noise_seed): the JDK specifies
 * Math.tanh only to 2.5 ulp and Math.log to 1 ulp, and the tests use this to count the frames whose
 * result depends on those unspecified bits.  0, 0 is the plain decode. */
int oracle_decode_llr_batch_f64(const oracle_graph* g, const double* llr, int64_t frames,
                                int max_iters, int early_term, int cn_rule, uint8_t* hard,
                                int32_t* iters, int32_t* syndrome_weight, int threads,
                                int tanh_ulps, int log_ulps, uint64_t noise_seed);

/* One frame with that noise, messages out: lets a test see that the noise reaches the messages. */
int oracle_decode_llr_noisy(const oracle_graph* g, const double* llr, int max_iters, int early_term,
                            int cn_rule, uint8_t* hard, int32_t* iters, int32_t* syndrome_weight,
                            double* alpha_out, double* beta_out, int tanh_ulps, int log_ulps,
                            uint64_t noise_seed);

/* lambda from a hard received word, LogDecoder.java:45-54 and variants. */
void oracle_bsc_llr(int m, int n, const uint8_t* r, double e, int llr_mode, double* llr);

/* Line-by-line dense mode: dense m x n message arrays allocated per call, adjacency
 * rebuilt from the dense H per call, d*(d-1) tanh per check, dense syndrome
 * (LogDecoder.java:17-123).  H is row-major int32 [m][n]; r, decoded are int32 [n]. */
int oracle_logdecoder_dense(const int32_t* H, int m, int n, const int32_t* r, double e,
                            int max_iters, int llr_mode, int cn_rule, int32_t* decoded,
                            int32_t* iters, int32_t* syndrome);

/* Batch of the dense mode over frames of hard words (the Java path's cost model). */
int oracle_logdecoder_dense_batch(const int32_t* H, int m, int n, const uint8_t* r, int64_t frames,
                                  double e, int max_iters, int llr_mode, uint8_t* hard,
                                  int32_t* iters, int32_t* syndrome, int threads);

/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator the device path uses. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Random codewords c = u * G (GenerateC.java:5-27); message bit i of frame f is bit (i mod 128)
 * of the Philox block (f, i / 128, stream 2).  G is [k][n] bytes, NULL = the all-zero word. */
void oracle_gen_codewords(int n, int k, const uint8_t* G, uint64_t seed, int64_t first_frame,
                          int64_t frames, uint8_t* cw);

/* Received hard words, frames first_frame..first_frame+frames-1; cw is the transmitted word
 * per frame ([frames][n]) or NULL for the all-zero word.
 * channel = ORACLE_CH_BSC_INFO flips only j < n-m (Channel.java:8-15), ORACLE_CH_BSC_ALL
 * flips every position (Channel.java:31-37).  r is [frames][n]. */
void oracle_gen_bsc(int m, int n, int channel, double e, uint64_t seed, int64_t first_frame,
                    int64_t frames, const uint8_t* cw, uint8_t* r);
/* BPSK/AWGN channel LLRs at Eb/N0 (dB), rate (n-m)/n. llr [frames][n]. */
void oracle_gen_awgn_llr(int m, int n, double ebn0_db, uint64_t seed, int64_t first_frame,
                         int64_t frames, const uint8_t* cw, float* llr);

/* Error accounting against the transmitted word (NULL = all-zero)
 * (LDPC_GallagerIntStreamSimu.java:136-183).  counters has 16 + max_iters entries and is
 * ACCUMULATED into.  rx_hard is the channel's hard decision per position. */
void oracle_count_errors(int m, int n, int64_t frames, const uint8_t* hard, const uint8_t* cw,
                         const uint8_t* rx_hard, const int32_t* iters, const int32_t* syndrome_weight,
                         int max_iters, uint64_t* counters);

/* Whole Monte-Carlo leg on the CPU: generate, decode, count.  G ([gen_k][n] bytes) may be NULL. */
int oracle_simulate(const oracle_graph* g, int gen_k, const uint8_t* G, int channel, double param,
                    uint64_t seed, int64_t first_frame, int64_t frames, int max_iters, int early_term,
                    int cn_rule, uint64_t* counters, int threads);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
