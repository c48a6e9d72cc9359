/*
 * ldpc_amd.h -- C ABI of the MI355X (gfx950) flooding belief-propagation LDPC decoder.
 *
 * The reference has no FFI seam; its decoder boundary is the Java static method
 *     LogDecoder.decode(int[][] encodedH, int[] r, double e, int maxL)
 *         -> DecodeResult(int[] decodedCode, int iterationNum, int syndrome)
 * (src/LDPC/LogDecoder.java:15-17; same shape in LogDecoderBefore.java:15-17 and
 * MinSumDecoder.java:9-11).  This header is what a Panama-FFM / JNI / ctypes binding of
 * that boundary binds: the parity-check matrix goes in once (create), frames go in as
 * batches, hard decisions + iteration count + syndrome weight come out per frame.
 *
 * Conventions
 *   - every function returns 0 on success and a negative LDPC_AMD_E* code on failure;
 *     ldpc_amd_last_error() gives the text.  Nothing throws or aborts across the ABI.
 *   - a context is bound to one GPU, owns its device buffers, and is not re-entrant;
 *     distinct contexts (one per GPU / per thread) are independent.
 *   - "host" entry points take host pointers and stage through the device;
 *     "_dev" entry points take device pointers on the context's GPU plus a hipStream_t
 *     (passed as void*, NULL = default stream) and only enqueue work on that stream.  The first
 *     decode of a context (and any later one that needs a larger workspace or other tables)
 *     allocates device memory with hipMalloc, which may wait for the device.
 *   - device buffers passed to "_dev" entry points must be 16-byte aligned (every hipMalloc'd base
 *     is): rows are read and written with 16-byte accesses when n is a multiple of 4.
 *   - per-frame outputs follow DecodeResult: hard[f][j] in {0,1}
 *     (estimatedCode, LogDecoder.java:104), iters[f] = l+1 at the first zero syndrome or
 *     max_iters (LogDecoder.java:118, :122), syndrome_weight[f] = number of unsatisfied
 *     checks of the returned word (sumSyndro, LogDecoder.java:108-116).
 */
#ifndef LDPC_AMD_H
#define LDPC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_AMD_VERSION 121 /* 0.1.21 */

/* error codes */
#define LDPC_AMD_OK 0
#define LDPC_AMD_EINVAL (-1)  /* bad argument */
#define LDPC_AMD_ENOMEM (-2)  /* host or device allocation failed */
#define LDPC_AMD_EHIP (-3)    /* a HIP runtime call failed */
#define LDPC_AMD_ENODEV (-4)  /* no usable gfx950 device */
#define LDPC_AMD_ELIMIT (-5)  /* the code exceeds a compiled-in limit (node degree) */

/* check-node rule */
#define LDPC_AMD_CN_SUMPRODUCT 0 /* 2*atanh(prod tanh(beta/2)), LogDecoder.java:82-95, evaluated on the
                                    complements 1 - tanh and 1 - prod: the same function, formed
                                    so that fp32 stays accurate where double does */
#define LDPC_AMD_CN_MINSUM 1     /* 0.7 * prod sign * min|beta|, MinSumDecoder.java:76-94 */
#define LDPC_AMD_CN_SUMPRODUCT_TANH 2 /* the literal fp32 tanh product / atanh of LogDecoder.java:82-95;
                                         saturates at |beta| ~ 17 instead of the reference's ~ 38 */

/* how decode_bsc_batch turns a hard received word into channel LLRs */
#define LDPC_AMD_LLR_UNIFORM 0       /* +-ln((1-e)/e) everywhere, LogDecoderBefore.java:45-50 */
#define LDPC_AMD_LLR_PARITY_LN1E20 1 /* j >= n-m get +-ln(1e20), LogDecoder.java:45-54 */
#define LDPC_AMD_LLR_PARITY_1EM7 2   /* j >= n-m get +-ln((1-1e-7)/1e-7), MinSumDecoder.java:41 */

/* channels of ldpc_amd_simulate */
#define LDPC_AMD_CH_BSC_INFO 0 /* flips on j < n-m only, Channel.java:5-17; LLRs as LogDecoder, or as
                                  MinSumDecoder (parity +-ln((1-1e-7)/1e-7)) under LDPC_AMD_CN_MINSUM */
#define LDPC_AMD_CH_BSC_ALL 1  /* flips everywhere, Channel.java:28-39; LLRs as LogDecoderBefore */
#define LDPC_AMD_CH_AWGN 2     /* BPSK + AWGN at Eb/N0 dB (not in the reference) */

/* layout of the counter vector of ldpc_amd_simulate / ldpc_amd_count_errors_dev:
 * the statistics of LDPC_GallagerIntStreamSimu.java:136-183 */
#define LDPC_AMD_CNT_FRAMES 0
#define LDPC_AMD_CNT_FRAME_ERRORS 1      /* !Arrays.equals(c, decodedC), :136 */
#define LDPC_AMD_CNT_INFO_BIT_ERRORS 2   /* residualsErrorInfoBits, :156-161 */
#define LDPC_AMD_CNT_CHECK_BIT_ERRORS 3  /* errorCheckBitsCounter, :167-172 */
#define LDPC_AMD_CNT_INFO_FRAME_ERRORS 4 /* errorInfoFrameCounter, :176 */
#define LDPC_AMD_CNT_CHECK_FRAME_ERRORS 5 /* errorCheckFrameCounter, :173 */
#define LDPC_AMD_CNT_UNDETECTED 6        /* syndrome == 0 && !correct, :182 */
#define LDPC_AMD_CNT_MISCORRECTED 7      /* errorCorrectionBits, :162-164 */
#define LDPC_AMD_CNT_ITERS_CORRECT 8     /* trueIterations[0], :148 */
#define LDPC_AMD_CNT_ITERS_WRONG 9       /* falseIterations[0], :151 */
#define LDPC_AMD_CNT_CHANNEL_ERRORS 10   /* sum over frames of info-bit channel errors, Channel.java:18-27 */
#define LDPC_AMD_CNT_CHANNEL_ERRORS_SQ 11 /* sum of their squares (for the variance, :221-228) */
#define LDPC_AMD_CNT_FIXED 16            /* histogram[it-1] starts here, :144; length max_iters */

typedef struct ldpc_amd_ctx ldpc_amd_ctx;

/* Library version (LDPC_AMD_VERSION of the build). */
int ldpc_amd_version(void);

/* Text of the last error on `ctx`, or of the last failed create when ctx is NULL. */
const char* ldpc_amd_last_error(const ldpc_amd_ctx* ctx);

/*
 * Build a decoder for the m x n matrix whose ones are the COO pairs (row_idx[t], col_idx[t]).
 * Replaces the per-frame adjacency build of LogDecoder.java:27-42: pairs may come in any
 * order; they are sorted row-major so A(i) and B(j) are ascending as the reference's dense
 * scan makes them; pairs outside the matrix are ignored and repeats collapse
 * (CheckMatrixIO.java:37-45).  device_id < 0 selects the current HIP device.
 */
int ldpc_amd_create(int32_t m, int32_t n, int64_t nnz, const int32_t* row_idx, const int32_t* col_idx,
                    int32_t device_id, ldpc_amd_ctx** out);
int ldpc_amd_destroy(ldpc_amd_ctx* ctx);

/* Shape queries: m, n, number of edges E, max check degree, max variable degree. */
int ldpc_amd_info(const ldpc_amd_ctx* ctx, int32_t* m, int32_t* n, int64_t* edges, int32_t* max_dc,
                  int32_t* max_dv);

/* Host copies of the adjacency the kernels walk: row_ptr[m+1], edge_col[E] (A(i), ascending
 * column, LogDecoder.java:35-42) and col_ptr[n+1], col_edge[E] (B(j) as CSR edge ids,
 * ascending row).  Any pointer may be NULL. */
int ldpc_amd_graph(const ldpc_amd_ctx* ctx, int32_t* row_ptr, int32_t* edge_col, int32_t* col_ptr,
                   int32_t* col_edge);

/* Check-node rule used by every later decode on this context (default sum-product). */
int ldpc_amd_set_cn_rule(ldpc_amd_ctx* ctx, int32_t cn_rule);

/* Cap on the bytes of device workspace a decode may hold (0 = default: 60 % of free HBM). */
int ldpc_amd_set_workspace_limit(ldpc_amd_ctx* ctx, uint64_t bytes);

/*
 * Per-context options.  Nothing in the library reads the process environment while decoding
 * (LDPC_AMD_VERBOSE alone is read once, by ldpc_amd_create, as the initial value of
 * LDPC_AMD_OPT_VERBOSE); which kernels run is a property of the context.  All engines return
 * identical bits; the options choose between them for measurements and tests.  The reference has
 * one decoder body (LogDecoder.java:64-120) and no counterpart of these.
 */
#define LDPC_AMD_OPT_ENGINE 1        /* LDPC_AMD_ENGINE_* below (default AUTO) */
#define LDPC_AMD_OPT_STREAM_KERNEL 2 /* streaming engine: 0 = one launch per sweep over (tile, row chunk)
                                        pairs (default), 1 = one launch per decode, a wavefront per tile */
#define LDPC_AMD_OPT_NODES 3         /* node-parallel kernel for batches of at most 32 frames: -1 = by its
                                        cost model (default), 0 = never, 1 = whenever the code allows */
#define LDPC_AMD_OPT_SIM_OVERLAP 4   /* ldpc_amd_simulate: chunks alternate between two streams (default 1) */
#define LDPC_AMD_OPT_VERBOSE 5       /* diagnostics on stderr */
#define LDPC_AMD_OPT_SIM_FUSED 6     /* ldpc_amd_simulate: the frame-resident kernels of the (6,3) and (8,2) codes count
                                        the errors themselves instead of writing hard words for a counting pass (default 1;
                                        the counters are the same either way) */
/* tuning knobs for measurements; 0 = the library's own choice */
#define LDPC_AMD_OPT_TUNE_LDS_WPE 100        /* frame-resident kernels: wavefronts per SIMD compiled for (4 or 6) */
#define LDPC_AMD_OPT_TUNE_LDS_FPW 101        /* ... frames one workgroup decodes in turn (early termination) */
#define LDPC_AMD_OPT_TUNE_GENERIC_LDS_KB 102 /* generic frame-resident kernel: LDS it may take, KiB (-1 = none) */
#define LDPC_AMD_OPT_TUNE_NODES_BLOCKS 103   /* node-parallel kernel: cap on its grid (never above what is resident) */
#define LDPC_AMD_OPT_TUNE_FPL 104            /* one-launch streaming kernel: frames per lane (1, 2, 4) */
#define LDPC_AMD_OPT_TUNE_LAYOUT 105         /* streaming arrays: 1 = [row][tile][T] instead of [tile][row][T] */
#define LDPC_AMD_OPT_TUNE_ROWS_WAVES 106     /* sweep launches: wavefronts per launch */
#define LDPC_AMD_OPT_TUNE_WPS 107            /* one-launch streaming kernel: wavefronts per SIMD (2 or 4) */
#define LDPC_AMD_OPT_TUNE_NOPIPE 108         /* one-launch streaming kernel: 1 = no software pipelining */
#define LDPC_AMD_OPT_TUNE_GENERIC_THREADS 110 /* generic frame-resident kernel: threads per frame (multiple of 64, 64..512), 0 = by code size */
#define LDPC_AMD_OPT_TUNE_LDS_VPT 111         /* frame-resident (6,3) kernels: variables per thread (8, 12 or 16; 0 = by code length) */
#define LDPC_AMD_OPT_TUNE_ROWS_FUSED 109     /* sweeps with early termination: syndrome inside the check-node launch
                                               of the next iteration (1), in launches of its own (0), -1 = the library's choice (inside, at every batch size) */
#define LDPC_AMD_OPT_TUNE_ROWS_LEAVES 112    /* sweeps at fixed iterations: 1 (default) = iterations 1 .. L-2 leave variables of
                                                degree 1 alone (same results, less traffic); 0 = every sweep touches them */
#define LDPC_AMD_OPT_TUNE_SIM_CHUNK_MB 113   /* ldpc_amd_simulate: MiB of received and decoded words per chunk of a leg; 0 (default) =
                                                a sixteenth of the free device memory, between 256 and 8192.  A leg holds two
                                                such sets when its chunks overlap, and three chunks of codewords */
#define LDPC_AMD_OPT_TUNE_NODES_FLAT_BARRIER 114 /* node-parallel kernel: 1 = every grid barrier is the one-level one (each workgroup
                                                releases and arrives at one counter); 0 (default) = workgroups meet per XCD first.
                                                Same results; the A/B also checks the two-level barrier's visibility assumptions */
#define LDPC_AMD_OPT_TUNE_SMALL_CALL_KB 115   /* host-pointer entry points: calls that move at most this many KiB in and out together run
                                                on a page-locked staging area of the context that the kernels read and write in place
                                                (one launch and one wait instead of four copies); 0 (default) = 1024, and 32768 when the
                                                caller's input is pageable memory; -1 = never; at most 65536 */
#define LDPC_AMD_OPT_TEST_NODES_OVERSIZE 190 /* tests: ask for a cooperative grid the runtime must refuse */
#define LDPC_AMD_OPT_TEST_NODES_TIMEOUT 191  /* tests: start the node-parallel kernel with its barrier time-out flag raised */
/* read-only */
#define LDPC_AMD_OPT_NODES_REFUSALS 200      /* cooperative launches the runtime refused on this context */
#define LDPC_AMD_OPT_NODES_LAST_HIP_ERROR 201 /* hipError_t of the last refusal (0 = none) */
#define LDPC_AMD_OPT_NOALLOC_MAX_FRAMES 202  /* largest batch ldpc_amd_decode_batch_dev_noalloc takes for this code */
#define LDPC_AMD_OPT_NODES_BARRIER_TIMED_OUT 203 /* 1 if a grid barrier of the most recent node-parallel launch ("bp_nodes_kernel")
                                                gave up waiting; read it after synchronising the stream of a "_dev" decode (see below) */

#define LDPC_AMD_ENGINE_AUTO 0    /* frame-resident kernels where the code fits in LDS, else streaming */
#define LDPC_AMD_ENGINE_GENERIC 1 /* as AUTO, but never the kernels specialised for (6,3) / (8,2) codes */
#define LDPC_AMD_ENGINE_STREAM 2  /* frame-major streaming through HBM for every code */

int ldpc_amd_set_option(ldpc_amd_ctx* ctx, int32_t key, int64_t value);
int ldpc_amd_get_option(const ldpc_amd_ctx* ctx, int32_t key, int64_t* value);

/*
 * Text of the last event on `ctx` that did not fail the call but that a caller should know of (e.g. the
 * runtime refused the node-parallel kernel's cooperative launch and the batch went through the sweep
 * launches instead); "" when there has been none.
 */
const char* ldpc_amd_last_warning(const ldpc_amd_ctx* ctx);

/*
 * Decode `frames` frames from channel LLRs (lambda_j > 0 <=> bit 0, LogDecoder.java:50).
 * Replaces the loop body LogDecoder.java:64-120 for a whole batch.
 *   llr      [frames][n] float
 *   early_term != 0: stop a frame at its first zero syndrome (the reference's behaviour);
 *   early_term == 0: run exactly max_iters iterations (iters[f] = max_iters).
 *   hard     [frames][n] uint8, iters [frames] int32, syndrome_weight [frames] int32
 *   (iters / syndrome_weight may be NULL).
 */
int ldpc_amd_decode_batch(ldpc_amd_ctx* ctx, const float* llr, int64_t frames, int32_t max_iters,
                          int32_t early_term, uint8_t* hard, int32_t* iters, int32_t* syndrome_weight);
int ldpc_amd_decode_batch_dev(ldpc_amd_ctx* ctx, const float* d_llr, int64_t frames, int32_t max_iters,
                              int32_t early_term, uint8_t* d_hard, int32_t* d_iters,
                              int32_t* d_syndrome_weight, void* stream);
/*
 * A "_dev" decode only enqueues work, so it cannot report what happens while that work runs.  One such event
 * exists: a batch of at most 32 frames of a code that does not fit in LDS runs in one cooperative launch whose
 * grid-wide barriers are bounded waits (about 0.2 s); if one gives up (it never has on a healthy device), the
 * launch finishes with invalid results and marks them: iters[f] = syndrome_weight[f] = -1 for every frame of
 * the batch, and ldpc_amd_get_option(LDPC_AMD_OPT_NODES_BARRIER_TIMED_OUT) returns 1.  After synchronising the
 * stream, a "_dev" caller checks either.  The host-pointer entry points, ldpc_amd_simulate and
 * ldpc_amd_last_decode_timing check for it themselves and fail with LDPC_AMD_EHIP.
 */

/*
 * ldpc_amd_decode_batch_dev for callers that cannot allocate device memory while they decode: tight
 * serving loops, harnesses that forbid hipMalloc, code under HIP graph capture (a call captured on a stream
 * replays on whatever the captured buffers hold then; the first call of a process is made outside a capture).  It has no counterpart in the
 * reference (whose decode() allocates all of its state per call, LogDecoder.java:57-58).  Same arguments,
 * same outputs bit for bit; it calls no hipMalloc / hipFree or other allocating routine.  The frame-major
 * message, LLR, decision and per-frame state arrays of the sweep kernels ("rows_cn_kernel"), which the
 * other entry points keep in a workspace they allocate, live in a static device array of
 * LDPC_AMD_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.
 * Limits: the batch must fit into that array in one piece, (E + n + ceil(n/32) + 35) * 256 bytes per 64
 * frames -- 1,984 frames of a (3,6) n = 4096 code, 192 of an NR-BG1-sized Z = 384 code, 128 of an
 * n = 64800 (3,6) code (ldpc_amd_get_option(LDPC_AMD_OPT_NOALLOC_MAX_FRAMES) gives the number for the
 * context's code); a larger batch fails with LDPC_AMD_ELIMIT and nothing is launched.  The engine options
 * are not consulted (always the sweep launches, one frame per lane); of the tuning options the sweeps' own still
 * apply (LDPC_AMD_OPT_TUNE_ROWS_WAVES, _ROWS_FUSED, _ROWS_LEAVES), with identical results for every setting.  It
 * records no timing events: ldpc_amd_last_decode_timing reports 0 launches after it.  All calls on one GPU, from any
 * context, share that GPU's array: calls that overlap in time on one GPU race, so the caller serialises
 * them (one stream, or events between streams).  The array keeps the last call's contents; what must
 * start at zero is cleared on the stream at the start of each call.
 */
#define LDPC_AMD_NOALLOC_SCRATCH_BYTES (128u << 20)
int ldpc_amd_decode_batch_dev_noalloc(ldpc_amd_ctx* ctx, const float* d_llr, int64_t frames, int32_t max_iters,
                                      int32_t early_term, uint8_t* d_hard, int32_t* d_iters,
                                      int32_t* d_syndrome_weight, void* stream);

/*
 * The reference's own calling convention: hard received words r[frames][n] (uint8, 0/1)
 * and the crossover probability e; channel LLRs are formed on the device exactly as
 * LogDecoder.java:45-54 (llr_mode LDPC_AMD_LLR_PARITY_LN1E20), LogDecoderBefore.java:45-50
 * (LDPC_AMD_LLR_UNIFORM) or MinSumDecoder.java:39-48 (LDPC_AMD_LLR_PARITY_1EM7) form them,
 * in double on the host and rounded once to float.
 */
int ldpc_amd_decode_bsc_batch(ldpc_amd_ctx* ctx, const uint8_t* r, int64_t frames, double e,
                              int32_t llr_mode, int32_t max_iters, int32_t early_term, uint8_t* hard,
                              int32_t* iters, int32_t* syndrome_weight);
int ldpc_amd_decode_bsc_batch_dev(ldpc_amd_ctx* ctx, const uint8_t* d_r, int64_t frames, double e,
                                  int32_t llr_mode, int32_t max_iters, int32_t early_term,
                                  uint8_t* d_hard, int32_t* d_iters, int32_t* d_syndrome_weight,
                                  void* stream);

/*
 * Generator matrix for random codewords.  G is [k][n] bytes (0/1, host), normally the
 * systematic matrix of TissueEncoder.encodeG (TissueEncoder.java:7-59); the library keeps a
 * bit-packed copy on the device.  k = 0 or G = NULL clears it, and the transmitted word is
 * again the all-zero word.  Replaces the per-frame "c = GenerateC.geneC(encodedG)" of
 * LDPC_GallagerIntStreamSimu.java:115 together with ldpc_amd_gen_codewords_dev.
 */
#define LDPC_AMD_MAX_GENERATOR_ROWS 10208 /* larger k fails with LDPC_AMD_ELIMIT: the codeword kernel keeps the message bits
                                            of 64 frames in one workgroup's LDS */
int ldpc_amd_set_generator(ldpc_amd_ctx* ctx, int32_t k, const uint8_t* G);

/*
 * d_cw[f][j] = (u_f * G)[j] mod 2 for frames first_frame .. first_frame+frames-1
 * (GenerateC.java:5-27).  Message bit i of frame f is bit (i mod 128) of the Philox4x32-10
 * block with counter (f, i / 128, stream 2) under `seed`; GenerateC's own Math.random() is
 * unseeded.  Without a generator matrix d_cw is filled with zeros.
 */
int ldpc_amd_gen_codewords_dev(ldpc_amd_ctx* ctx, uint64_t seed, int64_t first_frame, int64_t frames,
                               uint8_t* d_cw, void* stream);

/*
 * On-device channel, frames first_frame .. first_frame+frames-1.  d_cw is the transmitted
 * word per frame ([frames][n] uint8, device) or NULL for the all-zero word.
 * The generator is Philox4x32-10 keyed by `seed` with counter (frame, position/4, stream),
 * so a frame's noise does not depend on how frames are split over calls or GPUs.
 * Replaces Channel.GenerateR / GenerateRBefore (Channel.java:5-17, :28-39), whose
 * java.util.Random is unseeded.
 *   gen_bsc_dev : d_r   [frames][n] uint8, c[j] flipped iff u32 < floor(e * 2^32)
 *   gen_awgn_dev: d_llr [frames][n] float, 2*y/sigma^2 with y = (1 - 2 c[j]) + sigma*N(0,1),
 *                 sigma^2 = 1 / (2 * ((n-m)/n) * 10^(ebn0_db/10))
 */
int ldpc_amd_gen_bsc_dev(ldpc_amd_ctx* ctx, int32_t channel, double e, uint64_t seed, int64_t first_frame,
                         int64_t frames, const uint8_t* d_cw, uint8_t* d_r, void* stream);
int ldpc_amd_gen_awgn_llr_dev(ldpc_amd_ctx* ctx, double ebn0_db, uint64_t seed, int64_t first_frame,
                              int64_t frames, const uint8_t* d_cw, float* d_llr, void* stream);

/*
 * Accumulate the statistics of LDPC_GallagerIntStreamSimu.java:136-183 into
 * d_counters[LDPC_AMD_CNT_FIXED + max_iters] (uint64, device).  d_cw is the transmitted
 * word per frame or NULL for the all-zero word.  d_rx_hard is the channel's hard decision
 * per position ([frames][n] uint8); it may be NULL, then the miscorrected / channel-error
 * counters are left untouched.
 */
int ldpc_amd_count_errors_dev(ldpc_amd_ctx* ctx, int64_t frames, const uint8_t* d_hard, const uint8_t* d_cw,
                              const uint8_t* d_rx_hard, const int32_t* d_iters,
                              const int32_t* d_syndrome_weight, int32_t max_iters, uint64_t* d_counters,
                              void* stream);

/*
 * One Monte-Carlo leg on the GPU: generate, decode, count.  The whole frame loop of
 * LDPC_GallagerIntStreamSimu.java:99-184 for one (matrix, channel parameter) point.
 * `param` is e for the BSC channels and Eb/N0 in dB for AWGN.  counters is a HOST array
 * of LDPC_AMD_CNT_FIXED + max_iters uint64 and is ACCUMULATED into (zero it first);
 * summing the vectors of disjoint frame ranges gives the vector of their union.
 */
int ldpc_amd_simulate(ldpc_amd_ctx* ctx, int32_t channel, double param, uint64_t seed, int64_t first_frame,
                      int64_t frames, int32_t max_iters, int32_t early_term, uint64_t* counters);

/*
 * The same leg with the counters kept on the device: the counts of frames first_frame .. first_frame+frames-1 are
 * ADDED into d_counters (LDPC_AMD_CNT_FIXED + max_iters uint64 on the context's GPU) by the kernels, nothing is
 * copied to the host and the call does not wait for the GPU (apart from buffer allocations on its first use at
 * a given size).  The leg is ordered after the work already enqueued on `stream` (NULL = default stream), and
 * work enqueued on `stream` after the call is ordered after the leg, so a multi-GPU caller all-reduces
 * d_counters in place (one collective per batch instead of the reference's synchronized(lock),
 * LDPC_GallagerIntStreamSimu.java:139-183) and reads back the one word its stop rule needs.
 */
int ldpc_amd_simulate_dev(ldpc_amd_ctx* ctx, int32_t channel, double param, uint64_t seed, int64_t first_frame,
                          int64_t frames, int32_t max_iters, int32_t early_term, uint64_t* d_counters, void* stream);

/*
 * Page-locked host memory for the host-pointer entry points above.  They accept any host memory; from
 * pageable memory the runtime stages every copy (about 20 GB/s over the boundary measured), from memory
 * obtained here the copies run at link speed (55 GB/s, within 20 % of the rate with the data resident on
 * the device).  A JVM caller allocates its frame buffers once with these, wraps the addresses as off-heap
 * segments and reuses them.  The reference has no counterpart (its arrays never leave the Java heap).
 * Errors of these two calls are reported through ldpc_amd_last_error(NULL).
 */
int ldpc_amd_host_alloc(uint64_t bytes, void** out);
int ldpc_amd_host_free(void* ptr);

/*
 * Timing of the most recent decode on this context: number of launches of the
 * message-passing kernel, their summed duration in milliseconds (HIP events on the
 * stream the kernel ran on; synchronises that stream) and the frames they covered.
 */
int ldpc_amd_last_decode_timing(ldpc_amd_ctx* ctx, int32_t* launches, double* kernel_ms, int64_t* frames);

/*
 * Name of the message-passing kernel the most recent decode ran.  Frame-resident engines (one
 * workgroup per frame, the messages stay in the CU's LDS for all iterations): "bp_lds_kernel" (regular
 * (6,3) and (8,2) codes) and "bp_lds_gen_kernel" (any code whose messages fit in LDS).  Frame-major
 * streaming through HBM, for any code: "bp_nodes_kernel" (a batch of a few frames: one cooperative
 * launch, one thread per node), "rows_cn_kernel" (one launch per sweep; for the timing call
 * above, one "launch" is then the whole sequence of sweeps of a round) and "bp_decode_kernel" (all
 * iterations in one launch).  ldpc_amd_set_option chooses between them (LDPC_AMD_OPT_ENGINE,
 * LDPC_AMD_OPT_STREAM_KERNEL, LDPC_AMD_OPT_NODES).  All of them return identical bits.
 */
const char* ldpc_amd_last_decode_kernel(const ldpc_amd_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
