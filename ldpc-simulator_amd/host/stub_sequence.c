/*
 * The native calls of a Java (Panama FFM) binding of LogDecoder.decode, in its order, from C:
 * what a JVM caller does to the library over a run of the reference's driver
 * (LDPC_GallagerIntStreamSimu.java:55-76, :122).
 *
 *   - one context per matrix (ldpc_amd_create from the COO pairs of the dense scan, LogDecoder.java:27-42),
 *     kept in a least-recently-used cache of MAX_CTX entries: a fifth matrix destroys the oldest context;
 *   - frame and result buffers from ldpc_amd_host_alloc, allocated once and reused by every call, grown once;
 *   - ldpc_amd_decode_bsc_batch per batch, with the reference's LLR rule (LogDecoder.java:45-54);
 *   - ldpc_amd_last_error on failure; ldpc_amd_destroy for every cached context and ldpc_amd_host_free at the end.
 *
 * The matrices are (8,2) Gallager-style codes built from a linear congruential generator, the received words
 * random bits at a crossover probability of 0.02 from the same generator; per decode call the program prints
 * "call <i> matrix <k> frames <F> checksum <c> iters <sum> weight <sum>", which the test compares with the
 * same calls made through the Python binding.
 *
 *   gcc -std=c11 -Wall -Wextra -Werror stub_sequence.c -I ../../include -L .. -lldpc_amd -Wl,-rpath,$PWD/.. -o stub_sequence
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldpc_amd.h"

#define MAX_CTX 4
#define MATRICES 6

static uint64_t lcg_state;
static uint32_t lcg(void) {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(lcg_state >> 33);
}

/* (8,2) code: block 0 has row i = columns [8i, 8i+8); block 1 is a column permutation of it (GenerateMatrix.java:11-34) */
static void build_matrix(int n, uint64_t seed, int32_t* rows, int32_t* cols) {
    const int m2 = n / 8;
    int* perm = (int*)malloc((size_t)n * sizeof(int));
    lcg_state = seed;
    for (int j = 0; j < n; j++) perm[j] = j;
    for (int j = n - 1; j > 0; j--) {
        const int k = (int)(lcg() % (uint32_t)(j + 1));
        const int t = perm[j]; perm[j] = perm[k]; perm[k] = t;
    }
    int t = 0;
    for (int i = 0; i < m2; i++)
        for (int k = 0; k < 8; k++) { rows[t] = i; cols[t] = 8 * i + k; t++; }
    for (int i = 0; i < m2; i++)
        for (int k = 0; k < 8; k++) { rows[t] = m2 + i; cols[t] = perm[8 * i + k]; t++; }
    free(perm);
}

struct entry { int matrix; ldpc_amd_ctx* ctx; unsigned long used; };
static struct entry cache[MAX_CTX];
static unsigned long tick;
static int created, destroyed;

static ldpc_amd_ctx* ctx_for(int matrix, int n) {
    int free_slot = -1, oldest = 0;
    for (int s = 0; s < MAX_CTX; s++) {
        if (cache[s].ctx && cache[s].matrix == matrix) { cache[s].used = ++tick; return cache[s].ctx; }
        if (!cache[s].ctx && free_slot < 0) free_slot = s;
        if (cache[s].used < cache[oldest].used) oldest = s;
    }
    if (free_slot < 0) { /* evict the least recently used context */
        if (ldpc_amd_destroy(cache[oldest].ctx) != LDPC_AMD_OK) exit(3);
        destroyed++;
        cache[oldest].ctx = NULL;
        free_slot = oldest;
    }
    const int nnz = 2 * n;
    int32_t* rows = (int32_t*)malloc((size_t)nnz * sizeof(int32_t));
    int32_t* cols = (int32_t*)malloc((size_t)nnz * sizeof(int32_t));
    build_matrix(n, 1000u + (uint64_t)matrix, rows, cols);
    ldpc_amd_ctx* c = NULL;
    const int rc = ldpc_amd_create(n / 4, n, nnz, rows, cols, -1, &c);
    free(rows); free(cols);
    if (rc != LDPC_AMD_OK) { fprintf(stderr, "create: %d %s\n", rc, ldpc_amd_last_error(NULL)); exit(2); }
    created++;
    cache[free_slot].matrix = matrix; cache[free_slot].ctx = c; cache[free_slot].used = ++tick;
    return c;
}

static uint8_t *buf_in, *buf_hard;
static int32_t *buf_it, *buf_sw;
static long buf_frames, buf_n;
static int allocations;

static void ensure_buffers(long frames, long n) {
    if (frames * n <= buf_frames * buf_n && frames <= buf_frames) return;
    ldpc_amd_host_free(buf_in); ldpc_amd_host_free(buf_hard); ldpc_amd_host_free(buf_it); ldpc_amd_host_free(buf_sw);
    if (frames > buf_frames) buf_frames = frames;
    if (n > buf_n) buf_n = n;
    if (ldpc_amd_host_alloc((uint64_t)(buf_frames * buf_n), (void**)&buf_in) || ldpc_amd_host_alloc((uint64_t)(buf_frames * buf_n), (void**)&buf_hard) ||
        ldpc_amd_host_alloc((uint64_t)(4 * buf_frames), (void**)&buf_it) || ldpc_amd_host_alloc((uint64_t)(4 * buf_frames), (void**)&buf_sw)) {
        fprintf(stderr, "host_alloc: %s\n", ldpc_amd_last_error(NULL));
        exit(4);
    }
    allocations++;
}

int main(void) {
    const int n = 1024, maxL = 50;
    const double e = 0.02;
    /* the order in which a driver with a cache of four contexts meets six matrices, with returns to recent ones */
    const int order[] = {0, 1, 0, 2, 3, 1, 4, 0, 5, 4, 2};
    const long frames_of[] = {64, 200, 1, 200, 333, 64, 200, 17, 1000, 200, 64};
    for (unsigned call = 0; call < sizeof(order) / sizeof(order[0]); call++) {
        const int k = order[call];
        const long F = frames_of[call];
        ldpc_amd_ctx* ctx = ctx_for(k, n);
        ensure_buffers(F, n);
        lcg_state = 77u + call;
        for (long t = 0; t < F * n; t++) buf_in[t] = (t % n < n - n / 4 && lcg() % 1000u < 20u) ? 1 : 0; /* flips on information bits */
        const int rc = ldpc_amd_decode_bsc_batch(ctx, buf_in, F, e, LDPC_AMD_LLR_PARITY_LN1E20, maxL, 1, buf_hard, buf_it, buf_sw);
        if (rc != LDPC_AMD_OK) { fprintf(stderr, "decode: %d %s\n", rc, ldpc_amd_last_error(ctx)); return 5; }
        uint64_t sum = 1469598103934665603ull;
        long long its = 0, sws = 0;
        for (long t = 0; t < F * n; t++) sum = (sum ^ buf_hard[t]) * 1099511628211ull;
        for (long f = 0; f < F; f++) { its += buf_it[f]; sws += buf_sw[f]; }
        printf("call %u matrix %d frames %ld checksum %llu iters %lld weight %lld\n", call, k, F, (unsigned long long)sum, its, sws);
    }
    for (int s = 0; s < MAX_CTX; s++)
        if (cache[s].ctx) { if (ldpc_amd_destroy(cache[s].ctx) != LDPC_AMD_OK) return 6; destroyed++; }
    if (ldpc_amd_host_free(buf_in) || ldpc_amd_host_free(buf_hard) || ldpc_amd_host_free(buf_it) || ldpc_amd_host_free(buf_sw)) return 7;
    printf("contexts created %d destroyed %d buffer allocations %d\n", created, destroyed, allocations);
    return created == destroyed ? 0 : 8;
}
