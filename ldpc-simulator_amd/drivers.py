"""The reference's simulation mains as functions: build or load the code, run every (matrix, channel
error rate) leg on the GPU, write the H files and the result table.

  gallager_int_stream_simu   LDPC_GallagerIntStreamSimu.java:10-293 (LogDecoder, info-bit BSC)
                             with decoder="LogDecoderBefore" the *Before variant (all-bit BSC, uniform
                             LLRs; LDPC_QCIntBefore.java:104-123), with "MinSumDecoder" the sequential
                             min-sum driver (LDPC_GallagerLoopSimu.java:85-105)
  qc_int_stream_simu         LDPC_QCIntStreamSimu.java:10-293 / LDPC_QCIntBefore.java (random QC codes)
  load_int_stream            LDPC_LoadIntStream.java:10-276 / LDPC_LoadIntBefore.java: H read from
                             `<name>-HMatrix.txt`, one table

Differences a user of the Java tool will see: frames are decoded in batches on the GPU, the random
streams are seeded (same seed -> same table, on any number of GPUs), "decode time" is the wall time of
a leg rather than the sum of per-frame timers over all worker threads, and the table is UTF-8 unless
asked otherwise.  The frame loop shards over the ranks of an initialised torch.distributed group; rank 0
writes the files.
"""
import os
import time

import numpy as np

from . import capi, check_matrix_io, codes, encoder, result_csv, sim

E_VALUES = (0.001, 0.002, 0.003, 0.004, 0.005, 0.006, 0.007, 0.008, 0.009, 0.01, 0.02, 0.03, 0.04, 0.05)  # :30

_DECODERS = {  # decoder class -> (channel, check-node rule)
    "LogDecoder": (capi.CH_BSC_INFO, capi.CN_SUMPRODUCT),
    "LogDecoderBefore": (capi.CH_BSC_ALL, capi.CN_SUMPRODUCT),
    "MinSumDecoder": (capi.CH_BSC_INFO, capi.CN_MINSUM),
}


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def _legs(H, e_values, num_frames, max_l, decoder, seed, device, random_codewords, reduce_device):
    """All legs of one matrix: systematic pair (:64-76), then one simulate per error rate (:79-213)."""
    channel, rule = _DECODERS[decoder]
    encoded_h, encoded_g = encoder.systematic_pair(H)
    r, c = np.nonzero(encoded_h == 1)
    dec = capi.Decoder(H.shape[0], H.shape[1], r, c, device)
    dec.set_cn_rule(rule)
    if random_codewords:
        dec.set_generator(encoded_g)                                # c = GenerateC.geneC(encodedG), :102
    rank, world = _rank_world()
    counters, minutes = [], []
    for i, e in enumerate(e_values):
        t0 = time.perf_counter()
        leg_seed = (int(seed) << 8) + i                             # a fresh stream per leg, as an unseeded Random gives
        counters.append(sim.simulate_sharded(
            lambda first, count: dec.simulate(channel, e, leg_seed, first, count, max_l, True),
            0, num_frames, max_l, rank, world, reduce_device))
        minutes.append((time.perf_counter() - t0) / 60.0)
    dec.close()
    return counters, minutes


def _multi_matrix(file_name, make_code, n, label_wr, label_wc, max_l, num_frames, num_cm, e_values, decoder, seed,
                  out_dir, encoding, device, random_codewords, reduce_device):
    """The loop over parity-check matrices shared by the Gallager and QC mains (:55-218) and the table."""
    rank, _ = _rank_world()
    all_counters, totals, decodes = [], [], []
    m = None
    for column in range(num_cm):
        t0 = time.perf_counter()
        m, n_, rows, cols = make_code(column)                                    # same H on every rank
        assert n_ == n
        if rank == 0:
            check_matrix_io.save_check_matrix(os.path.join(out_dir, "%s%d-HMatrix.txt" % (file_name, column)),
                                              m, n_, rows, cols)                 # :70-71
        H = check_matrix_io.to_dense(m, n_, rows, cols)
        cnt, mins = _legs(H, e_values, num_frames, max_l, decoder, (int(seed) << 16) + column, device,
                          random_codewords, reduce_device)
        all_counters.append(cnt)
        decodes.append(mins)
        totals.append((time.perf_counter() - t0) / 60.0)
    if rank == 0:
        result_csv.write_result_csv(os.path.join(out_dir, file_name + "-result.csv"), n, label_wr, label_wc, max_l,
                                    num_frames, list(e_values), all_counters, m=m, total_minutes=totals,
                                    decode_minutes=decodes, encoding=encoding)
    return all_counters


def gallager_int_stream_simu(file_name, n=1024, wr=8, wc=2, max_l=50, num_frames=10_000, num_cm=25,
                             e_values=E_VALUES, decoder="LogDecoder", seed=1, out_dir=".", encoding="utf-8",
                             device=-1, random_codewords=True, reduce_device=None):
    """Returns counters[num_cm][len(e_values)]; writes `<file_name><i>-HMatrix.txt` and
    `<file_name>-result.csv` into out_dir (rank 0)."""
    return _multi_matrix(file_name, lambda column: codes.gallager(n, wr, wc, seed=int(seed) + column), n, wr, wc,
                         max_l, num_frames, num_cm, e_values, decoder, seed, out_dir, encoding, device,
                         random_codewords, reduce_device)


def qc_int_stream_simu(file_name, z=108, mb=5, nb=10, max_l=50, num_frames=10_000, num_cm=25, e_values=E_VALUES,
                       decoder="LogDecoder", seed=1, out_dir=".", encoding="utf-8", device=-1,
                       random_codewords=True, reduce_device=None):
    """LDPC_QCIntStreamSimu.java:10-293 / LDPC_QCIntBefore.java: random quasi-cyclic codes by
    GenerateMatrix.generateQC's rules (:91-153); the table's weight columns hold nb and mb (:238-239)."""
    return _multi_matrix(file_name, lambda column: codes.random_qc(z, mb, nb, seed=int(seed) + column)[1], nb * z,
                         nb, mb, max_l, num_frames, num_cm, e_values, decoder, seed, out_dir, encoding, device,
                         random_codewords, reduce_device)


def load_int_stream(file_name, wr, wc, max_l=50, num_frames=10_000, e_values=E_VALUES, decoder="LogDecoder", seed=1,
                    out_dir=".", encoding="utf-8", device=-1, random_codewords=True, reduce_device=None):
    """H from `<file_name>-HMatrix.txt` in out_dir (LDPC_LoadIntStream.java:19, :55); wr and wc only label
    the table, as in the reference."""
    rank, _ = _rank_world()
    t0 = time.perf_counter()
    loaded = check_matrix_io.load_check_matrix(os.path.join(out_dir, file_name + "-HMatrix.txt"))
    if loaded is None:
        raise FileNotFoundError(file_name + "-HMatrix.txt")
    m, n, rows, cols = loaded
    H = check_matrix_io.to_dense(m, n, rows, cols)
    cnt, mins = _legs(H, e_values, num_frames, max_l, decoder, int(seed) << 16, device, random_codewords,
                      reduce_device)
    if rank == 0:
        result_csv.write_result_csv(os.path.join(out_dir, file_name + "-result.csv"), n, wr, wc, max_l, num_frames,
                                    list(e_values), cnt, m, total_minutes=(time.perf_counter() - t0) / 60.0,
                                    decode_minutes=mins, encoding=encoding, single=True)
    return cnt
