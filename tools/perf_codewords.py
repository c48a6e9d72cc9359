#!/usr/bin/env python3
"""Time of the codeword generator (u * G, 64 frames per block) alone, (3,6) n = 4096 and the reference's
(8,2) n = 1024 shape, against the bytes it writes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ldpc_simulator_amd as L


def run(name, code, F):
    m, n, rows, cols = code
    eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
    r2, c2 = np.nonzero(eh)
    dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
    dec.set_generator(eg)
    d_cw = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(4):
        e0.record()
        dec.gen_codewords_dev(1, 0, F, d_cw.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-20s k=%5d %8d frames  %8.3f ms  %6.2f ns per frame  %6.1f GB/s written" % (
        name, eg.shape[0], F, best, best * 1e6 / F, F * n / best / 1e6), flush=True)
    dec.close()


run("(3,6) n=4096", L.codes.regular_ldpc(4096, 3, 6, seed=1), 1 << 20)
run("(8,2) n=1024", L.codes.gallager(1024, 8, 2, seed=1), 1 << 20)
