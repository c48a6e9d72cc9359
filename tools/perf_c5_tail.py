#!/usr/bin/env python3
"""C5 (n = 64800, 64 frames, early termination): does the tail of launches after the last frame has converged cost time?
The same batch with max_iters = 50 and with max_iters just above the slowest frame's count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.regular_ldpc(64800, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
for F in (64, 256, 8):
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(1.6, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
    worst = int(d_it.max())
    for L_ in (50, worst + 1, worst):
        best_wall, best_k = 1e9, 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dec.decode_batch_dev(d_llr.data_ptr(), F, L_, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            torch.cuda.synchronize(); best_wall = min(best_wall, time.perf_counter() - t0)
            best_k = min(best_k, dec.last_decode_timing()[1])
        print("%4d frames, slowest frame %d iterations, max_iters %2d: wall %.3f ms, launches between the events %.3f ms (%s)" % (
            F, worst, L_, best_wall * 1e3, best_k, dec.last_decode_kernel()), flush=True)
