#!/usr/bin/env python3
"""Frame-resident kernel: per-frame fixed cost (load, initialise, write out) from runs of 1, 2, 3 ... iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
F = 1 << 20
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for ebn0, early, iters_list in ((1.0, False, (1, 2, 3, 5, 10)), (4.0, True, (50,)), (3.0, True, (50,)), (2.0, True, (50,))):
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    for iters in iters_list:
        best = 1e30
        for _ in range(3):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        print("Eb/N0 %.1f early=%d max_iters=%2d %-14s %8.2f ms per %d frames, mean iters %.2f" % (
            ebn0, early, iters, dec.last_decode_kernel(), best, F, d_it.float().mean().item()), flush=True)
