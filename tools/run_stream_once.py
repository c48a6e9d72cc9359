#!/usr/bin/env python3
"""One decode through the row-parallel sweeps, for profilers:
    python tools/run_stream_once.py c4|c5|c2 <frames> early|fixed [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

case, F, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
code, ebn0 = {"c4": (lambda: L.codes.nr_bg1_shaped(384, 1)[1], 2.0), "c5": (lambda: L.codes.regular_ldpc(64800, 3, 6, 1), 1.6),
              "c2": (lambda: L.codes.regular_ldpc(4096, 3, 6, seed=1), 2.0)}[case]
m, n, rows, cols = code()
dec = L.Decoder(m, n, rows, cols)
dec.set_option("engine", "stream"); dec.set_option("nodes", 0)
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for _ in range(3):
    dec.decode_batch_dev(d_llr.data_ptr(), F, iters, mode == "early", d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
print("%s %d frames %s: %.3f ms, slowest frame %d iterations (%s)" % (case, F, mode, dec.last_decode_timing()[1], int(d_it.max()), dec.last_decode_kernel()))
