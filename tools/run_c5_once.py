#!/usr/bin/env python3
"""One early-termination decode of F frames of the (3,6) n = 64800 code for profilers: python tools/run_c5_once.py [F]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m, n, rows, cols = L.codes.regular_ldpc(64800, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(1.6, 1, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for _ in range(3):
    dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
print("%d frames: %.3f ms, slowest frame %d iterations (%s)" % (F, dec.last_decode_timing()[1], int(d_it.max()), dec.last_decode_kernel()))
