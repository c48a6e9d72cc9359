#!/usr/bin/env python3
"""One fixed-iteration decode of the reference's QC code (z=108, 5x10) through the generic frame-resident kernel (for profilers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.random_qc(108, 5, 10, 1)[1]
dec = L.Decoder(m, n, rows, cols)
F = 1 << 17
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(3.0, 1, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
dec.decode_batch_dev(d_llr.data_ptr(), F, 50, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
torch.cuda.synchronize()
print(dec.last_decode_kernel(), dec.last_decode_timing())
