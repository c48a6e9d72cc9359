#!/usr/bin/env python3
"""One early-termination decode of the NR-BG1-sized irregular code (Z = 384) through the row-parallel sweeps
(for profilers): which kernels the time of an early-exit run goes to."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.nr_bg1_shaped(384, 1)[1]
dec = L.Decoder(m, n, rows, cols)
F, ITERS = 1 << 15, 50
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(2.0, 1, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
dec.decode_batch_dev(d_llr.data_ptr(), F, ITERS, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
torch.cuda.synchronize()
it = d_it.cpu().numpy()
import numpy as np
print("kernel", dec.last_decode_kernel(), "timing", dec.last_decode_timing())
print("frames", F, "mean iterations", it.mean(), "histogram", np.bincount(it, minlength=51)[1:].tolist())
