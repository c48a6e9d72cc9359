#!/usr/bin/env python3
"""One decode of C2 for profilers: python tools/run_c2_once.py <ebn0> <early 0|1> <max_iters> [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

ebn0, early, iters = float(sys.argv[1]), int(sys.argv[2]) != 0, int(sys.argv[3])
F = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 18
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
torch.cuda.synchronize()
print("%.1f dB early=%d max_iters=%d: %d frames, %.2f ms, mean iterations %.3f, frame-iterations %.4g" % (
    ebn0, early, iters, F, dec.last_decode_timing()[1], d_it.float().mean().item(), d_it.double().sum().item()))
