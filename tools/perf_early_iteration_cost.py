#!/usr/bin/env python3
"""Cost of an iteration of the resident (6,3) kernel with early termination against fixed iterations, where nearly no
frame converges (Eb/N0 0 and 1 dB: every frame runs 47 to 50 iterations).  PERF_LIB=<library> measures another build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
F = 262144
for ebn0, early in ((1.0, True), (1.0, False), (0.0, True)):
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 0x5EED, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(3):
        dec.decode_batch_dev(d_llr.data_ptr(), F, 50, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        ts.append(dec.last_decode_timing()[1])
    its = d_it.float().mean().item()
    print("%s Eb/N0 %.1f early=%d: %.2f ms, mean iterations %.2f, %.3f ns per frame-iteration" % (os.environ.get("PERF_LIB", "shipped")[-12:], ebn0, early, min(ts), its, min(ts) * 1e6 / F / its), flush=True)
