#!/usr/bin/env python3
"""Small-batch latency of a long code (n = 64800, rate 1/2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.regular_ldpc(64800, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
for F in (1, 8, 64, 512):
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(1.6, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    ref = None
    for graph in ("-",):
        best = 1e30
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out = (d_hard.clone(), d_it.clone(), d_sw.clone())
        if ref is None:
            ref = out
        same = all(bool((a == b).all()) for a, b in zip(ref, out))
        print("F=%4d %.3f ms per batch (wall, host sync to sync)  %.3f ms/frame  mean iters %.1f  same=%s  %s" % (
            F, best * 1e3, best * 1e3 / F, d_it.float().mean().item(), same, dec.last_decode_kernel()), flush=True)
