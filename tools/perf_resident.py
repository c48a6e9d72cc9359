#!/usr/bin/env python3
"""Frame-resident kernel: register budget (4 or 6 wavefronts per SIMD) x configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


def run(name, code, F, iters, early, ebn0):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for wpe in ("4", "6"):
        L.capi.set_default_options(lds_wpe=int(wpe))
        best = 1e30
        for _ in range(3):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        print("%-28s early=%d wpe=%s %8.1f ms %7.3f Gbit/s" % (name, early, wpe, best, F * (n - m) / best / 1e6), flush=True)
    dec.close()


c1 = L.codes.gallager(1024, 8, 2, 1)
c2 = L.codes.regular_ldpc(4096, 3, 6, 1)
run("C1 3 dB", c1, 1 << 20, 50, False, 3.0)
run("C1 3 dB", c1, 1 << 20, 50, True, 3.0)
run("C1 5 dB", c1, 1 << 20, 50, True, 5.0)
run("C2 1 dB", c2, 1 << 19, 50, False, 1.0)
run("C2 1 dB", c2, 1 << 18, 50, True, 1.0)
run("C2 2 dB", c2, 1 << 20, 50, True, 2.0)
