#!/usr/bin/env python3
"""Frame-resident kernel: frames per workgroup (option lds_fpw) and waves per SIMD (option lds_wpe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

F = 1 << 20
for name, code, cases in (("C2", L.codes.regular_ldpc(4096, 3, 6, 1), ((1.0, False, 50), (1.0, False, 1), (4.0, True, 50), (2.0, True, 50))),
                          ("C1", L.codes.gallager(1024, 8, 2, 1), ((3.0, False, 50), (5.0, True, 50), (3.0, True, 50)))):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for ebn0, early, iters in cases:
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
        for wpe in (4, 6):
            line = "%s %.1f dB early=%d max_iters=%2d wpe=%d:" % (name, ebn0, early, iters, wpe)
            for fpw in (1, 2, 4, 8, 16):
                L.capi.set_default_options(lds_wpe=wpe, lds_fpw=fpw)
                best = 1e30
                for _ in range(3):
                    dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                    best = min(best, dec.last_decode_timing()[1])
                line += "  fpw=%d %7.2f ms" % (fpw, best)
            print(line, flush=True)
    dec.close()
