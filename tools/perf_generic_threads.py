#!/usr/bin/env python3
"""Generic frame-resident kernel: threads per frame (option generic_threads) against time, fixed 50 iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ldpc_simulator_amd as L


def qc_draw(pred):
    for s in range(1, 4000):
        base, code = L.codes.random_qc(108, 5, 10, s)
        if pred(np.asarray(base) < 0):
            return code


cases = (("QC z=108, two empty blocks in one row", qc_draw(lambda e: e.sum() == 2 and (e.sum(axis=1) == 2).any()), 3.0, 1 << 17),
         ("QC z=108 full, generic kernel", qc_draw(lambda e: e.sum() == 0), 3.0, 1 << 17),
         ("BG1-shaped Z=32", L.codes.nr_bg1_shaped(32, 1)[1], 2.0, 1 << 16),
         ("BG1-shaped Z=16", L.codes.nr_bg1_shaped(16, 4)[1], 2.0, 1 << 17),
         ("(3,6) n=2040 generic", L.codes.regular_ldpc(2040, 3, 6, 4), 1.6, 1 << 17),
         ("BG1-shaped Z=40", L.codes.nr_bg1_shaped(40, 1)[1], 2.0, 1 << 15),
         ("BG1-shaped Z=48", L.codes.nr_bg1_shaped(48, 1)[1], 2.0, 1 << 15),
         ("BG1-shaped Z=64", L.codes.nr_bg1_shaped(64, 1)[1], 2.0, 1 << 15),
         ("BG1-shaped Z=96", L.codes.nr_bg1_shaped(96, 1)[1], 2.0, 1 << 14),
         ("(3,4) n=4800 irregular-length", L.codes.regular_ldpc(4800, 3, 4, 2), 4.0, 1 << 15),
         ("(3,6) n=8192 generic", L.codes.regular_ldpc(8192, 3, 6, 4), 1.6, 1 << 15))
if len(sys.argv) > 1:
    cases = [c for c in cases if any(a in c[0] for a in sys.argv[1:])]
for name, (m, n, rows, cols), ebn0, F in cases:
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", "generic")
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    out = []
    for nt in (0, 128, 192, 256, 512) if len(sys.argv) > 1 else (0, 64, 128, 192, 256, 320, 384, 448, 512):
        dec2 = L.Decoder(m, n, rows, cols)
        dec2.set_option("engine", "generic")
        dec2.set_option("generic_threads", nt)
        try:
            best = 1e30
            for _ in range(2):
                dec2.decode_batch_dev(d_llr.data_ptr(), F, 50, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec2.last_decode_timing()[1])
            out.append("%d: %.1f ms (%s)" % (nt, best, dec2.last_decode_kernel()[3:10]))
        except L.LdpcError as e:
            out.append("%d: %s" % (nt, e.code))
        dec2.close()
    print("%-40s n=%d E=%d F=%d  %s" % (name, n, dec.E, F, "  ".join(out)), flush=True)
    dec.close()
