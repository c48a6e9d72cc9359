#!/usr/bin/env python3
"""Generic frame-resident engine against the streaming engine (and the specialised kernels where they exist)."""
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
    for eng in ("", "generic", "stream"):
        L.capi.set_default_options(engine=eng or "auto")
        best = 1e30
        for _ in range(3):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        print("%-30s E=%6d early=%d %-8s %-18s %8.1f ms %7.3f Gbit/s  mean iters %.1f FER %.4f" % (
            name, dec.E, early, eng or "default", dec.last_decode_kernel(), best, F * (n - m) / best / 1e6,
            d_it.float().mean().item(), (d_sw > 0).float().mean().item()), flush=True)
    L.capi.set_default_options()
    dec.close()


c2 = L.codes.regular_ldpc(4096, 3, 6, 1)
run("C2 (3,6) n=4096 1 dB", c2, 1 << 18, 50, False, 1.0)
run("C2 (3,6) n=4096 2 dB", c2, 1 << 18, 50, True, 2.0)
run("C1 gallager(1024,8,2) 3 dB", L.codes.gallager(1024, 8, 2, 1), 1 << 19, 50, True, 3.0)
qc = L.codes.random_qc(108, 5, 10, 1)[1]
run("QC z=108 5x10 n=1080 3 dB", qc, 1 << 18, 50, True, 3.0)
run("QC z=108 5x10 n=1080 3 dB", qc, 1 << 18, 50, False, 3.0)
bg = L.codes.nr_bg1_shaped(32, 1)[1]
run("BG1-shaped Z=32 n=2176 2 dB", bg, 1 << 17, 50, True, 2.0)
run("BG1-shaped Z=32 n=2176 2 dB", bg, 1 << 17, 20, False, 2.0)
run("(3,6) n=2048 2 dB", L.codes.regular_ldpc(2040, 3, 6, 1), 1 << 18, 50, True, 2.0)
