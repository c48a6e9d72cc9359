#!/usr/bin/env python3
"""Row-parallel sweeps: waves per launch (option rows_waves) on the irregular BG1-sized code and the long regular code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


BASE = {}


def run(name, code, F, iters, early, ebn0, settings):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for w in settings:
        L.capi.set_default_options(rows_waves=w, **BASE)
        best = 1e30
        for _ in range(2):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        bytes_ = F * d_it.float().mean().item() * (4 * dec.E * 4 + n * 4)
        print("%-24s F=%6d early=%d waves=%6d %-16s %8.1f ms %7.3f Gbit/s  algorithmic %6.0f GB/s" % (
            name, F, early, w, dec.last_decode_kernel(), best, F * (n - m) / best / 1e6, bytes_ / best / 1e6), flush=True)
    L.capi.set_default_options(**BASE)
    dec.close()


S = (4096, 8192, 1 << 18, 1 << 20, 1 << 22)
bg = L.codes.nr_bg1_shaped(384, 1)[1]
run("C4 BG1-shaped Z=384", bg, 1 << 16, 20, False, 2.0, S)
run("C4 BG1-shaped Z=384", bg, 1 << 16, 50, True, 2.0, S)
run("C4 BG1-shaped Z=384", bg, 1 << 14, 20, False, 2.0, S)
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
run("C5 (3,6) n=64800", c5, 1 << 16, 20, False, 1.6, S)
run("C5 (3,6) n=64800", c5, 64, 50, True, 1.6, S)
run("C5 (3,6) n=64800", c5, 1, 50, True, 1.6, S)
BASE = {"engine": "stream", "stream_kernel": 0}
run("C2 (3,6) n=4096 rows", L.codes.regular_ldpc(4096, 3, 6, 1), 1 << 17, 50, False, 1.0, S)
run("C2 (3,6) n=4096 rows", L.codes.regular_ldpc(4096, 3, 6, 1), 1 << 17, 50, True, 2.0, S)
run("QC z=108 rows", L.codes.random_qc(108, 5, 10, 1)[1], 1 << 16, 50, True, 3.0, S)
