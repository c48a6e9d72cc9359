#!/usr/bin/env python3
"""Throughput of the other configurations (not the bench line): C1, C4, C5, and early-exit simulate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
    best = 1e30
    for _ in range(2):
        dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        nl, ms, fr = dec.last_decode_timing()
        best = min(best, ms)
    it = d_it.float().mean().item()
    b = it * (4 * dec.E * 4 + n * 4) * F
    print("%-34s %-16s F=%7d early=%d mean iters %5.1f  %8.1f ms  %7.3f Gbit/s info  algorithmic %6.0f GB/s  FER %.4f" % (
        name, dec.last_decode_kernel(), F, early, it, best, F * (n - m) / best / 1e6, b / best / 1e6,
        (d_sw > 0).float().mean().item()), flush=True)
    dec.close()


run("C1 gallager(1024,8,2) 3 dB", L.codes.gallager(1024, 8, 2, 1), 1 << 20, 50, False, 3.0)
run("C1 gallager(1024,8,2) 3 dB", L.codes.gallager(1024, 8, 2, 1), 1 << 20, 50, True, 3.0)
run("C2 (3,6) n=4096 1.0 dB", L.codes.regular_ldpc(4096, 3, 6, 1), 1 << 18, 50, True, 1.0)
run("C2 (3,6) n=4096 2.0 dB", L.codes.regular_ldpc(4096, 3, 6, 1), 1 << 20, 50, True, 2.0)
L.capi.set_default_options(engine="stream")
run("C2 (3,6) n=4096 2.0 dB [stream]", L.codes.regular_ldpc(4096, 3, 6, 1), 1 << 18, 50, True, 2.0)
L.capi.set_default_options()
_, bg = L.codes.nr_bg1_shaped(384, 1)
run("C4 BG1-shaped Z=384 2 dB", bg, 65536, 50, False, 2.0)
run("C4 BG1-shaped Z=384 2 dB", bg, 65536, 50, True, 2.0)
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
run("C5 (3,6) n=64800 1.6 dB", c5, 64, 50, True, 1.6)
run("C5 (3,6) n=64800 1.6 dB", c5, 8192, 50, True, 1.6)
run("C5 (3,6) n=64800 1.6 dB", c5, 65536, 50, False, 1.6)
