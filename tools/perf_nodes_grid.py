#!/usr/bin/env python3
"""Node-parallel engine: kernel time against the number of blocks in the cooperative grid (option nodes_blocks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


def run(name, code, F, iters, early, ebn0, grids):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for g in grids:
        L.capi.set_default_options(nodes_blocks=g)
        best = 1e30
        for _ in range(3):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        its = d_it.float().max().item()
        print("%-24s F=%2d early=%d blocks<=%4d %-16s %8.3f ms  %6.1f us per iteration (max iters %d)" % (
            name, F, early, g, dec.last_decode_kernel(), best, best * 1e3 / its, its), flush=True)
    L.capi.set_default_options()
    dec.close()


G = (32, 64, 128, 256, 512, 1024)
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
bg = L.codes.nr_bg1_shaped(384, 1)[1]
for F in (1, 4, 16, 32):
    run("C5 (3,6) n=64800", c5, F, 50, False, 1.6, G)
for F in (1, 8, 32):
    run("C4 BG1-shaped Z=384", bg, F, 50, False, 2.0, G)
