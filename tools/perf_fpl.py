#!/usr/bin/env python3
"""Streaming engine at 1, 2 and 4 frames per lane (fixed iterations): what the width of a row access is worth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


def run(name, code, F, iters, ebn0):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for rows_mode in ("0", "1"):
        for fpl in ("1", "2", "4"):
            L.capi.set_default_options(engine="stream", stream_kernel=1 - int(rows_mode), fpl=int(fpl))
            best = 1e30
            for _ in range(2):
                dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            bytes_ = F * iters * (4 * dec.E * 4 + n * 4)
            print("%-24s rows=%s fpl=%s %-18s %8.1f ms %7.3f Gbit/s  algorithmic %6.0f GB/s  FER %.4f" % (
                name, rows_mode, fpl, dec.last_decode_kernel(), best, F * (n - m) / best / 1e6, bytes_ / best / 1e6,
                (d_sw > 0).float().mean().item()), flush=True)
    L.capi.set_default_options()
    dec.close()


run("C2 (3,6) n=4096 1 dB", L.codes.regular_ldpc(4096, 3, 6, 1), 1 << 18, 50, 1.0)
run("C5 (3,6) n=64800 1.6 dB", L.codes.regular_ldpc(64800, 3, 6, 1), 1 << 14, 20, 1.6)
bg = L.codes.nr_bg1_shaped(384, 1)[1]
run("C4 BG1-shaped Z=384", bg, 1 << 15, 20, 2.0)
