#!/usr/bin/env python3
"""Row-parallel sweeps on small batches: time per iteration against the number of waves a launch is cut into
(option rows_waves; 0 = the library's own choice), with a check that every cut returns the same bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpc_simulator_amd as L

def run(name, code, Fs, iters, ebn0, early=False, base=816):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", "stream"); dec.set_option("nodes", 0)
    for F in Fs:
        d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        ref = None
        for waves in (0, base * ((F + 63) // 64), 2048, 8192, 32768, 1 << 30):
            dec.set_option("rows_waves", waves)
            best = 1e30
            for _ in range(3):
                dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            torch.cuda.synchronize()
            out = (d_hard.clone(), d_it.clone(), d_sw.clone())
            if ref is None: ref = out
            same = all(bool((a == b).all()) for a, b in zip(ref, out))
            its = d_it.float().mean().item()
            b = its * (4 * dec.E * 4 + n * 4) * F
            print("%-20s F=%5d early=%d waves>=%6d  %8.3f ms  %7.1f us/iter  state %6.1f MB  algorithmic %6.0f GB/s  %.3f Gbit/s at these iterations  same=%s" % (
                name, F, early, waves, best, best * 1e3 / its, F * (dec.E + n) * 4 / 1e6, b / best / 1e6, F * (n - m) / best / 1e6, same), flush=True)
    dec.close()

bg = L.codes.nr_bg1_shaped(384, 1)[1]
run("C4 BG1-shaped Z=384", bg, (64, 128, 256, 512, 1024, 2048, 4096), 20, 2.0)
run("C4 BG1-shaped Z=384", bg, (64, 256, 512, 4096), 50, 2.0, early=True)
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
run("C5 (3,6) n=64800", c5, (64, 256), 50, 1.6, early=True, base=2025)
