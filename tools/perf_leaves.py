#!/usr/bin/env python3
"""Fixed-iteration sweeps on codes with variables of degree 1 (NR-BG1-sized Z = 384 and smaller): iterations
1 .. L-2 leave the leaves alone (option rows_leaves = 1, the default) against every sweep touching them (0).
Outputs must be identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


def run(name, code, F, iters, ebn0=2.0):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", "stream")
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    res = {}
    for leaves in (1, 0):
        dec.set_option("rows_leaves", leaves)
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        best = 1e9
        for _ in range(3):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        res[leaves] = (best, d_hard, d_sw)
    alg = F * iters * (4 * dec.E * 4 + n * 4)
    same = bool((res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all())
    print("%-34s %6d frames x %2d iterations: leaves left alone %8.2f ms (%.2f TB/s of the algorithmic bytes, %.3f Gbit/s), touched %8.2f ms (%.2f TB/s)  x%.3f  same bits: %s  undecoded %d"
          % (name, F, iters, res[1][0], alg / res[1][0] / 1e9, F * (n - m) / res[1][0] / 1e6, res[0][0], alg / res[0][0] / 1e9,
             res[0][0] / res[1][0], same, int((res[1][2] > 0).sum())), flush=True)
    dec.close()


run("NR-BG1-sized Z=384 n=26112", L.codes.nr_bg1_shaped(384, 1)[1], 1 << 15, 20)
run("NR-BG1-sized Z=384 n=26112", L.codes.nr_bg1_shaped(384, 1)[1], 1 << 16, 50)
run("NR-BG1-sized Z=384 n=26112", L.codes.nr_bg1_shaped(384, 1)[1], 1024, 20)
run("NR-BG1-shaped Z=32 n=2176", L.codes.nr_bg1_shaped(32, 1)[1], 1 << 17, 20)
