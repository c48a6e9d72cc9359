#!/usr/bin/env python3
"""Streaming engine (row-parallel sweeps and the one-launch kernel) at fixed iterations on the three codes that matter
for it: algorithmic TB/s against the HBM peak.  usage: tools/perf_stream.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):   # A/B runs against another build of the library
    L.capi.LIB_PATH = os.environ["PERF_LIB"]

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def run(name, code, F, iters, ebn0, early=False, kernels=(0, 1)):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for sk in kernels:
        dec.set_option("engine", "stream")
        dec.set_option("stream_kernel", sk)
        best = 1e30
        for _ in range(2):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        it = d_it.float().mean().item()
        b = it * (4 * dec.E * 4 + n * 4) * F
        print("%-30s %-16s F=%7d early=%d mean iters %5.1f  %8.1f ms  %7.3f Gbit/s info  algorithmic %6.0f GB/s = %.3f of 8 TB/s  checksum %d" % (
            name, dec.last_decode_kernel(), F, early, it, best, F * (n - m) / best / 1e6, b / best / 1e6, b / best / 1e6 / 8000,
            int(d_hard.sum().item()) + int(d_sw.sum().item())), flush=True)
    dec.close()


c2 = L.codes.regular_ldpc(4096, 3, 6, 1)
bg = L.codes.nr_bg1_shaped(384, 1)[1]
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
run("C2 (3,6) n=4096 1 dB", c2, 1 << (17 if quick else 19), 50 if not quick else 20, 1.0)
run("C4 BG1-shaped Z=384 2 dB", bg, 1 << (14 if quick else 16), 20, 2.0, kernels=(0,))
run("C5 (3,6) n=64800 1.6 dB", c5, 1 << (13 if quick else 15), 20, 1.6, kernels=(0,))
if not quick:
    run("C4 BG1-shaped Z=384 2 dB", bg, 1 << 16, 50, 2.0, early=True, kernels=(0,))
    run("C2 (3,6) n=4096 2 dB", c2, 1 << 18, 50, 2.0, early=True, kernels=(0,))
