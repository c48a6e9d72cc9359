#!/usr/bin/env python3
"""The frame-resident kernels on their workloads (A/B runs of library builds: PERF_LIB=<path to a libldpc_amd.so>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]


def run(name, code, F, iters, early, ebn0, reps=3):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 0x5EED, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(reps):
        dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        ts.append(dec.last_decode_timing()[1])
    print("%-24s %-14s early=%d  best %8.2f ms  all %s  %7.3f Gbit/s  checksum %d" % (
        name, dec.last_decode_kernel(), early, min(ts), ["%.1f" % t for t in ts], F * (n - m) / min(ts) / 1e6,
        int(d_hard.sum().item()) + int(d_sw.sum().item()) + int(d_it.sum().item())), flush=True)
    dec.close()


c1 = L.codes.gallager(1024, 8, 2, 1)
c2 = L.codes.regular_ldpc(4096, 3, 6, 1)
run("C2 1 dB fixed 50", c2, 1 << 20, 50, False, 1.0)
run("C2 2 dB early", c2, 1 << 20, 50, True, 2.0)
run("C2 4 dB early", c2, 1 << 20, 50, True, 4.0)
run("C1 3 dB fixed 50", c1, 1 << 20, 50, False, 3.0)
