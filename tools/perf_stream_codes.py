#!/usr/bin/env python3
"""The streaming sweeps on the codes that do not live in LDS, and on C2 with the streaming engine forced:
fixed iterations and early exit (A/B runs of library builds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]


def run(name, code, F, iters, early, ebn0, engine="auto"):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", engine)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(3):
        dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        ts.append(dec.last_decode_timing()[1])
    it = d_it.float().mean().item()
    b = it * (4 * dec.E * 4 + n * 4) * F
    best = min(ts)
    print("%-30s %-15s F=%6d early=%d iters %5.1f  %8.2f ms  %6.3f Gbit/s  algorithmic %5.0f GB/s  checksum %d" % (
        name, dec.last_decode_kernel(), F, early, it, best, F * (n - m) / best / 1e6, b / best / 1e6,
        int(d_hard.sum().item()) + int(d_sw.sum().item()) + int(d_it.sum().item())), flush=True)
    dec.close()


bg = L.codes.nr_bg1_shaped(384, 1)[1]
run("C4 BG1-sized Z=384 2 dB", bg, 65536, 50, False, 2.0)
run("C4 BG1-sized Z=384 2 dB", bg, 32768, 10, False, 2.0)
run("C4 BG1-sized Z=384 2 dB", bg, 65536, 50, True, 2.0)
run("C4 BG1-sized Z=384 2 dB", bg, 1024, 50, True, 2.0)
run("C4 BG1-sized Z=384 2 dB", bg, 64, 50, True, 2.0)
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
run("C5 (3,6) n=64800 1.6 dB", c5, 65536, 50, False, 1.6)
run("C5 (3,6) n=64800 1.6 dB", c5, 8192, 50, True, 1.6)
run("C5 (3,6) n=64800 1.6 dB", c5, 64, 50, True, 1.6)
dv = L.codes.dvbs2_like(64800, 1)
run("DVB-S2-shaped n=64800 1.5 dB", dv, 32768, 20, False, 1.5)
run("DVB-S2-shaped n=64800 1.5 dB", dv, 8192, 50, True, 1.5)
c2 = L.codes.regular_ldpc(4096, 3, 6, 1)
run("C2 (3,6) n=4096 1 dB stream", c2, 1 << 18, 50, False, 1.0, "stream")
run("C2 (3,6) n=4096 2 dB stream", c2, 1 << 18, 50, True, 2.0, "stream")
run("BG1-shaped Z=32 stream", L.codes.nr_bg1_shaped(32, 1)[1], 1 << 17, 20, False, 2.0, "stream")
