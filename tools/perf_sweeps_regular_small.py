#!/usr/bin/env python3
"""Row-parallel sweeps on regular codes over batch sizes, early termination (50 iterations at most) and 12 fixed
iterations, with a checksum of the outputs; PERF_LIB=<path> runs another build of the library (A/B)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpc_simulator_amd as L

if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]
for name, code, ebn0, Fs in (("(3,6) n=64800", L.codes.regular_ldpc(64800, 3, 6, 1), 1.6, (1, 8, 64, 256, 512, 4096)),
                             ("(3,6) n=4096", L.codes.regular_ldpc(4096, 3, 6, seed=1), 2.0, (64, 1024, 1984, 4096, 16384, 65536)),
                             ("(8,2) n=1024", L.codes.gallager(1024, 8, 2, 1), 3.0, (64, 1024, 16384, 65536))):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", "stream"); dec.set_option("nodes", 0)
    for F in Fs:
        d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda"); d_it = torch.empty(F, dtype=torch.int32, device="cuda"); d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        for early in (True, False):
            best = 1e30
            for _ in range(4):
                dec.decode_batch_dev(d_llr.data_ptr(), F, 50 if early else 12, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            torch.cuda.synchronize()
            crc = zlib.crc32(d_hard.cpu().numpy().tobytes() + d_it.cpu().numpy().tobytes() + d_sw.cpu().numpy().tobytes())
            print("%-16s F=%5d %-22s %8.3f ms  crc %08x" % (name, F, "early termination" if early else "12 fixed iterations", best, crc), flush=True)
    dec.close()
