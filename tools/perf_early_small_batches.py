#!/usr/bin/env python3
"""Early-termination decodes through the row-parallel sweeps over batch sizes, with a checksum of the outputs;
PERF_LIB=<path> runs another build of the library (A/B)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpc_simulator_amd as L

if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]

cases = (("C5 (3,6) n=64800", lambda: L.codes.regular_ldpc(64800, 3, 6, 1), 1.6, (8, 64, 256, 1024)),
         ("BG1-shaped Z=384", lambda: L.codes.nr_bg1_shaped(384, 1)[1], 2.0, (64, 512, 4096, 65536)),
         ("(3,6) n=4096 sweeps", lambda: L.codes.regular_ldpc(4096, 3, 6, seed=1), 2.0, (64, 4096, 262144)),
         ("Gallager (8,2) n=1024 sweeps", lambda: L.codes.gallager(1024, 8, 2, 1), 3.0, (64, 4096, 262144)))
for name, make, ebn0, Fs in cases:
    m, n, rows, cols = make()
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", "stream"); dec.set_option("nodes", 0)
    for F in Fs:
        d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        best = 1e30
        for _ in range(4):
            dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            best = min(best, dec.last_decode_timing()[1])
        torch.cuda.synchronize()
        crc = zlib.crc32(d_hard.cpu().numpy().tobytes() + d_it.cpu().numpy().tobytes() + d_sw.cpu().numpy().tobytes())
        print("%-30s F=%6d  %9.3f ms  mean iterations %5.1f  slowest %2d  %s  crc %08x" % (
            name, F, best, d_it.float().mean().item(), int(d_it.max()), dec.last_decode_kernel(), crc), flush=True)
        del d_llr, d_hard
    dec.close()
