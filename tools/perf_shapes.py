#!/usr/bin/env python3
"""The frame-resident kernels of the regular shapes beyond (6,3) and (8,2), and of lengths tiled as a larger code,
against the generic frame-resident kernel on the same inputs (identical outputs are asserted)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ldpc_simulator_amd as L


def full_qc(z, mb, nb, seed):
    for s in range(seed, seed + 200):
        base, code = L.codes.random_qc(z, mb, nb, s)
        if (np.asarray(base) >= 0).all():
            return code
    raise RuntimeError


def run(name, code, F, ebn0):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    if os.environ.get("PERF_LDS_WPE"):
        dec.set_option("lds_wpe", int(os.environ["PERF_LDS_WPE"]))   # measurements: register budget of the resident kernels
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 3, 0, F, d_llr.data_ptr())
    out = {}
    for early in (False, True):
        for engine in ("auto", "generic"):
            dec.set_option("engine", engine)
            d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
            d_it = torch.empty(F, dtype=torch.int32, device="cuda")
            d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
            best = 1e9
            for _ in range(3):
                dec.decode_batch_dev(d_llr.data_ptr(), F, 50, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            out[(early, engine)] = (best, dec.last_decode_kernel(), d_hard, d_it, d_sw)
        a, b = out[(early, "auto")], out[(early, "generic")]
        same = bool((a[2] == b[2]).all() and (a[3] == b[3]).all() and (a[4] == b[4]).all())
        its = a[3].float().mean().item()
        print("%-40s %s Eb/N0 %.1f dB  %-14s %7.2f ms %6.2f Gbit/s %5.2f T edge-it/s | %-17s %7.2f ms %6.2f Gbit/s | x%.2f  same bits: %s" % (
            name, "early" if early else "fixed", ebn0, a[1], a[0], F * (n - m) / a[0] / 1e6, F * dec.E * its / a[0] / 1e9,
            b[1], b[0], F * (n - m) / b[0] / 1e6, b[0] / a[0], same), flush=True)
    dec.close()


F = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
run("gallager (8,4) n=1024", L.codes.gallager(1024, 8, 4, 2), F, 3.0)
run("qc (10,5) z=108 n=1080 (tiled 1280)", full_qc(108, 5, 10, 1), F, 3.0)
run("(3,6) n=2040 (tiled 2048)", L.codes.regular_ldpc(2040, 3, 6, 4), F, 2.0)
run("gallager (8,2) n=1000 (tiled 1024)", L.codes.gallager(1000, 8, 2, 5), F, 4.0)
run("(3,6) n=3072 (6 wavefronts)", L.codes.regular_ldpc(3072, 3, 6, 4), F // 2, 2.0)
