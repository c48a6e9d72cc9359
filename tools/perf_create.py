#!/usr/bin/env python3
"""Time of ldpc_amd_create and of the first decode (tables of the engine are built then) for the reference's codes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ldpc_simulator_amd as L
import oracle.oracle_capi as oc
for name, code in (("(3,6) n=4096", L.codes.regular_ldpc(4096, 3, 6, 1)), ("gallager (8,2) n=1024", L.codes.gallager(1024, 8, 2, 1)),
                   ("gallager (8,4) n=1024", L.codes.gallager(1024, 8, 4, 2)), ("qc z=108", L.codes.random_qc(108, 5, 10, 3)[1])):
    m, n, rows, cols = code
    t0 = time.perf_counter()
    dec = L.Decoder(m, n, rows, cols)
    t1 = time.perf_counter()
    llr = oc.gen_awgn_llr(m, n, 3.0, 1, 0, 64)
    dec.decode_batch(llr, 10, True)
    t2 = time.perf_counter()
    dec.decode_batch(llr, 10, True)
    t3 = time.perf_counter()
    print("%-24s create %.1f ms, first decode of 64 frames %.1f ms (%s), second %.1f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, dec.last_decode_kernel(), (t3 - t2) * 1e3))
