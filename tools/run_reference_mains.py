#!/usr/bin/env python3
"""The reference's two simulation mains at their own sizes (LDPC_GallagerIntStreamSimu.java:20-30: n = 1024, (8,2),
25 matrices, 14 error rates, 10,000 frames per leg; LDPC_QCIntStreamSimu.java:20-31: z = 108, 5 x 10, 25 matrices,
19 error rates), end to end on one GPU: codes, generator matrices, every leg, the result tables.  Prints the
wall time of each and the CPU checker's time for one leg of each, for scale."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L

out = tempfile.mkdtemp()
qc_e = tuple(round(0.01 * i, 2) for i in range(1, 20))
for name, fn in (("gallager (8,2) n=1024: 25 x 14 legs", lambda: L.drivers.gallager_int_stream_simu(os.path.join(out, "g"))),
                 ("qc z=108 5x10: 25 x 19 legs", lambda: L.drivers.qc_int_stream_simu(os.path.join(out, "q"), e_values=qc_e))):
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    print("%s of 10,000 frames: %.1f s end to end (files: %s)" % (name, dt, sorted(os.listdir(out))[:3]), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    from oracle import oracle_capi as oc
    m, n, rows, cols = L.codes.gallager(1024, 8, 2, 1)
    g = oc.Graph(m, n, rows, cols)
    t0 = time.perf_counter()
    g.simulate(oc.CH_BSC_INFO, 0.03, 1, 0, 10000, 50)
    print("CPU checker (sparse, double, OpenMP), one gallager leg of 10,000 frames: %.2f s" % (time.perf_counter() - t0))
