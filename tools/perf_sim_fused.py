#!/usr/bin/env python3
"""ldpc_amd_simulate on the (3,6) n = 4096 code, random codewords, early termination: the leg with the error
accounting inside the resident decode kernel (option sim_fused = 1, the default) against the leg with the
counting pass of its own (0).  Same seeds, so the counters must be equal.

    python tools/perf_sim_fused.py [frames] [points]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
points = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
r2, c2 = np.nonzero(eh)
dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
dec.set_generator(eg)
if os.environ.get("PERF_SIM_CHUNK_MB"):
    dec.set_option("sim_chunk_mb", int(os.environ["PERF_SIM_CHUNK_MB"]))   # measurements: frames per chunk of a leg
dec.simulate(L.capi.CH_AWGN, 4.0, 1, 0, 1 << 17, 50, True)  # warm-up
print("frames per leg %d; M frames/s, best of 3" % frames)
for ebn0 in points:
    F = frames if ebn0 >= 2 else frames // 4
    res = {}
    for fused in (1, 0):
        dec.set_option("sim_fused", fused)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            cnt = dec.simulate(L.capi.CH_AWGN, ebn0, 2, 0, F, 50, True)
            best = min(best, time.perf_counter() - t0)
        res[fused] = (F / best / 1e6, np.asarray(cnt).copy())
    same = bool((res[0][1] == res[1][1]).all())
    print("Eb/N0 %.1f dB: accounting in the decode kernel %.2f, counting pass %.2f  (x%.3f), counters equal: %s, frame errors %d"
          % (ebn0, res[1][0], res[0][0], res[1][0] / res[0][0], same, int(res[1][1][1])), flush=True)
