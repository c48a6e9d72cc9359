#!/usr/bin/env python3
"""The counters of a Monte-Carlo leg must not depend on how the leg is cut: one leg against two half legs, several
chunk sizes (option sim_chunk_mb), repeated runs; (3,6) n = 4096, random codewords, 1.5 dB (undecoded frames present)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
r2, c2 = np.nonzero(eh)
dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
dec.set_generator(eg)
F = 1 << 20
ebn0 = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
ref = None
bad = 0
for name, mb, cuts in (("one leg, 8192 MiB", 8192, (0, F)), ("one leg, 8192 MiB, again", 8192, (0, F)), ("two half legs, 8192 MiB", 8192, (0, F // 2, F)),
                       ("one leg, 512 MiB", 512, (0, F)), ("one leg, 2048 MiB", 2048, (0, F)), ("three ragged legs, 1024 MiB", 1024, (0, 300001, 777777, F)),
                       ("one leg, automatic", 0, (0, F)), ("two half legs, automatic", 0, (0, F // 2, F))):
    dec.set_option("sim_chunk_mb", mb)
    cnt = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        c = np.asarray(dec.simulate(L.capi.CH_AWGN, ebn0, 7, a, b - a, 50, True)).astype(np.int64)
        cnt = c if cnt is None else cnt + c
    if ref is None:
        ref = cnt
    same = bool((cnt == ref).all())
    bad += 0 if same else 1
    print("%-32s frames %d frame errors %d info bit errors %d undetected %d  %s" % (name, cnt[0], cnt[1], cnt[2], cnt[6], "same" if same else "DIFFERENT " + str(np.nonzero(cnt != ref)[0].tolist())), flush=True)
print("mismatches", bad)
sys.exit(1 if bad else 0)
