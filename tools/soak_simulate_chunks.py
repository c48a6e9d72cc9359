#!/usr/bin/env python3
"""Soak of the Monte-Carlo leg's stream arrangement: the same 1,048,576 frames of the (3,6) n = 4096 code with random
codewords through legs of random lengths and chunk sizes from 16 MiB (about 800 frames a chunk: more than a
thousand chunks, every buffer set reused hundreds of times) to 8 GiB; every counter must equal the one-leg run's."""
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
rng = np.random.default_rng(5)
bad = 0
for ebn0 in (2.0, 3.5):
    dec.set_option("sim_chunk_mb", 8192)
    ref = np.asarray(dec.simulate(L.capi.CH_AWGN, ebn0, 3, 0, F, 50, True)).astype(np.int64)
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        mb = int(rng.choice([16, 16, 32, 64, 256, 1024]))
        cuts = [0] + sorted(int(x) for x in rng.integers(1, F, int(rng.integers(0, 5)))) + [F]
        dec.set_option("sim_chunk_mb", mb)
        tot = np.zeros_like(ref)
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                tot += np.asarray(dec.simulate(L.capi.CH_AWGN, ebn0, 3, a, b - a, 50, True)).astype(np.int64)
        same = bool((tot == ref).all())
        bad += 0 if same else 1
        print("%.1f dB  %4d MiB  legs at %s: frame errors %d info bit errors %d  %s" % (ebn0, mb, cuts[:-1], tot[1], tot[2], "same" if same else "DIFFERENT"), flush=True)
print("mismatches", bad)
sys.exit(1 if bad else 0)
