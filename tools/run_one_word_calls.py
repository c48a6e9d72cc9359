#!/usr/bin/env python3
"""2,000 calls of ldpc_amd_decode_bsc_batch with one received word each (for profilers): Gallager (8,2) n = 1024, e = 0.006."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L
from ldpc_simulator_amd.capi import _np_ptr

m, n, rows, cols = L.codes.gallager(1024, 8, 2, 1)
dec = L.Decoder(m, n, rows, cols)
dec._sync_options()
rng = np.random.default_rng(1)
r = np.zeros((1, n), np.uint8)
r[:, : n - m] = (rng.random((1, n - m)) < 0.006).astype(np.uint8)
hard, it, sw = np.zeros((1, n), np.uint8), np.zeros(1, np.int32), np.zeros(1, np.int32)
lib, h = dec._lib, dec._h
t0 = time.perf_counter()
for _ in range(2000):
    lib.ldpc_amd_decode_bsc_batch(h, _np_ptr(r), 1, 0.006, 1, 50, 1, _np_ptr(hard), _np_ptr(it), _np_ptr(sw))
print("%.1f us per call, %d iterations" % ((time.perf_counter() - t0) / 2000 * 1e6, int(it[0])))
