#!/usr/bin/env python3
"""Throughput of the host-pointer entry points (what a JVM caller uses): PCIe and staging included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
rng = np.random.default_rng(0)
F = 1 << 17
llr = (1.0 + rng.standard_normal((F, n), dtype=np.float32) * 0.9).astype(np.float32) * 2.5
r = (llr < 0).astype(np.uint8)
for name, fn, nbytes in (("decode_batch (float LLRs)", lambda: dec.decode_batch(llr, 50, False), 4 * n + n + 8),
                         ("decode_bsc_batch (hard words)", lambda: dec.decode_bsc_batch(r, 0.08, 50, 0, False), n + n + 8)):
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
    nl, ms, fr = dec.last_decode_timing()
    print("%-32s F=%d  %.1f ms wall  %.3f Gbit/s info  %.1f GB/s over the boundary  (last chunk: %d frames in %.1f ms of kernel)" % (
        name, F, best * 1e3, F * (n - m) / best / 1e9, F * nbytes / best / 1e9, fr, ms), flush=True)

# the same calls with page-locked caller buffers (ldpc_amd_host_alloc): what a caller that allocates its frame
# buffers through the library gets
bufs = {k: L.capi.HostBuffer(shape, dt) for k, (shape, dt) in (("llr", ((F, n), np.float32)), ("r", ((F, n), np.uint8)),
                                                                ("hard", ((F, n), np.uint8)), ("it", ((F,), np.int32)),
                                                                ("sw", ((F,), np.int32))).__iter__()}
bufs["llr"].array[:] = llr
bufs["r"].array[:] = r
from ldpc_simulator_amd.capi import _np_ptr
lib, h = dec._lib, dec._h
P = {k: _np_ptr(v.array) for k, v in bufs.items()}
for name, fn, nbytes in (
        ("decode_batch, page-locked", lambda: lib.ldpc_amd_decode_batch(h, P["llr"], F, 50, 0, P["hard"], P["it"], P["sw"]), 4 * n + n + 8),
        ("decode_bsc_batch, page-locked", lambda: lib.ldpc_amd_decode_bsc_batch(h, P["r"], F, 0.08, 0, 50, 0, P["hard"], P["it"], P["sw"]),
         n + n + 8)):
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        rc = fn()
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    print("%-32s F=%d  %.1f ms wall  %.3f Gbit/s info  %.1f GB/s over the boundary" % (
        name, F, best * 1e3, F * (n - m) / best / 1e9, F * nbytes / best / 1e9), flush=True)
for v in bufs.values():
    v.close()
