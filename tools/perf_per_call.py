#!/usr/bin/env python3
"""What the literal swap costs: ldpc_amd_decode_bsc_batch called as LogDecoder.decode is called by the reference's
drivers (LDPC_GallagerIntStreamSimu.java:122: one received word per call, from the threads of a parallel stream),
on the reference's default code (Gallager (8,2) n = 1024, BSC e = 0.03, at most 50 iterations, early termination):
calls of 1 ... 4096 frames from one thread, and one frame per call from 1 ... 16 threads with a context each."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L
from ldpc_simulator_amd.capi import _np_ptr

m, n, rows, cols = L.codes.gallager(1024, 8, 2, 1)
E_RATE, MAXL = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03, 50
SMALL_KB = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # option small_call_kb: 0 = the library's default, -1 = off
L.capi.set_default_options(small_call_kb=SMALL_KB)
rng = np.random.default_rng(1)


def make(F):
    r = np.zeros((F, n), np.uint8)
    r[:, : n - m] = (rng.random((F, n - m)) < E_RATE).astype(np.uint8)
    return r


def bench_calls(dec, F, seconds=1.0, pinned=False):
    dec._sync_options()
    lib, h = dec._lib, dec._h
    if pinned:
        bufs = [L.capi.HostBuffer((F, n), np.uint8), L.capi.HostBuffer((F, n), np.uint8), L.capi.HostBuffer((F,), np.int32),
                L.capi.HostBuffer((F,), np.int32)]
        r, hard, it, sw = (b.array for b in bufs)
        r[:] = make(F)
    else:
        bufs = []
        r, hard, it, sw = make(F), np.zeros((F, n), np.uint8), np.zeros(F, np.int32), np.zeros(F, np.int32)
    pr, ph, pi, ps = _np_ptr(r), _np_ptr(hard), _np_ptr(it), _np_ptr(sw)
    for _ in range(20):
        assert lib.ldpc_amd_decode_bsc_batch(h, pr, F, E_RATE, L.capi.LLR_PARITY_LN1E20, MAXL, 1, ph, pi, ps) == 0
    calls, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            lib.ldpc_amd_decode_bsc_batch(h, pr, F, E_RATE, L.capi.LLR_PARITY_LN1E20, MAXL, 1, ph, pi, ps)
        calls += 10
    dt = time.perf_counter() - t0
    mean_it = float(it.mean())
    for b in bufs:
        b.close()
    return calls / dt, mean_it


dec = L.Decoder(m, n, rows, cols)
print("BSC e = %g, small_call_kb = %d" % (E_RATE, SMALL_KB))
print("one thread, frames per call -> calls/s, frames/s, us per call (mean iterations), pageable | page-locked buffers")
for F in (1, 8, 64, 512, 2048, 4096, 16384):
    a, mi = bench_calls(dec, F)
    b, _ = bench_calls(dec, F, pinned=True)
    print("  %5d frames per call: %9.0f calls/s %11.0f frames/s %8.1f us (%.1f it) | %9.0f calls/s %11.0f frames/s %8.1f us  [%s]" % (
        F, a, a * F, 1e6 / a, mi, b, b * F, 1e6 / b, dec.last_decode_kernel()), flush=True)
dec.close()

print("one frame per call, a context per thread (page-locked buffers) -> frames/s over all threads")
for T in (1, 2, 4, 8, 16):
    decs = [L.Decoder(m, n, rows, cols) for _ in range(T)]
    out = [0.0] * T

    def work(i):
        out[i] = bench_calls(decs[i], 1, seconds=1.0, pinned=True)[0]

    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print("  %2d threads: %9.0f frames/s" % (T, sum(out)), flush=True)
    for d in decs:
        d.close()
