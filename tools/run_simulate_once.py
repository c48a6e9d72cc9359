#!/usr/bin/env python3
"""A Monte-Carlo leg through ldpc_amd_simulate on the (3,6) n=4096 code at a high Eb/N0 (for profilers):
channel generation, decode with early termination and error accounting, all on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L

ebn0 = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
dec.simulate(L.capi.CH_AWGN, ebn0, 1, 0, 1 << 16, 50, True)  # warm-up
t0 = time.perf_counter()
cnt = dec.simulate(L.capi.CH_AWGN, ebn0, 2, 0, frames, 50, True)
dt = time.perf_counter() - t0
cnt = np.asarray(cnt)
print("Eb/N0 %.1f dB: %d frames in %.1f ms = %.2f Mframes/s = %.2f Gbit/s info; frame errors %d" % (
    ebn0, frames, dt * 1e3, frames / dt / 1e6, frames * (n - m) / dt / 1e9, int(cnt[1])))
