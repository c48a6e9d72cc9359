#!/usr/bin/env python3
"""A Monte-Carlo leg through ldpc_amd_simulate on the (3,6) n=4096 code at a high Eb/N0 (for profilers):
channel generation, decode with early termination and error accounting, all on the device:
    python tools/run_simulate_once.py [Eb/N0] [frames] [codewords]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L

ebn0 = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
if "codewords" in sys.argv[3:]:  # random codewords through the systematic generator instead of the all-zero word
    eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
    r2, c2 = np.nonzero(eh)
    dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
    dec.set_generator(eg)
else:
    dec = L.Decoder(m, n, rows, cols)
if os.environ.get("SIM_CHUNK_MB"):  # staging memory per set of buffers (option sim_chunk_mb)
    dec.set_option("sim_chunk_mb", int(os.environ["SIM_CHUNK_MB"]))
for kv in os.environ.get("SIM_OPTIONS", "").split():  # e.g. SIM_OPTIONS="sim_overlap=0 sim_fused=0"
    name, value = kv.split("=")
    dec.set_option(name, int(value))
dec.simulate(L.capi.CH_AWGN, ebn0, 1, 0, frames, 50, True)  # warm-up: buffers of the measured leg's size
t0 = time.perf_counter()
cnt = dec.simulate(L.capi.CH_AWGN, ebn0, 2, 0, frames, 50, True)
dt = time.perf_counter() - t0
cnt = np.asarray(cnt)
print("Eb/N0 %.1f dB: %d frames in %.1f ms = %.2f Mframes/s = %.2f Gbit/s info; frame errors %d" % (
    ebn0, frames, dt * 1e3, frames / dt / 1e6, frames * (n - m) / dt / 1e9, int(cnt[1])))
