#!/usr/bin/env python3
"""The device's AWGN LLRs against the oracle's double-precision generator on the same Philox words:
largest absolute and relative differences, and the time of the kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ldpc_simulator_amd as L
from oracle import oracle_capi as oc

m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, 1)
dec = L.Decoder(m, n, rows, cols)
for ebn0 in (0.0, 2.0, 4.0):
    F = 2048
    d = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 0x5EED, 12345, F, d.data_ptr())
    got = d.cpu().numpy().astype(np.float64)
    want = oc.gen_awgn_llr(m, n, ebn0, 0x5EED, 12345, F).astype(np.float64)
    diff = np.abs(got - want)
    print("%.1f dB: %d LLRs, max |difference| %.3g (at LLR %.3f), max relative to max(|LLR|, 1) %.3g, signs that differ %d, mean %.6f vs %.6f, variance %.6f vs %.6f"
          % (ebn0, got.size, diff.max(), want.flat[diff.argmax()], (diff / np.maximum(np.abs(want), 1.0)).max(),
             int((np.signbit(got) != np.signbit(want)).sum()), got.mean(), want.mean(), got.var(), want.var()))
F = 1 << 20
d = torch.empty((F, n), dtype=torch.float32, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.gen_awgn_llr_dev(2.0, 1, 0, F, d.data_ptr()); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("1,048,576 frames of n = 4096: %.2f ms, %.2f TB/s of LLR stores" % (dt * 1e3, F * n * 4 / dt / 1e12))
