#!/usr/bin/env python3
"""One process, no process group: the shares that three ranks would decode of two batches of the 2.5 dB point
of the sweep, one after the other through ldpc_amd_simulate_dev, summed - must equal the one-rank counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ldpc_simulator_amd as L

m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
r2, c2 = np.nonzero(eh)
dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
dec.set_generator(eg)
total = torch.zeros(L.capi.CNT_FIXED + 50, dtype=torch.int64, device="cuda")
acc = L.sim.device_accumulator(dec, L.capi.CH_AWGN, 2.5, 0x5EED, 50, True)
for batch in range(2):
    for rank in range(3):
        start, share = L.sim.shard_range(batch << 20, 1 << 20, rank, 3)
        print("leg", start, share, flush=True)
        acc(start, share, total)
        torch.cuda.synchronize()
c = total.cpu().numpy()
print("frames", int(c[0]), "frame errors", int(c[1]), "undetected", int(c[6]))
assert int(c[0]) == 2097152 and int(c[1]) == 200
