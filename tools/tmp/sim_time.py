import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
r2, c2 = np.nonzero(eh)
dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
dec.set_generator(eg)
for ebn0, ov in ((4.0, 1), (3.0, 1), (2.0, 1)):
    dec.set_option('sim_overlap', ov)
    dec.simulate(L.capi.CH_AWGN, ebn0, 1, 0, 1 << 20, 50, True)
    best = 1e30
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tot = 0
        for b in range(4):
            c = dec.simulate(L.capi.CH_AWGN, ebn0, 1, b << 20, 1 << 20, 50, True)
            tot += int(c[1])
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("overlap=%d %s %.1f dB: %.2f M frames/s (%.1f ns per frame), frame errors %d" % (ov, os.environ.get("PERF_LIB", "current")[-22:], ebn0, 4 * (1 << 20) / best / 1e6, best * 1e9 / (4 << 20), tot), flush=True)
