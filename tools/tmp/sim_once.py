import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
import ldpc_simulator_amd as L
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
eh, eg = L.encoder.systematic_pair(L.check_matrix_io.to_dense(m, n, rows, cols))
r2, c2 = np.nonzero(eh)
dec = L.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
dec.set_generator(eg)
dec.set_option('sim_overlap', int(sys.argv[1]))
dec.simulate(L.capi.CH_AWGN, 4.0, 1, 0, 1 << 20, 50, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
dec.simulate(L.capi.CH_AWGN, 4.0, 1, 1 << 20, 1 << 20, 50, True)
torch.cuda.synchronize(); print("wall for 1M frames: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
