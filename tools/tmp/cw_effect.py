import os, sys
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
F = 1 << 20
d_cw = torch.empty((F, n), dtype=torch.uint8, device="cuda")
dec.gen_codewords_dev(1, 0, F, d_cw.data_ptr())
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for ebn0 in (4.0, 2.0):
    for name, cw in (("all-zero", 0), ("random", d_cw.data_ptr())):
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr(), d_cw=cw)
        for Fsub in (F, 104815):
            ts = []
            for _ in range(4):
                dec.decode_batch_dev(d_llr.data_ptr(), Fsub, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                ts.append(dec.last_decode_timing()[1])
            print("%s %.1f dB %s codewords, %d frames: %.3f ms = %.2f ns per frame, mean iters %.2f" % (
                os.environ.get("PERF_LIB", "current")[-22:], ebn0, name, Fsub, min(ts), min(ts) * 1e6 / Fsub, d_it[:Fsub].float().mean().item()), flush=True)
