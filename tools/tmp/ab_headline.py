import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]
tag = os.environ.get("PERF_LIB", "current")[-24:]
for name, code, ebn0 in (("C2 (3,6) n=4096", L.codes.regular_ldpc(4096, 3, 6, seed=1), 1.0), ("C2 at 4 dB", L.codes.regular_ldpc(4096, 3, 6, seed=1), 4.0), ("C1 (8,2) n=1024", L.codes.gallager(1024, 8, 2, 1), 3.0)):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    F = 1 << 20
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    for early, iters in ((False, 50), (True, 50), (False, 1)):
        ts = []
        for _ in range(5):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            ts.append(dec.last_decode_timing()[1])
        print("%s %s early=%d L=%d %s: min %.2f ms median %.2f ms  checksum %d" % (tag, name, early, iters, dec.last_decode_kernel(), min(ts), sorted(ts)[2],
              int(d_hard.sum().item()) + int(d_sw.sum().item())), flush=True)
    dec.close()
    del d_llr, d_hard
