import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
m, n, rows, cols = L.codes.gallager(1024, 8, 2, 1)
dec = L.Decoder(m, n, rows, cols)
F = 1 << 20
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for ebn0 in (3.0, 5.0):
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    for early in (False, True):
        for wpe in (4, 6, 4, 6):
            dec.set_option("lds_wpe", wpe)
            ts = []
            for _ in range(3):
                dec.decode_batch_dev(d_llr.data_ptr(), F, 50, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                ts.append(dec.last_decode_timing()[1])
            print("C1 %.0f dB early=%d wpe=%d: %.2f ms  mean iters %.2f" % (ebn0, early, wpe, min(ts), d_it.float().mean().item()), flush=True)
