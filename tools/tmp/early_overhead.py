import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
dec = L.Decoder(m, n, rows, cols)
F = 1 << 20
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for ebn0 in (4.0, 3.0, 2.0):
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    for early, iters in ((False, 1), (False, 50), (True, 1), (True, 50)):
        for fpw, wpe in ((0, 0), (0, 4)):
            dec.set_option("lds_fpw", fpw); dec.set_option("lds_wpe", wpe)
            best = 1e30
            for _ in range(3):
                dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            print("%.1f dB early=%d max_iters=%2d fpw=%2d wpe=%d %s  %8.3f ms  %6.2f ns/frame  mean iters %.2f" % (
                ebn0, early, iters, fpw, wpe, dec.last_decode_kernel(), best, best * 1e6 / F, d_it.float().mean().item()), flush=True)
