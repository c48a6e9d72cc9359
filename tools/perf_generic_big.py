import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
for n in (6144, 8192):
    m, n, rows, cols = L.codes.regular_ldpc(n, 3, 6, 1)
    dec = L.Decoder(m, n, rows, cols)
    F = 1 << 17
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda"); d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for early, ebn0, iters in ((False, 1.0, 50), (True, 2.0, 50)):
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
        for kb in ("80", "160"):
            L.capi.set_default_options(generic_lds_kb=int(kb))
            best = 1e9
            for _ in range(2):
                dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            print(n, "early", early, "lds limit", kb, dec.last_decode_kernel(), "%.1f ms %.3f Gbit/s" % (best, F*(n-m)/best/1e6), flush=True)
