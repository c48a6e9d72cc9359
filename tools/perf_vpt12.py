#!/usr/bin/env python3
"""The (6,3) resident kernels with 12 variables per thread against 8 on lengths that 8 per thread tile in 3, 5 or 6
wavefronts (option lds_vpt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldpc_simulator_amd as L
for n, F in ((3072, 1 << 19), (1536, 1 << 20), (2304, 1 << 19)):
    m, n, rows, cols = L.codes.regular_ldpc(n, 3, 6, 4)
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    for ebn0, early in ((1.0, False), (2.0, True)):
        dec.gen_awgn_llr_dev(ebn0, 0x5EED, 0, F, d_llr.data_ptr())
        ref = None
        for vpt in (8, 12):
            dec.set_option("lds_vpt", vpt)
            ts = []
            for _ in range(3):
                dec.decode_batch_dev(d_llr.data_ptr(), F, 50, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                ts.append(dec.last_decode_timing()[1])
            chk = int(d_hard.sum().item()) + int(d_sw.sum().item()) + int(d_it.sum().item())
            ref = chk if ref is None else ref
            print("(3,6) n=%d %s: %2d variables per thread %8.2f ms  %6.2f Gbit/s  same bits %s  %s" % (
                n, "early 2 dB" if early else "fixed 50", vpt, min(ts), F * (n - m) / min(ts) / 1e6, chk == ref, dec.last_decode_kernel()), flush=True)
