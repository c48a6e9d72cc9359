#!/usr/bin/env python3
"""Early termination in the sweeps of codes the specialised kernels do not take: the syndrome in launches of its own
(rows_fused = 0) against riding in the next check-node launch (rows_fused = 1), by batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

cases = (("NR-BG1-sized Z=384", L.codes.nr_bg1_shaped(384, 1)[1], 2.0, (256, 1024, 4096, 16384, 65536)),
         ("DVB-S2-shaped n=64800", L.codes.dvbs2_like(64800, 1), 1.5, (256, 2048, 16384)),
         ("NR-BG1-shaped Z=32, streaming forced", L.codes.nr_bg1_shaped(32, 1)[1], 2.0, (4096, 131072)))
for name, (m, n, rows, cols), ebn0, sizes in cases:
    for F in sizes:
        d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        out = []
        for fused in (0, 1):
            dec = L.Decoder(m, n, rows, cols)
            dec.set_option("engine", "stream")
            dec.set_option("nodes", 0)
            dec.set_option("rows_fused", fused)
            dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
            ts = []
            for _ in range(3):
                dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                ts.append(dec.last_decode_timing()[1])
            out.append(min(ts))
            dec.close()
        print("%-38s F=%6d  own launches %8.2f ms  riding %8.2f ms  (%+.1f %%)  mean iterations %.2f" % (
            name, F, out[0], out[1], 100.0 * (out[1] / out[0] - 1.0), d_it.float().mean().item()), flush=True)
