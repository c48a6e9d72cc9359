#!/usr/bin/env python3
"""What ordering the frames of a batch by a channel-quality measure before they are tiled would save an
early-termination run of the sweeps: a tile of 64 frames runs until its slowest frame has converged, so the cost of
a decode follows the sum over tiles of the slowest frame's iteration count.  Natural order against frames sorted by
their number of negative channel LLRs, by the sum of |LLR|, and by the iteration count itself (the bound no
predictor can beat)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

for name, code, ebn0 in (("NR-BG1-sized Z=384, 2.0 dB", L.codes.nr_bg1_shaped(384, 1)[1], 2.0),
                         ("DVB-S2-shaped n=64800, 1.5 dB", L.codes.dvbs2_like(64800, 1), 1.5),
                         ("(3,6) n=64800, 1.6 dB", L.codes.regular_ldpc(64800, 3, 6, 1), 1.6),
                         ("NR-BG1-shaped Z=32, 2.0 dB", L.codes.nr_bg1_shaped(32, 1)[1], 2.0)):
    m, n, rows, cols = code
    F = 32768
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
    it = d_it.float()
    neg = (d_llr < 0).sum(dim=1).float()
    base = it.view(-1, 64).max(dim=1).values.sum().item()
    res = ["%-30s mean iterations %.2f, tile-iterations in natural order %.0f" % (name, it.mean().item(), base)]
    for label, key in (("sorted by negative LLRs", neg), ("by sum of |LLR|", (-d_llr.abs()).sum(dim=1)), ("by the iteration count itself", it)):
        s = it[torch.argsort(key)].view(-1, 64).max(dim=1).values.sum().item()
        res.append("%s %.0f (%+.1f %%)" % (label, s, 100 * (s / base - 1)))
    res.append("correlation of negative LLRs with iterations %.2f" % torch.corrcoef(torch.stack([neg, it]))[0, 1].item())
    print("; ".join(res), flush=True)
    dec.close()
