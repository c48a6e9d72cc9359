#!/usr/bin/env python3
"""Node-parallel cooperative engine against the row-parallel sweeps on very small batches: identical outputs, latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


def run(name, code, F, iters, early, ebn0):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    res = {}
    for eng in ("nodes", "rows"):
        L.capi.set_default_options(nodes=1 if eng == "nodes" else 0)
        d_hard = torch.full((F, n), 7, dtype=torch.uint8, device="cuda")
        d_it = torch.full((F,), -1, dtype=torch.int32, device="cuda")
        d_sw = torch.full((F,), -1, dtype=torch.int32, device="cuda")
        best, wall = 1e30, 1e30
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            torch.cuda.synchronize()
            wall = min(wall, (time.perf_counter() - t0) * 1e3)
            best = min(best, dec.last_decode_timing()[1])
        res[eng] = (d_hard.cpu(), d_it.cpu(), d_sw.cpu())
        print("%-26s F=%3d early=%d %-6s %-16s kernel %8.3f ms  wall %8.3f ms  mean iters %.1f  unsatisfied frames %d" % (
            name, F, early, eng, dec.last_decode_kernel(), best, wall, d_it.float().mean().item(), int((d_sw > 0).sum())), flush=True)
    L.capi.set_default_options()
    same = all(bool((a == b).all()) for a, b in zip(res["nodes"], res["rows"]))
    print("    identical outputs:", same, flush=True)
    dec.close()
    return same


ok = True
bg = L.codes.nr_bg1_shaped(384, 1)[1]
c5 = L.codes.regular_ldpc(64800, 3, 6, 1)
for F in (1, 8, 32):
    ok &= run("C4 BG1-shaped Z=384 2 dB", bg, F, 50, True, 2.0)
ok &= run("C4 BG1-shaped Z=384 1 dB", bg, 5, 12, False, 1.0)
ok &= run("C4 BG1-shaped Z=384 0.5 dB", bg, 7, 30, True, 0.5)
for F in (1, 4, 8, 12, 16, 24, 32):
    ok &= run("C5 (3,6) n=64800 1.6 dB", c5, F, 50, True, 1.6)
ok &= run("C5 (3,6) n=64800 1.0 dB", c5, 3, 20, True, 1.0)
dvb = L.codes.dvbs2_like(64800, 1)
for F in (1, 4, 8, 16, 32):
    ok &= run("C5 DVB-S2-shaped n=64800 1.5 dB", dvb, F, 50, True, 1.5)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
