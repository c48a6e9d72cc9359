#!/usr/bin/env python3
"""One frame per workgroup (bp_lds_kernel) against two frames half an iteration apart (bp_lds2_kernel, option
lds_frames = 2) on the headline workload and other (6,3) lengths, fixed iterations; identical bits required."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L


def run(n, F, iters, ebn0, reps=3):
    m, n, rows, cols = L.codes.regular_ldpc(n, 3, 6, 1)
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    outs = []
    for name, opts in (("one frame, 8 var/thread", {}), ("two frames, 4 var/thread", {"lds_frames": 2}),
                       ("two frames, 8 var/thread", {"lds_frames": 2, "lds_vpt": 8})):
        dec = L.Decoder(m, n, rows, cols)
        for k, v in opts.items():
            dec.set_option(k, v)
        dec.gen_awgn_llr_dev(ebn0, 0x5EED, 0, F, d_llr.data_ptr())
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        ts = []
        for _ in range(reps):
            dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            ts.append(dec.last_decode_timing()[1])
        outs.append((d_hard, d_it, d_sw))
        same = all(bool((a == b).all()) for a, b in zip(outs[0], outs[-1]))
        print("n=%d F=%d L=%d %-26s %-15s best %8.2f ms  all %s  %7.3f Gbit/s  identical to the first: %s" % (
            n, F, iters, name, dec.last_decode_kernel(), min(ts), ["%.1f" % t for t in ts], F * (n - m) / min(ts) / 1e6, same), flush=True)
        dec.close()


run(4096, 1 << 20, 50, 1.0)
run(4096, (1 << 18) + 1, 7, 2.0)
run(2048, 1 << 19, 50, 1.0)
run(3072, 1 << 18, 20, 1.0)
run(1000, 1 << 18, 20, 2.0)
