#!/usr/bin/env python3
"""A/B timing of decode variants in one process (fixed 50 iterations, (3,6) n=4096)."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ldpc_simulator_amd as L

F = int(os.environ.get("SWEEP_FRAMES", 1 << 20))
iters = int(os.environ.get("SWEEP_ITERS", 50))
rounds = int(os.environ.get("SWEEP_ROUNDS", 2))
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
dec = L.Decoder(m, n, rows, cols)
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(1.0, 0x5EED, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
b_frame = iters * (4 * dec.E * 4 + n * 4)
variants = [tuple(v.split(":")) for v in os.environ.get("SWEEP_VARIANTS", "0:4:2:0:stream,0:4:2:0:lds,2:4:2:0:lds,1:4:2:0:lds").split(",")]
names = {0: "complement", 1: "minsum", 2: "tanh"}
res = {}
for r in range(rounds):
    for rule, fpl, wps, skip, eng in variants:
        # eng: "stream" = the one-launch streaming kernel, anything else = the library's choice; the skip
        # field of a variant is kept for the layout of old result files and must be 0
        assert skip == "0", "the library no longer has a switch that skips sweeps"
        layout = wps
        L.capi.set_default_options(engine="stream" if eng == "stream" else "auto", stream_kernel=1, wps=int(wps), fpl=int(fpl))
        dec.set_cn_rule(int(rule))
        dec.decode_batch_dev(d_llr.data_ptr(), F, iters, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        nl, ms, fr = dec.last_decode_timing()
        res.setdefault((rule, fpl, layout, skip, eng), []).append(ms)
for (rule, fpl, layout, skip, eng), v in res.items():
    best = min(v)
    print("engine=%-6s wps=%s skip=%s " % (eng, layout, skip), end="")
    print("rule=%-8s fpl=%s  ms=%s  best %.1f ms  %.0f GB/s  frac %.3f  %.3f Gbit/s" % (
        names[int(rule)], fpl, ["%.1f" % x for x in v], best, F * b_frame / best / 1e6, F * b_frame / best / 1e6 / 8000,
        F * (n - m) / best / 1e6))
