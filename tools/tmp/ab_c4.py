import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]
_, (m, n, rows, cols) = L.codes.nr_bg1_shaped(384, 1)
dec = L.Decoder(m, n, rows, cols)
for F, ebn0 in ((65536, 2.0), (16384, 2.0), (65536, 1.5)):
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    best = 1e30
    for _ in range(3):
        dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        best = min(best, dec.last_decode_timing()[1])
    print("%s F=%d %.1f dB early: %.1f ms mean iters %.2f checksum %d" % (os.environ.get("PERF_LIB", "current")[-24:], F, ebn0, best,
          d_it.float().mean().item(), int(d_hard.sum().item()) + int(d_it.sum().item())), flush=True)
    del d_llr, d_hard
