import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
if os.environ.get("PERF_LIB"):
    L.capi.LIB_PATH = os.environ["PERF_LIB"]
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
dec = L.Decoder(m, n, rows, cols)
F = 1 << 20
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e30
for _ in range(4):
    e0.record(); dec.gen_awgn_llr_dev(4.0, 1, 0, F, d_llr.data_ptr()); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("%s gen_awgn 1M frames n=4096: %.3f ms = %.2f ns/frame, %.0f GB/s written, checksum %.6f" % (os.environ.get("PERF_LIB", "current")[-20:], best, best * 1e6 / F, F * n * 4 / best / 1e6, d_llr[:4096].double().sum().item()))
d_hard = torch.zeros((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.ones(F, dtype=torch.int32, device="cuda")
d_sw = torch.zeros(F, dtype=torch.int32, device="cuda")
d_cnt = torch.zeros(16 + 50, dtype=torch.int64, device="cuda")
best = 1e30
for _ in range(4):
    e0.record(); dec.count_errors_dev(F, d_hard.data_ptr(), 0, d_it.data_ptr(), d_sw.data_ptr(), 50, d_cnt.data_ptr()); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("count_errors 1M frames: %.3f ms = %.2f ns/frame" % (best, best * 1e6 / F))
