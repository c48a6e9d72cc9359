import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
iters = int(sys.argv[1]); early = int(sys.argv[2]); ebn0 = float(sys.argv[3])
m, n, rows, cols = L.codes.regular_ldpc(4096, 3, 6, seed=1)
dec = L.Decoder(m, n, rows, cols)
F = 1 << 19
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
for _ in range(2):
    dec.decode_batch_dev(d_llr.data_ptr(), F, iters, bool(early), d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
torch.cuda.synchronize()
print(dec.last_decode_kernel(), dec.last_decode_timing(), d_it.float().mean().item())
