import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L
m, n, rows, cols = L.codes.gallager(1024, 8, 2, 1)
dec = L.Decoder(m, n, rows, cols)
dec.set_option("verbose", 1)
F = 1 << 20
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
dec.gen_awgn_llr_dev(3.0, 1, 0, F, d_llr.data_ptr())
for _ in range(2):
    dec.decode_batch_dev(d_llr.data_ptr(), F, 50, False, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
torch.cuda.synchronize()
print(dec.last_decode_kernel(), dec.last_decode_timing())
