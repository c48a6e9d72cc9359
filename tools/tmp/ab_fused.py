import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldpc_simulator_amd as L

def run(name, code, Fs, ebn0):
    m, n, rows, cols = code
    dec = L.Decoder(m, n, rows, cols)
    dec.set_option("engine", "stream"); dec.set_option("nodes", 0)
    for F in Fs:
        d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
        dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
        d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        d_it = torch.empty(F, dtype=torch.int32, device="cuda")
        d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
        res = {}
        for fused in (0, 1, 0, 1):
            dec.set_option("rows_fused", fused)
            best = 1e30
            for _ in range(2):
                dec.decode_batch_dev(d_llr.data_ptr(), F, 50, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
                best = min(best, dec.last_decode_timing()[1])
            res[fused] = min(res.get(fused, 1e30), best)
            chk = int(d_hard.sum().item()) + int(d_it.sum().item()) + int(d_sw.sum().item())
        print("%-22s F=%7d mean iters %5.2f  separate %9.3f ms  fused %9.3f ms  fused/separate %.3f  checksum %d" % (
            name, F, d_it.float().mean().item(), res[0], res[1], res[1] / res[0], chk), flush=True)
        del d_llr, d_hard
    dec.close()

bg = L.codes.nr_bg1_shaped(384, 1)[1]
run("C4 BG1 Z=384 2 dB", bg, (64, 256, 1024, 4096, 16384, 65536), 2.0)
run("C2 (3,6) n=4096 2 dB", L.codes.regular_ldpc(4096, 3, 6, 1), (4096, 65536, 262144), 2.0)
run("C5 (3,6) n=64800 1.6", L.codes.regular_ldpc(64800, 3, 6, 1), (64, 1024, 8192), 1.6)
run("BG1 Z=32 2 dB", L.codes.nr_bg1_shaped(32, 1)[1], (4096, 65536, 262144), 2.0)
