#!/usr/bin/env python3
"""One fixed-iteration decode through the generic frame-resident kernel, for profilers:
    python tools/run_generic_profile.py qc_empty | qc_full | bg1_z32 [frames] [early] [default]
(default: the library's own choice of kernel instead of the generic frame-resident one)
qc_empty: a draw of GenerateMatrix.generateQC(108, 5, 10) (LDPC_QCIntStreamSimu.java:20-23) with exactly one
empty block (check degrees 9 - 10, variable degrees 4 - 5); qc_full: a draw without one, forced onto the generic
kernel; bg1_z32: NR-BG1-shaped Z = 32 (n = 2176, degrees to 19 / 30)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ldpc_simulator_amd as L

if os.environ.get("PERF_LIB"):  # another build of the library, for A/B runs
    L.capi.LIB_PATH = os.environ["PERF_LIB"]


def qc_draw(empties):
    for s in range(1, 500):
        base, code = L.codes.random_qc(108, 5, 10, s)
        if int((np.asarray(base) < 0).sum()) == empties:
            return code
    raise RuntimeError("no such draw")


case = sys.argv[1] if len(sys.argv) > 1 else "qc_empty"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 17
early = "early" in sys.argv[3:]
code, ebn0 = {"qc_empty": (lambda: qc_draw(1), 3.0), "qc_full": (lambda: qc_draw(0), 3.0),
              "bg1_z32": (lambda: L.codes.nr_bg1_shaped(32, 1)[1], 2.0), "bg1_z16": (lambda: L.codes.nr_bg1_shaped(16, 1)[1], 2.0),
              "r36_2040": (lambda: L.codes.regular_ldpc(2040, 3, 6, seed=4), 1.6)}[case]
m, n, rows, cols = code()
dec = L.Decoder(m, n, rows, cols)
if "default" not in sys.argv[3:]:
    dec.set_option("engine", "generic")
d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
dec.gen_awgn_llr_dev(ebn0, 1, 0, F, d_llr.data_ptr())
d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
d_it = torch.empty(F, dtype=torch.int32, device="cuda")
d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
for _ in range(2):
    dec.decode_batch_dev(d_llr.data_ptr(), F, 50, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
ms = dec.last_decode_timing()[1]
print("%s: m=%d n=%d E=%d, %d frames, 50 iterations%s: %s %.2f ms, %.3f Gbit/s, %.3f T edge-iterations/s, mean iterations %.1f" % (
    case, m, n, dec.E, F, " (early exit)" if early else "", dec.last_decode_kernel(), ms, F * (n - m) / ms / 1e6,
    F * dec.E * float(d_it.float().mean()) / ms / 1e9, float(d_it.float().mean())))
# a checksum of the outputs, to compare builds of the library with each other
import zlib
print("   outputs crc32: words %08x iterations %08x syndrome weights %08x" % (
    zlib.crc32(d_hard.cpu().numpy().tobytes()), zlib.crc32(d_it.cpu().numpy().tobytes()), zlib.crc32(d_sw.cpu().numpy().tobytes())))
