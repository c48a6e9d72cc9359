"""Generator of the committed fixtures in this directory.

The reference has no golden vectors, cannot run here (no JVM) and seeds nothing, so these
vectors come from the independent NumPy restatement of its decoders
(oracle/np_restatement.py, double precision).  Each case stores the parity-check matrix in the
reference's own text format (CheckMatrixIO.java:8-24), the inputs and the expected
(decodedCode, iterationNum, syndrome) triples.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.txt and cases.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import ldpc_simulator_amd as L  # noqa: E402
from oracle import np_restatement as npr  # noqa: E402


def main():
    cases = []
    rng = np.random.default_rng(20240607)

    def add_bsc(name, code, hfile, mode, e, max_l, frames, rule="sum-product", info_only=True):
        m, n, rows, cols = code
        L.check_matrix_io.save_check_matrix(os.path.join(HERE, hfile), m, n, rows, cols)
        H = L.check_matrix_io.to_dense(m, n, rows, cols)
        rs, out = [], []
        for _ in range(frames):
            r = (rng.random(n) < e).astype(np.uint8)
            if info_only:
                r[n - m:] = 0  # Channel.GenerateR: parity part error free
            hard, it, sw = npr.log_decoder(H, r, e, max_l, mode=mode, rule=rule)
            rs.append(r.tolist())
            out.append({"hard": hard.tolist(), "iters": int(it), "syndrome": int(sw)})
        cases.append({"name": name, "kind": "bsc", "H": hfile, "llr_mode": mode, "rule": rule, "e": e,
                      "max_iters": max_l, "r": rs, "expected": out})

    def add_llr(name, code, hfile, sigma_mean, max_l, frames, early=True):
        m, n, rows, cols = code
        L.check_matrix_io.save_check_matrix(os.path.join(HERE, hfile), m, n, rows, cols)
        H = L.check_matrix_io.to_dense(m, n, rows, cols)
        llrs, out = [], []
        for _ in range(frames):
            llr = rng.normal(sigma_mean[1], sigma_mean[0], n).astype(np.float32)
            hard, it, sw = npr.decode_llr(H, llr.astype(np.float64), max_l, early)
            llrs.append([float(x) for x in llr])
            out.append({"hard": hard.tolist(), "iters": int(it), "syndrome": int(sw)})
        cases.append({"name": name, "kind": "llr", "H": hfile, "rule": "sum-product", "early_term": early,
                      "max_iters": max_l, "llr": llrs, "expected": out})

    gal = L.codes.gallager(96, 8, 2, seed=3)
    add_bsc("gallager96_logdecoder_e0.05", gal, "gallager_96_8_2.txt", "ln1e20", 0.05, 20, 6)
    add_bsc("gallager96_logdecoderbefore_e0.03", gal, "gallager_96_8_2.txt", "uniform", 0.03, 20, 6, info_only=False)
    add_bsc("gallager96_minsum_e0.04", gal, "gallager_96_8_2.txt", "1e-7", 0.04, 20, 6, rule="min-sum")
    reg = L.codes.regular_ldpc(120, 3, 6, seed=9)
    add_llr("regular120_awgn_like", reg, "regular_120_3_6.txt", (1.6, 2.2), 15, 6)
    add_llr("regular120_noisy_fixed_iters", reg, "regular_120_3_6.txt", (1.5, 0.9), 6, 4, early=False)
    _, bg = L.codes.nr_bg1_shaped(z=2, seed=4)
    add_llr("bg1shaped_z2_irregular", bg, "bg1_shaped_z2.txt", (2.0, 3.0), 12, 5)
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (oracle/np_restatement.py, float64)",
                   "cases": cases}, f)
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
