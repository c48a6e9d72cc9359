"""Committed golden vectors (tests/golden/, made by the NumPy restatement) against the C
oracle on the CPU and against the HIP path on the GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

MODES = {"uniform": 0, "ln1e20": 1, "1e-7": 2}
RULES = {"sum-product": 0, "min-sum": 1}


def _cases():
    with open(os.path.join(GOLD, "cases.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_golden(case, oracle, pkg):
    m, n, rows, cols = pkg.check_matrix_io.load_check_matrix(os.path.join(GOLD, case["H"]))
    g = oracle.Graph(m, n, rows, cols)
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    for f, exp in enumerate(case["expected"]):
        if case["kind"] == "bsc":
            r = np.array(case["r"][f], np.uint8)
            lam = oracle.bsc_llr(m, n, r, case["e"], MODES[case["llr_mode"]])
            got = g.decode_llr(lam, case["max_iters"], True, RULES[case["rule"]])
            dense = oracle.logdecoder_dense(H, r.astype(np.int32), case["e"], case["max_iters"],
                                            MODES[case["llr_mode"]], RULES[case["rule"]])
            assert dense[0].tolist() == exp["hard"] and dense[1] == exp["iters"] and dense[2] == exp["syndrome"]
        else:
            lam = np.array(case["llr"][f], np.float64)
            got = g.decode_llr(lam, case["max_iters"], case["early_term"])
        assert got[0].tolist() == exp["hard"] and got[1] == exp["iters"] and got[2] == exp["syndrome"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_gpu_reproduces_golden(case, pkg):
    m, n, rows, cols = pkg.check_matrix_io.load_check_matrix(os.path.join(GOLD, case["H"]))
    dec = pkg.Decoder(m, n, rows, cols)
    dec.set_cn_rule(RULES[case["rule"]])
    if case["kind"] == "bsc":
        r = np.array(case["r"], np.uint8)
        hard, it, sw = dec.decode_bsc_batch(r, case["e"], case["max_iters"], MODES[case["llr_mode"]], True)
    else:
        llr = np.array(case["llr"], np.float32)
        hard, it, sw = dec.decode_batch(llr, case["max_iters"], case["early_term"])
    for f, exp in enumerate(case["expected"]):
        assert hard[f].tolist() == exp["hard"], (case["name"], f)
        assert it[f] == exp["iters"] and sw[f] == exp["syndrome"], (case["name"], f)
