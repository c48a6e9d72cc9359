"""Pins for the CPU oracle.  The reference has no tests or golden vectors
(parity unpinned by the reference), so the pins are closed-form known answers, invariants
of LogDecoder.java, and agreement of three independently written restatements:
the sparse C mode, the dense line-by-line C mode and the NumPy restatement."""
import math

import numpy as np
import pytest

from oracle import np_restatement as npr


def _dense(m, n, rows, cols):
    H = np.zeros((m, n), np.int32)
    H[rows, cols] = 1
    return H


def test_closed_form_single_check(oracle):
    # H = [1 1 1], n-m = 2, r = 000, e = 0.1: lambda = (ln 9, ln 9, ln 1e20).
    # tanh(ln9 / 2) = 0.8, tanh(ln(1e20)/2) = 1.0 exactly in double, so
    # alpha->v0 = 2 atanh(0.8 * 1.0) = ln 9, alpha->v2 = 2 atanh(0.64) = ln(1.64/0.36).
    H = np.array([[1, 1, 1]], np.int32)
    g = oracle.Graph(1, 3, [0, 0, 0], [0, 1, 2])
    lam = oracle.bsc_llr(1, 3, np.zeros(3, np.uint8), 0.1, oracle.LLR_PARITY_LN1E20)
    assert lam[0] == pytest.approx(math.log(9.0), rel=1e-15)
    assert lam[2] == pytest.approx(math.log(1e20), rel=1e-15)
    hard, it, sw, alpha, beta = g.decode_llr(lam, 5, True, messages=True)
    assert hard.tolist() == [0, 0, 0] and it == 1 and sw == 0
    assert alpha[0] == pytest.approx(math.log(9.0), rel=1e-12)
    assert alpha[1] == pytest.approx(math.log(9.0), rel=1e-12)
    assert alpha[2] == pytest.approx(math.log(1.64 / 0.36), rel=1e-12)
    dec, it2, sy2 = oracle.logdecoder_dense(H, [0, 0, 0], 0.1, 5)
    assert dec.tolist() == [0, 0, 0] and it2 == 1 and sy2 == 0


def test_all_parity_neighbours_give_infinity(oracle):
    # check 0 = {v1, v2, v3}, all three are parity positions (n-m = 1): every other
    # member has tanh = 1.0 exactly, so the product is 1.0 and 2*atanh(1.0) = +Inf
    # (LogDecoder.java:9-11, :93).  A flipped parity bit gives -Inf.
    H = np.array([[0, 1, 1, 1], [1, 1, 0, 0], [1, 0, 1, 0]], np.int32)
    r, c = np.nonzero(H)
    g = oracle.Graph(3, 4, r, c)
    lam = oracle.bsc_llr(3, 4, np.array([0, 0, 0, 0], np.uint8), 0.05, oracle.LLR_PARITY_LN1E20)
    _, _, _, alpha, _ = g.decode_llr(lam, 1, True, messages=True)
    assert np.isposinf(alpha[0]) and np.isposinf(alpha[1]) and np.isposinf(alpha[2])
    lam2 = oracle.bsc_llr(3, 4, np.array([0, 1, 0, 0], np.uint8), 0.05, oracle.LLR_PARITY_LN1E20)
    _, _, _, alpha2, _ = g.decode_llr(lam2, 1, False, messages=True)
    assert np.isposinf(alpha2[0]) and np.isneginf(alpha2[1]) and np.isneginf(alpha2[2])


def test_nan_posterior_decodes_to_zero(oracle):
    # v0 sits in two degree-2 checks whose other members are parity bits received as 0 and 1:
    # it gets +Inf from one and -Inf from the other, gamma = NaN, and NaN < 0 is false
    # (LogDecoder.java:104) -> bit 0.
    H = np.array([[1, 1, 0], [1, 0, 1]], np.int32)
    r, c = np.nonzero(H)
    g = oracle.Graph(2, 3, r, c)
    lam = oracle.bsc_llr(2, 3, np.array([1, 0, 1], np.uint8), 0.1, oracle.LLR_PARITY_LN1E20)
    hard, it, sw = g.decode_llr(lam, 3, True)
    assert hard[0] == 0
    dec, it2, sy2 = oracle.logdecoder_dense(H, [1, 0, 1], 0.1, 3)
    assert dec.tolist() == hard.tolist() and it2 == it and sy2 == sw
    h3, it3, sw3 = npr.log_decoder(H, [1, 0, 1], 0.1, 3)
    assert h3.tolist() == hard.tolist() and it3 == it and sw3 == sw


def test_max_iters_zero(oracle):
    g = oracle.Graph(1, 3, [0, 0, 0], [0, 1, 2])
    hard, it, sw = g.decode_llr(np.array([-1.0, -2.0, -3.0]), 0, True)
    assert hard.tolist() == [0, 0, 0] and it == 0 and sw == 0  # LogDecoder.java:61, :122


def test_error_free_codeword_one_iteration(oracle, pkg):
    m, n, rows, cols = pkg.codes.gallager(96, 8, 2, seed=3)
    g = oracle.Graph(m, n, rows, cols)
    for mode in (oracle.LLR_UNIFORM, oracle.LLR_PARITY_LN1E20):
        lam = oracle.bsc_llr(m, n, np.zeros(n, np.uint8), 0.02, mode)
        hard, it, sw = g.decode_llr(lam, 50, True)
        assert not hard.any() and it == 1 and sw == 0


def test_degree_one_check_forces_zero(oracle):
    # a check with one member: the empty product is 1.0, so alpha = +Inf and the bit is 0
    g = oracle.Graph(1, 2, [0], [0])
    hard, it, sw, alpha, _ = g.decode_llr(np.array([-5.0, 3.0]), 2, True, messages=True)
    assert np.isposinf(alpha[0]) and hard.tolist() == [0, 0] and it == 1 and sw == 0


def test_adjacency_order_and_coo_permutation(oracle, pkg):
    m, n, rows, cols = pkg.codes.regular_ldpc(60, 3, 6, seed=5)
    g = oracle.Graph(m, n, rows, cols)
    rp, ec = g.csr()
    cp, ce = g.csc()
    H = _dense(m, n, rows, cols)
    A, B = npr.adjacency(H)
    for i in range(m):
        assert ec[rp[i]:rp[i + 1]].tolist() == A[i]
    edge_row = np.repeat(np.arange(m), np.diff(rp))
    for j in range(n):
        assert edge_row[ce[cp[j]:cp[j + 1]]].tolist() == B[j]
    # permuted, duplicated and out-of-range COO input builds the same graph (CheckMatrixIO.java:37-45)
    rng = np.random.default_rng(0)
    perm = rng.permutation(rows.size)
    r2 = np.concatenate([rows[perm], rows[:5], [m, 0]])
    c2 = np.concatenate([cols[perm], cols[:5], [0, n]])
    g2 = oracle.Graph(m, n, r2, c2)
    rp2, ec2 = g2.csr()
    assert g2.E == g.E and (rp2 == rp).all() and (ec2 == ec).all()


@pytest.mark.parametrize("mode,npmode", [(0, "uniform"), (1, "ln1e20")])
def test_three_restatements_agree(oracle, pkg, mode, npmode):
    m, n, rows, cols = pkg.codes.gallager(48, 8, 2, seed=11)
    H = _dense(m, n, rows, cols)
    g = oracle.Graph(m, n, rows, cols)
    rng = np.random.default_rng(7)
    for trial in range(12):
        e = [0.02, 0.05, 0.1, 0.15][trial % 4]
        r = (rng.random(n) < e).astype(np.uint8)
        if mode == 1:
            r[n - m:] = 0  # Channel.GenerateR keeps the parity part error free
        lam = oracle.bsc_llr(m, n, r, e, mode)
        h1, it1, sw1, a1, b1 = g.decode_llr(lam, 8, True, messages=True)
        d2, it2, sy2 = oracle.logdecoder_dense(H, r.astype(np.int32), e, 8, llr_mode=mode)
        h3, it3, sw3, a3, b3 = npr.decode_llr(H, npr.channel_llr(m, n, r, e, npmode), 8, True, trace=True)
        assert h1.tolist() == d2.tolist() == h3.tolist()
        assert it1 == it2 == it3 and sw1 == sy2 == sw3
        # messages of the two sparse-independent restatements: a few ulp (different libm tanh/log)
        rp, ec = g.csr()
        k = 0
        for i in range(m):
            for j in ec[rp[i]:rp[i + 1]]:
                x, y = a1[k], float(a3[(i, j)])
                assert (np.isnan(x) and np.isnan(y)) or x == y or abs(x - y) <= 1e-9 * max(1.0, abs(x))
                k += 1


def test_minsum_restatements_agree(oracle, pkg):
    m, n, rows, cols = pkg.codes.gallager(48, 8, 2, seed=2)
    H = _dense(m, n, rows, cols)
    g = oracle.Graph(m, n, rows, cols)
    rng = np.random.default_rng(3)
    for trial in range(6):
        e = 0.06
        r = (rng.random(n) < e).astype(np.uint8)
        r[n - m:] = 0
        lam = oracle.bsc_llr(m, n, r, e, oracle.LLR_PARITY_1EM7)
        h1, it1, sw1 = g.decode_llr(lam, 10, True, cn_rule=oracle.CN_MINSUM)
        d2, it2, sy2 = oracle.logdecoder_dense(H, r.astype(np.int32), e, 10, llr_mode=oracle.LLR_PARITY_1EM7,
                                               cn_rule=oracle.CN_MINSUM)
        h3, it3, sw3 = npr.log_decoder(H, r, e, 10, mode="1e-7", rule="min-sum")
        assert h1.tolist() == d2.tolist() == h3.tolist()
        assert it1 == it2 == it3 and sw1 == sy2 == sw3


def test_syndrome_weight_is_popcount_of_H_times_word(oracle, pkg):
    m, n, rows, cols = pkg.codes.regular_ldpc(120, 3, 6, seed=9)
    H = _dense(m, n, rows, cols)
    g = oracle.Graph(m, n, rows, cols)
    rng = np.random.default_rng(1)
    llr = (rng.normal(0.6, 1.4, size=(16, n))).astype(np.float32)  # noisy: many frames fail
    hard, it, sw = g.decode_llr_batch(llr, 6, True, threads=2)
    for f in range(16):
        w = int(((H @ hard[f].astype(np.int64)) % 2).sum())
        assert w == sw[f]
        assert (it[f] == 6) if w else (1 <= it[f] <= 6)
    assert (sw > 0).any()


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert oracle.philox4x32_10(ctr, key).tolist() == want


def test_channel_generators(oracle):
    m, n = 24, 96
    r = oracle.gen_bsc(m, n, oracle.CH_BSC_INFO, 0.1, seed=5, first_frame=0, frames=400)
    assert not r[:, n - m:].any()                       # Channel.java:15: parity part copied
    assert 0.07 < r[:, : n - m].mean() < 0.13
    # a frame's noise depends on its global index only, not on the batch split
    r2 = oracle.gen_bsc(m, n, oracle.CH_BSC_INFO, 0.1, seed=5, first_frame=100, frames=50)
    assert (r2 == r[100:150]).all()
    ra = oracle.gen_bsc(m, n, oracle.CH_BSC_ALL, 0.1, seed=5, first_frame=0, frames=400)
    assert (ra[:, : n - m] == r[:, : n - m]).all() and 0.07 < ra[:, n - m:].mean() < 0.13
    llr = oracle.gen_awgn_llr(m, n, 2.0, seed=9, first_frame=0, frames=2000)
    sigma2 = 1.0 / (2.0 * 0.75 * 10 ** 0.2)
    assert abs(llr.mean() - 2.0 / sigma2) < 0.05 * 2.0 / sigma2
    assert abs(llr.var() - 4.0 / sigma2) < 0.05 * 4.0 / sigma2
    assert np.isfinite(llr).all()


def test_count_errors_matches_python(oracle):
    m, n, F, L = 8, 24, 50, 5
    rng = np.random.default_rng(4)
    hard = (rng.random((F, n)) < 0.05).astype(np.uint8)
    hard[::3] = 0
    rx = (rng.random((F, n)) < 0.2).astype(np.uint8)
    it = rng.integers(1, L + 1, F).astype(np.int32)
    sw = rng.integers(0, 2, F).astype(np.int32)
    c = oracle.count_errors(m, n, hard, rx, it, sw, L)
    k = n - m
    correct = ~hard.any(axis=1)
    assert c[0] == F and c[1] == (~correct).sum()
    assert c[2] == hard[:, :k].sum() and c[3] == hard[:, k:].sum()
    assert c[4] == hard[:, :k].any(axis=1).sum() and c[5] == hard[:, k:].any(axis=1).sum()
    assert c[6] == ((sw == 0) & ~correct).sum()
    assert c[7] == ((rx[:, :k] == 0) & (hard[:, :k] == 1)).sum()
    assert c[8] == it[correct].sum() and c[9] == it[~correct].sum()
    ch = rx[:, :k].sum(axis=1)
    assert c[10] == ch.sum() and c[11] == (ch.astype(np.int64) ** 2).sum()
    assert c[16:].tolist() == np.bincount(it - 1, minlength=L).tolist()
