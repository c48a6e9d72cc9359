"""Encoder chain (GenerateMatrix.generatorMatrix, TissueEncoder.EncodeG/EncodeH): the package's
row-vectorised code against the literal-loop restatement, and the algebraic properties the reference's
unused ConfirmG.ConG checks (G * H^T = 0, src/LDPC/ConfirmG.java:6-25)."""
import numpy as np
import pytest

from oracle import np_restatement as npr


@pytest.mark.parametrize("shape", [("gallager", 48, 8, 2, 1), ("gallager", 60, 6, 3, 2), ("regular", 36, 3, 6, 3),
                                   ("gallager", 40, 4, 2, 5)])
def test_matches_literal_restatement(pkg, shape):
    kind, n, a, b, seed = shape
    m, n, rows, cols = (pkg.codes.gallager(n, a, b, seed) if kind == "gallager" else pkg.codes.regular_ldpc(n, a, b, seed))
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    G = pkg.encoder.generator_matrix(H)
    assert (G == npr.generator_matrix_ref(H)).all()
    assert G.shape == (n - m, n)
    assert not ((G.astype(np.int64) @ H.T) % 2).any()          # ConfirmG.ConG
    try:
        eg_ref, sw_ref = npr.encode_g_ref(G)
    except TypeError:
        with pytest.raises(IndexError):
            pkg.encoder.encode_g(G)
        return
    eg, sw = pkg.encoder.encode_g(G)
    assert sw == sw_ref and (eg == eg_ref).all()
    eh = pkg.encoder.encode_h(H, sw)
    assert (eh == npr.encode_h_ref(H, sw_ref)).all()
    k = n - m
    assert (eg[:, :k] == np.eye(k, dtype=np.int32)).all()       # [I_k | P]
    assert not ((eg.astype(np.int64) @ eh.T) % 2).any()          # codewords of encodedG satisfy encodedH
    u = np.random.default_rng(0).integers(0, 2, k)
    c = npr.gene_c_ref(eg, u)
    assert (c[:k] == u).all() and not ((eh.astype(np.int64) @ c) % 2).any()


def test_reference_default_code(pkg):
    m, n, rows, cols = pkg.codes.gallager(1024, 8, 2, seed=1)
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    eh, eg = pkg.encoder.systematic_pair(H)
    assert eg.shape == (768, 1024) and (eg[:, :768] == np.eye(768, dtype=np.int32)).all()
    assert not ((eg.astype(np.int64) @ eh.T) % 2).any()
    assert (np.sort(eh.sum(axis=0)) == np.sort(H.sum(axis=0))).all() and (eh.sum(axis=1) == 8).all()


def _small_pair(pkg, n=96, a=8, b=2, seed=3):
    m, n, rows, cols = pkg.codes.gallager(n, a, b, seed)
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    eh, eg = pkg.encoder.systematic_pair(H)
    return m, n, eh, eg


def test_checker_codewords_and_counts_with_a_transmitted_word(pkg):
    """The checker's own source / channel / accounting with random codewords (GenerateC.java:5-27,
    LDPC_GallagerIntStreamSimu.java:115, :136-183): words satisfy encodedH, the message is Philox stream 2,
    and shifting everything by the codeword leaves the counters where the all-zero word puts them."""
    from oracle import oracle_capi as oc
    m, n, eh, eg = _small_pair(pkg)
    k = n - m
    cw = oc.gen_codewords(eg, seed=11, first_frame=5, frames=40)
    assert not ((cw.astype(np.int64) @ eh.T) % 2).any()
    assert cw.any() and len({bytes(c) for c in cw}) > 30
    for f in (0, 17):   # systematic: the first k bits are the message bits, bit i of Philox block (frame, i/128, 2)
        x = oc.philox4x32_10([5 + f, 0, 0, 2], [11, 0])
        u = [(int(x[(i & 127) >> 5]) >> (i & 31)) & 1 for i in range(k)]
        assert list(cw[f, :k]) == u
    assert (oc.gen_codewords(eg, 11, 25, 4) == cw[20:24]).all()      # frames are keyed by global index
    r0 = oc.gen_bsc(m, n, 1, 0.05, 7, 5, 40)
    r1 = oc.gen_bsc(m, n, 1, 0.05, 7, 5, 40, cw=cw)
    assert ((r0 ^ cw) == r1).all()
    l0 = oc.gen_awgn_llr(m, n, 2.0, 7, 5, 40)
    l1 = oc.gen_awgn_llr(m, n, 2.0, 7, 5, 40, cw=cw)
    assert ((l1 < 0) != (l0 < 0))[cw == 1].mean() > 0.5 and (l1[cw == 0] == l0[cw == 0]).all()
    rng = np.random.default_rng(1)
    hard = (rng.random((40, n)) < 0.03).astype(np.uint8)
    it = rng.integers(1, 6, 40).astype(np.int32)
    sw = rng.integers(0, 2, 40).astype(np.int32)
    a = oc.count_errors(m, n, hard, r0, it, sw, 5)
    b = oc.count_errors(m, n, hard ^ cw, r1, it, sw, 5, cw=cw)
    assert (a == b).all()


def test_checker_simulation_does_not_depend_on_the_codeword(pkg):
    """Sum-product decoding of a linear code over a symmetric channel errs on the same positions whatever
    word was sent (tanh and atanh are odd, IEEE sums mirror exactly), so a leg with random codewords must
    return the counters of the all-zero leg bit for bit on the BSC, where r = c ^ flips mirrors the LLRs
    exactly (the AWGN leg adds the same noise to -1 as to +1, which is another realisation, so only its
    statistics agree)."""
    from oracle import oracle_capi as oc
    m, n, eh, eg = _small_pair(pkg, n=192, a=8, b=2, seed=4)
    r, c = np.nonzero(eh)
    g = oc.Graph(m, n, r, c)
    for ch, e in ((0, 0.08), (1, 0.045)):
        zero = g.simulate(ch, e, 21, 0, 600, 20)
        rand = g.simulate(ch, e, 21, 0, 600, 20, G=eg)
        assert 0 < zero[1] < 600 and (zero == rand).all(), (ch, zero[:12], rand[:12])
    za = g.simulate(2, 3.0, 21, 0, 600, 20)
    ra = g.simulate(2, 3.0, 21, 0, 600, 20, G=eg)
    assert abs(int(za[1]) - int(ra[1])) < 60 and za[0] == ra[0] == 600
