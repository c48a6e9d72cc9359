"""The node-parallel cooperative engine (very small batches of codes that do not live in LDS): against the
double-precision oracle, and bit for bit against the row-parallel sweeps on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def library_defaults(pkg):
    pkg.capi.set_default_options()
    yield
    pkg.capi.set_default_options()


def _syndrome_weight_sparse(m, rows, cols, hard):
    w = np.zeros((hard.shape[0], m), np.int64)
    np.add.at(w, (slice(None), rows), hard[:, cols].astype(np.int64))
    return (w % 2).sum(axis=1)


def _both(dec, call):
    """The same decode through the node-parallel engine and through the sweep launches."""
    dec.set_option("nodes", 1)
    a = call()
    assert dec.last_decode_kernel() == "bp_nodes_kernel"
    # the one-level grid barrier instead of the per-XCD one: a difference would mean that one of them lets a
    # workgroup read messages another has not made visible yet
    dec.set_option("nodes_flat_barrier", 1)
    flat = call()
    assert dec.last_decode_kernel() == "bp_nodes_kernel"
    dec.set_option("nodes_flat_barrier", 0)
    for x, y in zip(a, flat):
        assert (x == y).all(), "the two grid barriers of bp_nodes_kernel return different bits"
    dec.set_option("nodes", 0)
    b = call()
    assert dec.last_decode_kernel() == "rows_cn_kernel"
    dec.set_option("nodes", -1)
    return a, b


def test_small_irregular_code_against_the_oracle(pkg, oracle):
    """NR-BG1-shaped code at a small lifting size (degrees 3..19 / 1..30), kept out of the resident
    engines: every check-node rule, LLR and hard-word input, 1 to 32 frames, early exit and fixed."""
    _, (m, n, rows, cols) = pkg.codes.nr_bg1_shaped(z=8, seed=3)
    g = oracle.Graph(m, n, rows, cols)
    dec = pkg.Decoder(m, n, rows, cols)
    dec.set_option("generic_lds_kb", -1)
    for F, ebn0, early, L in ((1, 2.0, True, 30), (5, 1.0, True, 20), (32, 2.5, True, 40), (32, 1.0, False, 6),
                              (7, 3.0, True, 1), (3, 0.0, False, 1)):
        llr = oracle.gen_awgn_llr(m, n, ebn0, seed=21 + F, first_frame=0, frames=F)
        got, rows_out = _both(dec, lambda: dec.decode_batch(llr, L, early))
        for x, y in zip(got, rows_out):
            assert (x == y).all(), (F, early, L)
        want = g.decode_llr_batch(llr, L, early)
        for x, y in zip(got, want):
            assert (x == y).all(), (F, early, L)
        assert (_syndrome_weight_sparse(m, rows, cols, got[0]) == got[2]).all()
    # hard received words with the reference's parity LLRs (LogDecoder.java:45-54)
    r = oracle.gen_bsc(m, n, oracle.CH_BSC_INFO, 0.03, seed=5, first_frame=0, frames=20)
    got, rows_out = _both(dec, lambda: dec.decode_bsc_batch(r, 0.03, 50, oracle.LLR_PARITY_LN1E20, True))
    for x, y in zip(got, rows_out):
        assert (x == y).all()
    lam = np.stack([oracle.bsc_llr(m, n, r[f], 0.03, oracle.LLR_PARITY_LN1E20) for f in range(20)]).astype(np.float32)
    want = g.decode_llr_batch(lam, 50, True)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all() and (got[2] == want[2]).all()
    # normalised min-sum (MinSumDecoder.java:79-93)
    dec.set_cn_rule(pkg.capi.CN_MINSUM)
    llr = oracle.gen_awgn_llr(m, n, 2.0, seed=9, first_frame=0, frames=16)
    got, rows_out = _both(dec, lambda: dec.decode_batch(llr, 25, True))
    for x, y in zip(got, rows_out):
        assert (x == y).all()
    assert (_syndrome_weight_sparse(m, rows, cols, got[0]) == got[2]).all()
    dec.close()


def test_bg1_sized_and_long_codes_identical_to_the_sweeps(pkg):
    """The two codes this engine is for (too large for LDS): identical outputs, and the engine is the
    default choice for a single frame."""
    import torch
    for code, ebn0, L in ((pkg.codes.nr_bg1_shaped(z=384, seed=1)[1], 1.5, 30),
                          (pkg.codes.regular_ldpc(64800, 3, 6, 1), 1.6, 40)):
        m, n, rows, cols = code
        dec = pkg.Decoder(m, n, rows, cols)
        for F, early in ((1, True), (6, True), (4, False)):
            d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
            dec.gen_awgn_llr_dev(ebn0, 17, 0, F, d_llr.data_ptr())
            llr = d_llr.cpu().numpy()
            a, b = _both(dec, lambda: dec.decode_batch(llr, L if early else 7, early))
            for x, y in zip(a, b):
                assert (x == y).all(), (n, F, early)
            assert (_syndrome_weight_sparse(m, rows, cols, a[0]) == a[2]).all()
            if early:
                assert (a[1] < L).any()
        dec.decode_batch(llr[:1], L, True)
        assert dec.last_decode_kernel() == "bp_nodes_kernel"
        dec.close()


def test_refused_cooperative_launch_is_observable_and_not_final(pkg):
    """A cooperative grid the runtime refuses (here: four times what can be resident, asked for through the
    test option) must not fail the decode, must not be silent, and must not turn the engine off for good."""
    m, n, rows, cols = pkg.codes.nr_bg1_shaped(z=384, seed=1)[1]
    dec = pkg.Decoder(m, n, rows, cols)
    import torch
    F = 32
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(2.0, 3, 0, F, d_llr.data_ptr())
    llr = d_llr.cpu().numpy()
    dec.set_option("nodes", 1)
    want = dec.decode_batch(llr, 12, True)
    assert dec.last_decode_kernel() == "bp_nodes_kernel"
    assert dec.get_option("nodes_refusals") == 0 and dec.last_warning() == ""
    dec.set_option("test_nodes_oversize", 1)
    got = dec.decode_batch(llr, 12, True)
    assert dec.last_decode_kernel() == "rows_cn_kernel"          # the batch went through the sweeps
    assert dec.get_option("nodes_refusals") == 1
    assert dec.get_option("nodes_last_hip_error") != 0
    assert "refused" in dec.last_warning() and "bp_nodes_kernel" in dec.last_warning()
    for x, y in zip(got, want):
        assert (x == y).all()
    dec.set_option("test_nodes_oversize", 0)
    again = dec.decode_batch(llr, 12, True)
    assert dec.last_decode_kernel() == "bp_nodes_kernel"         # one refusal does not disable the engine
    for x, y in zip(again, want):
        assert (x == y).all()
    dec.close()


def test_a_grid_barrier_that_gave_up_is_an_error_everywhere(pkg, oracle):
    """bp_nodes_kernel's grid barriers are bounded waits; one that gives up leaves meaningless results.  With the
    time-out flag raised before the launch (test option), every path must say so: the host-pointer decode and a
    Monte-Carlo leg fail, and a caller of the device-pointer entry point finds -1 in every iteration count and
    syndrome weight and 1 in the read-only option after synchronising.  Without the flag the same calls succeed."""
    import torch
    _, (m, n, rows, cols) = pkg.codes.nr_bg1_shaped(z=8, seed=3)
    dec = pkg.Decoder(m, n, rows, cols)
    dec.set_option("generic_lds_kb", -1)
    dec.set_option("nodes", 1)
    F, L = 6, 10
    llr = oracle.gen_awgn_llr(m, n, 2.0, seed=3, first_frame=0, frames=F)
    good = dec.decode_batch(llr, L, True)
    assert dec.last_decode_kernel() == "bp_nodes_kernel" and dec.get_option("nodes_barrier_timed_out") == 0
    d_llr = torch.from_numpy(llr).cuda()
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.zeros(F, dtype=torch.int32, device="cuda")
    d_sw = torch.zeros(F, dtype=torch.int32, device="cuda")
    dec.set_option("test_nodes_timeout", 1)
    with pytest.raises(pkg.capi.LdpcError) as err:
        dec.decode_batch(llr, L, True)
    assert err.value.code == pkg.capi.EHIP and "barrier" in str(err.value)
    dec.decode_batch_dev(d_llr.data_ptr(), F, L, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
    assert dec.last_decode_kernel() == "bp_nodes_kernel"
    assert dec.get_option("nodes_barrier_timed_out") == 1
    assert (d_it.cpu().numpy() == -1).all() and (d_sw.cpu().numpy() == -1).all()
    with pytest.raises(pkg.capi.LdpcError) as err:
        dec.simulate(pkg.capi.CH_AWGN, 2.0, 1, 0, F, L, True)
    assert err.value.code == pkg.capi.EHIP
    dec.set_option("test_nodes_timeout", 0)
    dec.decode_batch_dev(d_llr.data_ptr(), F, L, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
    assert dec.get_option("nodes_barrier_timed_out") == 0
    assert (d_hard.cpu().numpy() == good[0]).all() and (d_it.cpu().numpy() == good[1]).all() and (d_sw.cpu().numpy() == good[2]).all()
    c = dec.simulate(pkg.capi.CH_AWGN, 2.0, 1, 0, F, L, True)
    assert c[0] == F
    dec.close()


def test_row_chunking_of_small_batches_does_not_change_a_bit(pkg, oracle):
    """A small batch of a code the specialised kernels do not take is cut into short waves, down to 8 variables
    each, which store their share of a packed decision word as bytes, and early termination forms its syndrome
    either in launches of its own or inside the next check-node launch: every combination returns the words of
    the double oracle, on a length that is no multiple of 8 as well."""
    rng = np.random.default_rng(11)
    _, bg = pkg.codes.nr_bg1_shaped(z=8, seed=3)
    m2, n2 = 150, 301
    H = (rng.random((m2, n2)) < 0.02).astype(np.uint8)
    H[np.arange(m2), rng.integers(0, n2, m2)] = 1
    H[rng.integers(0, m2, n2), np.arange(n2)] = 1
    r2, c2 = np.nonzero(H)
    for (m, n, rows, cols), F in ((bg, 70), ((m2, n2, r2.astype(np.int32), c2.astype(np.int32)), 130)):
        g = oracle.Graph(m, n, rows, cols)
        dec = pkg.Decoder(m, n, rows, cols)
        dec.set_option("engine", "stream")
        dec.set_option("nodes", 0)
        llr = oracle.gen_awgn_llr(m, n, 2.5, seed=9, first_frame=0, frames=F)
        for early, L in ((True, 30), (False, 5)):
            want = g.decode_llr_batch(llr, L, early)
            for waves, fused in ((0, -1), (256, 0), (256, 1), (1024, -1), (4096, 0), (4096, 1), (1 << 20, -1)):
                dec.set_option("rows_waves", waves)
                dec.set_option("rows_fused", fused)   # the syndrome in launches of its own / inside the next check-node launch
                got = dec.decode_batch(llr, L, early)
                assert dec.last_decode_kernel() == "rows_cn_kernel"
                for x, y in zip(got, want):
                    assert (x == y).all(), (n, early, waves, fused)
        dec.close()
