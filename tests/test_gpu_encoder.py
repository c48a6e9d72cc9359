"""Device side of the encoder chain: random codewords u*G (GenerateC.java:5-27), the channel and the
accounting around a transmitted word (LDPC_GallagerIntStreamSimu.java:115-183), against the CPU checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sys_code(pkg, oracle):
    """The reference's default shape in systematic form: Gallager (8,2) n=1024 -> encodedH, encodedG."""
    m, n, rows, cols = pkg.codes.gallager(1024, 8, 2, seed=1)
    H = pkg.check_matrix_io.to_dense(m, n, rows, cols)
    eh, eg = pkg.encoder.systematic_pair(H)
    r, c = np.nonzero(eh)
    return dict(m=m, n=n, eh=eh, eg=eg, dec=pkg.Decoder(m, n, r, c), g=oracle.Graph(m, n, r, c))


def test_codewords_bit_exact(sys_code, oracle):
    import torch
    c = sys_code
    dec, n = c["dec"], c["n"]
    dec.set_generator(c["eg"])
    for first, F in ((0, 8), (123456789012, 301), (7, 1), (5, 64), (5, 65)):   # ragged: a block encodes 64 frames
        d_cw = torch.full((F, n), 9, dtype=torch.uint8, device="cuda")
        dec.gen_codewords_dev(42, first, F, d_cw.data_ptr())
        torch.cuda.synchronize()
        got = d_cw.cpu().numpy()
        assert (got == oracle.gen_codewords(c["eg"], 42, first, F)).all()
        assert not ((got.astype(np.int64) @ c["eh"].T) % 2).any()
    dec.gen_codewords_dev(42, 0, 0, 0)                           # empty batch
    dec.set_generator(None)
    d_cw = torch.full((5, n), 9, dtype=torch.uint8, device="cuda")
    dec.gen_codewords_dev(42, 0, 5, d_cw.data_ptr())
    torch.cuda.synchronize()
    assert not d_cw.any()                                        # no generator: the all-zero word


def test_codewords_odd_length(pkg, oracle):
    """n not a multiple of 4, 32 or 256 and k not a multiple of 32, 64 or 128: the tail paths; dense G.
    The last two shapes have columns of more than 32 words (k > 2048): the lookup tables are rebuilt per pass,
    with a full last pass and with a short one."""
    import torch
    rng = np.random.default_rng(5)
    for m, n, k, F in ((30, 75, 45, 19), (100, 301, 130, 70), (200, 777, 577, 130), (60, 2301, 2241, 70),
                       (40, 1100, 4096, 65)):
        G = rng.integers(0, 2, (k, n)).astype(np.uint8)
        rows, cols = np.nonzero(rng.random((m, n)) < 0.1)
        dec = pkg.Decoder(m, n, rows, cols)
        dec.set_generator(G)
        d_cw = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        dec.gen_codewords_dev(3, 100, F, d_cw.data_ptr())
        torch.cuda.synchronize()
        assert (d_cw.cpu().numpy() == oracle.gen_codewords(G, 3, 100, F)).all(), (m, n, k)
        dec.close()


def test_channel_and_counts_with_codewords(sys_code, oracle):
    import torch
    c = sys_code
    dec, m, n = c["dec"], c["m"], c["n"]
    F, L = 400, 30
    dec.set_generator(c["eg"])
    d_cw = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    dec.gen_codewords_dev(1, 50, F, d_cw.data_ptr())
    cw = oracle.gen_codewords(c["eg"], 1, 50, F)
    for ch in (oracle.CH_BSC_INFO, oracle.CH_BSC_ALL):
        d_r = torch.empty((F, n), dtype=torch.uint8, device="cuda")
        dec.gen_bsc_dev(ch, 0.03, 8, 50, F, d_r.data_ptr(), d_cw=d_cw.data_ptr())
        torch.cuda.synchronize()
        r = oracle.gen_bsc(m, n, ch, 0.03, 8, 50, F, cw=cw)
        assert (d_r.cpu().numpy() == r).all()
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(2.0, 8, 50, F, d_llr.data_ptr(), d_cw=d_cw.data_ptr())
    torch.cuda.synchronize()
    assert np.allclose(d_llr.cpu().numpy(), oracle.gen_awgn_llr(m, n, 2.0, 8, 50, F, cw=cw), rtol=2e-5, atol=2e-5)
    # decode the BSC_ALL words and count against the transmitted words, on the device and in the checker
    d_hard = torch.empty((F, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(F, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(F, dtype=torch.int32, device="cuda")
    dec.decode_bsc_batch_dev(d_r.data_ptr(), F, 0.03, oracle.LLR_UNIFORM, L, True, d_hard.data_ptr(), d_it.data_ptr(),
                             d_sw.data_ptr())
    d_cnt = torch.zeros(16 + L, dtype=torch.int64, device="cuda")
    dec.count_errors_dev(F, d_hard.data_ptr(), d_r.data_ptr(), d_it.data_ptr(), d_sw.data_ptr(), L, d_cnt.data_ptr(),
                         d_cw=d_cw.data_ptr())
    torch.cuda.synchronize()
    want = oracle.count_errors(m, n, d_hard.cpu().numpy(), r, d_it.cpu().numpy(), d_sw.cpu().numpy(), L, cw=cw)
    got = d_cnt.cpu().numpy().astype(np.uint64)
    assert (got == want).all() and 0 < got[1] < F and got[10] > 0
    dec.set_generator(None)


def test_simulate_with_generator(sys_code, oracle):
    """The whole leg with random codewords: equal to the all-zero leg on the BSC (the decoder is symmetric
    and r = c ^ flips mirrors the LLRs exactly) and to the checker's leg with the same G."""
    c = sys_code
    dec, g = c["dec"], c["g"]
    for ch, e in ((oracle.CH_BSC_INFO, 0.035), (oracle.CH_BSC_ALL, 0.022)):
        dec.set_generator(None)
        zero = dec.simulate(ch, e, seed=3, first_frame=10, frames=3000, max_iters=40)
        dec.set_generator(c["eg"])
        rand = dec.simulate(ch, e, seed=3, first_frame=10, frames=3000, max_iters=40)
        assert 0 < zero[1] < 3000 and (zero == rand).all(), (zero[:12], rand[:12])
        want = g.simulate(ch, e, seed=3, first_frame=10, frames=3000, max_iters=40, G=c["eg"])
        # every counter, histogram included: 3,000 frames leave no room for a differing frame at the 99.99 % bar
        assert (rand == want).all(), (rand[:12], want[:12], np.nonzero(rand != want)[0])
    rand = dec.simulate(oracle.CH_AWGN, 3.5, seed=3, first_frame=0, frames=3000, max_iters=40)
    want = g.simulate(oracle.CH_AWGN, 3.5, seed=3, first_frame=0, frames=3000, max_iters=40, G=c["eg"])
    # AWGN: the channel LLRs come from the device's logf/sinf/cosf and the C library's (ulps apart), so a
    # received sign within an ulp of zero may differ: the channel-error counter may differ by that much, the
    # decoded words may not
    assert abs(int(rand[10]) - int(want[10])) <= 2 and (rand[:10] == want[:10]).all(), (rand[:12], want[:12])
    dec.set_generator(None)


def test_count_errors_word_and_byte_paths(pkg, oracle):
    """The accounting kernel reads four positions per lane when n is a multiple of 4 (also across the
    information/check boundary) and one otherwise."""
    import torch
    rng = np.random.default_rng(4)
    for n, m in ((24, 9), (24, 12), (22, 9), (256, 95)):
        H = (rng.random((m, n)) < 0.2).astype(np.uint8)
        H[np.arange(m), rng.integers(0, n, m)] = 1
        rows, cols = np.nonzero(H)
        dec = pkg.Decoder(m, n, rows.astype(np.int32), cols.astype(np.int32))
        F, L = 333, 12
        hard = (rng.random((F, n)) < 0.3).astype(np.uint8)
        rx = (rng.random((F, n)) < 0.3).astype(np.uint8)
        cw = (rng.random((F, n)) < 0.5).astype(np.uint8)
        it = rng.integers(1, L + 1, F).astype(np.int32)
        sw = (rng.random(F) < 0.5).astype(np.int32) * rng.integers(1, 5, F).astype(np.int32)
        d = {k: torch.from_numpy(v).cuda() for k, v in (("hard", hard), ("rx", rx), ("cw", cw), ("it", it), ("sw", sw))}
        for use_cw in (False, True):
            d_cnt = torch.zeros(16 + L, dtype=torch.int64, device="cuda")
            dec.count_errors_dev(F, d["hard"].data_ptr(), d["rx"].data_ptr(), d["it"].data_ptr(), d["sw"].data_ptr(), L,
                                 d_cnt.data_ptr(), d_cw=d["cw"].data_ptr() if use_cw else 0)
            torch.cuda.synchronize()
            want = oracle.count_errors(m, n, hard, rx, it, sw, L, cw=cw if use_cw else None)
            assert (d_cnt.cpu().numpy().astype(np.uint64) == want).all(), (n, m, use_cw)
        dec.close()


def test_accounting_inside_the_resident_kernels_equals_the_counting_pass(pkg, oracle):
    """ldpc_amd_simulate on the (8,2) and (6,3) codes: the frame-resident kernels compare their decisions with
    the transmitted word themselves (option sim_fused, the default) instead of writing hard words for
    count_errors_kernel.  Every counter must equal the two-pass leg's: both channels families, random and
    all-zero codewords, early termination and fixed iterations, ragged batches, one and several frames per
    workgroup; and the checker's leg on the BSC."""
    for shape, nvar, early_cases in (((8, 2), 1024, (True, False)), ((6, 3), 1536, (True, False))):
        dc, dv = shape
        m, n, rows, cols = (pkg.codes.gallager(nvar, dc, dv, seed=3) if dc == 8 else pkg.codes.regular_ldpc(nvar, dv, dc, seed=3))
        eh, eg = pkg.encoder.systematic_pair(pkg.check_matrix_io.to_dense(m, n, rows, cols))
        r, c = np.nonzero(eh)
        dec = pkg.Decoder(m, n, r.astype(np.int32), c.astype(np.int32))
        g = oracle.Graph(m, n, r, c)
        # "wrong": a generator whose rows are not codewords of this H (nothing checks that they are): the transmitted
        # words then have a non-zero syndrome, no frame can be decoded to them, and the fused leg must still count
        # exactly what the two-pass leg and the checker count.
        wrong = eg.copy()
        wrong[::3, n - m + 5] ^= 1
        for with_g in (False, True, "wrong"):
            dec.set_generator(wrong if with_g == "wrong" else eg if with_g else None)
            for early in early_cases:
                for ch, param, frames, L in ((oracle.CH_BSC_INFO, 0.03, 1000, 30), (oracle.CH_BSC_ALL, 0.04, 777, 20),
                                             (oracle.CH_AWGN, 2.0, 1501, 25), (oracle.CH_AWGN, 4.5, 64, 10)):
                    got = {}
                    for fused, fpw in ((1, 0), (1, 5), (0, 0)):
                        dec.set_option("sim_fused", fused)
                        dec.set_option("lds_fpw", fpw)
                        got[(fused, fpw)] = dec.simulate(ch, param, 17, 123, frames, L, early)
                        assert dec.last_decode_kernel() == "bp_lds_kernel"
                    dec.set_option("lds_fpw", 0)
                    ref = got[(0, 0)]
                    assert ref[0] == frames and ref[16:].sum() == frames
                    for key, val in got.items():
                        assert (val == ref).all(), (shape, with_g, early, ch, key, val[:12], ref[:12], np.nonzero(val != ref)[0])
                    # (the (8,2) code only: on the (6,3) code one frame in a thousand takes an iteration more or less in
                    # fp32 than in the checker's double, with either leg)
                    if ch != oracle.CH_AWGN and early and dc == 8:
                        want = g.simulate(ch, param, 17, 123, frames, L, G=wrong if with_g == "wrong" else eg if with_g else None)
                        assert (ref == want).all(), (shape, with_g, ch, ref[:12], want[:12])
        # no frames, and a rule the fused kernels are not built for: the counting pass takes over
        dec.set_option("sim_fused", 1)
        assert not dec.simulate(oracle.CH_AWGN, 2.0, 17, 0, 0, 10, True).any()
        dec.set_cn_rule(pkg.capi.CN_MINSUM)
        a = dec.simulate(oracle.CH_BSC_INFO, 0.03, 17, 0, 300, 10, True)
        dec.set_option("sim_fused", 0)
        assert (a == dec.simulate(oracle.CH_BSC_INFO, 0.03, 17, 0, 300, 10, True)).all()
        dec.close()


def test_generator_larger_than_the_codeword_kernel_takes_is_refused_at_once(pkg):
    """The codeword kernel keeps the message bits of 64 frames in LDS: k above LDPC_AMD_MAX_GENERATOR_ROWS is refused
    by ldpc_amd_set_generator itself, not in the middle of a leg."""
    n, k = 10400, 10209
    dec = pkg.Decoder(n - k, n, np.arange(n - k, dtype=np.int32), np.arange(n - k, dtype=np.int32))
    with pytest.raises(pkg.capi.LdpcError) as err:
        dec.set_generator(np.zeros((k, n), np.uint8))
    assert err.value.code == pkg.capi.ELIMIT
    dec.set_generator(np.zeros((k - 1, n), np.uint8))        # the limit itself is accepted
    c = dec.simulate(pkg.capi.CH_BSC_ALL, 0.01, 1, 0, 64, 2, True)
    assert c[0] == 64
    dec.close()
