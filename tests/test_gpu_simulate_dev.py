"""ldpc_amd_simulate_dev: a leg of the reference's Monte-Carlo loop (LDPC_GallagerIntStreamSimu.java:99-184) whose
counters are ADDED into a vector on the device, ordered with the caller's stream, nothing copied to the host.  Its
counts must be those of ldpc_amd_simulate; legs must add up; work queued on the stream before and after the call
must see the vector as the stream order says; and an error-rate point run through it must stop where the
host-counter path stops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_simulate_dev_adds_what_simulate_returns(pkg, oracle):
    import torch
    for name, (m, n, rows, cols), gen in (("bp_lds_kernel", pkg.codes.gallager(1024, 8, 2, seed=3), True),
                                           ("bp_lds_gen_kernel", pkg.codes.nr_bg1_shaped(8, 3)[1], False),
                                           ("rows_cn_kernel", pkg.codes.regular_ldpc(1536, 3, 6, seed=5), False)):
        if gen:
            eh, eg = pkg.encoder.systematic_pair(pkg.check_matrix_io.to_dense(m, n, rows, cols))
            r, c = np.nonzero(eh)
            dec = pkg.Decoder(m, n, r.astype(np.int32), c.astype(np.int32))
            dec.set_generator(eg)
        else:
            dec = pkg.Decoder(m, n, rows, cols)
        if name == "rows_cn_kernel":
            dec.set_option("engine", "stream")
        L = 20
        stream = torch.cuda.current_stream().cuda_stream
        for ch, param, legs in ((pkg.capi.CH_AWGN, 2.5, ((0, 700), (700, 333), (5000, 1))),
                                (pkg.capi.CH_BSC_INFO, 0.02, ((11, 512), (523, 100)))):
            want = sum(dec.simulate(ch, param, 9, first, count, L, True).astype(np.int64) for first, count in legs)
            # queued before the legs on the same stream: the vector starts at 7 everywhere ...
            d_cnt = torch.empty(pkg.capi.CNT_FIXED + L, dtype=torch.int64, device="cuda")
            d_cnt.fill_(7)
            for first, count in legs:
                dec.simulate_dev(ch, param, 9, first, count, L, True, d_cnt.data_ptr(), stream)
            # ... and queued after them, without any synchronisation in between: a copy and a doubling
            snapshot = d_cnt.clone()
            d_cnt.mul_(2)
            torch.cuda.synchronize()
            assert dec.last_decode_kernel() == name
            assert np.array_equal(snapshot.cpu().numpy() - 7, want), (name, ch)
            assert np.array_equal(d_cnt.cpu().numpy(), 2 * (want + 7)), (name, ch)
            assert int(want[0]) == sum(count for _, count in legs) and int(want[pkg.capi.CNT_FIXED:].sum()) == int(want[0])
        # an error-rate point with the counters on the device stops where the host-counter path stops
        fn = lambda first, count: dec.simulate(pkg.capi.CH_AWGN, 1.5, 4, first, count, L, True)
        a = pkg.sim.simulate_until(fn, L, min_frame_errors=30, max_frames=20000, batch=1024)
        b = pkg.sim.simulate_until(fn, L, min_frame_errors=30, max_frames=20000, batch=1024, device=torch.device("cuda"),
                                   accumulate_fn=pkg.sim.device_accumulator(dec, pkg.capi.CH_AWGN, 1.5, 4, L, True))
        assert np.array_equal(a, b), name
        assert int(a[0]) % 1024 == 0 and (int(a[1]) >= 30 or int(a[0]) == 20480)
        dec.close()


def test_simulate_dev_rejects_what_simulate_rejects(pkg):
    import torch
    m, n, rows, cols = pkg.codes.gallager(1024, 8, 2, seed=3)
    dec = pkg.Decoder(m, n, rows, cols)
    d_cnt = torch.zeros(pkg.capi.CNT_FIXED + 10, dtype=torch.int64, device="cuda")
    with pytest.raises(pkg.LdpcError):
        dec.simulate_dev(99, 0.1, 1, 0, 10, 10, True, d_cnt.data_ptr(), 0)          # unknown channel
    with pytest.raises(pkg.LdpcError):
        dec.simulate_dev(pkg.capi.CH_BSC_INFO, 1.5, 1, 0, 10, 10, True, d_cnt.data_ptr(), 0)   # e outside (0, 1)
    with pytest.raises(pkg.LdpcError):
        dec.simulate_dev(pkg.capi.CH_AWGN, 2.0, 1, 0, 10, 10, True, 0, 0)            # no counter vector
    dec.simulate_dev(pkg.capi.CH_AWGN, 2.0, 1, 0, 0, 10, True, d_cnt.data_ptr(), 0)  # an empty leg adds nothing
    torch.cuda.synchronize()
    assert not d_cnt.any()
    dec.close()


def test_leg_counters_do_not_depend_on_the_chunking(pkg):
    """A leg is cut into chunks that are generated ahead of their decodes on streams of their own (random codewords two
    chunks ahead, the channel one); every counter must be the same however the leg is cut: one chunk, hundreds of small
    ones (the buffer sets are reused many times), ragged legs at odd first frames, the library's own choice, with
    random codewords and with the all-zero word, on the resident kernels and on the sweeps."""
    m, n, rows, cols = pkg.codes.regular_ldpc(1536, 3, 6, 2)
    eh, eg = pkg.encoder.systematic_pair(pkg.check_matrix_io.to_dense(m, n, rows, cols))
    r2, c2 = np.nonzero(eh)
    F = 200_000
    for generator in (True, False):
        for engine in ("auto", "stream"):
            dec = pkg.Decoder(m, n, r2.astype(np.int32), c2.astype(np.int32))
            if generator:
                dec.set_generator(eg)
            dec.set_option("engine", engine)
            ref = None
            for mb, cuts in ((4096, (0, F)), (16, (0, F)), (16, (0, 33_333, 100_001, F)), (64, (0, F // 2, F)), (0, (0, F))):
                if engine == "stream" and mb == 16 and len(cuts) == 2:
                    continue  # hundreds of sweep decodes of a thousand frames: slow, and one stream anyway
                dec.set_option("sim_chunk_mb", mb)
                total = None
                for a, b in zip(cuts[:-1], cuts[1:]):
                    c = np.asarray(dec.simulate(pkg.capi.CH_AWGN, 1.9, 11, a, b - a, 30, True)).astype(np.int64)
                    total = c if total is None else total + c
                if ref is None:
                    ref = total
                    assert 0 < ref[1] < F and ref[0] == F    # some frames fail, some do not
                assert (total == ref).all(), (generator, engine, mb, cuts, np.nonzero(total != ref)[0].tolist())
            assert not np.asarray(dec.simulate(pkg.capi.CH_AWGN, 1.9, 11, 5, 0, 30, True)).any()   # an empty leg
            dec.close()
