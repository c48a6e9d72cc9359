"""Host-pointer calls that move little: the library runs them on page-locked host memory that the kernels read and
write in place (LDPC_AMD_OPT_TUNE_SMALL_CALL_KB) instead of staging through device buffers.  The reference's
drivers make exactly such calls, one received word each (LDPC_GallagerIntStreamSimu.java:122).  Both routes must
return the same triples, on every engine, on both sides of the size limit and when the staging area has to grow."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(pkg, dec, fn):
    out = []
    for kb in (-1, 0):
        dec.set_option("small_call_kb", kb)
        out.append(fn())
    dec.set_option("small_call_kb", 0)
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    return out[1]


def test_small_calls_equal_staged_calls_on_every_engine(pkg, oracle):
    cases = (("bp_lds_kernel", pkg.codes.gallager(1024, 8, 2, seed=1), 3.0, {}),
             ("bp_lds_gen_kernel", pkg.codes.nr_bg1_shaped(8, 3)[1], 2.5, {}),
             ("rows_cn_kernel", pkg.codes.regular_ldpc(1536, 3, 6, seed=2), 2.0, {"engine": "stream", "nodes": 0}),
             ("bp_nodes_kernel", pkg.codes.nr_bg1_shaped(8, 3)[1], 2.5, {"generic_lds_kb": -1, "nodes": 1}))
    for name, (m, n, rows, cols), ebn0, options in cases:
        dec = pkg.Decoder(m, n, rows, cols)
        for k, v in options.items():
            dec.set_option(k, v)
        g = oracle.Graph(m, n, rows, cols)
        for F in (1, 2, 9, 31):
            llr = oracle.gen_awgn_llr(m, n, ebn0, seed=17, first_frame=F, frames=F)
            for early in (True, False):
                got = _both(pkg, dec, lambda: dec.decode_batch(llr, 20, early))
                assert dec.last_decode_kernel() == name
                want = g.decode_llr_batch(llr, 20, early)
                assert all(np.array_equal(x, y) for x, y in zip(got, want)), (name, F, early)
            r = oracle.gen_bsc(m, n, oracle.CH_BSC_INFO, 0.01, seed=5, first_frame=3, frames=F)
            _both(pkg, dec, lambda: dec.decode_bsc_batch(r, 0.01, 20, oracle.LLR_PARITY_LN1E20, True))
        dec.close()


def test_small_call_limit_and_growth_of_the_staging_area(pkg, oracle):
    m, n, rows, cols = pkg.codes.gallager(1024, 8, 2, seed=1)
    dec = pkg.Decoder(m, n, rows, cols)
    g = oracle.Graph(m, n, rows, cols)
    per_frame = n * 4 + n + 8                      # bytes in and out of one frame of float LLRs
    at_limit = (256 << 10) // per_frame            # frames that fit a limit of 256 KiB (every array rounded up to 256 bytes)
    results = {}
    # 1 frame (64 KiB area), then more than 64 KiB (the area grows), then around an explicit limit, a larger limit, a
    # small call again, and the library's own choice for a batch of 3.5 MiB from pageable memory
    for F, kb in ((1, 0), (40, 0), (at_limit - 2, 256), (at_limit + 2, 256), (at_limit + 2, 1024), (3, 0), (700, 4096), (700, 0)):
        dec.set_option("small_call_kb", kb)
        llr = oracle.gen_awgn_llr(m, n, 3.5, seed=23, first_frame=0, frames=F)
        got = dec.decode_batch(llr, 15, True)
        want = g.decode_llr_batch(llr, 15, True)
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), (F, kb)
        results[(F, kb)] = got
    assert all(np.array_equal(x, y) for x, y in zip(results[(at_limit + 2, 256)], results[(at_limit + 2, 1024)]))
    # page-locked caller buffers (the library then keeps the in-place route to calls of at most 1 MiB): both sides of it
    for F in (100, 400):
        bufs = [pkg.capi.HostBuffer((F, n), np.float32), pkg.capi.HostBuffer((F, n), np.uint8),
                pkg.capi.HostBuffer((F,), np.int32), pkg.capi.HostBuffer((F,), np.int32)]
        bufs[0].array[:] = oracle.gen_awgn_llr(m, n, 3.5, seed=29, first_frame=0, frames=F)
        dec.set_option("small_call_kb", 0)
        ptr = pkg.capi._np_ptr
        rc = dec._lib.ldpc_amd_decode_batch(dec._h, ptr(bufs[0].array), F, 15, 1, ptr(bufs[1].array), ptr(bufs[2].array), ptr(bufs[3].array))
        assert rc == 0
        want = g.decode_llr_batch(bufs[0].array, 15, True)
        assert all(np.array_equal(x, y.array) for x, y in zip(want, bufs[1:])), F
        for b in bufs:
            b.close()
    with pytest.raises(pkg.LdpcError):
        dec.set_option("small_call_kb", 65537)
    dec.close()
