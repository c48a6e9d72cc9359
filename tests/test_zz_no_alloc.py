"""ldpc_amd_decode_batch_dev_noalloc (the entry point that allocates nothing: its workspace is a static
device array) against ldpc_amd_decode_batch_dev on the same inputs, through the C ABI: identical hard
decisions, iteration counts and syndrome weights, on two calls in a row, the second with a smaller
batch (the array keeps the first call's contents), then on other codes and at the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(dec, torch, ebn0, seed, F, L, early):
    """(outputs of the allocating entry point, outputs of the no-alloc entry point) on the same LLRs"""
    n = dec.n
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    dec.gen_awgn_llr_dev(ebn0, seed, 0, F, d_llr.data_ptr())
    outs = []
    for fn in (dec.decode_batch_dev, dec.decode_batch_dev_noalloc):
        d_hard = torch.full((F, n), 9, dtype=torch.uint8, device="cuda")
        d_it = torch.full((F,), -7, dtype=torch.int32, device="cuda")
        d_sw = torch.full((F,), -7, dtype=torch.int32, device="cuda")
        fn(d_llr.data_ptr(), F, L, early, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
        torch.cuda.synchronize()
        outs.append((d_hard.cpu().numpy(), d_it.cpu().numpy(), d_sw.cpu().numpy()))
    return outs


def _same(a, b, what):
    for x, y, name in zip(a, b, ("hard", "iters", "syndrome weight")):
        assert (x == y).all(), "%s: %s differs on %d frames" % (what, name, int((x != y).reshape(len(x), -1).any(axis=1).sum()))


def test_no_alloc_entry_point_equals_decode_batch_dev(pkg):
    import torch
    pkg.capi.set_default_options()
    m, n, rows, cols = pkg.codes.regular_ldpc(4096, 3, 6, seed=1)
    dec = pkg.Decoder(m, n, rows, cols)
    cap = dec.get_option("noalloc_max_frames")
    assert cap == 1984
    # two calls in a row, the second with a smaller batch: waterfall first (frames fail, every state word is
    # left non-zero), then a cleaner and shorter batch with early termination
    want, got = _pair(dec, torch, 1.0, 5, 1500, 20, True)
    assert dec.last_decode_kernel() == "rows_cn_kernel"
    _same(want, got, "first call, 1500 frames")
    assert (want[2] > 0).any() and (want[2] == 0).any()
    want, got = _pair(dec, torch, 2.2, 6, 333, 30, True)
    _same(want, got, "second call, 333 frames")
    assert (want[1] < 30).all()
    # fixed iterations, a ragged single frame, the largest batch the array takes, and one frame too many
    want, got = _pair(dec, torch, 1.2, 7, 1, 9, False)
    _same(want, got, "one frame, fixed iterations")
    want, got = _pair(dec, torch, 1.5, 8, cap, 6, False)
    _same(want, got, "largest batch")
    d_llr = torch.zeros((cap + 1, n), dtype=torch.float32, device="cuda")
    d_hard = torch.zeros((cap + 1, n), dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.LdpcError) as ei:
        dec.decode_batch_dev_noalloc(d_llr.data_ptr(), cap + 1, 5, True, d_hard.data_ptr())
    assert ei.value.code == pkg.capi.ELIMIT and "at most 1984 frames" in str(ei.value)
    # maxL = 0 and an empty batch behave as in the allocating entry point
    want, got = _pair(dec, torch, 1.5, 9, 10, 0, True)
    _same(want, got, "maxL = 0")
    dec.decode_batch_dev_noalloc(0, 0, 5, True, d_hard.data_ptr())
    dec.close()
    # an irregular code (degrees to 19 / 30) right after, on the same array, then the min-sum rule
    m, n, rows, cols = pkg.codes.nr_bg1_shaped(z=16, seed=2)[1]
    dec = pkg.Decoder(m, n, rows, cols)
    want, got = _pair(dec, torch, 2.0, 11, 700, 15, True)
    _same(want, got, "irregular code")
    dec.set_cn_rule(pkg.capi.CN_MINSUM)
    want, got = _pair(dec, torch, 2.0, 12, 130, 15, False)
    _same(want, got, "irregular code, min-sum")
    dec.close()


def test_no_alloc_entry_point_under_graph_capture(pkg):
    """The callers the entry point is for include code under HIP graph capture: the call is captured into a graph
    (torch.cuda.graph on a side stream), then replayed three times on new LLRs written into the captured buffers;
    every replay must return what ldpc_amd_decode_batch_dev returns for those LLRs."""
    import torch
    m, n, rows, cols = pkg.codes.regular_ldpc(4096, 3, 6, seed=1)
    dec = pkg.Decoder(m, n, rows, cols)
    F, L = 777, 25
    d_llr = torch.empty((F, n), dtype=torch.float32, device="cuda")
    d_hard = torch.full((F, n), 9, dtype=torch.uint8, device="cuda")
    d_it = torch.full((F,), -7, dtype=torch.int32, device="cuda")
    d_sw = torch.full((F,), -7, dtype=torch.int32, device="cuda")
    dec.gen_awgn_llr_dev(1.8, 3, 0, F, d_llr.data_ptr())
    # once outside any capture: module loading, function attributes
    dec.decode_batch_dev_noalloc(d_llr.data_ptr(), F, L, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dec.decode_batch_dev_noalloc(d_llr.data_ptr(), F, L, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr(), s.cuda_stream)
    for ebn0, seed in ((1.8, 3), (2.4, 4), (1.2, 5)):
        dec.gen_awgn_llr_dev(ebn0, seed, 0, F, d_llr.data_ptr())      # new inputs in the captured buffers
        d_hard.fill_(9); d_it.fill_(-7); d_sw.fill_(-7)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got = (d_hard.cpu().numpy(), d_it.cpu().numpy(), d_sw.cpu().numpy())
        h2 = torch.empty_like(d_hard); i2 = torch.empty_like(d_it); w2 = torch.empty_like(d_sw)
        dec.decode_batch_dev(d_llr.data_ptr(), F, L, True, h2.data_ptr(), i2.data_ptr(), w2.data_ptr())
        torch.cuda.synchronize()
        for a, b in zip(got, (h2.cpu().numpy(), i2.cpu().numpy(), w2.cpu().numpy())):
            assert np.array_equal(a, b), (ebn0, seed)
    dec.close()
