"""The high-SNR half of the Eb/N0 sweep on the (3,6) n = 4096 code, frame by frame against the CPU oracle.

Above 2 dB every frame error of that sweep is an *undetected* one: the reference never clips a message
(LogDecoder.java:93), +Inf and -Inf meet in a variable, the NaN posterior decodes to 0 (LogDecoder.java:104)
and spreads, and the frame ends as the all-zero word with syndrome 0 (LogDecoder.java:117-119) although a
random codeword was sent (the loop of LDPC_GallagerIntStreamSimu.java:79-213).  These events are the whole
curve up there, so they are pinned here one by one: the GPU decodes the very frames of the committed sweep
(same code, seed and global frame indices as tools/run_c3_sweep.py) through gen_codewords_dev ->
gen_awgn_llr_dev -> decode_batch_dev, every failing frame and 2,000 random other frames per point are
decoded again by the double oracle from the same LLRs, and the whole (word, iterations, syndrome weight)
triple must be equal for each of them.  The oracle also regenerates the failing frames on its own from the
global frame index (its own codeword and channel generators).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED                      # tools/run_c3_sweep.py --seed
# Eb/N0, frames decoded here, (frames, frame errors) of the committed sweep's point (profiles/r02/c3_sweep.json)
POINTS = ((2.5, 2097152, (2097152, 200)), (3.0, 3145728, (3145728, 130)), (3.5, 8388608, (6291456, 106)),
          (4.0, 18874368, (18874368, 102)))
CONTROLS = 2000
CHUNK = 1 << 18


def test_every_frame_error_above_2_db_equals_the_oracle(pkg, oracle):
    import torch
    m, n, rows, cols = pkg.codes.regular_ldpc(4096, 3, 6, seed=1)
    eh, eg = pkg.encoder.systematic_pair(pkg.check_matrix_io.to_dense(m, n, rows, cols))
    r2, c2 = (a.astype(np.int32) for a in np.nonzero(eh))
    dec = pkg.Decoder(m, n, r2, c2)
    dec.set_generator(eg)
    stream_dec = pkg.Decoder(m, n, r2, c2)
    stream_dec.set_option("engine", "stream")
    g = oracle.Graph(m, n, r2, c2)
    L = 50
    d_cw = torch.empty((CHUNK, n), dtype=torch.uint8, device="cuda")
    d_llr = torch.empty((CHUNK, n), dtype=torch.float32, device="cuda")
    d_hard = torch.empty((CHUNK, n), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(CHUNK, dtype=torch.int32, device="cuda")
    d_sw = torch.empty(CHUNK, dtype=torch.int32, device="cuda")
    report = []
    for ebn0, frames, (sweep_frames, sweep_errors) in POINTS:
        rng = np.random.default_rng(int(ebn0 * 10))
        controls = np.sort(rng.choice(frames, CONTROLS, replace=False))
        kept = {}                                     # global frame -> (llr, cw, hard, iters, weight, failed)
        for first in range(0, frames, CHUNK):
            F = min(CHUNK, frames - first)
            dec.gen_codewords_dev(SEED, first, F, d_cw.data_ptr())
            dec.gen_awgn_llr_dev(ebn0, SEED, first, F, d_llr.data_ptr(), d_cw=d_cw.data_ptr())
            dec.decode_batch_dev(d_llr.data_ptr(), F, L, True, d_hard.data_ptr(), d_it.data_ptr(), d_sw.data_ptr())
            failed = (d_hard[:F] != d_cw[:F]).any(dim=1)
            bad = torch.nonzero(failed).flatten().cpu().numpy()
            ctl = controls[(controls >= first) & (controls < first + F)] - first
            for local in np.union1d(bad, ctl):
                local = int(local)
                kept[first + local] = (d_llr[local].cpu().numpy(), d_cw[local].cpu().numpy(), d_hard[local].cpu().numpy(),
                                       int(d_it[local]), int(d_sw[local]), bool(failed[local]))
        assert dec.last_decode_kernel() == "bp_lds_kernel"
        idx = np.array(sorted(kept))
        failing = np.array([f for f in idx if kept[f][5]])
        llr = np.stack([kept[f][0] for f in idx])
        got = (np.stack([kept[f][2] for f in idx]), np.array([kept[f][3] for f in idx], np.int32),
               np.array([kept[f][4] for f in idx], np.int32))
        # the oracle on the same LLRs: every failing frame and every control, the whole triple
        want = g.decode_llr_batch(llr, L, True)
        differ = [int(f) for t, f in enumerate(idx)
                  if not ((got[0][t] == want[0][t]).all() and got[1][t] == want[1][t] and got[2][t] == want[2][t])]
        assert not differ, "%.1f dB: frames whose triple differs from the oracle's: %s" % (ebn0, differ[:10])
        # the streaming engine on the same frames: identical bits
        again = stream_dec.decode_batch(llr, L, True)
        assert all((a == b).all() for a, b in zip(again, got))
        # the oracle's failing set on the decoded range = the GPU's: among the frames looked at, those whose
        # oracle word differs from the transmitted word are exactly the frames the GPU reported
        cw = np.stack([kept[f][1] for f in idx])
        o_failed = (want[0] != cw).any(axis=1)
        assert np.array_equal(idx[o_failed], failing)
        # the same events from the global frame index alone, with the oracle's own generators
        own_equal = with_nan = undetected = 0
        for f in failing:
            f = int(f)
            o_cw = oracle.gen_codewords(eg, SEED, f, 1)
            assert (o_cw[0] == kept[f][1]).all()                       # the transmitted word is integer work: bit-exact
            o_llr = oracle.gen_awgn_llr(m, n, ebn0, SEED, f, 1, cw=o_cw)
            assert np.allclose(o_llr[0], kept[f][0], rtol=1e-4, atol=1e-4)
            hard, it, sw, alpha, _ = g.decode_llr(o_llr[0], L, True, messages=True)
            own_equal += int((hard == kept[f][2]).all() and it == kept[f][3] and sw == kept[f][4])
            with_nan += int(np.isnan(alpha).any())
            undetected += int(kept[f][4] == 0)
        assert own_equal >= len(failing) - 1       # float channel arithmetic of two libraries: the last bit of an LLR may differ
        # the per-frame path and the leg (ldpc_amd_simulate, accounting inside the decode kernel) count the same events,
        # and they are the events of the committed sweep
        leg = dec.simulate(pkg.capi.CH_AWGN, ebn0, SEED, 0, sweep_frames, L, True)
        assert int(leg[0]) == sweep_frames and int(leg[1]) == sweep_errors == int((failing < sweep_frames).sum())
        assert int(leg[6]) == int(sum(kept[int(f)][4] == 0 for f in failing if f < sweep_frames))
        report.append((ebn0, frames, len(failing), undetected, with_nan, own_equal, len(idx) - len(failing)))
        if ebn0 >= 3.0:
            assert len(failing) >= 100
    print("Eb/N0 dB, frames, failing frames (all equal to the oracle), undetected among them, with NaN among the oracle's "
          "check-to-variable messages, equal from the oracle's own generators, controls (all equal):", report)
    dec.close()
    stream_dec.close()
