"""One context per worker thread, all on one GPU, decoding at the same time: what a binding gets when the
reference's parallel frame loop (IntStream.range(...).parallel(), LDPC_GallagerIntStreamSimu.java:99, one
LogDecoder.decode per frame at :122) calls into the library from its pool threads.  Contexts share nothing but
the device, so every thread must get the results of a lone run of its own batches."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = 6
ROUNDS = 5


def test_one_context_per_thread_decoding_at_the_same_time(pkg, oracle):
    # three kinds of code, so that the resident, the generic resident, the node-parallel and the sweep engines
    # all run side by side (each context picks its engine per batch)
    codes = (pkg.codes.gallager(1024, 8, 2, seed=3), pkg.codes.regular_ldpc(1536, 3, 6, seed=5),
             pkg.codes.nr_bg1_shaped(8, 2)[1])
    plans = []
    for w in range(THREADS):
        m, n, rows, cols = codes[w % len(codes)]
        batches = []
        for r in range(ROUNDS):
            F = (1, 7, 64, 300, 1000)[(w + r) % 5]
            ebn0 = (2.0, 3.0, 4.0)[(w + r) % 3]
            batches.append((oracle.gen_awgn_llr(m, n, ebn0, seed=100 + w, first_frame=1000 * r, frames=F), 30, (w + r) % 2 == 0))
        plans.append(((m, n, rows, cols), batches))

    def run(plan, out, barrier=None):
        (m, n, rows, cols), batches = plan
        dec = pkg.Decoder(m, n, rows, cols)
        try:
            for llr, L, early in batches:
                if barrier is not None:
                    barrier.wait(timeout=120)      # every round starts in all threads at once
                out.append((dec.decode_batch(llr, L, early), dec.last_decode_kernel()))
        finally:
            dec.close()

    alone = [[] for _ in plans]
    for plan, out in zip(plans, alone):
        run(plan, out)
    together = [[] for _ in plans]
    errors = []
    barrier = threading.Barrier(THREADS)

    def worker(i):
        try:
            run(plans[i], together[i], barrier)
        except BaseException as e:  # noqa: BLE001 - reported below, in the main thread
            errors.append((i, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(THREADS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    kernels = set()
    for w in range(THREADS):
        assert len(together[w]) == ROUNDS
        for (a, ka), (b, kb) in zip(alone[w], together[w]):
            assert ka == kb
            kernels.add(ka)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (w, ka)
    assert {"bp_lds_kernel", "bp_lds_gen_kernel"} <= kernels
    # and one of the lone runs against the oracle, so that "equal" is not "equally wrong"
    (m, n, rows, cols), batches = plans[1]
    g = oracle.Graph(m, n, rows, cols)
    for (llr, L, early), (got, _) in zip(batches, alone[1]):
        want = g.decode_llr_batch(llr, L, early)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
