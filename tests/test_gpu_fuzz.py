"""Differential fuzz across the engines (tools/fuzz_engines.py): random regular, quasi-cyclic, NR-shaped and
unstructured codes, batch sizes around the engine thresholds, all three check-node rules, LLR and hard-word
input, early exit and fixed iterations; every engine must return the bits of the row-parallel sweeps."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engines_agree_on_random_cases(pkg):
    spec = importlib.util.spec_from_file_location("fuzz_engines", os.path.join(ROOT, "tools", "fuzz_engines.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pkg.capi.set_default_options()
    seen, bad = mod.fuzz(250, 7)
    assert bad == 0, seen
    for kernel in ("bp_lds_kernel", "bp_lds_gen_kernel", "bp_nodes_kernel", "rows_cn_kernel", "bp_decode_kernel"):
        assert seen.get(kernel, 0) > 0, seen


def test_generic_resident_engine_dealing(pkg, oracle):
    """The generic frame-resident kernel deals its nodes out over the lanes by degree (pairs of nodes of one
    degree, single nodes for small degree classes, any number of wavefronts per frame): whatever the dealing, the
    bits are those of the row-parallel sweeps.  Codes with many small degree classes, empty rows and columns, a
    regular code (nothing but pairs), every thread count the option takes, all three check-node rules."""
    import numpy as np
    rng = np.random.default_rng(3)
    H = rng.random((150, 400)) < 0.02
    H[7, :] = False                      # a check without variables
    H[:, 11] = False                     # a variable without checks
    H[20, :] = False
    H[20, rng.permutation(400)[:20]] = True   # a check at the degree limit of the engine
    sparse = (150, 400, *[a.astype(np.int32) for a in np.nonzero(H)])
    base = None
    for s in range(1, 4000):             # a quasi-cyclic draw with two empty blocks in one row: degrees 8 and 10, 4 and 5
        b, code = pkg.codes.random_qc(108, 5, 10, s)
        e = np.asarray(b) < 0
        if e.sum() == 2 and (e.sum(axis=1) == 2).any():
            base = code
            break
    assert base is not None
    cases = (("NR-BG1-shaped Z=16", pkg.codes.nr_bg1_shaped(16, 2)[1], 2.0, 300),
             ("NR-BG1-shaped Z=32", pkg.codes.nr_bg1_shaped(32, 3)[1], 2.0, 200),
             ("QC z=108, two empty blocks in one row", base, 3.0, 300),
             ("sparse random", sparse, 6.0, 300),
             ("(3,4) n=480", pkg.codes.regular_ldpc(480, 3, 4, 9), 4.0, 300))
    try:
        for name, (m, n, rows, cols), ebn0, F in cases:
            dec = pkg.Decoder(m, n, rows, cols)
            llr = oracle.gen_awgn_llr(m, n, ebn0, seed=21, first_frame=0, frames=F)
            for rule in (0, 1, 2):
                dec.set_cn_rule(rule)
                for early, L in ((True, 25), (False, 6)):
                    pkg.capi.set_default_options(engine="stream", stream_kernel=0)
                    ref = dec.decode_batch(llr, L, early)
                    assert dec.last_decode_kernel() == "rows_cn_kernel"
                    for threads in (0, 64, 128, 192, 256, 320, 512):
                        pkg.capi.set_default_options(engine="generic", generic_threads=threads)
                        out = dec.decode_batch(llr, L, early)
                        if dec.last_decode_kernel() != "bp_lds_gen_kernel":
                            assert threads == 64, (name, threads, dec.last_decode_kernel())  # too many steps for one wavefront
                            continue
                        for x, y in zip(ref, out):
                            assert (x == y).all(), (name, rule, early, threads)
            dec.close()
    finally:
        pkg.capi.set_default_options()
