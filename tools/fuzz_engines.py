#!/usr/bin/env python3
"""Differential fuzz of the engines: random codes, batch sizes, rules and modes; every engine that takes
the case must return the same bits as the row-parallel sweeps.

usage: tools/fuzz_engines.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ldpc_simulator_amd as L


def random_code(rng):
    kind = rng.integers(0, 5)
    if kind == 4:  # the shapes the specialised frame-resident kernels take, whole wavefronts or tiled as a larger code
        shape = int(rng.integers(0, 6))
        s = int(rng.integers(1, 1000))
        if shape == 0:
            return L.codes.regular_ldpc(512 * int(rng.integers(1, 9)), 3, 6, s)
        if shape == 1:
            return L.codes.gallager(512 * int(rng.integers(1, 9)), 8, 2, s)
        if shape == 2:
            return L.codes.gallager(8 * int(rng.integers(4, 300)), 8, 2 if rng.random() < 0.5 else 4, s)
        if shape == 3:
            return L.codes.regular_ldpc(12 * int(rng.integers(8, 340)), 3, 6, s)
        if shape == 4:
            return L.codes.regular_ldpc(20 * int(rng.integers(4, 200)), 5, 10, s)
        z = int(rng.integers(8, 128))
        base = rng.integers(0, z, (5, 10))
        if rng.random() < 0.5:   # empty blocks in distinct rows and columns: one edge short of (10,5) in places
            for r_, c_ in zip(rng.permutation(5)[: int(rng.integers(1, 4))], rng.permutation(10)):
                base[r_, c_] = -1
        return L.codes.qc_expand(base, z)   # a quasi-cyclic (10,5) code, with or without empty blocks
    if kind == 0:  # regular (dv, dc)
        dv, dc = [(3, 6), (2, 8), (3, 4), (4, 8), (2, 4)][rng.integers(0, 5)]
        n = int(rng.integers(2, 40)) * dc * (1 if rng.random() < 0.5 else 8)
        return L.codes.regular_ldpc(n, dv, dc, int(rng.integers(1, 1000)))
    if kind == 1:  # quasi-cyclic, random base
        z = int(rng.integers(2, 64))
        mb, nb = int(rng.integers(2, 6)), int(rng.integers(6, 12))
        base = np.where(rng.random((mb, nb)) < 0.6, rng.integers(0, z, (mb, nb)), -1)
        return L.codes.qc_expand(base, z)
    if kind == 2:  # NR-shaped, degrees to 19 / 30
        return L.codes.nr_bg1_shaped(int(rng.integers(1, 12)), int(rng.integers(1, 1000)))[1]
    # sparse random H with empty rows / columns allowed
    m, n = int(rng.integers(1, 200)), int(rng.integers(2, 400))
    H = rng.random((m, n)) < min(0.5, 3.0 / max(m, 1) + 2.0 / n)
    rows, cols = np.nonzero(H)
    return m, n, rows.astype(np.int32), cols.astype(np.int32)


def run(dec, call, options):
    L.capi.set_default_options(**options)
    out = call()
    kernel = dec.last_decode_kernel()
    L.capi.set_default_options()
    return out, kernel


def fuzz(cases, seed):
    """Runs the cases; returns ({kernel name: decodes}, number of mismatching decodes)."""
    rng = np.random.default_rng(seed)
    bad = 0
    seen = {}
    for c in range(cases):
        if c and c % 250 == 0:
            print("... %d cases, %d mismatches so far" % (c, bad), file=sys.stderr, flush=True)
        m, n, rows, cols = random_code(rng)
        if len(rows) == 0:
            continue
        dec = L.Decoder(m, n, rows, cols)
        if max(dec.max_dc, dec.max_dv) > 128:  # documented limit of the library
            dec.close()
            continue
        rule = [L.capi.CN_SUMPRODUCT, L.capi.CN_MINSUM, L.capi.CN_SUMPRODUCT_TANH][rng.integers(0, 3)] if rng.random() < 0.4 else L.capi.CN_SUMPRODUCT
        dec.set_cn_rule(rule)
        F = int([1, 2, 7, 31, 32, 33, 64, 65, 200, 700][rng.integers(0, 10)])
        iters = int([1, 2, 5, 12, 30][rng.integers(0, 5)])
        early = bool(rng.integers(0, 2))
        sigma = float(rng.uniform(0.5, 1.2))
        llr = (2.0 / sigma ** 2 * (1.0 + sigma * rng.standard_normal((F, n)))).astype(np.float32)
        if rng.random() < 0.1:
            llr[rng.integers(0, F), rng.integers(0, n)] = np.inf
        if rng.random() < 0.1:
            llr[rng.integers(0, F), rng.integers(0, n)] = 0.0
        if rng.random() < 0.3:  # hard received words, the reference's own input (LogDecoder.java:45-54)
            r = (rng.random((F, n)) < 0.04).astype(np.uint8)
            e = float(rng.uniform(0.005, 0.08))
            mode = int(rng.integers(0, 3))
            call = lambda: dec.decode_bsc_batch(r, e, iters, mode, early)
        else:
            call = lambda: dec.decode_batch(llr, iters, early)
        ref, _ = run(dec, call, {"engine": "stream", "stream_kernel": 0})
        for name, env in (("default", {}), ("generic", {"engine": "generic"}),
                          ("sweeps, leaves touched", {"engine": "stream", "stream_kernel": 0, "rows_leaves": 0}),
                          ("one-launch", {"engine": "stream", "stream_kernel": 1}),
                          ("nodes", {"generic_lds_kb": -1, "engine": "generic", "nodes": 1}),
                          ("nodes, one-level barrier", {"generic_lds_kb": -1, "engine": "generic", "nodes": 1, "nodes_flat_barrier": 1})):
            out, kernel = run(dec, call, env)
            seen[kernel] = seen.get(kernel, 0) + 1
            same = all(bool((a == b).all()) for a, b in zip(out, ref))
            if not same:
                bad += 1
                if os.environ.get("FUZZ_VERBOSE"):
                    fr = np.nonzero((out[0] != ref[0]).any(axis=1) | (out[1] != ref[1]) | (out[2] != ref[2]))[0]
                    print("  frames", fr[:10], "iters", out[1][fr[:10]], ref[1][fr[:10]], "synd", out[2][fr[:10]], ref[2][fr[:10]],
                          "bits differing", [(int((out[0][f] != ref[0][f]).sum())) for f in fr[:10]])
                    deg = np.bincount(rows, minlength=m)
                    print("  check degrees", np.bincount(deg), "has inf/zero llr", bool(np.isinf(llr).any()), bool((llr == 0).any()))
                print("MISMATCH case %d %s (%s): m=%d n=%d E=%d dc<=%d dv<=%d rule=%d F=%d iters=%d early=%d" % (
                    c, name, kernel, m, n, len(rows), dec.max_dc, dec.max_dv, rule, F, iters, early), flush=True)
        dec.close()
    return seen, bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    kernels, mismatches = fuzz(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("cases %d, kernels seen %s, mismatches %d" % (n_cases, kernels, mismatches))
    sys.exit(1 if mismatches else 0)
