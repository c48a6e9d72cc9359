"""Seeded parity-check matrix builders (host side, NumPy).

Every builder returns (m, n, rows, cols): the COO entries of H sorted row-major, which
is the order a dense row-major scan of H visits them (src/LDPC/LogDecoder.java:35-42)
and the order the text format stores them (src/LDPC/CheckMatrixIO.java:14-19).
"""
import numpy as np


def _sorted_coo(m, n, rows, cols):
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    key = np.unique(rows * n + cols)
    return int(m), int(n), (key // n).astype(np.int32), (key % n).astype(np.int32)


def gallager(n, wr, wc, seed):
    """Gallager regular code, the construction of src/LDPC/GenerateMatrix.java:11-34.

    Block 0: row i has ones in columns [i*wr, (i+1)*wr).  Block b >= 1 is block 0 with
    its columns permuted: H[blockRows*b + j][i] = H[j][index[i]], index a shuffled
    identity (the reference reshuffles the same list for every block; here each block
    draws a fresh seeded permutation, which has the same distribution).
    Requires n % wr == 0 (src/LDPC/Divisibility.java:4-7)."""
    if n % wr:
        raise ValueError("n must be a multiple of wr")
    block_rows = n // wr
    m = n * wc // wr
    rng = np.random.default_rng(seed)
    rows = [np.repeat(np.arange(block_rows), wr)]
    cols = [np.arange(n)]
    for b in range(1, wc):
        index = rng.permutation(n)
        # column i of block b carries a one in row (index[i] // wr)
        rows.append(block_rows * b + index // wr)
        cols.append(np.arange(n))
    return _sorted_coo(m, n, np.concatenate(rows), np.concatenate(cols))


def regular_ldpc(n, dv, dc, seed, max_repair_rounds=200):
    """(dv, dc)-regular code from a seeded socket permutation with repeated-edge repair.

    The reference generator cannot build this family when n % dc != 0
    (src/LDPC/GenerateMatrix.java:12-21 leaves rows empty); this is the build's own
    construction.  m = n*dv/dc."""
    if (n * dv) % dc:
        raise ValueError("n*dv must be a multiple of dc")
    m = n * dv // dc
    rng = np.random.default_rng(seed)
    vsock = np.repeat(np.arange(n), dv)
    csock = np.repeat(np.arange(m), dc)
    perm = rng.permutation(n * dv)
    rows = csock[perm].copy()
    cols = vsock.copy()
    for _ in range(max_repair_rounds):
        key = rows.astype(np.int64) * n + cols
        order = np.argsort(key, kind="stable")
        dup = np.zeros(key.size, bool)
        dup[order[1:]] = key[order[1:]] == key[order[:-1]]
        bad = np.nonzero(dup)[0]
        if bad.size == 0:
            break
        # swap the check end of every repeated edge with that of a random other edge
        other = rng.integers(0, key.size, bad.size)
        for a, b in zip(bad, other):
            rows[a], rows[b] = rows[b], rows[a]
    else:
        raise RuntimeError("could not remove repeated edges")
    return _sorted_coo(m, n, rows, cols)


def dvbs2_like(n=64800, seed=1, heavy_fraction=0.4, heavy_degree=8, light_degree=3, max_repair_rounds=200):
    """Irregular repeat-accumulate code of rate 1/2 with the degree profile of the long DVB-S2 rate-1/2 code
    (ETSI EN 302 307: n = 64800, 12960 information columns of degree 8, 19440 of degree 3, 32400 parity columns
    in a staircase of degree 2, the last of degree 1; E = 226,799).  The connections of the information part
    are a seeded socket permutation with repeated-edge repair, not the standard's address tables (which the
    reference does not contain either): the shape, not the code.  Check i holds parity columns k + i and
    k + i - 1; every check has the same number of information edges."""
    k = n // 2
    m = n - k
    heavy = int(round(k * heavy_fraction))
    deg = np.where(np.arange(k) < heavy, heavy_degree, light_degree)
    total = int(deg.sum())
    if total % m:
        raise ValueError("information edges must divide evenly over the checks")
    rng = np.random.default_rng(seed)
    cols = np.repeat(np.arange(k), deg)
    rows = np.repeat(np.arange(m), total // m)[rng.permutation(total)]
    for _ in range(max_repair_rounds):
        key = rows.astype(np.int64) * n + cols
        order = np.argsort(key, kind="stable")
        dup = np.zeros(key.size, bool)
        dup[order[1:]] = key[order[1:]] == key[order[:-1]]
        bad = np.nonzero(dup)[0]
        if bad.size == 0:
            break
        other = rng.integers(0, key.size, bad.size)
        for a, b in zip(bad, other):
            rows[a], rows[b] = rows[b], rows[a]
    else:
        raise RuntimeError("could not remove repeated edges")
    stair_r = np.concatenate([np.arange(m), np.arange(1, m)])
    stair_c = np.concatenate([k + np.arange(m), k + np.arange(m - 1)])
    return _sorted_coo(m, n, np.concatenate([rows, stair_r]), np.concatenate([cols, stair_c]))


def qc_expand(base, z):
    """Expand a base matrix of shifts (-1 = zero block) by circulants of size z:
    H[i*z + k][j*z + (k + s) % z] = 1  (src/LDPC/GenerateMatrix.java:139-150)."""
    base = np.asarray(base, dtype=np.int64)
    mb, nb = base.shape
    bi, bj = np.nonzero(base >= 0)
    k = np.arange(z)
    rows = (bi[:, None] * z + k[None, :]).reshape(-1)
    cols = (bj[:, None] * z + (k[None, :] + base[bi, bj][:, None]) % z).reshape(-1)
    return _sorted_coo(mb * z, nb * z, rows, cols)


def has_base_4cycle(base, z):
    """The base-level 4-cycle test of src/LDPC/GenerateMatrix.java:155-177."""
    base = np.asarray(base, dtype=np.int64)
    mb, nb = base.shape
    for i1 in range(mb):
        for i2 in range(i1 + 1, mb):
            both = np.nonzero((base[i1] >= 0) & (base[i2] >= 0))[0]
            if both.size < 2:
                continue
            d = (base[i1, both] - base[i2, both]) % z
            if np.unique(d).size < d.size:
                return True
    return False


def random_qc(z, mb, nb, seed, max_tries=100000):
    """Random QC code with the acceptance rules of src/LDPC/GenerateMatrix.java:91-137:
    shifts uniform in {-1, 0..z-1}, every base row and column has >= 2 blocks, no
    base-level 4-cycle."""
    rng = np.random.default_rng(seed)
    for _ in range(max_tries):
        base = rng.integers(0, z + 1, size=(mb, nb)) - 1
        if ((base >= 0).sum(axis=1) < 2).any() or ((base >= 0).sum(axis=0) < 2).any():
            continue
        if has_base_4cycle(base, z):
            continue
        return base, qc_expand(base, z)
    raise RuntimeError("no acceptable base matrix found")


def nr_bg1_shaped_base(z, seed):
    """A 46 x 68 base matrix with the shape of 5G NR base graph 1 (316 blocks, check
    degrees 3..19, variable degrees 1..30): 4 dense core rows over 22 information
    columns with a dual-diagonal 4-column parity core, then 42 sparse extension rows
    each closed by its own degree-1 parity column.  The shifts are seeded random
    values; the 3GPP tables themselves are not part of the reference or of this build."""
    rng = np.random.default_rng(seed)
    mb, nb, kb = 46, 68, 22
    base = -np.ones((mb, nb), np.int64)
    # core: rows 0-3, 19 blocks each = 15 of the 22 information columns + parity core
    core_parity = [(0, 22), (0, 23), (1, 22), (1, 23), (1, 24), (2, 23), (2, 24), (2, 25), (3, 22), (3, 24), (3, 25)]
    for r in range(4):
        npar = sum(1 for (rr, _) in core_parity if rr == r)
        info = rng.choice(kb, 19 - npar, replace=False)
        base[r, info] = rng.integers(0, z, info.size)
    # columns 0 and 1 take part in every core row (high-degree columns as in BG1)
    for r in range(4):
        for c in (0, 1):
            if base[r, c] < 0:
                drop = [x for x in np.nonzero(base[r, :kb] >= 0)[0] if x > 1]
                base[r, rng.choice(drop)] = -1
                base[r, c] = rng.integers(0, z)
    for (r, c) in core_parity:
        base[r, c] = 0 if (r, c) != (1, 22) else 1 % z
    # extension rows 4..45: own parity column 26 + (r - 4), the rest over columns 0..25
    used = int((base >= 0).sum())
    ext_rows = mb - 4
    budget = 316 - used - ext_rows
    # row weight decays down the graph as in BG1: 9 blocks near the core, 2 at the bottom
    weights = np.round(np.linspace(9, 2, ext_rows)).astype(np.int64)
    t = 0
    while weights.sum() != budget:
        step = 1 if weights.sum() < budget else -1
        if 2 <= weights[t % ext_rows] + step <= 9:
            weights[t % ext_rows] += step
        t += 1
    hub_left = {0: 26, 1: 26}  # columns 0/1: 4 core rows + at most 26 extension rows = degree 30
    for t in range(ext_rows):
        r = 4 + t
        base[r, 26 + t] = 0
        w = int(weights[t])
        picks = []
        for c in (0, 1):
            if len(picks) < w and hub_left[c] > 0 and rng.random() < 0.66:
                picks.append(c)
                hub_left[c] -= 1
        extra = rng.choice(np.arange(2, 26), w - len(picks), replace=False)
        for c in list(picks) + list(extra):
            base[r, c] = rng.integers(0, z)
    return base


def nr_bg1_shaped(z=384, seed=1):
    base = nr_bg1_shaped_base(z, seed)
    return base, qc_expand(base, z)


def degrees(m, n, rows, cols):
    dc = np.bincount(np.asarray(rows), minlength=m)
    dv = np.bincount(np.asarray(cols), minlength=n)
    return dc, dv
