"""Parity-check matrix text files in the reference's format.

Format (src/LDPC/CheckMatrixIO.java:8-24): first line "m n", then one line holding an
"i j " pair (each followed by one space) for every 1 of H in row-major order.  The
reader (CheckMatrixIO.java:27-52) takes whitespace-separated integers, stops at the
first non-integer token, drops a trailing unpaired integer and ignores pairs with
i >= m or j >= n; a repeated pair sets the same entry again.
"""
import numpy as np


def save_check_matrix(path, m, n, rows, cols):
    """Write the COO entries (rows, cols) of an m x n matrix, sorted row-major."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    order = np.lexsort((cols, rows))
    pairs = np.stack([rows[order], cols[order]], axis=1).reshape(-1)
    with open(path, "w") as f:
        f.write("%d %d\n" % (m, n))
        f.write("".join("%d " % v for v in pairs))


def save_dense(path, H):
    H = np.asarray(H)
    r, c = np.nonzero(H == 1)
    save_check_matrix(path, H.shape[0], H.shape[1], r, c)


def load_check_matrix(path):
    """Return (m, n, rows, cols) with rows/cols int32, row-major sorted, no repeats.

    Returns None when the file does not start with an integer, as the reference
    returns null (CheckMatrixIO.java:30)."""
    with open(path, "r") as f:
        tokens = f.read().split()
    vals = []
    for t in tokens:
        try:
            vals.append(int(t))
        except ValueError:
            break  # Scanner.hasNextInt() turns false at the first non-integer token
    if len(vals) < 2:
        return None
    m, n = vals[0], vals[1]
    body = vals[2:]
    if len(body) % 2:
        body = body[:-1]
    pairs = np.asarray(body, dtype=np.int64).reshape(-1, 2)
    keep = (pairs[:, 0] < m) & (pairs[:, 1] < n) & (pairs[:, 0] >= 0) & (pairs[:, 1] >= 0)
    pairs = pairs[keep]
    if pairs.size:
        key = np.unique(pairs[:, 0] * max(n, 1) + pairs[:, 1])
        rows, cols = key // max(n, 1), key % max(n, 1)
    else:
        rows = cols = np.zeros(0, np.int64)
    return m, n, rows.astype(np.int32), cols.astype(np.int32)


def to_dense(m, n, rows, cols):
    H = np.zeros((m, n), np.int32)
    H[np.asarray(rows), np.asarray(cols)] = 1
    return H
