"""Generator matrix, systematic form and the column-permuted parity-check matrix (host side).

Same results as the reference's dense integer code, computed on whole rows at a time:
  generator_matrix(H)  src/LDPC/GenerateMatrix.java:37-89   forward elimination on [H^T | I],
                                                           G = rows m.. of the right block
  encode_g(G)          src/LDPC/TissueEncoder.java:7-52     Gauss-Jordan with logged column swaps
  encode_h(H, swaps)   src/LDPC/TissueEncoder.java:55-70    replay of the swaps on H's columns
The decoder works on encode_h's output and codewords are u * encode_g's output (GenerateC.java:17-24).
Quirks kept: G always has n - m rows, also when H is rank deficient (every Gallager H is), and the
swap search of encode_g starts at column 0, not at the diagonal.
"""
import numpy as np


def generator_matrix(H):
    H = (np.asarray(H) != 0).astype(np.uint8)
    m, n = H.shape
    A = np.zeros((n, m + n), np.uint8)          # [H^T | I]  (GenerateMatrix.java:44-53)
    A[:, :m] = H.T
    A[np.arange(n), m + np.arange(n)] = 1
    for i in range(m):                          # :56-80
        below = np.nonzero(A[i + 1:, i])[0] + i + 1
        if below.size == 0:
            continue
        if A[i, i] != 1:                        # swap row i with the first row below that has a 1
            A[[i, below[0]]] = A[[below[0], i]]
        sel = below[A[below, i] == 1]           # rows of the list that (still) have a 1 in column i
        A[sel] ^= A[i]
    return A[m:, m:].astype(np.int32)           # :83-87


def encode_g(G):
    """Returns (encodedG, swaps); swaps is the flat list (i, jc, i, jc, ...) the reference logs."""
    E = (np.asarray(G) != 0).astype(np.uint8)
    k = E.shape[0]
    swaps = []
    for i in range(k):                          # TissueEncoder.java:13-50
        if E[i, i] != 1:
            ones = np.nonzero(E[i])[0]
            if ones.size == 0:
                raise IndexError("row %d of G is zero (the reference throws IndexOutOfBounds here)" % i)
            jc = int(ones[0])
            swaps += [i, jc]
            E[:, [i, jc]] = E[:, [jc, i]]
        rows = np.nonzero(E[:, i])[0]
        rows = rows[rows != i]
        E[rows] ^= E[i]
    return E.astype(np.int32), swaps


def encode_h(H, swaps):
    E = np.array(H, dtype=np.int32, copy=True)
    for t in range(len(swaps) // 2):            # TissueEncoder.java:62-68
        a, b = swaps[2 * t], swaps[2 * t + 1]
        E[:, [a, b]] = E[:, [b, a]]
    return E


def systematic_pair(H):
    """(encodedH, encodedG) as the drivers build them (LDPC_GallagerIntStreamSimu.java:64-76)."""
    G = generator_matrix(H)
    eg, swaps = encode_g(G)
    return encode_h(H, swaps), eg
