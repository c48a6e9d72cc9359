"""Generator matrix, systematic form and the column-permuted parity-check matrix (host side).

Same results as the reference's dense integer code, computed on bit-packed rows (64 columns per word):
  generator_matrix(H)  src/LDPC/GenerateMatrix.java:37-89   forward elimination on [H^T | I],
                                                           G = rows m.. of the right block
  encode_g(G)          src/LDPC/TissueEncoder.java:7-52     Gauss-Jordan with logged column swaps
  encode_h(H, swaps)   src/LDPC/TissueEncoder.java:55-70    replay of the swaps on H's columns
The decoder works on encode_h's output and codewords are u * encode_g's output (GenerateC.java:17-24).
Quirks kept: G always has n - m rows, also when H is rank deficient (every Gallager H is), and the
swap search of encode_g starts at column 0, not at the diagonal.
"""
import numpy as np

_U1 = np.uint64(1)


def _pack_rows(bits):
    """[rows][cols] 0/1 -> [rows][ceil(cols/64)] uint64, column j at bit j % 64 of word j // 64."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    rows, cols = bits.shape
    words = (cols + 63) // 64
    by = np.zeros((rows, words * 8), np.uint8)
    pk = np.packbits(bits, axis=1, bitorder="little")
    by[:, :pk.shape[1]] = pk
    return by.view("<u8").astype(np.uint64, copy=False)


def _unpack_rows(words, cols):
    by = np.ascontiguousarray(words).astype("<u8", copy=False).view(np.uint8)
    return np.unpackbits(by, axis=1, bitorder="little")[:, :cols]


def _column(words, j):
    return (words[:, j >> 6] >> np.uint64(j & 63)) & _U1


def generator_matrix(H):
    """Rows are held 64 columns to a word, and the two halves of [H^T | I] in separate arrays that take
    the same row operations."""
    H = (np.asarray(H) != 0).astype(np.uint8)
    m, n = H.shape
    L = _pack_rows(H.T)                          # [n][m]  H^T           (GenerateMatrix.java:44-48)
    R = _pack_rows(np.eye(n, dtype=np.uint8))    # [n][n]  I             (:49-53)
    for i in range(m):                           # :56-80
        col = _column(L, i)
        below = np.nonzero(col[i + 1:])[0] + i + 1
        if below.size == 0:
            continue
        if col[i] != 1:                          # swap row i with the first row below that has a 1;
            f = below[0]                         # that row then holds a 0 in column i and drops out
            L[[i, f]] = L[[f, i]]
            R[[i, f]] = R[[f, i]]
            below = below[1:]
        L[below] ^= L[i]
        R[below] ^= R[i]
    return _unpack_rows(R[m:], n).astype(np.int32)   # :83-87


def encode_g(G):
    """Returns (encodedG, swaps); swaps is the flat list (i, jc, i, jc, ...) the reference logs."""
    G = (np.asarray(G) != 0).astype(np.uint8)
    k, n = G.shape
    E = _pack_rows(G)
    swaps = []
    for i in range(k):                           # TissueEncoder.java:13-50
        if (E[i, i >> 6] >> np.uint64(i & 63)) & _U1 != 1:
            nz = np.nonzero(E[i])[0]
            if nz.size == 0:
                raise IndexError("row %d of G is zero (the reference throws IndexOutOfBounds here)" % i)
            w = int(E[i, nz[0]])
            jc = int(nz[0]) * 64 + ((w & -w).bit_length() - 1)   # first 1 of the row, searched from column 0
            swaps += [i, jc]
            diff = _column(E, i) ^ _column(E, jc)                # rows where the two columns differ
            E[:, i >> 6] ^= diff << np.uint64(i & 63)
            E[:, jc >> 6] ^= diff << np.uint64(jc & 63)
        rows = np.nonzero(_column(E, i))[0]
        rows = rows[rows != i]
        E[rows] ^= E[i]
    return _unpack_rows(E, n).astype(np.int32), swaps


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
