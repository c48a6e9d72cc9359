"""Second, independent double-precision restatement of the reference decoders (NumPy).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED by the reference (no tests or fixtures
there, no JVM here).  This file exists so that two restatements written separately
(this one and oracle/ldpc_oracle.c) can be checked against each other, and it is the
generator of the fixtures under tests/golden/.

Follows src/LDPC/LogDecoder.java:17-123 (parity positions get +-ln(1e20)),
src/LDPC/LogDecoderBefore.java:45-50 (every position +-l0) and
src/LDPC/MinSumDecoder.java:39-48, :76-94.

Java arithmetic is IEEE-754: 2.0/0.0 = +Infinity, Math.log(0.0) = -Infinity,
Math.tanh(+-Infinity) = +-1.0, NaN < 0 is false.  NumPy float64 scalars under
errstate(ignore) behave the same; plain Python floats raise instead, so every value
here is an np.float64.
"""
import numpy as np

F = np.float64


def adjacency(H):
    """A(i), B(j) by the row-major scan of LogDecoder.java:35-42."""
    H = np.asarray(H)
    m, n = H.shape
    A = [[] for _ in range(m)]
    B = [[] for _ in range(n)]
    for i in range(m):
        for j in range(n):
            if H[i, j] == 1:
                B[j].append(i)
                A[i].append(j)
    return A, B


def channel_llr(m, n, r, e, mode="ln1e20"):
    g_length = n - m
    with np.errstate(all="ignore"):
        l0 = np.log(F(1 - e) / F(e))
        if mode == "uniform":
            l1 = l0
        elif mode == "ln1e20":
            l1 = np.log(F(1 - 0.00000000000000000001) / F(0.00000000000000000001))
        elif mode == "1e-7":
            l1 = np.log(F(1 - 0.0000001) / F(0.0000001))
        else:
            raise ValueError(mode)
    lam = np.zeros(n, F)
    for j in range(n):
        mag = l0 if (mode == "uniform" or j < g_length) else l1
        lam[j] = mag if r[j] == 0 else -mag
    return lam


def _atanh(x):
    return F(0.5) * np.log((F(1.0) + x) / (F(1.0) - x))


def _signum(x):
    if x > 0:
        return F(1.0)
    if x < 0:
        return F(-1.0)
    return x


def decode_llr(H, lam, max_iters, early_term=True, rule="sum-product", trace=False):
    """Returns (hard[n] uint8, iterations, syndrome weight[, alpha dict, beta dict])."""
    H = np.asarray(H)
    m, n = H.shape
    A, B = adjacency(H)
    lam = np.asarray(lam, dtype=F)
    alpha = {(i, j): F(0.0) for i in range(m) for j in A[i]}
    beta = {(j, i): F(0.0) for j in range(n) for i in B[j]}
    est = np.zeros(n, np.uint8)
    sum_syndro = 0
    with np.errstate(all="ignore"):
        for l in range(max_iters):
            for j in range(n):
                for i in B[j]:
                    s = lam[j]
                    for k in B[j]:
                        if k == i:
                            continue
                        s = s + alpha[(k, j)]
                    beta[(j, i)] = s
            for i in range(m):
                for j in A[i]:
                    if rule == "sum-product":
                        p = F(1.0)
                        for k in A[i]:
                            if k == j:
                                continue
                            p = p * np.tanh(F(0.5) * beta[(k, i)])
                        alpha[(i, j)] = F(2) * _atanh(p)
                    else:
                        p = F(1.0)
                        for k in A[i]:
                            if k == j:
                                continue
                            p = p * _signum(beta[(k, i)])
                        mn = F(np.finfo(F).max)
                        for k in A[i]:
                            if k != j and mn > np.abs(beta[(k, i)]):
                                mn = np.abs(beta[(k, i)])
                        alpha[(i, j)] = p * mn * F(0.7)
            for j in range(n):
                g = lam[j]
                for k in B[j]:
                    g = g + alpha[(k, j)]
                est[j] = 1 if g < 0 else 0
            synd = (H.astype(np.int64) @ est.astype(np.int64)) % 2
            sum_syndro = int(synd.sum())
            if early_term and sum_syndro == 0:
                out = (est.copy(), l + 1, 0)
                return out + (alpha, beta) if trace else out
    out = (est.copy(), max_iters, sum_syndro)
    return out + (alpha, beta) if trace else out


def log_decoder(H, r, e, max_l, mode="ln1e20", rule="sum-product"):
    """LogDecoder.decode(encodedH, r, e, maxL) -> (decodedCode, iterationNum, syndrome)."""
    H = np.asarray(H)
    m, n = H.shape
    return decode_llr(H, channel_llr(m, n, r, e, mode), max_l, True, rule)


# ---- encoder chain, literal loops (GenerateMatrix.java:37-89, TissueEncoder.java:7-70, GenerateC.java:5-27)
def generator_matrix_ref(H):
    H = [[int(v) for v in row] for row in np.asarray(H)]
    m, n = len(H), len(H[0])
    A = [[0] * (n + m) for _ in range(n)]
    for i in range(n):
        for j in range(m):
            A[i][j] = H[j][i]
    for i in range(n):
        for j in range(m, m + n):
            A[i][j] = 1 if i + m == j else 0
    for i in range(m):
        tempo = [j for j in range(n) if j > i and A[j][i] == 1]
        if len(tempo) == 0:
            continue
        if A[i][i] != 1:
            A[i], A[tempo[0]] = A[tempo[0]], A[i]
        for j in tempo:
            if A[j][i] == 1:
                for k in range(n + m):
                    if A[i][k] == 1 and A[j][k] == 0:
                        A[j][k] = 1
                        continue
                    if A[i][k] == 1 and A[j][k] == 1:
                        A[j][k] = 0
    return np.array([[A[m + i][m + j] for j in range(n)] for i in range(n - m)], np.int32)


def encode_g_ref(G):
    E = [[int(v) for v in row] for row in np.asarray(G)]
    k, n = len(E), len(E[0])
    column = []
    for i in range(k):
        if E[i][i] != 1:
            col = None
            for jc in range(n):
                if E[i][jc] == 1:
                    col = jc
                    column += [i, jc]
                    break
            for r in range(k):
                E[r][col], E[r][i] = E[r][i], E[r][col]
        rows = [jr for jr in range(k) if E[jr][i] == 1 and jr != i]
        for jr in rows:
            for a in range(n):
                if E[i][a] == 1 and E[jr][a] == 1:
                    E[jr][a] = 0
                    continue
                if E[i][a] == 1 and E[jr][a] == 0:
                    E[jr][a] = 1
    return np.array(E, np.int32), column


def encode_h_ref(H, column):
    E = [[int(v) for v in row] for row in np.asarray(H)]
    for i in range(len(column) // 2):
        for j in range(len(E)):
            E[j][column[2 * i]], E[j][column[2 * i + 1]] = E[j][column[2 * i + 1]], E[j][column[2 * i]]
    return np.array(E, np.int32)


def gene_c_ref(encodedG, message):
    """GenerateC.geneC with the random message supplied by the caller."""
    G = np.asarray(encodedG, dtype=np.int64)
    return (np.asarray(message, dtype=np.int64) @ G % 2).astype(np.int32)
