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
