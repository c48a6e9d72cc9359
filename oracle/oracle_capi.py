"""ctypes binding of the CPU oracle (oracle/libldpc_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  PARITY UNPINNED by the reference (it has no tests,
fixtures or seeded runs, and no JVM exists here); see ldpc_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libldpc_oracle.so")

CN_SUMPRODUCT, CN_MINSUM = 0, 1
LLR_UNIFORM, LLR_PARITY_LN1E20, LLR_PARITY_1EM7 = 0, 1, 2
CH_BSC_INFO, CH_BSC_ALL, CH_AWGN = 0, 1, 2
NUM_FIXED_COUNTERS = 16


def build():
    """Compile the oracle with gcc (no GPU, no reference sources involved)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    vp, i32, i64, u64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
    lib.oracle_graph_create.argtypes = [i32, i32, i64, vp, vp, C.POINTER(vp)]
    lib.oracle_graph_destroy.argtypes = [vp]
    lib.oracle_graph_edges.argtypes = [vp]
    lib.oracle_graph_edges.restype = i64
    lib.oracle_graph_csr.argtypes = [vp, vp, vp]
    lib.oracle_graph_csc.argtypes = [vp, vp, vp]
    lib.oracle_decode_llr.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.oracle_decode_llr_batch_f32.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, i32]
    lib.oracle_decode_llr_batch_f32_stored.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, i32, i32]
    lib.oracle_decode_llr_batch_f64.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, i32, i32, i32, u64]
    lib.oracle_decode_llr_noisy.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, u64]
    lib.oracle_bsc_llr.argtypes = [i32, i32, vp, f64, i32, vp]
    lib.oracle_logdecoder_dense.argtypes = [vp, i32, i32, vp, f64, i32, i32, i32, vp, vp, vp]
    lib.oracle_logdecoder_dense_batch.argtypes = [vp, i32, i32, vp, i64, f64, i32, i32, vp, vp, vp, i32]
    lib.oracle_philox4x32_10.argtypes = [vp, vp, vp]
    lib.oracle_gen_codewords.argtypes = [i32, i32, vp, u64, i64, i64, vp]
    lib.oracle_gen_bsc.argtypes = [i32, i32, i32, f64, u64, i64, i64, vp, vp]
    lib.oracle_gen_awgn_llr.argtypes = [i32, i32, f64, u64, i64, i64, vp, vp]
    lib.oracle_count_errors.argtypes = [i32, i32, i64, vp, vp, vp, vp, vp, i32, vp]
    lib.oracle_simulate.argtypes = [vp, i32, vp, i32, f64, u64, i64, i64, i32, i32, i32, vp, i32]
    lib.oracle_max_threads.restype = i32
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_threads():
    return int(lib().oracle_max_threads())


class Graph:
    """Tanner graph built from COO pairs (A(i)/B(j) ascending, LogDecoder.java:27-42)."""

    def __init__(self, m, n, rows, cols):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        assert rows.shape == cols.shape
        self.m, self.n = int(m), int(n)
        h = C.c_void_p()
        rc = lib().oracle_graph_create(self.m, self.n, rows.size, _p(rows), _p(cols), C.byref(h))
        if rc != 0:
            raise RuntimeError("oracle_graph_create failed: %d" % rc)
        self._h = h
        self.E = int(lib().oracle_graph_edges(h))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:   # at interpreter exit the module globals may be gone already
            lib().oracle_graph_destroy(self._h)
            self._h = None

    def csr(self):
        rp = np.zeros(self.m + 1, np.int32)
        ec = np.zeros(max(self.E, 1), np.int32)
        lib().oracle_graph_csr(self._h, _p(rp), _p(ec))
        return rp, ec[: self.E]

    def csc(self):
        cp = np.zeros(self.n + 1, np.int32)
        ce = np.zeros(max(self.E, 1), np.int32)
        lib().oracle_graph_csc(self._h, _p(cp), _p(ce))
        return cp, ce[: self.E]

    def decode_llr(self, llr, max_iters, early_term=True, cn_rule=CN_SUMPRODUCT, messages=False):
        llr = np.ascontiguousarray(llr, dtype=np.float64)
        assert llr.shape == (self.n,)
        hard = np.zeros(self.n, np.uint8)
        it, sw = C.c_int(0), C.c_int(0)
        al = np.zeros(max(self.E, 1)) if messages else None
        be = np.zeros(max(self.E, 1)) if messages else None
        rc = lib().oracle_decode_llr(self._h, _p(llr), int(max_iters), int(bool(early_term)), int(cn_rule),
                                     _p(hard), C.byref(it), C.byref(sw),
                                     _p(al) if messages else None, _p(be) if messages else None)
        if rc != 0:
            raise RuntimeError("oracle_decode_llr failed: %d" % rc)
        if messages:
            return hard, it.value, sw.value, al[: self.E], be[: self.E]
        return hard, it.value, sw.value

    def decode_llr_noisy(self, llr, max_iters, early_term, tanh_ulps, log_ulps, noise_seed, cn_rule=CN_SUMPRODUCT):
        """One frame with every tanh / log result moved by up to that many ulps; returns (hard, iters, synd, alpha, beta)."""
        llr = np.ascontiguousarray(llr, dtype=np.float64)
        assert llr.shape == (self.n,)
        hard = np.zeros(self.n, np.uint8)
        it, sw = C.c_int(0), C.c_int(0)
        al = np.zeros(max(self.E, 1))
        be = np.zeros(max(self.E, 1))
        rc = lib().oracle_decode_llr_noisy(self._h, _p(llr), int(max_iters), int(bool(early_term)), int(cn_rule),
                                           _p(hard), C.byref(it), C.byref(sw), _p(al), _p(be), int(tanh_ulps),
                                           int(log_ulps), int(noise_seed))
        if rc != 0:
            raise RuntimeError("oracle_decode_llr_noisy failed: %d" % rc)
        return hard, it.value, sw.value, al[: self.E], be[: self.E]

    def decode_llr_batch(self, llr, max_iters, early_term=True, cn_rule=CN_SUMPRODUCT, threads=None, storage=0):
        """storage 0: the reference's double messages.  1 / 2: every message rounded to a 4-byte float (in nats /
        in bits) when it is stored, all operations still double -- what float storage alone does to a decode."""
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        frames = llr.shape[0]
        assert llr.shape == (frames, self.n)
        hard = np.zeros((frames, self.n), np.uint8)
        it = np.zeros(frames, np.int32)
        sw = np.zeros(frames, np.int32)
        rc = lib().oracle_decode_llr_batch_f32_stored(self._h, _p(llr), frames, int(max_iters), int(bool(early_term)),
                                                      int(cn_rule), _p(hard), _p(it), _p(sw),
                                                      int(threads or max_threads()), int(storage))
        if rc != 0:
            raise RuntimeError("oracle_decode_llr_batch_f32_stored failed: %d" % rc)
        return hard, it, sw

    def decode_llr_batch_f64(self, llr, max_iters, early_term=True, cn_rule=CN_SUMPRODUCT, threads=None,
                             tanh_ulps=0, log_ulps=0, noise_seed=0):
        """Batch from double LLRs (the reference's own lambda in its BSC modes, LogDecoder.java:45-54).
        tanh_ulps / log_ulps > 0: every tanh / log result is moved by a random whole number of ulps of at
        most that size (the JDK bounds Math.tanh by 2.5 ulp and Math.log by 1 ulp)."""
        llr = np.ascontiguousarray(llr, dtype=np.float64)
        frames = llr.shape[0]
        assert llr.shape == (frames, self.n)
        hard = np.zeros((frames, self.n), np.uint8)
        it = np.zeros(frames, np.int32)
        sw = np.zeros(frames, np.int32)
        rc = lib().oracle_decode_llr_batch_f64(self._h, _p(llr), frames, int(max_iters), int(bool(early_term)),
                                               int(cn_rule), _p(hard), _p(it), _p(sw), int(threads or max_threads()),
                                               int(tanh_ulps), int(log_ulps), int(noise_seed))
        if rc != 0:
            raise RuntimeError("oracle_decode_llr_batch_f64 failed: %d" % rc)
        return hard, it, sw

    def simulate(self, channel, param, seed, first_frame, frames, max_iters, early_term=True,
                 cn_rule=CN_SUMPRODUCT, threads=None, G=None):
        counters = np.zeros(NUM_FIXED_COUNTERS + int(max_iters), np.uint64)
        if G is not None:
            G = np.ascontiguousarray(G, dtype=np.uint8)
            assert G.ndim == 2 and G.shape[1] == self.n
        rc = lib().oracle_simulate(self._h, 0 if G is None else G.shape[0], None if G is None else _p(G),
                                   int(channel), float(param), int(seed), int(first_frame), int(frames),
                                   int(max_iters), int(bool(early_term)), int(cn_rule), _p(counters),
                                   int(threads or max_threads()))
        if rc != 0:
            raise RuntimeError("oracle_simulate failed: %d" % rc)
        return counters


def bsc_llr(m, n, r, e, llr_mode):
    r = np.ascontiguousarray(r, dtype=np.uint8)
    out = np.zeros(n, np.float64)
    lib().oracle_bsc_llr(int(m), int(n), _p(r), float(e), int(llr_mode), _p(out))
    return out


def bsc_llr_batch(m, n, r, e, llr_mode):
    """double lambda of every frame of r [frames][n], as the reference forms it (LogDecoder.java:45-54)"""
    r = np.ascontiguousarray(r, dtype=np.uint8)
    return np.stack([bsc_llr(m, n, r[f], e, llr_mode) for f in range(r.shape[0])]) if r.shape[0] else np.zeros((0, n))


def logdecoder_dense(H, r, e, max_iters, llr_mode=LLR_PARITY_LN1E20, cn_rule=CN_SUMPRODUCT):
    """Line-by-line dense mode of LogDecoder.decode (LogDecoder.java:17-123)."""
    H = np.ascontiguousarray(H, dtype=np.int32)
    m, n = H.shape
    r = np.ascontiguousarray(r, dtype=np.int32)
    dec = np.zeros(n, np.int32)
    it, sy = C.c_int(0), C.c_int(0)
    rc = lib().oracle_logdecoder_dense(_p(H), m, n, _p(r), float(e), int(max_iters), int(llr_mode), int(cn_rule),
                                       _p(dec), C.byref(it), C.byref(sy))
    if rc != 0:
        raise RuntimeError("oracle_logdecoder_dense failed: %d" % rc)
    return dec, it.value, sy.value


def logdecoder_dense_batch(H, r, e, max_iters, llr_mode=LLR_PARITY_LN1E20, threads=None):
    H = np.ascontiguousarray(H, dtype=np.int32)
    m, n = H.shape
    r = np.ascontiguousarray(r, dtype=np.uint8)
    frames = r.shape[0]
    hard = np.zeros((frames, n), np.uint8)
    it = np.zeros(frames, np.int32)
    sy = np.zeros(frames, np.int32)
    rc = lib().oracle_logdecoder_dense_batch(_p(H), m, n, _p(r), frames, float(e), int(max_iters), int(llr_mode),
                                             _p(hard), _p(it), _p(sy), int(threads or max_threads()))
    if rc != 0:
        raise RuntimeError("oracle_logdecoder_dense_batch failed: %d" % rc)
    return hard, it, sy


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, np.uint32)
    lib().oracle_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def _cw(cw, frames, n):
    if cw is None:
        return None, None
    cw = np.ascontiguousarray(cw, dtype=np.uint8)
    assert cw.shape == (frames, n)
    return cw, _p(cw)


def gen_codewords(G, seed, first_frame, frames):
    """c = u * G for Philox-drawn messages u (GenerateC.java:5-27); G None -> zeros (needs n: pass an (0, n) array)."""
    G = np.ascontiguousarray(G, dtype=np.uint8)
    k, n = G.shape
    cw = np.zeros((frames, n), np.uint8)
    lib().oracle_gen_codewords(int(n), int(k), _p(G) if k else None, int(seed), int(first_frame), int(frames), _p(cw))
    return cw


def gen_bsc(m, n, channel, e, seed, first_frame, frames, cw=None):
    r = np.zeros((frames, n), np.uint8)
    cw, pc = _cw(cw, frames, n)
    lib().oracle_gen_bsc(int(m), int(n), int(channel), float(e), int(seed), int(first_frame), int(frames), pc, _p(r))
    return r


def gen_awgn_llr(m, n, ebn0_db, seed, first_frame, frames, cw=None):
    llr = np.zeros((frames, n), np.float32)
    cw, pc = _cw(cw, frames, n)
    lib().oracle_gen_awgn_llr(int(m), int(n), float(ebn0_db), int(seed), int(first_frame), int(frames), pc, _p(llr))
    return llr


def count_errors(m, n, hard, rx_hard, iters, synd, max_iters, counters=None, cw=None):
    hard = np.ascontiguousarray(hard, dtype=np.uint8)
    rx_hard = np.ascontiguousarray(rx_hard, dtype=np.uint8)
    iters = np.ascontiguousarray(iters, dtype=np.int32)
    synd = np.ascontiguousarray(synd, dtype=np.int32)
    cw, pc = _cw(cw, hard.shape[0], n)
    if counters is None:
        counters = np.zeros(NUM_FIXED_COUNTERS + int(max_iters), np.uint64)
    lib().oracle_count_errors(int(m), int(n), hard.shape[0], _p(hard), pc, _p(rx_hard), _p(iters), _p(synd),
                              int(max_iters), _p(counters))
    return counters
