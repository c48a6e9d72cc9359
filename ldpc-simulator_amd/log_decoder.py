"""Host-side mirror of the reference's decoder classes over the C ABI.

Same names, argument meaning and result triple as
    LogDecoder.decode(int[][] encodedH, int[] r, double e, int maxL)
        -> DecodeResult(int[] decodedCode, int iterationNum, int syndrome)
(src/LDPC/LogDecoder.java:15-17), LogDecoderBefore (src/LDPC/LogDecoderBefore.java:15-17)
and MinSumDecoder (src/LDPC/MinSumDecoder.java:9-11).  A per-frame call is a batch of one
on the GPU; `decode_frames` is the batch form a driver should use instead of the per-frame
loop of LDPC_GallagerIntStreamSimu.java:99-184.
"""
import hashlib
from collections import OrderedDict, namedtuple

import numpy as np

from . import capi

DecodeResult = namedtuple("DecodeResult", ["decodedCode", "iterationNum", "syndrome"])

_CACHE = OrderedDict()
_CACHE_MAX = 8


def _decoder_for(encodedH, cn_rule, device=-1):
    H = np.ascontiguousarray(np.asarray(encodedH, dtype=np.int32))
    if H.ndim != 2:
        raise ValueError("encodedH must be a 2-D 0/1 matrix")
    key = (H.shape, hashlib.sha1(H.tobytes()).hexdigest(), cn_rule, device)
    dec = _CACHE.get(key)
    if dec is None:
        dec = capi.Decoder.from_dense(H, device)
        dec.set_cn_rule(cn_rule)
        _CACHE[key] = dec
        while len(_CACHE) > _CACHE_MAX:
            _, old = _CACHE.popitem(last=False)
            old.close()
    else:
        _CACHE.move_to_end(key)
    return dec


class _Base:
    LLR_MODE = capi.LLR_PARITY_LN1E20
    CN_RULE = capi.CN_SUMPRODUCT

    @classmethod
    def decode(cls, encodedH, r, e, maxL):
        """One frame, exactly the reference's signature and result."""
        dec = _decoder_for(encodedH, cls.CN_RULE)
        r = np.asarray(r).reshape(1, -1)
        if r.shape[1] != dec.n:
            # the reference would throw ArrayIndexOutOfBoundsException here (LogDecoder.java:50)
            raise IndexError("received word length %d does not match n = %d" % (r.shape[1], dec.n))
        hard, it, sw = dec.decode_bsc_batch(r.astype(np.uint8), e, maxL, cls.LLR_MODE, True)
        return DecodeResult(hard[0].astype(np.int32), int(it[0]), int(sw[0]))

    @classmethod
    def decode_frames(cls, encodedH, r, e, maxL):
        """Batch form: r is [frames][n]; returns (decodedCode[frames][n], iterationNum[frames],
        syndrome[frames])."""
        dec = _decoder_for(encodedH, cls.CN_RULE)
        return dec.decode_bsc_batch(np.asarray(r, dtype=np.uint8), e, maxL, cls.LLR_MODE, True)


class LogDecoder(_Base):
    """Sum-product; parity positions j >= n-m get +-ln(1e20) (LogDecoder.java:45-54)."""
    LLR_MODE = capi.LLR_PARITY_LN1E20


class LogDecoderBefore(_Base):
    """Sum-product; every position gets +-ln((1-e)/e) (LogDecoderBefore.java:45-50)."""
    LLR_MODE = capi.LLR_UNIFORM


class MinSumDecoder(_Base):
    """Normalised min-sum (x0.7); parity positions get +-ln((1-1e-7)/1e-7)
    (MinSumDecoder.java:39-48, :76-94)."""
    LLR_MODE = capi.LLR_PARITY_1EM7
    CN_RULE = capi.CN_MINSUM
