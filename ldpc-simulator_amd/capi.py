"""ctypes binding of libldpc_amd.so (the C ABI declared in include/ldpc_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, loading or
creating a decoder raises.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldpc_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ldpc_amd.h")

OK, EINVAL, ENOMEM, EHIP, ENODEV, ELIMIT = 0, -1, -2, -3, -4, -5
CN_SUMPRODUCT, CN_MINSUM, CN_SUMPRODUCT_TANH = 0, 1, 2
LLR_UNIFORM, LLR_PARITY_LN1E20, LLR_PARITY_1EM7 = 0, 1, 2
CH_BSC_INFO, CH_BSC_ALL, CH_AWGN = 0, 1, 2
CNT_FRAMES, CNT_FRAME_ERRORS, CNT_INFO_BIT_ERRORS, CNT_CHECK_BIT_ERRORS = 0, 1, 2, 3
CNT_INFO_FRAME_ERRORS, CNT_CHECK_FRAME_ERRORS, CNT_UNDETECTED, CNT_MISCORRECTED = 4, 5, 6, 7
CNT_ITERS_CORRECT, CNT_ITERS_WRONG, CNT_CHANNEL_ERRORS, CNT_CHANNEL_ERRORS_SQ = 8, 9, 10, 11
CNT_FIXED = 16
# option keys of ldpc_amd_set_option / ldpc_amd_get_option (include/ldpc_amd.h)
OPT = {
    "engine": 1, "stream_kernel": 2, "nodes": 3, "sim_overlap": 4, "verbose": 5, "sim_fused": 6,
    "lds_wpe": 100, "lds_fpw": 101, "generic_lds_kb": 102, "nodes_blocks": 103, "fpl": 104, "layout": 105,
    "rows_waves": 106, "wps": 107, "nopipe": 108, "rows_fused": 109, "generic_threads": 110, "lds_vpt": 111, "rows_leaves": 112, "sim_chunk_mb": 113, "nodes_flat_barrier": 114, "small_call_kb": 115,
    "test_nodes_oversize": 190, "test_nodes_timeout": 191,
    "nodes_refusals": 200, "nodes_last_hip_error": 201, "noalloc_max_frames": 202, "nodes_barrier_timed_out": 203,
}
OPT_READ_ONLY = ("nodes_refusals", "nodes_last_hip_error", "noalloc_max_frames", "nodes_barrier_timed_out")
OPT_LIBRARY_DEFAULT = {"nodes": -1, "sim_overlap": 1, "sim_fused": 1, "rows_leaves": 1}   # every other option defaults to 0
ENGINE_AUTO, ENGINE_GENERIC, ENGINE_STREAM = 0, 1, 2
ENGINES = {"auto": ENGINE_AUTO, "generic": ENGINE_GENERIC, "stream": ENGINE_STREAM}

# Options every Decoder of this process applies before its next call, unless it has been given its own
# (Decoder.set_option).  Tests and measurement scripts switch engines here, for decoders that already
# exist as well; the library itself keeps options per context and reads no environment while decoding.
_default_options = {}
_default_options_version = 0


def set_default_options(**options):
    """Replace the process-wide option defaults, e.g. set_default_options(engine="stream", stream_kernel=1);
    called without arguments it goes back to the library's own defaults."""
    global _default_options, _default_options_version
    for k in options:
        if k not in OPT or k in OPT_READ_ONLY:
            raise KeyError("unknown or read-only option %r" % k)
    _default_options = dict(options)
    _default_options_version += 1

_vp, _i32, _i64, _u64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double

SIGNATURES = {
    "ldpc_amd_version": (C.c_int, []),
    "ldpc_amd_last_error": (C.c_char_p, [_vp]),
    "ldpc_amd_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(_vp)]),
    "ldpc_amd_host_free": (C.c_int, [_vp]),
    "ldpc_amd_create": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _i32, C.POINTER(_vp)]),
    "ldpc_amd_destroy": (C.c_int, [_vp]),
    "ldpc_amd_info": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ldpc_amd_graph": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ldpc_amd_set_cn_rule": (C.c_int, [_vp, _i32]),
    "ldpc_amd_set_workspace_limit": (C.c_int, [_vp, _u64]),
    "ldpc_amd_set_option": (C.c_int, [_vp, _i32, _i64]),
    "ldpc_amd_get_option": (C.c_int, [_vp, _i32, C.POINTER(_i64)]),
    "ldpc_amd_last_warning": (C.c_char_p, [_vp]),
    "ldpc_amd_decode_batch": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "ldpc_amd_decode_batch_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ldpc_amd_decode_batch_dev_noalloc": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ldpc_amd_decode_bsc_batch": (C.c_int, [_vp, _vp, _i64, _f64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ldpc_amd_decode_bsc_batch_dev": (C.c_int, [_vp, _vp, _i64, _f64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ldpc_amd_set_generator": (C.c_int, [_vp, _i32, _vp]),
    "ldpc_amd_gen_codewords_dev": (C.c_int, [_vp, _u64, _i64, _i64, _vp, _vp]),
    "ldpc_amd_gen_bsc_dev": (C.c_int, [_vp, _i32, _f64, _u64, _i64, _i64, _vp, _vp, _vp]),
    "ldpc_amd_gen_awgn_llr_dev": (C.c_int, [_vp, _f64, _u64, _i64, _i64, _vp, _vp, _vp]),
    "ldpc_amd_count_errors_dev": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "ldpc_amd_simulate": (C.c_int, [_vp, _i32, _f64, _u64, _i64, _i64, _i32, _i32, _vp]),
    "ldpc_amd_simulate_dev": (C.c_int, [_vp, _i32, _f64, _u64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "ldpc_amd_last_decode_timing": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ldpc_amd_last_decode_kernel": (C.c_char_p, [_vp]),
}

_lib = None


def _preload_hip_runtime():
    """Keep one HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so with
    the same SONAME as ROCm's; whichever copy is loaded first serves every later library that
    names that SONAME, but `import torch` after ROCm's copy would load the bundled file as a
    second runtime and then fail to see the GPU.  So when PyTorch is installed its copy is
    loaded first (without importing torch); otherwise libldpc_amd.so pulls in ROCm's."""
    if "torch" in sys.modules or os.environ.get("LDPC_AMD_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        path = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load_library():
    """dlopen libldpc_amd.so and declare every entry point; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s not found: build it with `make -C ldpc-simulator_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class LdpcError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("ldpc_amd error %d: %s" % (code, text))
        self.code = code


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HostBuffer:
    """Page-locked host memory from ldpc_amd_host_alloc, viewed as a NumPy array: the host-buffer entry
    points copy from and to it at link speed.  close() (or the context manager) releases it; views taken
    from `.array` must not be used afterwards."""

    def __init__(self, shape, dtype):
        lib = load_library()
        self.array = None
        self._ptr = _vp()
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        rc = lib.ldpc_amd_host_alloc(max(nbytes, 1), C.byref(self._ptr))
        if rc:
            raise LdpcError(rc, (lib.ldpc_amd_last_error(None) or b"").decode())
        raw = (C.c_char * max(nbytes, 1)).from_address(self._ptr.value)
        self.array = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._ptr is not None and self._ptr.value:
            self.array = None
            load_library().ldpc_amd_host_free(self._ptr)
            self._ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """One parity-check matrix on one GPU (an ldpc_amd_ctx)."""

    def __init__(self, m, n, rows, cols, device=-1):
        lib = load_library()
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        if rows.shape != cols.shape:
            raise ValueError("rows and cols must have the same length")
        h = C.c_void_p()
        rc = lib.ldpc_amd_create(int(m), int(n), rows.size, _np_ptr(rows), _np_ptr(cols), int(device), C.byref(h))
        if rc != OK:
            raise LdpcError(rc, (lib.ldpc_amd_last_error(None) or b"").decode())
        self._lib, self._h = lib, h
        self._own_options = {}
        self._options_version = 0   # the library's defaults are in force
        mm, nn, ee, dc, dv = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32()
        lib.ldpc_amd_info(h, C.byref(mm), C.byref(nn), C.byref(ee), C.byref(dc), C.byref(dv))
        self.m, self.n, self.E, self.max_dc, self.max_dv = mm.value, nn.value, ee.value, dc.value, dv.value

    @classmethod
    def from_dense(cls, H, device=-1):
        H = np.asarray(H)
        r, c = np.nonzero(H == 1)
        return cls(H.shape[0], H.shape[1], r, c, device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ldpc_amd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise LdpcError(rc, (self._lib.ldpc_amd_last_error(self._h) or b"").decode())

    @property
    def handle(self):
        return self._h

    def graph(self):
        rp = np.zeros(self.m + 1, np.int32)
        ec = np.zeros(max(self.E, 1), np.int32)
        cp = np.zeros(self.n + 1, np.int32)
        ce = np.zeros(max(self.E, 1), np.int32)
        self._check(self._lib.ldpc_amd_graph(self._h, _np_ptr(rp), _np_ptr(ec), _np_ptr(cp), _np_ptr(ce)))
        return rp, ec[: self.E], cp, ce[: self.E]

    def set_cn_rule(self, rule):
        self._check(self._lib.ldpc_amd_set_cn_rule(self._h, int(rule)))

    def set_option(self, name, value):
        """An option of this decoder alone; it overrides the process-wide defaults (set_default_options)."""
        if isinstance(value, str):
            value = ENGINES[value]
        self._check(self._lib.ldpc_amd_set_option(self._h, OPT[name], int(value)))
        self._own_options[name] = int(value)

    def get_option(self, name):
        v = _i64()
        self._check(self._lib.ldpc_amd_get_option(self._h, OPT[name], C.byref(v)))
        return v.value

    def last_warning(self):
        return (self._lib.ldpc_amd_last_warning(self._h) or b"").decode()

    def _sync_options(self):
        if self._options_version == _default_options_version:
            return
        for name, key in OPT.items():
            if name in OPT_READ_ONLY or name in self._own_options:
                continue
            value = _default_options.get(name, OPT_LIBRARY_DEFAULT.get(name, 0))
            if isinstance(value, str):
                value = ENGINES[value]
            self._check(self._lib.ldpc_amd_set_option(self._h, key, int(value)))
        self._options_version = _default_options_version

    def set_workspace_limit(self, nbytes):
        self._check(self._lib.ldpc_amd_set_workspace_limit(self._h, int(nbytes)))

    def set_generator(self, G):
        """G [k][n] 0/1 (e.g. TissueEncoder.encodeG's matrix): simulate() and gen_codewords_dev() then
        send random codewords u*G (LDPC_GallagerIntStreamSimu.java:115); None goes back to the all-zero word."""
        if G is None:
            self._check(self._lib.ldpc_amd_set_generator(self._h, 0, None))
            return
        G = np.ascontiguousarray(G, dtype=np.uint8)
        if G.ndim != 2 or G.shape[1] != self.n:
            raise ValueError("G must be [k][n]")
        self._check(self._lib.ldpc_amd_set_generator(self._h, G.shape[0], _np_ptr(G)))

    # ---- host-buffer entry points ------------------------------------------------------
    def decode_batch(self, llr, max_iters, early_term=True):
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        if llr.ndim != 2 or llr.shape[1] != self.n:
            raise ValueError("llr must be [frames][n]")
        self._sync_options()
        f = llr.shape[0]
        hard = np.zeros((f, self.n), np.uint8)
        it = np.zeros(f, np.int32)
        sw = np.zeros(f, np.int32)
        self._check(self._lib.ldpc_amd_decode_batch(self._h, _np_ptr(llr), f, int(max_iters), int(bool(early_term)),
                                                    _np_ptr(hard), _np_ptr(it), _np_ptr(sw)))
        return hard, it, sw

    def decode_bsc_batch(self, r, e, max_iters, llr_mode=LLR_PARITY_LN1E20, early_term=True):
        r = np.ascontiguousarray(r, dtype=np.uint8)
        if r.ndim != 2 or r.shape[1] != self.n:
            raise ValueError("r must be [frames][n]")
        self._sync_options()
        f = r.shape[0]
        hard = np.zeros((f, self.n), np.uint8)
        it = np.zeros(f, np.int32)
        sw = np.zeros(f, np.int32)
        self._check(self._lib.ldpc_amd_decode_bsc_batch(self._h, _np_ptr(r), f, float(e), int(llr_mode), int(max_iters),
                                                        int(bool(early_term)), _np_ptr(hard), _np_ptr(it), _np_ptr(sw)))
        return hard, it, sw

    def simulate(self, channel, param, seed, first_frame, frames, max_iters, early_term=True, counters=None):
        if counters is None:
            counters = np.zeros(CNT_FIXED + int(max_iters), np.uint64)
        assert counters.dtype == np.uint64 and counters.size >= CNT_FIXED + int(max_iters)
        self._sync_options()
        self._check(self._lib.ldpc_amd_simulate(self._h, int(channel), float(param), int(seed), int(first_frame),
                                                int(frames), int(max_iters), int(bool(early_term)), _np_ptr(counters)))
        return counters

    def simulate_dev(self, channel, param, seed, first_frame, frames, max_iters, early_term, d_counters, stream=0):
        """simulate() with the counters added into a device vector (address of 16 + max_iters 64-bit words);
        nothing is copied to the host and the call does not wait for the GPU."""
        self._sync_options()
        self._check(self._lib.ldpc_amd_simulate_dev(self._h, int(channel), float(param), int(seed), int(first_frame),
                                                    int(frames), int(max_iters), int(bool(early_term)), d_counters,
                                                    stream or None))

    # ---- device-pointer entry points (addresses as ints, e.g. torch.Tensor.data_ptr()) ---
    def decode_batch_dev(self, d_llr, frames, max_iters, early_term, d_hard, d_iters=0, d_synd=0, stream=0):
        self._sync_options()
        self._check(self._lib.ldpc_amd_decode_batch_dev(self._h, d_llr, int(frames), int(max_iters),
                                                        int(bool(early_term)), d_hard, d_iters or None,
                                                        d_synd or None, stream or None))

    def decode_batch_dev_noalloc(self, d_llr, frames, max_iters, early_term, d_hard, d_iters=0, d_synd=0, stream=0):
        """decode_batch_dev on the library's static device scratch array: allocates nothing, takes at most
        get_option("noalloc_max_frames") frames."""
        self._sync_options()
        self._check(self._lib.ldpc_amd_decode_batch_dev_noalloc(self._h, d_llr, int(frames), int(max_iters),
                                                                int(bool(early_term)), d_hard, d_iters or None,
                                                                d_synd or None, stream or None))

    def decode_bsc_batch_dev(self, d_r, frames, e, llr_mode, max_iters, early_term, d_hard, d_iters=0, d_synd=0,
                             stream=0):
        self._sync_options()
        self._check(self._lib.ldpc_amd_decode_bsc_batch_dev(self._h, d_r, int(frames), float(e), int(llr_mode),
                                                            int(max_iters), int(bool(early_term)), d_hard,
                                                            d_iters or None, d_synd or None, stream or None))

    # d_cw: the transmitted words [frames][n] uint8 on the device, 0 = the all-zero word
    def gen_codewords_dev(self, seed, first_frame, frames, d_cw, stream=0):
        self._check(self._lib.ldpc_amd_gen_codewords_dev(self._h, int(seed), int(first_frame), int(frames), d_cw,
                                                         stream or None))

    def gen_bsc_dev(self, channel, e, seed, first_frame, frames, d_r, stream=0, d_cw=0):
        self._check(self._lib.ldpc_amd_gen_bsc_dev(self._h, int(channel), float(e), int(seed), int(first_frame),
                                                   int(frames), d_cw or None, d_r, stream or None))

    def gen_awgn_llr_dev(self, ebn0_db, seed, first_frame, frames, d_llr, stream=0, d_cw=0):
        self._check(self._lib.ldpc_amd_gen_awgn_llr_dev(self._h, float(ebn0_db), int(seed), int(first_frame),
                                                        int(frames), d_cw or None, d_llr, stream or None))

    def count_errors_dev(self, frames, d_hard, d_rx_hard, d_iters, d_synd, max_iters, d_counters, stream=0, d_cw=0):
        self._check(self._lib.ldpc_amd_count_errors_dev(self._h, int(frames), d_hard, d_cw or None,
                                                        d_rx_hard or None, d_iters, d_synd, int(max_iters),
                                                        d_counters, stream or None))

    def last_decode_kernel(self):
        return (self._lib.ldpc_amd_last_decode_kernel(self._h) or b"").decode()

    def last_decode_timing(self):
        n, ms, fr = C.c_int32(), C.c_double(), C.c_int64()
        self._check(self._lib.ldpc_amd_last_decode_timing(self._h, C.byref(n), C.byref(ms), C.byref(fr)))
        return n.value, ms.value, fr.value
