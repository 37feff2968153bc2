"""ctypes front-end of oracle/libdensity_oracle.so (C restatement of density-rs 0.16.6).

TEST INFRASTRUCTURE ONLY — see oracle/density_oracle.c.  Builds the library on first use if missing.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdensity_oracle.so")
ALGOS = ("chameleon", "cheetah", "lion")
BLOCK_BYTES = {"chameleon": 256, "cheetah": 128, "lion": 64}
SIG_BYTES = {"chameleon": 8, "cheetah": 8, "lion": 6}


class Stats(ctypes.Structure):
    _fields_ = [("copy_blocks", ctypes.c_uint64), ("coded_blocks", ctypes.c_uint64), ("flags", ctypes.c_uint64 * 8)]


def build(force=False):
    src = os.path.join(_HERE, "density_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdensity_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        for a in ALGOS:
            for f in ("encode", "decode"):
                fn = getattr(L, f"oracle_{a}_{f}")
                fn.restype = ctypes.c_size_t
                fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            fn = getattr(L, f"oracle_{a}_encode_stats")
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Stats)]
            fn = getattr(L, f"oracle_{a}_safe_encode_buffer_size")
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.c_size_t]
        L.oracle_encode_chunks_mt.restype = ctypes.c_int
        L.oracle_encode_chunks_mt.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.oracle_decode_chunks_mt.restype = ctypes.c_int
        L.oracle_decode_chunks_mt.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
        _lib = L
    return _lib


def encode_chunks_mt(algo, src_addr, n, chunk, dst_addr, stride, sizes_addr, threads):
    """All chunks of a buffer, one chunk per OpenMP task (bench.py's all-cores baseline).  Returns the number of failed chunks."""
    return lib().oracle_encode_chunks_mt(ALGOS.index(algo), src_addr, n, chunk, dst_addr, stride, sizes_addr, threads)


def decode_chunks_mt(algo, src_addr, stride, sizes_addr, dst_addr, n, chunk, threads):
    return lib().oracle_decode_chunks_mt(ALGOS.index(algo), src_addr, stride, sizes_addr, dst_addr, n, chunk, threads)


def _as_buf(data):
    """bytes / bytearray / numpy uint8 array -> (address, length, keepalive)."""
    if isinstance(data, (bytes, bytearray)):
        b = (ctypes.c_char * len(data)).from_buffer_copy(data) if len(data) else (ctypes.c_char * 1)()
        return ctypes.addressof(b), len(data), b
    # numpy array
    return data.ctypes.data, data.size, data


def safe_encode_buffer_size(algo, n):
    return getattr(lib(), f"oracle_{algo}_safe_encode_buffer_size")(n)


def encode(algo, data, cap=None):
    addr, n, keep = _as_buf(data)
    cap = safe_encode_buffer_size(algo, n) if cap is None else cap
    out = ctypes.create_string_buffer(max(cap, 1))
    m = getattr(lib(), f"oracle_{algo}_encode")(addr, n, out, cap)
    return out.raw[:m]


def encode_stats(algo, data):
    addr, n, keep = _as_buf(data)
    cap = safe_encode_buffer_size(algo, n)
    out = ctypes.create_string_buffer(max(cap, 1))
    st = Stats()
    m = getattr(lib(), f"oracle_{algo}_encode_stats")(addr, n, out, cap, ctypes.byref(st))
    return out.raw[:m], {"copy_blocks": st.copy_blocks, "coded_blocks": st.coded_blocks, "flags": list(st.flags)}


def decode(algo, data, out_len):
    addr, n, keep = _as_buf(data)
    out = ctypes.create_string_buffer(max(out_len, 1))
    m = getattr(lib(), f"oracle_{algo}_decode")(addr, n, out, out_len)
    return out.raw[:m]


def encode_into(algo, src_addr, n, dst_addr, cap):
    """Raw-pointer form used by bench.py's cpu_baseline leg (no copies)."""
    return getattr(lib(), f"oracle_{algo}_encode")(src_addr, n, dst_addr, cap)


def decode_into(algo, src_addr, n, dst_addr, cap):
    return getattr(lib(), f"oracle_{algo}_decode")(src_addr, n, dst_addr, cap)
