"""Loads libdensity_hip.so (the HIP product library) and declares its C ABI (include/density_hip.h).

There is no fallback: if the library is missing this raises, and if no gfx950 device is usable every call returns
its error value (0 / error code) with density_hip_last_error() explaining why.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdensity_hip.so")

ALGO_IDS = {"chameleon": 0, "cheetah": 1, "lion": 2}
ALGO_NAMES = {v: k for k, v in ALGO_IDS.items()}
DEFAULT_CHUNK = 1 << 20

OK, ERR_ARGUMENT, ERR_CAPACITY, ERR_FORMAT, ERR_RUNTIME, ERR_UNSUPPORTED = range(6)


class Header(ctypes.Structure):
    """density_hip_header_t"""
    _fields_ = [("magic", ctypes.c_uint32), ("algo", ctypes.c_uint8), ("version", ctypes.c_uint8), ("flags", ctypes.c_uint16),
                ("chunk_size", ctypes.c_uint32), ("n_chunks", ctypes.c_uint32), ("total_len", ctypes.c_uint64),
                ("container_len", ctypes.c_uint64)]


class Shard(ctypes.Structure):
    """density_hip_shard_t"""
    _fields_ = [(k, ctypes.c_uint64) for k in ("chunk_first", "chunk_end", "byte_first", "byte_end")]


class GlobalLayout(ctypes.Structure):
    """density_hip_global_layout_t"""
    _fields_ = [(k, ctypes.c_uint64) for k in ("n_chunks", "total_len", "index_at", "index_bytes", "payload_at", "container_len",
                                                "chunk_offset", "payload_offset", "input_offset", "payload_bytes_padded")]


class MultiHeader(ctypes.Structure):
    """density_hip_multi_header_t"""
    _fields_ = [("magic", ctypes.c_uint32), ("version", ctypes.c_uint8), ("algo", ctypes.c_uint8), ("flags", ctypes.c_uint16),
                ("n_ranks", ctypes.c_uint32), ("chunk_size", ctypes.c_uint32), ("total_len", ctypes.c_uint64), ("container_len", ctypes.c_uint64)]


class MultiRow(ctypes.Structure):
    """density_hip_multi_row_t"""
    _fields_ = [(k, ctypes.c_uint64) for k in ("offset", "length", "input_bytes")]


_lib = None

# every symbol include/density_hip.h declares: name -> (restype, argtypes)
_SZ, _VP, _I = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
SYMBOLS = {}
for _a in ("chameleon", "cheetah", "lion"):
    SYMBOLS[f"{_a}_encode"] = (_SZ, [_VP, _SZ, _VP, _SZ])
    SYMBOLS[f"{_a}_decode"] = (_SZ, [_VP, _SZ, _VP, _SZ])
    SYMBOLS[f"{_a}_safe_encode_buffer_size"] = (_SZ, [_SZ])
SYMBOLS.update({
    "density_hip_auto_chunk": (_SZ, [_SZ]),
    "density_hip_stream_stats": (None, [ctypes.POINTER(ctypes.c_uint64)]),
    "density_hip_stage_stats": (None, [ctypes.POINTER(ctypes.c_uint64)]),
    "density_hip_auto_chunk_for": (_SZ, [_I, _SZ]),
    "density_hip_container_bound": (_SZ, [_I, _SZ, _SZ]),
    "density_hip_encode": (_SZ, [_I, _VP, _SZ, _VP, _SZ, _SZ]),
    "density_hip_decode": (_SZ, [_VP, _SZ, _VP, _SZ]),
    "density_hip_decoded_size": (_SZ, [_VP, _SZ]),
    "density_hip_encode_workspace_size": (_SZ, [_I, _SZ, _SZ]),
    "density_hip_decode_workspace_size": (_SZ, [ctypes.c_uint32]),
    "density_hip_decode_workspace_size_for": (_SZ, [_I, _SZ, _SZ]),
    "density_hip_decode_pass_count": (ctypes.c_uint64, []),
    "density_hip_encode_device": (_I, [_I, _VP, _SZ, _VP, _SZ, _SZ, _VP, _SZ, _VP, ctypes.POINTER(Header)]),
    "density_hip_decode_device": (_I, [_VP, _SZ, ctypes.POINTER(Header), _VP, _SZ, _VP, _SZ, _VP, ctypes.POINTER(_SZ)]),
    "density_hip_container_bound_slotted": (_SZ, [_I, _SZ, _SZ]),
    "density_hip_encode_device_slotted": (_I, [_I, _VP, _SZ, _VP, _SZ, _SZ, _VP, _SZ, _VP, ctypes.POINTER(Header)]),
    "density_hip_container_bound_paged": (_SZ, [_I, _SZ, _SZ]),
    "density_hip_paged_pages_per_chunk": (_SZ, [_SZ]),
    "density_hip_encode_device_paged": (_I, [_I, _VP, _SZ, _VP, _SZ, _SZ, _VP, _SZ, _VP, ctypes.POINTER(Header)]),
    "density_hip_pack_device": (_I, [_VP, _SZ, ctypes.POINTER(Header), _VP, _SZ, _VP, _SZ, _VP, ctypes.POINTER(Header)]),
    "density_hip_stream_encode_device": (_I, [_I, _VP, _SZ, _VP, _SZ, _VP, ctypes.POINTER(_SZ)]),
    "density_hip_stream_decode_device": (_I, [_I, _VP, _SZ, _VP, _SZ, _VP, ctypes.POINTER(_SZ)]),
    "density_hip_set_profiling": (None, [_I]),
    "density_hip_set_kernel_variant": (None, [_I]),
    "density_hip_last_timings": (_I, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p), _I]),
    "density_hip_selftest": (_I, []),
    "density_hip_selftest_bits": (_I, []),
    "density_hip_last_error": (ctypes.c_char_p, []),
    "density_hip_version": (ctypes.c_char_p, []),
    "density_hip_shard_range": (_I, [_SZ, _SZ, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Shard)]),
    "density_hip_global_layout": (_I, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(GlobalLayout)]),
    "density_hip_multi_layout": (_I, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32, _I, _SZ,
                                      ctypes.POINTER(MultiHeader), ctypes.POINTER(MultiRow)]),
    "density_hip_multi_row": (_I, [_VP, _SZ, _SZ, ctypes.c_uint32, ctypes.POINTER(MultiHeader), ctypes.POINTER(MultiRow)]),
    "density_hip_shutdown": (None, []),
})


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m density_amd.build` "
                               "(there is no CPU fallback in the product path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def use_debug_build():
    """Profiling / tuning tools only: switch this process to libdensity_hip_debug.so (-DDENSITY_HIP_DEBUG: the one build that reads DENSITY_HIP_PROF,
    DENSITY_HIP_TUNE, ... from the environment), building it if it is not there.  Call before the first lib()."""
    global _lib, LIB_PATH
    from . import build
    LIB_PATH = build.build(debug=True)
    _lib = None
    return lib()


def last_error():
    return lib().density_hip_last_error().decode()
