"""Chunked container API (include/density_hip.h section 2): the data-parallel path.

Host-buffer calls stage through device memory; the *_device calls take device pointers (e.g. torch CUDA tensors'
data_ptr()) and a HIP stream handle and enqueue kernels only.
"""
import ctypes

from . import _lib
from .codec import DecodeError, EncodeError, _ro, _rw


def container_bound(algo, input_size, chunk_size=0):
    return _lib.lib().density_hip_container_bound(_lib.ALGO_IDS[algo], input_size, chunk_size)


def encode(algo, input, output, chunk_size=0):
    ia, n, k1 = _ro(input)
    oa, cap, k2 = _rw(output)
    r = _lib.lib().density_hip_encode(_lib.ALGO_IDS[algo], ia, n, oa, cap, chunk_size)
    if r == 0:
        raise EncodeError(_lib.last_error())
    return r


def decoded_size(container):
    ia, n, k = _ro(container)
    return _lib.lib().density_hip_decoded_size(ia, n)


def decode(container, output):
    ia, n, k1 = _ro(container)
    oa, cap, k2 = _rw(output)
    r = _lib.lib().density_hip_decode(ia, n, oa, cap)
    if r == 0 and _lib.last_error():          # 0 with no error == a valid, empty container
        raise DecodeError(_lib.last_error())
    return r


def parse_header(raw32):
    h = _lib.Header.from_buffer_copy(bytes(raw32[:32]))
    return h


FLAG_BLOCK_INDEX = 1
FLAG_SLOTTED = 2
FLAG_PAGED = 4
PAGE_BYTES = 65536


def slot_stride(algo, chunk_size):
    """Distance between the payload slots of a slotted container (include/density_hip.h: DENSITY_HIP_FLAG_SLOTTED)."""
    safe = getattr(_lib.lib(), f"{algo}_safe_encode_buffer_size")(chunk_size)
    return (safe + 255) // 256 * 256


def block_index(container):
    """The container's block index (bytes, one per 256-byte input block) or None."""
    b = bytes(container)
    h = parse_header(b)
    if not (h.flags & FLAG_BLOCK_INDEX):
        return None
    base = (32 + 4 * h.n_chunks + 15) // 16 * 16
    return b[base:base + (h.total_len + 255) // 256]


def chunk_payloads(container):
    """Splits a host-resident container into its per-chunk reference streams (for parity checks)."""
    b = bytes(container)
    h = parse_header(b)
    sizes = [int.from_bytes(b[32 + 4 * i:36 + 4 * i], "little") for i in range(h.n_chunks)]
    off = (32 + 4 * h.n_chunks + 15) // 16 * 16
    if h.flags & FLAG_BLOCK_INDEX:
        off = (off + (h.total_len + 255) // 256 + 15) // 16 * 16
    if h.flags & FLAG_PAGED:
        # a chunk's stream = the used bytes of its pages, in directory order (include/density_hip.h): what a CPU reader does before it calls the crate
        ppc = int(_lib.lib().density_hip_paged_pages_per_chunk(h.chunk_size))
        pages_base = (off + 16 * (ppc + 1) * h.n_chunks + 255) // 256 * 256
        out = []
        for i, s in enumerate(sizes):
            d = off + 16 * (ppc + 1) * i
            n_pages = int.from_bytes(b[d:d + 4], "little")
            parts = []
            for k in range(n_pages):
                e = d + 16 * (k + 1)
                page, used = int.from_bytes(b[e:e + 4], "little"), int.from_bytes(b[e + 8:e + 12], "little")
                parts.append(b[pages_base + page * PAGE_BYTES:pages_base + page * PAGE_BYTES + used])
            stream = b"".join(parts)
            assert len(stream) == s, f"chunk {i}: the directory's bytes ({len(stream)}) are not the size table's ({s})"
            out.append(stream)
        return h, out
    out = []
    stride = slot_stride(_lib.ALGO_NAMES[h.algo], h.chunk_size) if h.flags & FLAG_SLOTTED else 0
    for i, s in enumerate(sizes):
        if stride:
            out.append(b[off + i * stride:off + i * stride + s])
        else:
            out.append(b[off:off + s])
            off = (off + s + 15) // 16 * 16
    return h, out


def _check(rc, exc):
    if rc != _lib.OK:
        raise exc(f"density_hip error {rc}: {_lib.last_error()}")


def encode_device(algo, d_in, n, d_out, cap, chunk_size=0, stream=0, workspace=(0, 0), want_header=True):
    """Enqueue a container encode of device memory.  Returns the header (synchronises) or None."""
    hdr = _lib.Header() if want_header else None
    rc = _lib.lib().density_hip_encode_device(_lib.ALGO_IDS[algo], d_in, n, d_out, cap, chunk_size, workspace[0], workspace[1], stream,
                                              ctypes.byref(hdr) if want_header else None)
    _check(rc, EncodeError)
    return hdr


def container_bound_slotted(algo, input_size, chunk_size=0):
    return _lib.lib().density_hip_container_bound_slotted(_lib.ALGO_IDS[algo], input_size, chunk_size)


def encode_device_slotted(algo, d_in, n, d_out, cap, chunk_size=0, stream=0, workspace=(0, 0), want_header=True):
    """As encode_device, but every chunk stream stays in its slot inside the container (no stitch pass): DENSITY_HIP_FLAG_SLOTTED."""
    hdr = _lib.Header() if want_header else None
    rc = _lib.lib().density_hip_encode_device_slotted(_lib.ALGO_IDS[algo], d_in, n, d_out, cap, chunk_size, workspace[0], workspace[1], stream,
                                                      ctypes.byref(hdr) if want_header else None)
    _check(rc, EncodeError)
    return hdr


def container_bound_paged(algo, n, chunk_size=0):
    return int(_lib.lib().density_hip_container_bound_paged(_lib.ALGO_IDS[algo], n, chunk_size))


def encode_device_paged(algo, d_in, n, d_out, cap, chunk_size=0, stream=0, workspace=(0, 0), want_header=True):
    """As encode_device, but wire-ready WITHOUT a stitch pass: the streams in pages taken from one counter (DENSITY_HIP_FLAG_PAGED; what the paged form
    is not for comes out slotted: see the header's flags)."""
    hdr = _lib.Header() if want_header else None
    rc = _lib.lib().density_hip_encode_device_paged(_lib.ALGO_IDS[algo], d_in, n, d_out, cap, chunk_size, workspace[0], workspace[1], stream,
                                                    ctypes.byref(hdr) if want_header else None)
    _check(rc, EncodeError)
    return hdr


def pack_device(d_container, container_size, d_out, cap, header=None, stream=0, workspace=(0, 0), want_header=True):
    """Slotted container -> packed container (the wire form).  Returns the packed container's header (synchronises) or None."""
    hdr = _lib.Header() if want_header else None
    rc = _lib.lib().density_hip_pack_device(d_container, container_size, ctypes.byref(header) if header is not None else None, d_out, cap,
                                            workspace[0], workspace[1], stream, ctypes.byref(hdr) if want_header else None)
    _check(rc, EncodeError)
    return hdr


def decode_device(d_container, container_size, d_out, cap, header=None, stream=0, workspace=(0, 0), sync=True):
    size = ctypes.c_size_t(0)
    rc = _lib.lib().density_hip_decode_device(d_container, container_size, ctypes.byref(header) if header is not None else None, d_out, cap,
                                              workspace[0], workspace[1], stream, ctypes.byref(size) if sync else None)
    _check(rc, DecodeError)
    return size.value if sync else None


def stream_encode_device(algo, d_in, n, d_out, cap, stream=0):
    size = ctypes.c_size_t(0)
    _check(_lib.lib().density_hip_stream_encode_device(_lib.ALGO_IDS[algo], d_in, n, d_out, cap, stream, ctypes.byref(size)), EncodeError)
    return size.value


def stream_decode_device(algo, d_in, n, d_out, cap, stream=0):
    size = ctypes.c_size_t(0)
    _check(_lib.lib().density_hip_stream_decode_device(_lib.ALGO_IDS[algo], d_in, n, d_out, cap, stream, ctypes.byref(size)), DecodeError)
    return size.value


def set_kernel_variant(variant):
    """Bit mask (test hook): 1 = simple one-wavefront kernels, 2 = containers without the block index."""
    _lib.lib().density_hip_set_kernel_variant(int(variant))


def set_profiling(on):
    _lib.lib().density_hip_set_profiling(1 if on else 0)


def last_timings(cap=8192):
    ms = (ctypes.c_float * cap)()
    names = (ctypes.c_char_p * cap)()
    n = _lib.lib().density_hip_last_timings(ms, names, cap)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]
