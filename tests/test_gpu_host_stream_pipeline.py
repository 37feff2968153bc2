"""The reference's own symbols on host pointers (chameleon.rs:70-78), long streams: `chameleon_encode` / `chameleon_decode` move the caller's buffers in
slices beside the kernels (api_stream.hip::host_stream_{encode,decode}_pipelined).  Whatever the data does to the speculation the slices ride on,
the stream is the oracle's stream byte for byte and decodes to the input; kernel variant 512 (no pipelining) gives the same bytes."""
import ctypes

import numpy as np
import pytest

import datagen
from density_amd import _lib, Chameleon, container
from density_amd.codec import DecodeError
from oracle import pyoracle

pytestmark = pytest.mark.gpu
ALGO = "chameleon"


def _stats():
    a = (ctypes.c_uint64 * 4)()
    _lib.lib().density_hip_stream_stats(a)
    return list(a)


def _inputs():
    n = (40 << 20) + 1234
    text = datagen.rep_text(n, period=1_000_003)
    island = text.copy()                                                  # a stretch of random bytes and one of zeros in the middle of the text
    island[(17 << 20):(20 << 20)] = datagen.random_bytes(3 << 20, seed=5)
    island[(29 << 20) + 77:(30 << 20)] = 0
    early = text.copy()                                                   # raw copies inside the very first slice
    early[(1 << 20):(1 << 20) + 70_000] = datagen.random_bytes(70_000, seed=6)
    return {"text": text, "island": island, "early": early, "random": datagen.random_bytes((33 << 20) + 2, seed=7),
            "prose": datagen.prose((36 << 20) + 255, seed=8)}


@pytest.mark.parametrize("kind", ["text", "island", "early", "random", "prose"])
def test_long_host_stream_is_the_oracle_stream(kind):
    data = _inputs()[kind]
    n = data.size
    want = np.frombuffer(pyoracle.encode(ALGO, data), dtype=np.uint8)
    out = np.zeros(Chameleon.safe_encode_buffer_size(n), dtype=np.uint8)
    s0 = _stats()
    m = Chameleon.encode(data, out)
    s1 = _stats()
    assert m == want.size and np.array_equal(out[:m], want), kind
    assert s1[0] - s0[0] == 1                                             # (in segments, not on one work-group)
    back = np.zeros(n, dtype=np.uint8)
    assert Chameleon.decode(out[:m], back) == n and np.array_equal(back, data), kind
    # a generous output buffer, and one that is too small
    big = np.zeros(n + (5 << 20), dtype=np.uint8)
    assert Chameleon.decode(out[:m], big) == n and np.array_equal(big[:n], data), kind
    with pytest.raises(DecodeError):
        Chameleon.decode(out[:m], np.zeros(n - 4096, dtype=np.uint8))
    # the staged path writes the same stream
    L = _lib.lib()
    L.density_hip_set_kernel_variant(512)
    try:
        out2 = np.zeros_like(out)
        assert Chameleon.encode(data, out2) == m and np.array_equal(out2[:m], want)
        back2 = np.zeros(n, dtype=np.uint8)
        assert Chameleon.decode(out[:m], back2) == n and np.array_equal(back2, data)
    finally:
        L.density_hip_set_kernel_variant(0)


def test_truncated_long_stream_ends_like_the_sequential_path():
    """Truncated, corrupted behind its head, or given too small an output: the same bytes back as on the one-work-group path, or DecodeError."""
    data = datagen.rep_text(48 << 20, period=1_000_003)
    enc = np.frombuffer(pyoracle.encode(ALGO, data), dtype=np.uint8)
    out = np.zeros(data.size, dtype=np.uint8)

    def both(stream, buf):
        res = []
        for variant in (0, 4):                                            # 4: the role pipelines -> the sequential stream path
            container.set_kernel_variant(variant)
            try:
                m = Chameleon.decode(stream, buf)
                res.append(("ok", m, buf[:m].tobytes()))
            except DecodeError:
                res.append(("error",))
        container.set_kernel_variant(0)
        return res

    for cut in (3, 1000, enc.size // 2 + 1):
        a, b = both(enc[:-cut].copy(), out)
        assert a[0] == b[0], (cut, a[0], b[0])
        if a[0] == "ok":
            assert a[1:] == b[1:], cut
    broken = enc.copy()
    broken[enc.size // 2 + 12345] ^= 0x5A
    a, b = both(broken, out)
    assert a[0] == b[0]
    if a[0] == "ok":
        assert a[1:] == b[1:]
