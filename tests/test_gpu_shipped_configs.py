"""Full-coverage parity at the configurations the library SHIPS — the chunk size density_hip_auto_chunk_for() picks — and on the hostile inputs
SURVEY.md 8d lists, default kernels, device-resident like the bench.  The shape is the reference bench's own: a full round-trip assertion
before anything is timed (benches/density.rs:41-45,83-87,125-129), here with EVERY chunk stream held against the oracle's stream of that chunk.

  * config 1: Chameleon, 10,192,446 B (dickens' size; non-periodic synthetic prose) at the automatic chunk (64 KiB)
  * config 3: Cheetah, 100,000,000 B at the automatic chunk (393,216 B: 96 trips of 4 KiB, not a power of two)
  * config 4: Lion, 100,000,000 B at the automatic chunk
  * all-zero 256 MiB (maximum hit rate, the zero-quad special case: chameleon.rs:88-100), xorshift-random 256 MiB (raw copies dominate:
    protection_state.rs:19-47), 64 MiB whose quads all share one hash (worst case inside a block) through the Chameleon container path;
    the same kinds at 32 MiB through Cheetah and Lion.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import datagen
from density_amd import _lib, container
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def xorshift_bytes(n, seed=0x9E3779B97F4A7C15):
    """n bytes of xorshift64* output (SURVEY.md Appendix C's generator), 65,536 independent lanes seeded by splitmix64 so that numpy can run
    them side by side; lane-major interleave of 8-byte words."""
    lanes = 1 << 16
    with np.errstate(over="ignore"):
        z = (np.arange(1, lanes + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        s = z ^ (z >> np.uint64(31))
        s[s == 0] = np.uint64(1)
        steps = -(-n // (8 * lanes))
        out = np.empty((steps, lanes), dtype=np.uint64)
        for k in range(steps):
            s ^= s >> np.uint64(12)
            s ^= s << np.uint64(25)
            s ^= s >> np.uint64(27)
            out[k] = s * np.uint64(0x2545F4914F6CDD1D)
    return out.reshape(-1).view(np.uint8)[:n].copy()


def hostile(kind, n):
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if kind == "random":
        return xorshift_bytes(n)
    if kind == "samehash":
        return datagen.same_hash_quads(n // 4 + 1, seed=3)[:n].copy()
    raise ValueError(kind)


def every_chunk_is_the_oracle_stream(algo, host, chunk):
    """container encode on the device at `chunk` (0: the library's choice): decode == input, and ALL chunk payloads == pyoracle.encode(chunk)."""
    import torch
    n = host.size
    auto = int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS[algo], n))
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound(algo, n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hdr = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    c = int(hdr.chunk_size)
    assert c == (chunk or auto) and hdr.n_chunks == -(-n // c) and hdr.total_len == n
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s) == n
    assert torch.equal(back, x)
    _, payloads = container.chunk_payloads(cont[:hdr.container_len].cpu().numpy())

    def check(i):
        return payloads[i] == pyoracle.encode(algo, host[i * c:(i + 1) * c])

    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        ok = list(ex.map(check, range(hdr.n_chunks)))
    assert all(ok), (algo, c, [i for i, v in enumerate(ok) if not v][:8])
    # the slotted form the bench times holds the same streams
    cap_s = container.container_bound_slotted(algo, n, c)
    cont_s = torch.empty(cap_s, dtype=torch.uint8, device="cuda")
    hs = container.encode_device_slotted(algo, x.data_ptr(), n, cont_s.data_ptr(), cap_s, c, stream=s)
    back.zero_()
    assert container.decode_device(cont_s.data_ptr(), hs.container_len, back.data_ptr(), n, header=hs, stream=s) == n and torch.equal(back, x)
    _, slotted = container.chunk_payloads(cont_s[:hs.container_len].cpu().numpy())
    assert slotted == payloads
    return hdr


_PROSE = {}


def prose_100m():
    if "d" not in _PROSE:
        _PROSE["d"] = datagen.prose(100_000_000, seed=0xD1B54A32D192ED03)
    return _PROSE["d"]


def test_config1_chameleon_at_the_shipped_chunk():
    host = datagen.prose(10_192_446, seed=0x9E3779B97F4A7C15)
    hdr = every_chunk_is_the_oracle_stream("chameleon", host, 0)
    assert hdr.chunk_size == 65536 and hdr.n_chunks == 156


def test_config3_cheetah_at_the_shipped_chunk():
    hdr = every_chunk_is_the_oracle_stream("cheetah", prose_100m(), 0)
    assert hdr.chunk_size == 393_216 and hdr.n_chunks == 255                       # one chunk per CU, 4 KiB trips: not a power of two


def test_config4_lion_at_the_shipped_chunk():
    hdr = every_chunk_is_the_oracle_stream("lion", prose_100m(), 0)
    assert hdr.chunk_size == int(_lib.lib().density_hip_auto_chunk_for(2, 100_000_000))


@pytest.mark.parametrize("kind,n", [("zeros", 256 << 20), ("random", 256 << 20), ("samehash", 64 << 20)])
def test_hostile_inputs_through_the_chameleon_container(kind, n):
    hdr = every_chunk_is_the_oracle_stream("chameleon", hostile(kind, n), 0)
    if kind == "zeros":
        assert hdr.container_len < n * 136 // 256 + (n >> 7)                         # every record is its signature + 64 hashes (chameleon.rs:88-100: a zero quad hits at once)
    if kind == "random":
        assert hdr.container_len > n                                                # README's "never exceeds" is an aim; safe_encode_buffer_size the guarantee


@pytest.mark.parametrize("algo", ["cheetah", "lion"])
@pytest.mark.parametrize("kind", ["zeros", "random", "samehash"])
def test_hostile_inputs_through_cheetah_and_lion(algo, kind):
    every_chunk_is_the_oracle_stream(algo, hostile(kind, 32 << 20), 0)


def test_xorshift_generator_is_the_appendix_c_generator():
    """one lane of xorshift_bytes against tests/datagen.py::xs_bytes, same seed state"""
    lanes = 1 << 16
    z = (1 * 0x9E3779B97F4A7C15 + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
    s0 = z ^ (z >> 31)
    got = xorshift_bytes(8 * lanes * 4).reshape(4, lanes, 8)[:, 0, :].reshape(-1).tobytes()
    assert got == datagen.xs_bytes(s0, 32)
