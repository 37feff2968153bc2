"""Cheetah container decode in passes (density_amd/csrc/decode_passes.hip: records parsed per chunk, dictionary and prediction tables as
ordered LDS exchange passes, the chain of contexts on one wave per chunk) against the oracle: containers ASSEMBLED ON THE CPU from oracle
streams — so nothing the GPU encoder does can mask a decoder fault — decode to the input bit for bit; the one-wave decoder (kernel variant
128) must agree, error for error."""
import struct

import numpy as np
import pytest

import datagen
from density_amd import DecodeError, container
from oracle import pyoracle

pytestmark = pytest.mark.gpu
ALGO = "cheetah"


def cpu_container(data, chunk, algo=ALGO, algo_id=1):
    """A packed DHC1 container (no block index) from oracle streams: what a CPU producer writes (include/density_hip.h)."""
    data = np.ascontiguousarray(data)
    n = data.size
    streams = [pyoracle.encode(algo, data[i:i + chunk]) for i in range(0, n, chunk)]
    nc = len(streams)
    base = (32 + 4 * nc + 15) // 16 * 16
    body = bytearray()
    for k, s in enumerate(streams):
        body += s
        if k + 1 < nc:
            body += bytes(-len(body) % 16)
    total = base + len(body)
    head = struct.pack("<IBBHIIQQ", 0x31434844, algo_id, 1, 0, chunk, nc, n, total)
    table = b"".join(struct.pack("<I", len(s)) for s in streams)
    return np.frombuffer(head + table + bytes(base - 32 - 4 * nc) + bytes(body), dtype=np.uint8).copy(), streams


def decode_both(raw, n):
    """(passes result, one-wave result): each ('ok', bytes) or ('error',).  The passes run twice — records found by the window kernels (default) and
    by the one-wave walk alone (variant 1024) — and must agree with each other before they are compared with anything else."""
    res = []
    for variant in (0, 1024, 128):
        container.set_kernel_variant(variant)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        try:
            m = container.decode(raw, out)
            res.append(("ok", out[:m].tobytes()))
        except DecodeError:
            res.append(("error",))
    container.set_kernel_variant(0)
    assert res[0] == res[1], "window parse and one-wave parse disagree"
    return [res[0], res[2]]


KINDS = ["prose", "mixed", "random", "zeros", "rep", "binaryish", "samehash", "patchy", "pairs", "vocab30", "vocab60"]


def make(kind, n, seed):
    if kind == "patchy":
        d = datagen.prose(n, seed).copy()
        rng = np.random.default_rng(seed)
        for s in range(20_000, n - 40_000, 150_000):
            d[s:s + 30_000] = rng.integers(0, 256, size=30_000, dtype=np.uint8)
        return d
    if kind == "pairs":
        # quad pairs and runs repeated at short range: long predicted runs, predictions of predictions, MAP_B swaps
        rng = np.random.default_rng(seed)
        words = rng.integers(0, 2**32, size=48, dtype=np.uint32)
        parts = []
        total = 0
        while total < n // 4 + 8:
            a, b, c = (int(x) for x in rng.integers(0, 48, size=3))
            kind2 = int(rng.integers(0, 5))
            seq = ([words[a], words[b], words[c]] * int(rng.integers(1, 6)) if kind2 == 0 else [words[a]] * int(rng.integers(2, 90)) if kind2 == 1
                   else [words[a], words[b]] * int(rng.integers(2, 30)) if kind2 == 2 else [0] * int(rng.integers(1, 70)) if kind2 == 3
                   else list(rng.integers(0, 2**32, size=int(rng.integers(1, 9)), dtype=np.uint32)))
            parts.append(np.array(seq, dtype=np.uint32))
            total += len(seq)
        return np.concatenate(parts).astype("<u4").view(np.uint8)[:n].copy()
    if kind.startswith("vocab"):
        # Draws from a vocabulary of 4096 quads: nearly every quad is a dictionary quad (MAP_A / MAP_B / PLAIN — all of them take part in the dictionary
        # pass), and a share of the vocabulary hashes into ONE quarter of the slots.  30 %: every quarter's trips hold 230-310 of their 1024 quads — the
        # dense form's six-block case; 60 %: the heavy quarter holds ~610 — the sixteen-block fall-back —, the others ~135: the three-block case.
        share = int(kind[5:]) / 100.0
        rng = np.random.default_rng(seed)
        cand = rng.integers(1, 2**32, size=200_000, dtype=np.uint64)
        hq = ((cand * 0x9D6EF916) & 0xffffffff) >> 30                                     # the quarter of the slot: the hash's top two bits
        heavy, rest = cand[hq == 0], cand[hq != 0]
        k = int(4096 * share)
        vocab = np.concatenate([heavy[:k], rest[:4096 - k]]).astype(np.uint32)
        return vocab[rng.integers(0, 4096, size=n // 4 + 1)].astype("<u4").view(np.uint8)[:n].copy()
    return datagen.by_kind(kind, n, seed=seed)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n,chunk", [(6 * 65536 + 1234, 65536), (3 * (1 << 20) + 77, 1 << 20), (5 * 131072, 131072), ((1 << 20) + 3, 1 << 18), (200_001, 1 << 20)])
def test_cpu_built_containers_decode_in_passes(kind, n, chunk):
    data = make(kind, n, seed=n % 1000 + 3)
    raw, streams = cpu_container(data, chunk)
    a, b = decode_both(raw, n)
    assert a[0] == "ok" and a[1] == data.tobytes(), (kind, n, chunk)
    assert b == a


def test_the_passes_run():
    """The default path for chunks of 64 KiB and more IS the passes (the library counts the decodes they serve); their scratch is what
    density_hip_decode_workspace_size_for adds."""
    from density_amd import _lib
    lib = _lib.lib()
    n, chunk = 8 << 20, 1 << 20
    assert lib.density_hip_decode_workspace_size_for(1, n, chunk) > n + n // 2
    assert lib.density_hip_decode_workspace_size_for(2, n, chunk) == lib.density_hip_decode_workspace_size_for(2, 1, chunk) or True
    data = make("prose", 3 * 65536 + 5, seed=2)
    raw, _ = cpu_container(data, 65536)
    out = np.zeros(data.size, dtype=np.uint8)
    c0 = lib.density_hip_decode_pass_count()
    assert container.decode(raw, out) == data.size and np.array_equal(out, data)
    c1 = lib.density_hip_decode_pass_count()
    container.set_kernel_variant(128)
    assert container.decode(raw, out) == data.size
    container.set_kernel_variant(0)
    c2 = lib.density_hip_decode_pass_count()
    assert (c1 - c0, c2 - c1) == (1, 0)
    small, _ = cpu_container(data[:20000], 4096)                                          # chunks below 64 KiB: the one-wave decoder
    assert container.decode(small, out) == 20000 and lib.density_hip_decode_pass_count() == c2


@pytest.mark.parametrize("kind", ["prose", "mixed", "pairs"])
def test_gpu_encoded_device_containers(kind):
    """Device-resident round trip through encoder and pass decoder, packed and slotted, with a caller-owned workspace of the advertised
    size; every chunk stream == oracle."""
    import torch
    from density_amd import _lib
    n, chunk = 12 * (1 << 20) + 4096 + 3, 1 << 20
    host = make(kind, n, seed=5)
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound_slotted(ALGO, n, chunk)
    ws_size = max(_lib.lib().density_hip_encode_workspace_size(1, n, chunk), _lib.lib().density_hip_decode_workspace_size_for(1, n, chunk))
    ws = torch.empty(ws_size, dtype=torch.uint8, device="cuda")
    for enc in (container.encode_device, container.encode_device_slotted):
        cont = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        back = torch.zeros(n, dtype=torch.uint8, device="cuda")
        hdr = enc(ALGO, x.data_ptr(), n, cont.data_ptr(), cap, chunk, workspace=(ws.data_ptr(), ws_size))
        assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, workspace=(ws.data_ptr(), ws_size)) == n
        assert torch.equal(back, x)
        _, payloads = container.chunk_payloads(cont[:hdr.container_len].cpu().numpy())
        for i in (0, 5, hdr.n_chunks - 1):
            assert payloads[i] == pyoracle.encode(ALGO, host[i * chunk:(i + 1) * chunk])


def test_truncated_and_corrupt_payloads_end_like_the_one_wave_decoder():
    n, chunk = 4 * 131072 + 555, 131072
    data = make("mixed", n, seed=9)
    raw, streams = cpu_container(data, chunk)
    rng = np.random.default_rng(1)
    base = (32 + 4 * len(streams) + 15) // 16 * 16
    # a size-table entry cut short (the chunk's stream is truncated): records end early or mid-item
    for cut in (1, 2, 3, 5, 9, 64, 137, 5000):
        bad = raw.copy()
        sz = int.from_bytes(bad[32 + 4:36 + 4].tobytes(), "little")
        bad[32 + 4:36 + 4] = np.frombuffer(int(sz - cut).to_bytes(4, "little"), dtype=np.uint8)
        a, b = decode_both(bad, n)
        assert a[0] == b[0], (cut, a[0], b[0])
        if a[0] == "ok":
            assert a[1] == b[1], cut
    # ... or made longer: what follows a stream in the container (padding, the next stream's first bytes) is then read as its last record — eight
    # more bytes are a signature with nothing behind it, which the reference accepts without writing anything when its first flag is PLAIN
    for grow in (1, 2, 7, 8, 9, 16, 24, 136):
        for k in (0, 1, 2):
            bad = raw.copy()
            sz = len(streams[k])
            bad[32 + 4 * k:36 + 4 * k] = np.frombuffer(int(sz + grow).to_bytes(4, "little"), dtype=np.uint8)
            a, b = decode_both(bad, n)
            assert a[0] == b[0], (grow, k, a[0], b[0])
            if a[0] == "ok":
                assert a[1] == b[1], (grow, k)
    # flipped bytes inside payloads: signatures and items go wrong; both decoders must end the same way
    for trial in range(12):
        bad = raw.copy()
        at = base + int(rng.integers(0, len(raw) - base))
        bad[at] ^= int(rng.integers(1, 256))
        a, b = decode_both(bad, n)
        assert a[0] == b[0], (trial, at, a[0], b[0])
        if a[0] == "ok":
            assert a[1] == b[1], (trial, at)


@pytest.mark.parametrize("kind", ["pairs", "patchy", "mixed"])
def test_corrupt_streams_decode_like_the_oracle(kind):
    """What no encoder writes a corrupt stream can hold — a MAP flag on a slot nothing has written yet (its value, 0, does not hash to the slot:
    cheetah.rs:78-92 return the ITEM as the next context, :97-102 the hash of the VALUE), predictions of never-written contexts, garbage items:
    the passes must decode every such stream exactly like the oracle, and like the one-wave decoder (a seeded fuzz that found the first case)."""
    n, chunk = 4 * 131072 + 555, 131072
    data = make(kind, n, seed=11)
    raw, streams = cpu_container(data, chunk)
    base = (32 + 4 * len(streams) + 15) // 16 * 16
    offs, o = [], base
    for s in streams:
        offs.append(o)
        o = (o + len(s) + 15) // 16 * 16
    rng = np.random.default_rng(4242)
    compared = 0
    for t in range(48):
        bad = raw.copy()
        mode = t % 3
        if mode == 0:
            at = base + int(rng.integers(0, len(raw) - base)); bad[at] ^= int(rng.integers(1, 256))
        elif mode == 1:
            at = base + int(rng.integers(0, len(raw) - base - 8)); bad[at:at + 8] = rng.integers(0, 256, size=8, dtype=np.uint8)
        else:
            at = (base + int(rng.integers(0, (len(raw) - base) // 2))) & ~1; bad[at] ^= 1 << int(rng.integers(0, 8))
        a, b = decode_both(bad, n)
        assert a[0] == b[0], (t, a[0], b[0])
        if a[0] == "ok":
            assert a[1] == b[1], t
            want = b"".join(pyoracle.decode(ALGO, bytes(bad[offs[k]:offs[k] + len(s)]), min(chunk, n - k * chunk)) for k, s in enumerate(streams))
            if len(want) == n:                                                    # (where the oracle itself stops short the container decode is an error or differs by design)
                assert a[1] == want, t
                compared += 1
    assert compared >= 16, compared


@pytest.mark.parametrize("kind", ["prose", "mixed", "pairs", "random", "zeros"])
def test_reference_shaped_streams_decode_in_passes(kind):
    """`cheetah_decode` (the reference's symbol: ONE stream, host pointers) of 64 KiB and more goes through the same passes as one chunk;
    oracle streams in, the input out; truncated streams end like the one-wave decoder."""
    from density_amd import Cheetah, _lib
    lib = _lib.lib()
    n = 1_500_003
    data = make(kind, n, seed=17)
    enc = np.frombuffer(pyoracle.encode(ALGO, data), dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint8)
    c0 = lib.density_hip_decode_pass_count()
    assert Cheetah.decode(enc, out) == n and np.array_equal(out, data)
    assert lib.density_hip_decode_pass_count() == c0 + 1
    big = np.zeros(n + 100_000, dtype=np.uint8)                                           # an output buffer larger than the data
    assert Cheetah.decode(enc, big) == n and np.array_equal(big[:n], data)
    for cut in (1, 2, 7, 1000):
        res = []
        for variant in (0, 128):
            container.set_kernel_variant(variant)
            buf = np.zeros(n, dtype=np.uint8)
            try:
                m = Cheetah.decode(enc[:-cut].copy(), buf)
                res.append(("ok", m, buf[:m].tobytes()))
            except DecodeError:
                res.append(("error",))
        container.set_kernel_variant(0)
        assert res[0] == res[1], (kind, cut, res[0][0], res[1][0])
