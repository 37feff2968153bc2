"""GPU parity tests for Cheetah and Lion (density_amd/csrc/serial_codec.hip, exchange_stages.hip) through the same C ABI, bit-exact
against the CPU oracle: every test runs on the default kernels (one wave per stream; containers of 64 / 128 KiB chunks and more encode
in passes of ordered LDS exchanges), on the one-lane-per-stream kernels (kernel variant 16) and with the exchange passes off (variant 32)."""
import hashlib
import json
import os

import numpy as np
import pytest

import datagen
from density_amd import BY_NAME, DecodeError, EncodeError, container
from oracle import pyoracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
ALGOS = ["cheetah", "lion"]
VARIANTS = {"default": 0, "lane": 16, "wave": 32}


@pytest.fixture(autouse=True, params=list(VARIANTS))
def kernel_variant(request):
    container.set_kernel_variant(VARIANTS[request.param])
    yield request.param
    container.set_kernel_variant(0)


def gpu_encode(algo, data):
    C = BY_NAME[algo]
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    out = np.zeros(max(C.safe_encode_buffer_size(data.size), 1), dtype=np.uint8)
    n = C.encode(data, out)
    return out[:n].tobytes()


def gpu_decode(algo, enc, n):
    out = np.zeros(max(n, 1), dtype=np.uint8)
    m = BY_NAME[algo].decode(np.frombuffer(bytes(enc), dtype=np.uint8), out)
    return out[:m].tobytes()


@pytest.mark.parametrize("algo", ALGOS)
def test_reference_golden_vector_on_gpu(algo):
    """src/lib.rs:44-64 (cheetah), :66-86 (lion): exact bytes into a len(input)-byte buffer, then decode == input."""
    data = bytes.fromhex(KAT["reference_input_hex"])
    out = bytearray(len(data))
    n = BY_NAME[algo].encode(data, out)
    assert bytes(out[:n]) == bytes.fromhex(KAT["reference"][algo])
    back = bytearray(len(data))
    m = BY_NAME[algo].decode(bytes(out[:n]), back)
    assert bytes(back[:m]) == data


@pytest.mark.parametrize("algo", ALGOS)
def test_committed_kat_fixtures_on_gpu(algo):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_kat", os.path.join(HERE, "golden", "make_kat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, data in mod.derived_inputs().items():
        if not data:
            continue
        row = KAT["derived"][name][algo]
        enc = gpu_encode(algo, data)
        assert (len(enc), hashlib.sha256(enc).hexdigest()) == (row["len"], row["sha256"]), name
        assert gpu_decode(algo, enc, len(data)) == data, name


EDGE = sorted(set(list(range(1, 20)) + [63, 64, 65, 69, 70, 71, 127, 128, 129, 135, 136, 137, 255, 256, 257, 263, 264, 265, 1023, 1024, 1025, 4093, 4096, 4099]))


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("kind", ["prose", "random", "zeros", "mixed", "samehash"])
def test_stream_parity_edge_sizes(algo, kind):
    big = datagen.by_kind(kind, 5000, seed=31)
    for n in EDGE:
        data = big[:n].copy()
        want = pyoracle.encode(algo, data)
        assert gpu_encode(algo, data) == want, (kind, n)
        assert gpu_decode(algo, want, n) == data.tobytes(), (kind, n)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("kind,n", [("prose", 200_003), ("mixed", 300_000), ("random", 50_001), ("rep", 250_000), ("binaryish", 120_001)])
def test_stream_parity_large(algo, kind, n):
    """Whole-stream parity: multi-block dictionary/predictor carry, MAP_B, PREDICTED (B..E for Lion), copy mode in and out."""
    data = datagen.by_kind(kind, n, seed=78)
    want, st = pyoracle.encode_stats(algo, data)
    assert gpu_encode(algo, data) == want
    if kind in ("mixed", "random"):
        assert st["copy_blocks"] > 0
    assert gpu_decode(algo, want, n) == data.tobytes()


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("chunk", [256, 4096, 65536])
def test_container_chunks_match_oracle(algo, chunk):
    n = 24 * chunk + 77 if chunk <= 4096 else 6 * chunk + 1234
    data = datagen.by_kind("mixed", n, seed=chunk + 1)
    cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
    cn = container.encode(algo, data, cont, chunk)
    hdr, payloads = container.chunk_payloads(cont[:cn])
    assert (hdr.algo, hdr.total_len, hdr.chunk_size, hdr.n_chunks, hdr.flags) == ({"cheetah": 1, "lion": 2}[algo], n, chunk, -(-n // chunk), 0)
    for i, p in enumerate(payloads):
        assert p == pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk]), (chunk, i)
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == n
    assert np.array_equal(back, data)


@pytest.mark.parametrize("algo", ALGOS)
def test_errors_are_reported(algo):
    data = datagen.prose(3000, 41)
    enc = pyoracle.encode(algo, data)
    C = BY_NAME[algo]
    with pytest.raises(DecodeError):
        C.decode(enc, np.zeros(100, dtype=np.uint8))
    with pytest.raises(EncodeError):
        C.encode(datagen.random_bytes(3000, 1), np.zeros(100, dtype=np.uint8))
    # a truncated stream ends exactly like the oracle's decode of it: the same bytes, or an error where the oracle returns 0
    # (the reference panics there: io/read_buffer.rs:22)
    wrong = []
    for kind, n in (("prose", 3000), ("mixed", 70_000), ("random", 5000)):
        data = datagen.by_kind(kind, n, seed=41)
        enc = pyoracle.encode(algo, data)
        out = np.zeros(n, dtype=np.uint8)
        for cut in (1, 2, 3, 5, 9, 100, 257, 1000):
            want = pyoracle.decode(algo, enc[:-cut], n)
            try:
                m = C.decode(enc[:-cut], out)
                got = out[:m].tobytes()
            except DecodeError:
                got = b""
            if got != want:
                wrong.append((kind, cut, len(got), len(want)))
    assert not wrong, wrong


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("kind", ["prose", "rep", "random", "zeros", "samehash", "binaryish"])
def test_container_chunks_match_oracle_by_kind(algo, kind):
    chunk = 65536
    n = 5 * chunk + 4321
    data = datagen.by_kind(kind, n, seed=7)
    cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
    cn = container.encode(algo, data, cont, chunk)
    hdr, payloads = container.chunk_payloads(cont[:cn])
    for i, p in enumerate(payloads):
        assert p == pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk]), (kind, i)
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == n
    assert np.array_equal(back, data)


_PROSE_100M = {}


def synth_prose_100m():
    """BASELINE configs 3 / 4 stand-in (enwik8 is not available): 100,000,000 bytes of non-periodic synthetic prose, SURVEY.md §8d seed."""
    if "d" not in _PROSE_100M:
        _PROSE_100M["d"] = datagen.prose(100_000_000, seed=0xD1B54A32D192ED03)
    return _PROSE_100M["d"]


@pytest.mark.parametrize("algo", ["chameleon", "cheetah", "lion"])
def test_config34_full_coverage_parity(algo, kernel_variant):
    if kernel_variant != "default":
        pytest.skip("full-size configs run on the default kernels")
    """BASELINE configs 3 and 4 at full size, at the chunk size the library ships as default: EVERY chunk stream equals the oracle's
    stream of that chunk bit for bit, and decode(container) == input (device-resident, like the bench)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    host = synth_prose_100m()
    n, chunk = host.size, 1 << 20
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound(algo, n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hdr = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    assert hdr.n_chunks == -(-n // chunk) and hdr.total_len == n
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s) == n
    assert torch.equal(back, x)
    _, payloads = container.chunk_payloads(cont[:hdr.container_len].cpu().numpy())

    def check(i):
        return payloads[i] == pyoracle.encode(algo, host[i * chunk:(i + 1) * chunk])

    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        ok = list(ex.map(check, range(hdr.n_chunks)))
    assert all(ok), [i for i, v in enumerate(ok) if not v][:8]


@pytest.mark.parametrize("algo", ALGOS)
def test_repeated_pairs_inside_a_record(algo):
    """Quad pairs that repeat within one record: the second occurrence is predicted from a predictor entry written earlier in the
    SAME record, which the wave decoder's speculative predictor reads cannot see (scalar re-decode of the record), and chains of
    equal quads (one resolution round per link in both directions)."""
    rng = np.random.default_rng(5)
    words = rng.integers(0, 2**32, size=64, dtype=np.uint32)
    parts = []
    for rep in range(400):
        a, b, c = (int(x) for x in rng.integers(0, 64, size=3))
        kind = rep % 4
        if kind == 0:
            seq = [words[a], words[b], words[c], words[a], words[b], words[c]]            # repeated triple
        elif kind == 1:
            seq = [words[a]] * int(rng.integers(2, 40))                                   # run of one quad
        elif kind == 2:
            seq = [words[a], words[b]] * int(rng.integers(2, 12))                         # alternating pair
        else:
            seq = list(words[rng.integers(0, 64, size=int(rng.integers(1, 9)))])          # filler
        parts.append(np.array(seq, dtype=np.uint32))
    data = np.concatenate(parts).view(np.uint8)
    want = pyoracle.encode(algo, data)
    assert gpu_encode(algo, data) == want
    assert gpu_decode(algo, want, data.size) == data.tobytes()


def stage_stats():
    import ctypes
    from density_amd import _lib
    a = (ctypes.c_uint64 * 2)()
    _lib.lib().density_hip_stage_stats(a)
    return list(a)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("kind", ["prose", "rep", "zeros", "lowzero", "mixed", "random", "samehash", "binaryish"])
@pytest.mark.parametrize("chunk", [262144, 524288])
def test_exchange_passes_match_oracle(algo, kind, chunk, kernel_variant):
    """Containers at chunk sizes the exchange passes take (whole 4 KiB trips, four heads and more): text (the cold-dictionary head of
    every chunk in order, raw copies and all, the rest in passes), data whose records meet the blow-up protection later on (those
    chunks are handed back to the in-order kernel), zero quads against empty tables, one-slot pile-ups, a short ragged last chunk —
    every chunk stream == the oracle's, decode == input; and the passes keep every chunk of calm data."""
    n = 5 * chunk + 3 * 4096 + 1001
    data = datagen.by_kind(kind, n, seed=chunk + 7)
    cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
    if kernel_variant == "default":
        container.set_kernel_variant(64)                          # audit: count the chunks kept / handed back
    s0 = stage_stats()
    cn = container.encode(algo, data, cont, chunk)
    s1 = stage_stats()
    hdr, payloads = container.chunk_payloads(cont[:cn])
    assert hdr.n_chunks == 6
    copies = 0
    for i, p in enumerate(payloads):
        want, st = pyoracle.encode_stats(algo, data[i * chunk:(i + 1) * chunk])
        assert p == want, (kind, chunk, i)
        copies += st["copy_blocks"]
    if kernel_variant == "default":
        assert s1[0] - s0[0] == hdr.n_chunks
        back_to_in_order = s1[1] - s0[1]                          # (the short last chunk is finished by the head kernel: not "back")
        assert 0 <= back_to_in_order <= hdr.n_chunks
        if kind in ("prose", "rep", "zeros", "lowzero"):
            assert back_to_in_order == 0                          # raw copies (if any) only while the dictionary is cold: the head's
        if kind == "random":
            assert copies > 0 and back_to_in_order == 0           # never calm: the head kernel goes on to the end of the chunk
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == n
    assert np.array_equal(back, data)


@pytest.mark.parametrize("algo", ALGOS)
def test_exchange_passes_repeats_inside_a_block(algo, kernel_variant):
    """the same slot many times within one 64-quad exchange, predicted quads rewriting their prediction, chains through every level"""
    rng = np.random.default_rng(17)
    words = rng.integers(0, 2**32, size=9, dtype=np.uint32)
    words[0] = 0
    data = words[rng.integers(0, 9, size=3 * 65536)].view(np.uint8)
    chunk = 262144
    cont = np.zeros(container.container_bound(algo, data.size, chunk), dtype=np.uint8)
    if kernel_variant == "default":
        container.set_kernel_variant(64)
    s0 = stage_stats()
    cn = container.encode(algo, data, cont, chunk)
    s1 = stage_stats()
    _, payloads = container.chunk_payloads(cont[:cn])
    for i, p in enumerate(payloads):
        assert p == pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk]), i
    if kernel_variant == "default":
        assert (s1[0] - s0[0], s1[1] - s0[1]) == (len(payloads), 0)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("kind", ["prose", "rep", "mixed", "zeros"])
@pytest.mark.parametrize("ragged", [134 * 1024 + 1001, 192 * 1024 + 4096 - 1, 200 * 1024 + 128, 250 * 1024 + 2])
def test_exchange_passes_ragged_end(algo, kind, ragged, kernel_variant):
    """A last chunk that is not whole 4 KiB trips: the passes take its whole trips, write their tables back, and the in-order kernel goes
    on from there (blow-up protection counters advanced over the calm blocks in between) — no chunk is handed back for calm data."""
    chunk = 262144
    n = 2 * chunk + ragged
    data = datagen.by_kind(kind, n, seed=ragged)
    cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
    if kernel_variant == "default":
        container.set_kernel_variant(64)
    s0 = stage_stats()
    cn = container.encode(algo, data, cont, chunk)
    s1 = stage_stats()
    _, payloads = container.chunk_payloads(cont[:cn])
    for i, p in enumerate(payloads):
        assert p == pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk]), (kind, ragged, i)
    if kernel_variant == "default" and kind != "mixed":
        assert (s1[0] - s0[0], s1[1] - s0[1]) == (3, 0)
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == n
    assert np.array_equal(back, data)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("kind,n", [("prose", 3_000_003), ("rep", 2_500_000), ("mixed", 1_200_001)])
def test_long_stream_encode_in_exchange_passes(algo, kind, n, kernel_variant):
    """The reference symbols `cheetah_encode` / `lion_encode` on one long stream: the whole stream is one chunk for the exchange passes
    (in-order head, stages over the whole trips, the ragged end in order again) — byte for byte the reference's stream."""
    data = datagen.by_kind(kind, n, seed=n)
    want = pyoracle.encode(algo, data)
    if kernel_variant == "default":
        container.set_kernel_variant(64)
    s0 = stage_stats()
    got = gpu_encode(algo, data)
    s1 = stage_stats()
    assert len(got) == len(want) and got == want
    if kernel_variant == "default":
        assert s1[0] - s0[0] == 1                                  # one chunk through the passes ...
        if kind != "mixed":
            assert s1[1] - s0[1] == 0                              # ... and kept
    assert gpu_decode(algo, want, n) == data.tobytes()
