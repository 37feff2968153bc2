"""GPU parity tests (run with -m gpu on an MI355X): the HIP Chameleon path, called through the C ABI, against the
CPU oracle on the same inputs — bit-exact (integer/byte work, no tolerance)."""
import hashlib
import json
import os

import numpy as np
import pytest

import datagen
from density_amd import Chameleon, DecodeError, EncodeError, container
from oracle import pyoracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
ALGO = "chameleon"


VARIANTS = {"rotor": 0, "rotor-noindex": 2, "pipelined": 4, "pipelined-noindex": 6, "simple": 1, "rotor-hostpipe": 256,
            "rotor-alt": 2048}   # 2048: the OTHER rotation encoder (kernels.hpp kRotorSplitDefault: the split one — 8 chain + 8 emit waves — or the 8-wave one)   # 256: the host-pointer container calls pipelined whatever the size, a slice per chunk


@pytest.fixture(autouse=True, params=list(VARIANTS))
def kernel_variant(request):
    """Every test runs against the default wave-rotation kernels (rotor.hip; containers with the block index = index-fed rotation
    decoder, without it = record-walking decoder), the 16-wave role pipelines (chameleon.hip, with and without the index) and the
    one-wavefront kernels: three independent implementations of the same stream semantics cross-checking each other."""
    container.set_kernel_variant(VARIANTS[request.param])
    yield request.param
    container.set_kernel_variant(0)


def expected_block_index(data, chunk):
    """What the block index must say, derived from the oracle's streams by walking their records with the reference FSM
    (codec/codec.rs:88-123, protection_state.rs)."""
    from oracle import pymodel
    out = bytearray()
    for c0 in range(0, len(data), chunk):
        part = bytes(data[c0:c0 + chunk])
        enc = pyoracle.encode(ALGO, part)
        g, pos = pymodel.Guard(), 0
        for b0 in range(0, len(part), 256):
            blen = min(256, len(part) - b0)
            ragged = blen < 256
            if g.next_is_copy():
                out.append(0x80 | (0x7F if ragged else 0))
                pos += blen
                g.decay()
            else:
                sig = int.from_bytes(enc[pos:pos + 8], "little")
                hits = bin(sig).count("1")
                out.append(0x7F if ragged else hits)
                reclen = 8 + 4 * (blen // 4) - 2 * hits + blen % 4
                g.update(reclen >= 256)
                pos += reclen
        assert pos == len(enc)
    return bytes(out)


def gpu_encode(data):
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    out = np.zeros(max(Chameleon.safe_encode_buffer_size(data.size), 1), dtype=np.uint8)
    n = Chameleon.encode(data, out)
    return out[:n].tobytes()


def gpu_decode(enc, n):
    enc = np.frombuffer(bytes(enc), dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    m = Chameleon.decode(enc, out)
    return out[:m].tobytes()


def test_reference_golden_vector_on_gpu():
    """src/lib.rs:22-42: exact 73 bytes, output buffer of len(input) bytes like the reference test, then decode."""
    data = bytes.fromhex(KAT["reference_input_hex"])
    out = bytearray(len(data))
    n = Chameleon.encode(data, out)
    assert bytes(out[:n]) == bytes.fromhex(KAT["reference"][ALGO])
    back = bytearray(len(data))
    m = Chameleon.decode(bytes(out[:n]), back)
    assert bytes(back[:m]) == data


def test_committed_kat_fixtures_on_gpu():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_kat", os.path.join(HERE, "golden", "make_kat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, data in mod.derived_inputs().items():
        row = KAT["derived"][name][ALGO]
        enc = gpu_encode(data) if data else b""
        assert (len(enc), hashlib.sha256(enc).hexdigest()) == (row["len"], row["sha256"]), name
        assert gpu_decode(enc, len(data)) == data if data else True


EDGE_SIZES = sorted(set(list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 251, 252, 253, 254, 255, 256, 257, 258, 259, 260, 263, 264, 265,
                                              271, 272, 273, 511, 512, 513, 519, 520, 521, 767, 768, 769, 4093, 4094, 4095, 4096, 4097, 4098, 4099]))


@pytest.mark.parametrize("kind", ["prose", "random", "zeros", "mixed", "samehash", "lowzero", "saltzero"])
def test_stream_parity_edge_sizes(kind):
    big = {"prose": datagen.prose(5000, 21), "random": datagen.random_bytes(5000, 22), "zeros": np.zeros(5000, np.uint8),
           "mixed": datagen.mixed(5000, 23), "samehash": datagen.same_hash_quads(1250, 24),
           "lowzero": datagen.low_zero_quads(1250, 25), "saltzero": datagen.salted_zero_quads(1250, 26)}[kind]
    for n in EDGE_SIZES:
        data = big[:n].copy()
        want = pyoracle.encode(ALGO, data)
        got = gpu_encode(data)
        assert got == want, (kind, n)
        assert gpu_decode(want, n) == data.tobytes(), (kind, n)


@pytest.mark.parametrize("kind,n", [("prose", 1_000_003), ("random", 300_001), ("zeros", 262_144 + 2), ("mixed", 2_000_000),
                                    ("samehash", 400_000), ("lowzero", 500_002), ("saltzero", 600_002), ("binaryish", 700_001), ("rep", 1_500_000)])
def test_stream_parity_large(kind, n):
    """Whole-stream (single chunk) parity at sizes the oracle finishes in well under a second: exercises multi-block
    dictionary carry, copy mode entering/leaving (random, mixed), the all-one-slot hazard and the zero-entry map."""
    data = datagen.by_kind(kind, n, seed=77)
    want, st = pyoracle.encode_stats(ALGO, data)
    got = gpu_encode(data)
    assert len(got) == len(want)
    assert got == want
    if kind in ("random", "mixed"):
        assert st["copy_blocks"] > 0
    assert gpu_decode(want, n) == data.tobytes()


@pytest.mark.parametrize("chunk", [256, 512, 4096, 65536, 1 << 20])
@pytest.mark.parametrize("kind", ["prose", "mixed", "saltzero"])
def test_container_chunks_match_oracle(kind, chunk, kernel_variant):
    n = 3 * (1 << 20) + 12345 if chunk >= 65536 else 40 * chunk + 77
    data = datagen.by_kind(kind, n, seed=chunk)
    cont = np.zeros(container.container_bound(ALGO, n, chunk), dtype=np.uint8)
    cn = container.encode(ALGO, data, cont, chunk)
    hdr, payloads = container.chunk_payloads(cont[:cn])
    assert (hdr.total_len, hdr.chunk_size, hdr.n_chunks, hdr.container_len) == (n, chunk, -(-n // chunk), cn)
    for i, p in enumerate(payloads):
        assert p == pyoracle.encode(ALGO, data[i * chunk:(i + 1) * chunk]), (kind, chunk, i)
    idx = container.block_index(cont[:cn])
    if kernel_variant.endswith("-noindex"):
        assert idx is None and hdr.flags == 0
    elif chunk <= 65536:                                      # the python record walk is slow; small-chunk cases cover every marker
        assert idx == expected_block_index(data, chunk)
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == n
    assert np.array_equal(back, data)


def test_gpu_decodes_cpu_built_container():
    """A container assembled on the CPU from oracle streams (what a CPU producer would write) decodes on the GPU."""
    chunk, n = 8192, 100_000
    data = datagen.mixed(n, seed=5)
    streams = [pyoracle.encode(ALGO, data[i:i + chunk]) for i in range(0, n, chunk)]
    nc = len(streams)
    body = bytearray()
    base = (32 + 4 * nc + 15) // 16 * 16
    for s in streams:
        body += s
        body += bytes(-len(body) % 16)
    body = body[:len(body) - (-len(streams[-1]) % 16)] if streams else body
    import struct
    total = base + len(body)
    head = struct.pack("<IBBHIIQQ", 0x31434844, 0, 1, 0, chunk, nc, n, total)
    table = b"".join(struct.pack("<I", len(s)) for s in streams)
    raw = head + table + bytes(base - 32 - 4 * nc) + bytes(body)
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(raw, back) == n
    assert np.array_equal(back, data)


def test_errors_are_reported_not_crashes():
    data = datagen.prose(3000, 41)
    enc = pyoracle.encode(ALGO, data)
    out = np.zeros(3000, dtype=np.uint8)
    # truncated stream whose last record ends on a MAP item with < 2 bytes -> reference panics, we raise
    bad = 0
    for cut in (1, 2, 3, 5, 7, 9, 100):
        try:
            m = Chameleon.decode(enc[:-cut], out)
            assert out[:m].tobytes() != data.tobytes()
        except DecodeError:
            bad += 1
    assert bad >= 1
    with pytest.raises(DecodeError):
        Chameleon.decode(enc, np.zeros(100, dtype=np.uint8))          # output too small
    with pytest.raises(EncodeError):
        Chameleon.encode(datagen.random_bytes(3000, 1), np.zeros(100, dtype=np.uint8))
    # corrupted container header
    cont = np.zeros(container.container_bound(ALGO, 3000, 1024), dtype=np.uint8)
    cn = container.encode(ALGO, data, cont, 1024)
    broken = cont[:cn].copy()
    broken[0] ^= 0xFF
    with pytest.raises(DecodeError):
        container.decode(broken, out)
    short = cont[:cn - 40].copy()
    with pytest.raises(DecodeError):
        container.decode(short, out)


def test_selftest_bits_all_clear():
    """Every LDS ordering assumption holds on this device, so the default (wave-rotation) kernels are what the other tests ran."""
    from density_amd import _lib
    assert _lib.lib().density_hip_selftest_bits() == 0


def test_corrupt_block_index_is_a_format_error(kernel_variant):
    """The index-fed decoder cross-checks every index entry it uses against the signature it describes: a container whose index
    lies is DENSITY_HIP_ERR_FORMAT (DecodeError), never silently wrong bytes."""
    if kernel_variant != "rotor":
        pytest.skip("index validation is a property of the default decoder")
    n, chunk = 600_000, 65536
    data = datagen.prose(n, seed=9)
    cont = np.zeros(container.container_bound(ALGO, n, chunk), dtype=np.uint8)
    cn = container.encode(ALGO, data, cont, chunk)
    good = cont[:cn].copy()
    hdr = container.parse_header(good)
    base = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
    out = np.zeros(n, dtype=np.uint8)
    assert container.decode(good, out) == n and np.array_equal(out, data)
    rng = np.random.default_rng(3)
    for trial in range(12):
        bad = good.copy()
        b = int(rng.integers(0, n // 256))
        if bad[base + b] & 0x80 or (bad[base + b] & 0x7F) == 0x7F:
            continue
        bad[base + b] = (int(bad[base + b]) + int(rng.integers(1, 5))) % 65
        with pytest.raises(DecodeError):
            container.decode(bad, out)
    # the raw-copy bits are checked against the blow-up protection (rotor.hip::index_fsm_consistent): a coded block flagged raw, a raw block
    # flagged coded, on calm text and on data with raw copies in it — wrong bytes of the right length would otherwise come back as OK
    for kind, seed in (("prose", 9), ("mixed", 19), ("random", 29)):
        data = datagen.by_kind(kind, n, seed=seed)
        cn = container.encode(ALGO, data, cont, chunk)
        good = cont[:cn].copy()
        assert container.decode(good, out) == n and np.array_equal(out, data), kind
        ix = good[base:base + (n + 255) // 256]
        raws, coded = np.flatnonzero(ix & 0x80), np.flatnonzero((ix & 0x80) == 0)
        assert raws.size > 0 or kind == "prose"                  # (a chunk's cold start can give even text a raw block or two)
        picks = [("to-raw", int(b)) for b in rng.choice(coded, size=6)] + [("to-coded", int(b)) for b in (rng.choice(raws, size=6) if raws.size else [])]
        if raws.size:
            picks += [("to-raw", int(raws[-1]) + 1), ("to-coded", int(raws[0]))]        # a run one block longer / one block shorter
        for how, b in picks:
            if b >= ix.size or (good[base + b] & 0x7F) == 0x7F:
                continue
            bad = good.copy()
            bad[base + b] = 0x80 if how == "to-raw" else 0
            if bad[base + b] == good[base + b]:
                continue
            with pytest.raises(DecodeError):
                container.decode(bad, out)


def test_corrupt_size_table_is_a_format_error():
    """A valid header with a size-table entry that runs past the container must not be followed by the kernels."""
    n, chunk = 300_000, 65536
    data = datagen.prose(n, seed=10)
    cont = np.zeros(container.container_bound(ALGO, n, chunk), dtype=np.uint8)
    cn = container.encode(ALGO, data, cont, chunk)
    out = np.zeros(n, dtype=np.uint8)
    for entry, value in ((0, 0xFFFFFFFF), (2, 0x7FFFFFF0), (4, cn)):
        bad = cont[:cn].copy()
        bad[32 + 4 * entry:36 + 4 * entry] = np.frombuffer(int(value).to_bytes(4, "little"), dtype=np.uint8)
        with pytest.raises(DecodeError):
            container.decode(bad, out)


def test_corrupt_streams_under_a_true_index(kernel_variant):
    """The payload is corrupted, the block index is not.  A signature whose MAP count no longer is what the index says must be a format error
    whichever record of a round it belongs to (the index-fed decoders take record lengths from the index: without the check a zeroed
    signature decodes to wrong bytes of the right length); every corruption the decoder accepts must decode like the oracle decodes the
    corrupted chunk streams — items changed, MAP flags moved: quads that were never written, zero entries, garbage slots."""
    if kernel_variant not in ("rotor", "pipelined"):
        pytest.skip("the index-fed decoders")
    n, chunk = 3 * 262144 + 999, 262144
    for kind in ("lowzero", "prose", "zeros"):
        data = datagen.by_kind(kind, n, seed=21)
        cont = np.zeros(container.container_bound(ALGO, n, chunk), dtype=np.uint8)
        cn = container.encode(ALGO, data, cont, chunk)
        good = cont[:cn].copy()
        hdr, payloads = container.chunk_payloads(good)
        assert hdr.flags & 1
        off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
        ix = good[off:off + (n + 255) // 256].copy()
        off = (off + (hdr.total_len + 255) // 256 + 15) // 16 * 16
        offs = []
        for p in payloads:
            offs.append(off)
            off = (off + len(p) + 15) // 16 * 16
        out = np.zeros(n, dtype=np.uint8)
        # record starts of chunk 0 from its index (coded: 8 + 256 - 2 * MAP count, raw: 256)
        starts, pos = [], 0
        for e in ix[:chunk // 256]:
            starts.append(pos)
            pos += 256 if e & 0x80 else 8 + 256 - 2 * int(e & 0x7F)
        coded = [k for k in range(len(starts)) if not ix[k] & 0x80 and (ix[k] & 0x7F) not in (0, 0x7F)]
        for k in [c for c in coded if c % 24 in (1, 4, 7, 10, 11, 12, 13, 23)][:16] + coded[-3:]:      # every place in a round of 8 or 12, the last rounds
            bad = good.copy()
            bad[offs[0] + starts[k]:offs[0] + starts[k] + 8] = 0                                          # MAP count 0 != the index's
            with pytest.raises(DecodeError):
                container.decode(bad, out)
        rng = np.random.default_rng(77)
        accepted = 0
        for trial in range(40):
            bad = good.copy()
            k = int(rng.integers(0, len(payloads)))
            at = offs[k] + int(rng.integers(0, len(payloads[k])))
            if trial % 2:
                bad[at] ^= int(rng.integers(1, 256))
            else:
                bad[at:at + 4] = rng.integers(0, 256, size=min(4, len(bad) - at), dtype=np.uint8)
            try:
                m = container.decode(bad, out)
            except DecodeError:
                continue
            accepted += 1
            want = b"".join(pyoracle.decode(ALGO, bytes(bad[offs[i]:offs[i] + len(payloads[i])]), min(chunk, n - i * chunk)) for i in range(len(payloads)))
            assert out[:m].tobytes() == want, (kind, trial, k, at - offs[k])
        assert accepted >= 20, (kind, accepted)


def test_a_signature_with_more_map_flags_in_the_last_partial_round(kernel_variant):
    """Round 6 (found by tools/gpu_fuzz_tail.py): in the decoder's in-order epilogue — the records of a chunk's last, partial round — record lengths came
    from the index alone; a corrupted signature with MORE MAP flags makes its record 8 or more bytes shorter, and the bytes left over passed for a signature
    with no items behind it, so the container was accepted where the reference reads its next record from the wrong place.  The signature's MAP count is now
    held against the index there too.  Whatever is accepted must decode like the oracle decodes the corrupted chunk stream."""
    if kernel_variant not in ("rotor", "pipelined"):
        pytest.skip("the index-fed decoders")
    for kind in ("prose", "mixed"):
        for tail in (263, 999, 1500):
            n, chunk = 3 * 262144 + tail, 262144
            data = datagen.by_kind(kind, n, seed=21)
            cont = np.zeros(container.container_bound(ALGO, n, chunk), dtype=np.uint8)
            cn = container.encode(ALGO, data, cont, chunk)
            good = cont[:cn].copy()
            hdr, payloads = container.chunk_payloads(good)
            off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
            off = (off + (hdr.total_len + 255) // 256 + 15) // 16 * 16
            offs = []
            for p in payloads:
                offs.append(off)
                off = (off + len(p) + 15) // 16 * 16
            last = len(payloads) - 1
            out = np.zeros(n, dtype=np.uint8)
            sig = int.from_bytes(good[offs[last]:offs[last] + 8].tobytes(), "little")
            clear = [b for b in range(64) if not (sig >> b) & 1]
            assert len(clear) >= 8, (kind, tail)
            for extra in (4, 5, 8):                                                     # 4 more MAP flags: the record is 8 bytes shorter — exactly a signature's worth left over
                bad = good.copy()
                sig2 = sig
                for b in clear[:extra]:
                    sig2 |= 1 << b
                bad[offs[last]:offs[last] + 8] = np.frombuffer(sig2.to_bytes(8, "little"), dtype=np.uint8)
                want = pyoracle.decode(ALGO, bytes(bad[offs[last]:offs[last] + len(payloads[last])]), tail)
                try:
                    m = container.decode(bad, out)
                except DecodeError:
                    continue                                                            # (the reference panics or returns other bytes: refusing is right)
                assert out[:m].tobytes() == data[:3 * chunk].tobytes() + want, (kind, tail, extra)


def test_pipelined_host_calls_make_the_same_container(kernel_variant):
    """density_hip_encode / _decode through host pointers: inputs of 8 MiB and more go up, through the kernels and down in slices on separate
    streams (api.hip: *_container_pipelined).  The container must be byte for byte what the staged path (kernel variant 512) writes — header,
    size table, block index, payloads, zeroed gaps — and both decoders must return the input from either."""
    if kernel_variant != "rotor":
        pytest.skip("one configuration is enough: the paths differ on the host side only")
    for n, chunk in ((40 * (1 << 20) + 12345, 1 << 20), (9 * (1 << 20), 262144), (33 * (1 << 20) + 1, 4 << 20)):
        data = datagen.by_kind("mixed" if chunk < (4 << 20) else "rep", n, seed=31)
        made = {}
        for variant in (0, 512):
            container.set_kernel_variant(variant)
            cont = np.full(container.container_bound(ALGO, n, chunk), 0xA5, dtype=np.uint8)
            cn = container.encode(ALGO, data, cont, chunk)
            made[variant] = cont[:cn].copy()
        assert made[0].tobytes() == made[512].tobytes(), (n, chunk)
        for variant in (0, 512):
            container.set_kernel_variant(variant)
            back = np.zeros(n, dtype=np.uint8)
            assert container.decode(made[0], back) == n and np.array_equal(back, data), (n, chunk, variant)
        # errors come back as errors from the pipelined path too: a corrupt size table, an output buffer too small
        container.set_kernel_variant(0)
        bad = made[0].copy()
        bad[32:36] = np.frombuffer((0x7FFFFFF0).to_bytes(4, "little"), dtype=np.uint8)
        with pytest.raises(DecodeError):
            container.decode(bad, np.zeros(n, dtype=np.uint8))
        with pytest.raises(EncodeError):
            container.encode(ALGO, datagen.random_bytes(n, 5), np.zeros(n // 2, dtype=np.uint8), chunk)
    container.set_kernel_variant(0)


def test_abort_and_recovery_paths(kernel_variant):
    """Inputs that flip between compressible and incompressible regions every few KiB: the rotation encoder's speculation
    ("no raw-copy block in this round") fails again and again, so roll-back, slow mode and the way back to fast mode all run."""
    rng = np.random.default_rng(17)
    text = datagen.prose(1 << 20, seed=31)
    parts, pos = [], 0
    while pos < (3 << 20):
        k = int(rng.integers(1, 40)) * 256 + int(rng.integers(0, 2)) * int(rng.integers(0, 256))
        if rng.integers(0, 2):
            parts.append(text[(pos % (1 << 19)):(pos % (1 << 19)) + k])
        else:
            parts.append(rng.integers(0, 256, size=k, dtype=np.uint8))
        pos += k
    data = np.concatenate(parts)
    want, st = pyoracle.encode_stats(ALGO, data)
    assert st["copy_blocks"] > 100
    got = gpu_encode(data)
    assert got == want
    assert gpu_decode(want, data.size) == data.tobytes()
    chunk = 1 << 19
    cont = np.zeros(container.container_bound(ALGO, data.size, chunk), dtype=np.uint8)
    cn = container.encode(ALGO, data, cont, chunk)
    _, payloads = container.chunk_payloads(cont[:cn])
    for i, p in enumerate(payloads):
        assert p == pyoracle.encode(ALGO, data[i * chunk:(i + 1) * chunk]), i
    back = np.zeros(data.size, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == data.size and np.array_equal(back, data)


def test_long_incompressible_stretches_and_their_ends(kernel_variant):
    """Stretches of incompressible data LONG enough for the encoder's ordered rounds to run ahead of their commit (rotor.hip: three rounds of one
    unbroken stretch, 12 KiB, start it), ended by data that compresses — where the round that ran ahead finds its assumption wrong, raises the abort and
    the waves behind it take back exactly the blocks that exchanged — by zeros, by low-entropy noise, by the chunk's end; at chunk sizes from one round of
    run-ahead to the headline's 4 MiB, as container (every chunk stream == the oracle's) and as one stream."""
    rng = np.random.default_rng(23)
    text = datagen.prose(1 << 20, seed=41)
    pieces = []
    for k, ln in enumerate([300_000, 70_000, 49_152, 4096, 1_000_000, 256, 131_072, 20_000, 700_000, 12_288 + 256, 65_536]):
        kind = k % 4
        if kind in (0, 2):
            pieces.append(rng.integers(0, 256, size=ln + int(rng.integers(0, 300)), dtype=np.uint8))              # incompressible
        elif kind == 1:
            pieces.append(text[k * 4099:k * 4099 + ln])                                                            # compresses
        else:
            pieces.append(np.zeros(ln, dtype=np.uint8) if k % 8 == 3 else rng.integers(0, 4, size=ln, dtype=np.uint8))
    data = np.concatenate(pieces)
    want, st = pyoracle.encode_stats(ALGO, data)
    assert st["copy_blocks"] > 3000
    assert gpu_encode(data) == want
    assert gpu_decode(want, data.size) == data.tobytes()
    for chunk in (1 << 16, 1 << 18, 1 << 20, 4 << 20):
        cont = np.zeros(container.container_bound(ALGO, data.size, chunk), dtype=np.uint8)
        cn = container.encode(ALGO, data, cont, chunk)
        _, payloads = container.chunk_payloads(cont[:cn])
        for i, p in enumerate(payloads):
            assert p == pyoracle.encode(ALGO, data[i * chunk:(i + 1) * chunk]), (chunk, i)
        back = np.zeros(data.size, dtype=np.uint8)
        assert container.decode(cont[:cn], back) == data.size and np.array_equal(back, data), chunk


def test_full_size_properties_device_resident():
    """256 MiB device-resident container round trip (BASELINE config-2 shape at a quarter of the size): decode(encode(x))
    == x bit for bit, header arithmetic, and a sample of chunk streams equal to the oracle."""
    import torch
    n, chunk = 1 << 28, 1 << 20
    host = datagen.rep_text(n)
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound(ALGO, n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hdr = container.encode_device(ALGO, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    assert hdr.n_chunks == n // chunk and hdr.total_len == n
    got = container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s)
    assert got == n
    assert torch.equal(back, x)
    raw = cont[:hdr.container_len].cpu().numpy()
    _, payloads = container.chunk_payloads(raw)
    for i in (0, 1, 117, hdr.n_chunks - 1):
        assert payloads[i] == pyoracle.encode(ALGO, host[i * chunk:(i + 1) * chunk]), i


@pytest.mark.parametrize("kind,n", [("rep", 48 * 1024 * 1024 + 12345), ("prose", 20 * 1024 * 1024), ("patchy", 40 * 1024 * 1024 + 7), ("random", 17 * 1024 * 1024),
                                    ("zeros", 16 * 1024 * 1024 + 4), ("saltzero", 24 * 1024 * 1024 + 258), ("samehash", 16 * 1024 * 1024), ("zeropatch", 32 * 1024 * 1024),
                                    ("prose", 6 * 1024 * 1024 + 300), ("rep", 5 * 1024 * 1024), ("hash0patch", 24 * 1024 * 1024)])
def test_long_stream_encode_in_segments_is_the_reference_stream(kind, n, kernel_variant):
    """`chameleon_encode` of ONE long stream runs in parallel segments (api.hip::run_stream_encode_segmented) and must still be the
    reference's single stream, byte for byte: calm text (one pass), text with incompressible patches (raw-copy blocks break the
    speculation of the segments behind them: several passes, then the sequential remainder), random bytes (raw copies throughout).
    `chameleon_decode` of the same stream runs in parallel segments as well when the stream is calm (stream_parse.hip), on one
    work-group otherwise; density_hip_stream_stats says which."""
    if kernel_variant != "rotor":
        pytest.skip("the segmented stream encode belongs to the default kernels")
    if kind == "patchy":
        data = datagen.by_kind("prose", n, seed=11).copy()
        rng = np.random.default_rng(12)
        for start in (3 << 20, (9 << 20) + 512, 21 << 20, (33 << 20) + 77 * 256):
            data[start:start + (96 << 10)] = rng.integers(0, 256, size=96 << 10, dtype=np.uint8)
    elif kind == "zeropatch":
        # zero quads over slot 0 and genuine zero-entry quads, overwritten and re-written across segment borders
        data = datagen.by_kind("prose", n, seed=21).copy()
        salted = datagen.by_kind("saltzero", 1 << 20, seed=22)
        for i, start in enumerate(range(1 << 20, n - (2 << 20), 3 << 20)):
            if i % 2 == 0:
                data[start:start + (64 << 10)] = 0
            else:
                data[start:start + (256 << 10)] = salted[:256 << 10]
    elif kind == "hash0patch":
        # non-zero quads whose hash is 0 (slot 0 then holds them), zero quads written over them (PLAIN: the LAST WRITER of slot 0 in that
        # segment is the zero quad, whose entry is 0), zero quads read back in later segments (MAP of slot 0 must give 0, not the old quad)
        data = datagen.by_kind("prose", n, seed=13).copy()
        half = 0x9D6EF916 >> 1
        inv = pow(half, -1, 1 << 31)
        quads = []
        for P in (2, 6, 1000, 65534):
            for top in (0, 1):
                q = (((P >> 1) * inv) % (1 << 31)) | (top << 31)
                assert q != 0 and ((q * 0x9D6EF916) & 0xFFFFFFFF) >> 16 == 0
                quads.append(q)
        for i, start in enumerate(range(1 << 20, n - (4 << 20), 5 << 19)):
            q = quads[i % len(quads)]
            data[start:start + 4] = np.frombuffer(int(q).to_bytes(4, "little"), dtype=np.uint8)           # slot 0 := q
            data[start + (1 << 19):start + (1 << 19) + 64] = 0                                            # zero quad PLAIN over it, then MAPs
            data[start + (3 << 19):start + (3 << 19) + 32] = 0                                            # MAPs of slot 0, segments later
    else:
        data = datagen.by_kind(kind, n, seed=13)
    import ctypes
    from density_amd import _lib
    def stats():
        a = (ctypes.c_uint64 * 4)()
        _lib.lib().density_hip_stream_stats(a)
        return list(a)
    want, st = pyoracle.encode_stats("chameleon", data)
    s0 = stats()
    got = gpu_encode(data)
    s1 = stats()
    assert len(got) == len(want)
    assert got == want
    assert s1[0] == s0[0] + 1                                     # encoded in segments ...
    calm = st["copy_blocks"] == 0                                 # no raw-copy block anywhere in the reference's stream
    assert calm == (kind in ("rep", "prose", "zeros", "saltzero", "zeropatch", "hash0patch"))
    assert (s1[1] - s0[1] == 1) == calm, (kind, s1[1] - s0[1])    # ... in one pass iff nothing breaks the speculation
    assert gpu_decode(want, n) == data.tobytes()
    s2 = stats()
    # decoded in parallel unless raw copies run on and on: a few incompressible patches are walked with the real FSM, the parallel
    # parse resumes behind each of them (stream_parse.hip); random bytes and one-slot collisions never calm down
    assert len(want) >= 2 << 20                                  # (the stream is long enough for the parallel decode to be tried)
    parallel = kind not in ("random", "samehash")
    assert (s2[2] - s1[2], s2[3] - s1[3]) == ((1, 0) if parallel else (0, 1)), (kind, s2, s1)


def test_long_stream_decode_errors_match_the_sequential_path(kernel_variant):
    """A long calm stream that is truncated, corrupted behind its head, or given too small an output must end like the same
    call on the one-work-group path: the same bytes back, or DecodeError — never a crash or silence."""
    if kernel_variant != "rotor":
        pytest.skip("the segmented stream decode belongs to the default kernels")
    n = 24 * 1024 * 1024 + 1000
    data = datagen.by_kind("rep", n, seed=17)
    enc = np.frombuffer(pyoracle.encode("chameleon", data), dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint8)

    def both(stream, buf):
        res = []
        for variant in (0, 4):                                                    # 4: the role pipelines -> the sequential stream path
            container.set_kernel_variant(variant)
            try:
                m = Chameleon.decode(stream, buf)
                res.append(("ok", m, buf[:m].tobytes()))
            except DecodeError:
                res.append(("error",))
        container.set_kernel_variant(0)
        return res

    a, b = both(enc, out)
    assert a == b and a[0] == "ok" and a[2] == data.tobytes()
    for cut in (1, 3, 137, 100_000):                                              # truncated
        a, b = both(enc[:-cut].copy(), out)
        assert a[0] == b[0], (cut, a[0], b[0])
        if a[0] == "ok":
            assert a[1:] == b[1:], cut
    a, b = both(enc, np.zeros(n - 5000, dtype=np.uint8))                          # output too small
    assert a[0] == b[0] == "error"
    broken = enc.copy()                                                           # a signature byte flipped deep inside: the record chain derails
    broken[9_000_001] ^= 0x5A
    a, b = both(broken, out)
    assert a[0] == b[0], (a[0], b[0])
    if a[0] == "ok":
        assert a[1:] == b[1:]


def _stream_stats():
    import ctypes
    from density_amd import _lib
    a = (ctypes.c_uint64 * 4)()
    _lib.lib().density_hip_stream_stats(a)
    return list(a)


def test_config2_full_size_every_chunk_is_the_oracle_stream(kernel_variant):
    """BASELINE config 2 at its real size — 1 GiB of rep-text (period 1,000,003), 4 MiB chunks, device-resident like the bench:
    ALL 256 chunk payloads equal the oracle's stream of that chunk byte for byte (one oracle encode per host thread), the block
    index describes them, and decode(container) == input."""
    if kernel_variant != "rotor":
        pytest.skip("full-size config runs on the default kernels")
    import torch
    from concurrent.futures import ThreadPoolExecutor
    n, chunk = 1 << 30, 4 << 20
    host = datagen.rep_text(n)
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound(ALGO, n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hdr = container.encode_device(ALGO, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    assert (hdr.n_chunks, hdr.total_len, hdr.chunk_size) == (256, n, chunk)
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s) == n
    assert torch.equal(back, x)
    raw = cont[:hdr.container_len].cpu().numpy()
    _, payloads = container.chunk_payloads(raw)
    idx = container.block_index(raw)

    def check(i):
        want = pyoracle.encode(ALGO, host[i * chunk:(i + 1) * chunk])
        if payloads[i] != want:
            return False
        # the index bytes of the chunk's first records: MAP count == popcount of the signature found there
        pos = 0
        for b in range(64):
            e = idx[i * (chunk // 256) + b]
            if e & 0x80:
                pos += 256
                continue
            if bin(int.from_bytes(want[pos:pos + 8], "little")).count("1") != e:
                return False
            pos += 8 + 256 - 2 * e
        return True

    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        ok = list(ex.map(check, range(hdr.n_chunks)))
    assert all(ok), [i for i, v in enumerate(ok) if not v][:8]


def test_config2_full_size_strict_stream_is_the_reference_stream(kernel_variant):
    """`chameleon_encode` of the whole 1 GiB as ONE reference stream (device-resident entry point of the same code path) equals
    `oracle_encode(1 GiB)` byte for byte, `chameleon_decode` of it gives the input back, and density_hip_stream_stats says the parallel
    segment paths served both (one encode pass: every speculation held)."""
    if kernel_variant != "rotor":
        pytest.skip("full-size config runs on the default kernels")
    import torch
    n = 1 << 30
    host = datagen.rep_text(n)
    want = np.frombuffer(pyoracle.encode(ALGO, host), dtype=np.uint8)
    x = torch.from_numpy(host).cuda()
    enc = torch.zeros(Chameleon.safe_encode_buffer_size(n), dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s0 = _stream_stats()
    m = container.stream_encode_device(ALGO, x.data_ptr(), n, enc.data_ptr(), enc.numel())
    s1 = _stream_stats()
    assert m == want.size
    assert torch.equal(enc[:m].cpu(), torch.from_numpy(want.copy()))
    assert (s1[0] - s0[0], s1[1] - s0[1]) == (1, 1)
    k = container.stream_decode_device(ALGO, enc.data_ptr(), m, back.data_ptr(), n)
    s2 = _stream_stats()
    assert k == n and torch.equal(back, x)
    assert (s2[2] - s1[2], s2[3] - s1[3]) == (1, 0)
    # ... and the reference's own symbols (host pointers: chameleon.rs:70-78) on the same gigabyte
    out = np.zeros(Chameleon.safe_encode_buffer_size(n), dtype=np.uint8)
    m2 = Chameleon.encode(host, out)
    assert m2 == want.size and np.array_equal(out[:m2], want)
    hback = np.zeros(n, dtype=np.uint8)
    assert Chameleon.decode(out[:m2], hback) == n and np.array_equal(hback, host)
    s3 = _stream_stats()
    assert (s3[0] - s2[0], s3[2] - s2[2]) == (1, 1)


def test_strict_stream_beyond_2_gib(kernel_variant):
    """The reference takes any usize (codec/codec.rs:72-80): ONE stream of 3.5 GiB — the input above 2^31 bytes, and its stream too — is
    still encoded and decoded in parallel segments, byte for byte the oracle's stream."""
    if kernel_variant != "rotor":
        pytest.skip("long streams belong to the default kernels")
    import torch
    n = (7 << 29) + 777
    host = datagen.rep_text(n, period=1_000_003, seed=5)
    cap = pyoracle.safe_encode_buffer_size(ALGO, n)
    want = np.empty(cap, dtype=np.uint8)
    m_want = pyoracle.encode_into(ALGO, host.ctypes.data, n, want.ctypes.data, cap)
    assert m_want > (1 << 31)
    x = torch.from_numpy(host).cuda()
    enc = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    s0 = _stream_stats()
    m = container.stream_encode_device(ALGO, x.data_ptr(), n, enc.data_ptr(), cap)
    s1 = _stream_stats()
    assert m == m_want and s1[0] == s0[0] + 1
    w = torch.from_numpy(want[:m_want]).cuda()
    assert torch.equal(enc[:m], w)
    del w
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    k = container.stream_decode_device(ALGO, enc.data_ptr(), m, back.data_ptr(), n)
    s2 = _stream_stats()
    assert k == n and torch.equal(back, x)
    assert (s2[2] - s1[2], s2[3] - s1[3]) == (1, 0)
