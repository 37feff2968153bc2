"""Pins the CPU oracle: the reference's three golden vectors (src/lib.rs:19-72), SURVEY.md Appendix C's
independently derived known answers, the committed KAT fixtures, and round trips on the edge sizes the block
codec cares about (codec/codec.rs:42-63,88-123)."""
import hashlib
import json
import os

import pytest

import datagen
from oracle import pyoracle

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
REF_INPUT = bytes.fromhex(KAT["reference_input_hex"])

# "encoded_len : first 16 hex of SHA-256(encoded) : raw-copied blocks", produced by the
# surveyor's scratch model (a third implementation, independent of oracle/ and oracle/pymodel.py).
APPENDIX_C = {
    "empty": {"chameleon": (0, "e3b0c44298fc1c14", 0), "cheetah": (0, "e3b0c44298fc1c14", 0), "lion": (0, "e3b0c44298fc1c14", 0)},
    "zeros1024": {"chameleon": (544, "228db5081a7fe348", 0), "cheetah": (64, "8667e718294e9e0d", 0), "lion": (96, "3bf0ac90e81f122b", 0)},
    "abcd300xyz": {"chameleon": (645, "42d3fedbd0191833", 0), "cheetah": (89, "04ad9c6a79b08d85", 0), "lion": (123, "2227abc5e736c893", 0)},
    "xs1_4099": {"chameleon": (4147, "4e47bbdf23d1007c", 11), "cheetah": (4171, "a8e0ed9d1282dda0", 24), "lion": (4189, "a1546dd4c44d4dc2", 50)},
    "words65536": {"chameleon": (35964, "17bf2ad1acbc8220", 0), "cheetah": (37156, "2d9ae37fc1f4f2e1", 0), "lion": (36280, "c4bf6cc6d7e4e0fd", 3)},
}


def _derived_inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_kat", os.path.join(HERE, "golden", "make_kat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.derived_inputs()


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
def test_reference_golden_vector(algo):
    """src/lib.rs:22-42 (chameleon), :44-64 (cheetah), :66-86 (lion): exact bytes, then decode == input.
    Like the reference test, the output buffer is only len(input) bytes (smaller than the safe size)."""
    want = bytes.fromhex(KAT["reference"][algo])
    got = pyoracle.encode(algo, REF_INPUT, cap=len(REF_INPUT))
    assert got == want
    assert pyoracle.decode(algo, got, len(REF_INPUT)) == REF_INPUT


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
def test_appendix_c_and_committed_kat(algo):
    inputs = _derived_inputs()
    for name, data in inputs.items():
        row = KAT["derived"][name]
        assert len(data) == row["len"] and hashlib.sha256(data).hexdigest() == row["sha256_input"], name
        enc, st = pyoracle.encode_stats(algo, data)
        assert (len(enc), hashlib.sha256(enc).hexdigest(), st["copy_blocks"]) == (row[algo]["len"], row[algo]["sha256"], row[algo]["copy_blocks"]), name
        if name in APPENDIX_C:
            ln, sha16, copied = APPENDIX_C[name][algo]
            assert (len(enc), hashlib.sha256(enc).hexdigest()[:16], st["copy_blocks"]) == (ln, sha16, copied), name
        assert pyoracle.decode(algo, enc, len(data)) == data, name


def test_sanity_identities():
    """1024 zeros -> 4x(8+64*2), 8x8 (all PREDICTED), 16x6; hash("test") = 0xfb70."""
    z = bytes(1024)
    assert len(pyoracle.encode("chameleon", z)) == 4 * (8 + 128)
    assert len(pyoracle.encode("cheetah", z)) == 8 * 8
    assert len(pyoracle.encode("lion", z)) == 16 * 6
    assert ((int.from_bytes(b"test", "little") * 0x9D6EF916) & 0xFFFFFFFF) >> 16 == 0xFB70


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
def test_safe_encode_buffer_size(algo):
    """codec/codec.rs:18-21"""
    B, S = pyoracle.BLOCK_BYTES[algo], pyoracle.SIG_BYTES[algo]
    for n in [0, 1, 3, 4, B - 1, B, B + 1, 10 * B, 10 * B + 5, 1 << 30]:
        assert pyoracle.safe_encode_buffer_size(algo, n) == n + (n // B) * S + (S if n % B else 0)


EDGE_SIZES = sorted(set(list(range(0, 41)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 263, 264, 265, 271, 272, 273,
                                              511, 512, 513, 519, 520, 521, 4093, 4094, 4095, 4096, 4097, 4098, 4099]))


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
@pytest.mark.parametrize("kind", ["prose", "random", "zeros", "mixed", "samehash"])
def test_round_trip_edge_sizes(algo, kind):
    big = {"prose": datagen.prose(5000, 21), "random": datagen.random_bytes(5000, 22), "zeros": bytes(5000),
           "mixed": datagen.mixed(5000, 23), "samehash": datagen.same_hash_quads(1250, 24)}[kind]
    big = bytes(big)
    for n in EDGE_SIZES:
        data = big[:n]
        enc = pyoracle.encode(algo, data)
        assert len(enc) <= pyoracle.safe_encode_buffer_size(algo, n)
        assert pyoracle.decode(algo, enc, n) == data, (kind, n)


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
def test_flag_and_copy_coverage(algo):
    """The reference's vectors never reach copy mode, MAP_B or PREDICTED_B..E; make sure the inputs
    used for differential/GPU parity do."""
    _, st = pyoracle.encode_stats(algo, datagen.mixed(400_000, 31))
    assert st["copy_blocks"] > 0 and st["coded_blocks"] > 0
    nflags = {"chameleon": 2, "cheetah": 4, "lion": 8}[algo]
    _, st2 = pyoracle.encode_stats(algo, datagen.prose(400_000, 32))
    merged = [a + b for a, b in zip(st["flags"], st2["flags"])]
    assert all(merged[f] > 0 for f in range(nflags)), merged


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
def test_decode_rejects_truncation_without_crashing(algo):
    """The reference panics on truncated input (read_buffer.rs:22); the oracle returns 0 or a short result, never reads OOB."""
    data = bytes(datagen.prose(3000, 41))
    enc = pyoracle.encode(algo, data)
    for cut in (1, 2, 3, 5, 7, len(enc) // 2):
        out = pyoracle.decode(algo, enc[:-cut], len(data))
        assert out != data
