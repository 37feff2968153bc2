"""Cheetah / Lion encoding as passes of blind exchanges (tools/exchange_stage_model.py — the formulation of
density_amd/csrc/exchange_stages.hip) against the oracle: stage after stage over the whole stream must give the reference's stream
byte for byte wherever the reference takes no raw-copy block, and the record sizes must say where it would.  CPU only."""
import importlib.util
import os

import numpy as np
import pytest

import datagen
from oracle import pyoracle

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("exchange_stage_model", os.path.join(os.path.dirname(HERE), "tools", "exchange_stage_model.py"))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)

BLOCK = {"cheetah": 128, "lion": 64}


@pytest.mark.parametrize("algo", ["cheetah", "lion"])
@pytest.mark.parametrize("kind,n", [("prose", 48 * 1024), ("rep", 64 * 1024), ("zeros", 8 * 1024), ("lowzero", 16 * 1024), ("mixed", 32 * 1024)])
def test_stages_reproduce_the_reference_stream(algo, kind, n):
    data = datagen.by_kind(kind, n, seed=5)
    want, st = pyoracle.encode_stats(algo, data)
    got, sizes = model.encode(algo, bytes(data))
    pairs = [i for i in range(len(sizes) - 1) if sizes[i] >= BLOCK[algo] and sizes[i + 1] >= BLOCK[algo]]
    if st["copy_blocks"] == 0:
        assert got == want
    else:                                                             # the stages agree up to the pair of incompressible records that starts the copies
        assert pairs
        upto = sum(sizes[: pairs[0] + 2])
        assert got[:upto] == want[:upto]
    # a pair of incompressible records that still has a block behind it is exactly what makes the reference copy (protection_state.rs:38-47)
    assert (st["copy_blocks"] > 0) == any(p + 2 < len(sizes) for p in pairs)


@pytest.mark.parametrize("algo", ["cheetah", "lion"])
def test_repeats_inside_one_exchange(algo):
    """the same slot many times within 64 quads, predicted quads rewriting their prediction, zero quads against empty tables"""
    rng = np.random.default_rng(11)
    words = rng.integers(0, 2**32, size=7, dtype=np.uint32)
    words[0] = 0
    q = words[rng.integers(0, 7, size=4096)]
    data = q.astype("<u4").tobytes()
    want, st = pyoracle.encode_stats(algo, np.frombuffer(data, dtype=np.uint8))
    assert st["copy_blocks"] == 0
    got, _ = model.encode(algo, data)
    assert got == want


@pytest.mark.parametrize("kind,n,handed_back", [("prose", 96 * 1024, False), ("rep", 80 * 1024, False), ("zeros", 64 * 1024, False),
                                                ("mixed", 128 * 1024, None), ("random", 64 * 1024, True)])
def test_in_order_head_then_passes(kind, n, handed_back):
    """What exchange_stages.hip does per chunk: the first 16 KiB in order (cold dictionary: incompressible records, raw copies), the
    rest in passes from the tables the head left; a chunk whose raw copies are not over with the head goes back whole."""
    data = datagen.by_kind(kind, n, seed=23)
    want, st = pyoracle.encode_stats("cheetah", data)
    got = model.cheetah_encode_head_then_passes(bytes(data), 16 * 1024)
    if handed_back is not None:
        assert (got is None) == handed_back
    if got is not None:
        assert got == want
    if kind == "prose":
        assert st["copy_blocks"] > 0                              # ... all of them inside the head


@pytest.mark.parametrize("kind,n", [("prose", 160 * 1024), ("rep", 128 * 1024), ("zeros", 64 * 1024)])
def test_lion_in_order_head_then_passes(kind, n):
    """The same hand-over for Lion: seven tables (the prediction row level by level) seeded from an in-order head of 48 KiB."""
    data = datagen.by_kind(kind, n, seed=29)
    want, st = pyoracle.encode_stats("lion", data)
    got = model.encode_head_then_passes("lion", bytes(data), 48 * 1024)
    if got is not None:
        assert got == want
    else:
        assert st["copy_blocks"] > 0
    assert model.encode_head_then_passes("cheetah", bytes(data), 16 * 1024) in (None, pyoracle.encode("cheetah", data))
