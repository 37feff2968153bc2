"""The record-boundary parse of density_amd/csrc/stream_parse.hip, restated in numpy (tools/parse_prototype.py), against an FSM walk of
the oracle's streams: the per-window "entry -> exit, records" tables, their composition and the forward walk must reproduce the
block index, the number of whole blocks and the offset where they end — and a stream with raw-copy blocks must be reported as not
calm.  CPU only; the GPU kernels are checked end to end by tests/test_gpu_chameleon.py::test_long_stream_*."""
import importlib.util
import os

import numpy as np
import pytest

import datagen
from oracle import pyoracle

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("parse_prototype", os.path.join(os.path.dirname(HERE), "tools", "parse_prototype.py"))
proto = importlib.util.module_from_spec(spec)
spec.loader.exec_module(proto)


@pytest.mark.parametrize("kind,n", [("prose", 600_000 + 77), ("rep", 700_000), ("zeros", 300_000 + 2), ("prose", 256 * 1200)])
def test_parse_reproduces_the_fsm_walk(kind, n):
    data = datagen.by_kind(kind, n, seed=9)
    enc = pyoracle.encode("chameleon", data)
    r = proto.parse(enc)
    assert r is not None and r[0] != "fallback"
    p0, b0, total, endpos, index = r
    want_index, want_end = proto.truth(enc, n)
    assert total == n // 256 == len(want_index)
    assert index == want_index
    assert endpos == want_end
    assert (endpos < len(enc)) == (n % 256 != 0)                  # a ragged end, if any, starts where the whole blocks stop


def test_streams_with_raw_copies_are_not_calm():
    data = datagen.by_kind("prose", 400_000, seed=3).copy()
    data[100_000:140_000] = np.random.default_rng(1).integers(0, 256, size=40_000, dtype=np.uint8)
    enc, st = pyoracle.encode_stats("chameleon", data)
    assert st["copy_blocks"] > 0
    r = proto.parse(enc)
    assert r is None or r[0] == "fallback"
