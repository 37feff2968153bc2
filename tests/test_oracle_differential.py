"""Differential test: C oracle (value-table restatement) vs the independently structured Python model
(position formulation / list LRUs).  This is the protection against a shared misreading of the reference,
whose own tests pin a single 125-byte input."""
import random

import pytest

import datagen
from oracle import pymodel, pyoracle


def _inputs():
    rnd = random.Random(7)
    yield b""
    for n in (1, 5, 64, 255, 256, 257, 700, 1500, 4099):
        yield bytes(rnd.randrange(256) for _ in range(n))                      # incompressible -> copy mode
        yield bytes(rnd.choice(b"ab") for _ in range(n))                        # tiny alphabet: collisions + predictions
        yield bytes(rnd.choice([0, 0, 0, 1]) for _ in range(n))                 # zero quads (hit on first sight)
        yield b"".join(rnd.choice([b"the ", b"and ", b"of  ", b"cat ", b"dog ", b"xyzw"]) for _ in range(n // 4 + 1))[:n]
    yield bytes(datagen.mixed(30_000, 8))
    yield bytes(datagen.prose(20_000, 9))
    yield bytes(datagen.same_hash_quads(3000, 10))
    yield bytes(datagen.random_bytes(8000, 11))


@pytest.mark.parametrize("algo", pyoracle.ALGOS)
def test_oracle_matches_independent_model(algo):
    for data in _inputs():
        enc_c, st = pyoracle.encode_stats(algo, data)
        enc_py, copied = pymodel.encode(algo, data)
        assert enc_c == enc_py, (algo, len(data))
        assert st["copy_blocks"] == copied
        assert pymodel.decode(algo, enc_c) == data
        assert pyoracle.decode(algo, enc_py, len(data)) == data
