"""The local rules by which the rotation decoder validates the raw-copy bits of a block index (rotor.hip::index_fsm_consistent,
restated in tools/index_fsm_model.py) accept exactly the indexes a walk of the reference's blow-up protection FSM accepts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import index_fsm_model as m


def test_generated_indexes_are_accepted_and_mutations_judged_like_the_fsm_walk():
    rng = np.random.default_rng(5)
    judged = rejected = 0
    for trial in range(400):
        n = int(rng.integers(1, 700))
        density = float(rng.choice([0.02, 0.2, 0.5, 0.8, 0.97, 1.0]))
        incs = rng.random(n) < density
        if trial % 7 == 0:                                    # long incompressible stretches between calm ones
            incs[:] = False
            for a in rng.integers(0, n, size=4):
                incs[a:a + int(rng.integers(2, 200))] = True
        ix = m.make_index(incs)
        if rng.integers(0, 3) == 0 and n > 1:
            ix[-1] = (ix[-1] & 0x80) | 0x7F                   # a ragged last block (raw or coded)
        assert m.fsm_walk_ok(ix) and m.index_consistent(ix), trial
        for _ in range(12):
            bad = bytearray(ix)
            k = int(rng.integers(0, n))
            how = int(rng.integers(0, 4))
            if how == 0:
                bad[k] ^= 0x80
                if not (bad[k] & 0x80):
                    bad[k] = int(rng.choice([0, 3, 4, 5, 40]))
                else:
                    bad[k] = 0x80
            elif how == 1:
                bad[k] = int(rng.choice([0, 4, 5, 64]))
            elif how == 2:
                del bad[k]
            else:
                bad.insert(k, int(rng.choice([0x80, 1, 33])))
            if len(bad) == 0:
                continue
            want = m.fsm_walk_ok(bad)
            assert m.index_consistent(bad) == want, (trial, k, how, bytes(ix).hex(), bytes(bad).hex())
            judged += 1
            rejected += not want
    assert judged > 3000 and rejected > 1000


def test_index_of_oracle_streams_is_consistent():
    """Indexes derived from the oracle's Chameleon streams of mixed and random data (raw copies in and out)."""
    import datagen
    from oracle import pyoracle
    for kind, n, seed in (("random", 200_000, 1), ("mixed", 600_000, 2), ("prose", 100_000, 3)):
        data = datagen.by_kind(kind, n, seed=seed)
        enc, st = pyoracle.encode_stats("chameleon", data)
        ix = bytearray()
        penalty, start, prev, counter, pos = 0, 1, False, 0, 0
        for b0 in range(0, n, 256):
            blen = min(256, n - b0)
            if (counter & 15) == 0 and start > 1:
                start >>= 1
            counter += 1
            if penalty > 0:
                ix.append(0x80 | (0x7F if blen < 256 else 0))
                pos += blen
                penalty -= 1
                if penalty == 0:
                    start += 1
            else:
                hits = bin(int.from_bytes(enc[pos:pos + 8], "little")).count("1")
                ix.append(0x7F if blen < 256 else hits)
                reclen = 8 + 4 * (blen // 4) - 2 * hits + blen % 4
                inc = reclen >= 256
                if inc and prev:
                    penalty = start
                prev = inc
                pos += reclen
        assert pos == len(enc)
        assert m.fsm_walk_ok(ix) and m.index_consistent(ix), kind
        if kind != "prose":
            assert st["copy_blocks"] > 0 and any(e & 0x80 for e in ix)
