"""The C-ABI placement helpers (include/density_hip.h: density_hip_shard_range, density_hip_global_layout — what a C / Rust caller with its
own RCCL all-gather uses) against density_amd/parallel.py, the torch.distributed form of the same arithmetic.  Host
arithmetic only: runs without a GPU."""
import ctypes
import random

import pytest

from density_amd import _lib, parallel


def _shard(total, chunk, rank, world):
    out = _lib.Shard()
    rc = _lib.lib().density_hip_shard_range(total, chunk, rank, world, ctypes.byref(out))
    return rc, (out.chunk_first, out.chunk_end, out.byte_first, out.byte_end)


def test_shard_range_matches_parallel_py():
    rnd = random.Random(7)
    cases = [(0, 256, 1), (1, 256, 1), (1 << 30, 4 << 20, 8), (8 << 30, 4 << 20, 8), (10_192_446, 65536, 8), (100_000_000, 393_216, 8),
             (255, 256, 4), (257, 256, 3), (5 * 65536 + 123, 65536, 4)]
    for _ in range(300):
        chunk = 256 * rnd.randint(1, 20000)
        cases.append((rnd.randint(0, 1 << 36), chunk, rnd.randint(1, 16)))
    for total, chunk, world in cases:
        prev_end = (0, 0)
        for rank in range(world):
            rc, got = _shard(total, chunk, rank, world)
            assert rc == _lib.OK
            assert got == parallel.shard_chunks(total, chunk, rank, world), (total, chunk, rank, world)
            assert (got[0], got[2]) == prev_end                      # contiguous
            prev_end = (got[1], got[3])
        assert prev_end == ((total + chunk - 1) // chunk, total)     # and complete


def test_shard_range_rejects_bad_arguments():
    assert _shard(1000, 100, 0, 1)[0] == _lib.ERR_ARGUMENT           # chunk: a multiple of 256
    assert _shard(1000, 0, 0, 1)[0] == _lib.ERR_ARGUMENT
    assert _shard(1000, 256, 1, 1)[0] == _lib.ERR_ARGUMENT           # rank < world
    assert _shard(1000, 256, 0, 0)[0] == _lib.ERR_ARGUMENT
    assert _lib.lib().density_hip_shard_range(1000, 256, 0, 1, None) == _lib.ERR_ARGUMENT


def _global(chunks, pays, lens, rank, flags):
    w = len(chunks)
    A = ctypes.c_uint64 * w
    out = _lib.GlobalLayout()
    rc = _lib.lib().density_hip_global_layout(A(*chunks), A(*pays), A(*lens), w, rank, flags, ctypes.byref(out))
    return rc, out


def _py_layout(chunks, pays, lens, rank, chunk_size, flags):
    """parallel.exchange_layout's bookkeeping without the collective + parallel.global_layout"""
    rows = list(zip(chunks, pays, lens))
    last = max([i for i, r in enumerate(rows) if r[1] > 0], default=-1)
    pay = [(r[1] if i == last else parallel._align16(r[1])) for i, r in enumerate(rows)]
    lay = dict(chunks=list(chunks), payload_bytes=pay, input_bytes=list(lens), chunk_offset=sum(chunks[:rank]), payload_offset=sum(pay[:rank]),
               input_offset=sum(lens[:rank]))
    return lay, parallel.global_layout(lay, chunk_size, flags)


@pytest.mark.parametrize("flags", [0, 1])
def test_global_layout_matches_parallel_py(flags):
    rnd = random.Random(11)
    for _ in range(200):
        world = rnd.randint(1, 9)
        chunk = 256 * rnd.randint(1, 4096)
        chunks = [rnd.randint(0, 40) for _ in range(world)]
        lens = [c * chunk for c in chunks]
        if any(chunks):                                              # the last non-empty shard may be ragged
            k = max(i for i, c in enumerate(chunks) if c)
            lens[k] -= rnd.randint(0, chunk - 1)
        pays = [rnd.randint(1, 300) * c + rnd.randint(0, 15) if c else 0 for c in chunks]
        for rank in range(world):
            rc, got = _global(chunks, pays, lens, rank, flags)
            assert rc == _lib.OK
            lay, glob = _py_layout(chunks, pays, lens, rank, chunk, flags)
            assert (got.n_chunks, got.total_len, got.index_at, got.index_bytes, got.payload_at, got.container_len) == \
                   (glob["n_chunks"], glob["total_len"], glob["index_at"], glob["index_bytes"], glob["payload_at"], glob["container_len"])
            assert (got.chunk_offset, got.payload_offset, got.input_offset) == (lay["chunk_offset"], lay["payload_offset"], lay["input_offset"])
            assert got.payload_bytes_padded == lay["payload_bytes"][rank]


def test_global_layout_describes_a_real_stitched_container():
    """Two oracle-built local containers, concatenated by hand at the offsets the C helper gives: the bytes parallel.py's gloo test stitches."""
    import numpy as np
    import datagen
    from test_parallel_gloo import cpu_container
    chunk = 4096
    data = datagen.mixed(9 * chunk + 777, seed=5)
    world = 2
    locals_ = []
    for r in range(world):
        rc, (c0, c1, b0, b1) = _shard(data.size, chunk, r, world)
        locals_.append(cpu_container(data[b0:b1], chunk))
    whole = cpu_container(data, chunk)
    import torch
    parts = [parallel.parse_local(torch.frombuffer(bytearray(c), dtype=torch.uint8)) for c in locals_]
    chunks = [p[0]["n_chunks"] for p in parts]
    pays = [p[3].numel() for p in parts]
    lens = [p[0]["total_len"] for p in parts]
    flags = parts[0][0]["flags"]
    out = None
    for r in range(world):
        rc, g = _global(chunks, pays, lens, r, flags)
        assert rc == _lib.OK
        if out is None:
            out = np.zeros(g.container_len, dtype=np.uint8)
        hdr, table, index, payload = parts[r]
        out[32 + 4 * g.chunk_offset:32 + 4 * (g.chunk_offset + chunks[r])] = table.numpy()
        if index is not None:
            at = g.index_at + g.input_offset // 256
            out[at:at + index.numel()] = index.numpy()
        out[g.payload_at + g.payload_offset:g.payload_at + g.payload_offset + pays[r]] = payload.numpy()
    import struct
    out[:32] = np.frombuffer(struct.pack("<IBBHIIQQ", parallel.MAGIC, parts[0][0]["algo"], 1, flags, chunk, g.n_chunks, g.total_len, g.container_len), dtype=np.uint8)
    assert bytes(out) == bytes(whole)


# ---- the multi-rank container "DHCM" (config 5's wire form) ----
def _c_multi(lengths, inputs, algo, chunk):
    w = len(lengths)
    A = ctypes.c_uint64 * w
    hdr = _lib.MultiHeader()
    rows = (_lib.MultiRow * w)()
    rc = _lib.lib().density_hip_multi_layout(A(*lengths), A(*inputs), w, algo, chunk, ctypes.byref(hdr), rows)
    return rc, hdr, [(r.offset, r.length, r.input_bytes) for r in rows]


def test_multi_layout_matches_parallel_py_and_validates():
    rnd = random.Random(11)
    for _ in range(300):
        w = rnd.randint(1, 16)
        lengths = [rnd.choice([0, rnd.randint(1, 1 << 34)]) for _ in range(w)]
        inputs = [rnd.randint(0, 1 << 34) for _ in range(w)]
        algo, chunk = rnd.randint(0, 2), 256 * rnd.randint(1, 1 << 14)
        rc, hdr, rows = _c_multi(lengths, inputs, algo, chunk)
        front, prows, total = parallel.multi_layout(lengths, inputs, algo, chunk)
        assert rc == _lib.OK and rows == prows and hdr.container_len == total and hdr.total_len == sum(inputs) and hdr.n_ranks == w
        assert bytes(hdr) == front[:32]
        assert all(o % 256 == 0 for o, _, _ in rows) and all(rows[i][0] + rows[i][1] <= rows[i + 1][0] for i in range(w - 1))
        # the reader's check of the front matter gives every row back ...
        got_h, got_r = _lib.MultiHeader(), _lib.MultiRow()
        for r in range(w):
            assert _lib.lib().density_hip_multi_row(front, len(front), total, r, ctypes.byref(got_h), ctypes.byref(got_r)) == _lib.OK
            assert (got_r.offset, got_r.length, got_r.input_bytes) == rows[r] and got_h.container_len == total
        assert _lib.lib().density_hip_multi_row(front, len(front), total, w, ctypes.byref(got_h), ctypes.byref(got_r)) == _lib.ERR_ARGUMENT
        # ... and refuses a front that lies: a truncated container, a row moved inside its predecessor, input bytes that do not add up, a wrong magic
        if total:
            assert _lib.lib().density_hip_multi_row(front, len(front), total - 1, 0, ctypes.byref(got_h), ctypes.byref(got_r)) == _lib.ERR_FORMAT
        bad = bytearray(front); bad[0] ^= 1
        assert _lib.lib().density_hip_multi_row(bytes(bad), len(bad), total, 0, ctypes.byref(got_h), ctypes.byref(got_r)) == _lib.ERR_FORMAT
        bad = bytearray(front); bad[32 + 16:32 + 24] = (inputs[0] + 1).to_bytes(8, "little")
        assert _lib.lib().density_hip_multi_row(bytes(bad), len(bad), total, 0, ctypes.byref(got_h), ctypes.byref(got_r)) == _lib.ERR_FORMAT
        if w > 1 and lengths[0] > 256:
            bad = bytearray(front); bad[32 + 24:32 + 32] = (rows[0][0]).to_bytes(8, "little")      # rank 1's blob on top of rank 0's
            assert _lib.lib().density_hip_multi_row(bytes(bad), len(bad), total, 1, ctypes.byref(got_h), ctypes.byref(got_r)) == _lib.ERR_FORMAT
    assert _c_multi([1], [1], 3, 256)[0] == _lib.ERR_ARGUMENT
    assert _lib.lib().density_hip_multi_layout(None, None, 1, 0, 256, None, None) == _lib.ERR_ARGUMENT
