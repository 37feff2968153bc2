"""world_size-2 gloo test (CPU) of the multi-GPU sharding path: chunk-range arithmetic, the metadata all-gather and the
optional gather-to-rank-0 stitch.  Local containers are built on the CPU from oracle streams (test infrastructure) in the
exact DHC1 layout the GPU encoder writes, so the test exercises only the distributed logic, which is device-agnostic."""
import os
import struct
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen
from density_amd import parallel
from oracle import pyoracle

ALGO = "chameleon"


def cpu_container(data, chunk, with_index=True):
    """DHC1 container of `data` assembled from oracle streams (mirrors include/density_hip.h)."""
    from oracle import pymodel
    data = bytes(data)
    n = len(data)
    streams = [pyoracle.encode(ALGO, data[i:i + chunk]) for i in range(0, n, chunk)]
    nc = len(streams)
    index = bytearray()
    if with_index:
        for c0 in range(0, n, chunk):
            part = data[c0:c0 + chunk]
            enc = streams[c0 // chunk]
            g, pos = pymodel.Guard(), 0
            for b0 in range(0, len(part), 256):
                blen = min(256, len(part) - b0)
                if g.next_is_copy():
                    index.append(0x80 | (0x7F if blen < 256 else 0)); pos += blen; g.decay()
                else:
                    hits = bin(int.from_bytes(enc[pos:pos + 8], "little")).count("1")
                    index.append(0x7F if blen < 256 else hits)
                    rl = 8 + 4 * (blen // 4) - 2 * hits + blen % 4
                    g.update(rl >= 256); pos += rl
    idx_at = (32 + 4 * nc + 15) // 16 * 16
    pay_at = (idx_at + len(index) + 15) // 16 * 16
    body = bytearray()
    for k, s in enumerate(streams):
        body += s
        if k + 1 < nc:
            body += bytes(-len(body) % 16)
    total = pay_at + len(body)
    head = struct.pack("<IBBHIIQQ", 0x31434844, 0, 1, 1 if with_index else 0, chunk, nc, n, total)
    raw = head + b"".join(struct.pack("<I", len(s)) for s in streams)
    raw += bytes(idx_at - len(raw)) + bytes(index)
    raw += bytes(pay_at - len(raw)) + bytes(body)
    return raw


def cpu_decode_container(raw):
    hdr, table, index, payload = parallel.parse_local(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    sizes = np.frombuffer(bytes(table.numpy()), dtype="<u4")
    out, off = bytearray(), 0
    pay = bytes(payload.numpy())
    for i, s in enumerate(sizes):
        want = min(hdr["chunk_size"], hdr["total_len"] - i * hdr["chunk_size"])
        out += pyoracle.decode(ALGO, pay[off:off + int(s)], want)
        off = (off + int(s) + 15) // 16 * 16
    return bytes(out)


def test_shard_chunks_partition():
    for total, chunk, world in [(0, 256, 2), (1000, 256, 2), (10 * 4096 + 5, 4096, 4), (1 << 20, 1 << 16, 8), (777, 1 << 20, 8)]:
        prev_c, prev_b = 0, 0
        for r in range(world):
            c0, c1, b0, b1 = parallel.shard_chunks(total, chunk, r, world)
            assert c0 == prev_c and b0 == prev_b and c1 >= c0 and b1 >= b0
            assert b0 % chunk == 0 or b0 == total
            prev_c, prev_b = c1, b1
        assert prev_c == (total + chunk - 1) // chunk and prev_b == total


def _worker(rank, world, initfile, total, chunk, with_index, ret):
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        data = datagen.mixed(total, seed=3)
        c0, c1, b0, b1 = parallel.shard_chunks(total, chunk, rank, world)
        local = torch.frombuffer(bytearray(cpu_container(data[b0:b1], chunk, with_index)), dtype=torch.uint8)
        hdr, table, index, payload = parallel.parse_local(local)
        assert hdr["n_chunks"] == c1 - c0 and hdr["total_len"] == b1 - b0
        lay = parallel.exchange_layout(hdr["n_chunks"], payload.numel(), hdr["total_len"], torch.device("cpu"))
        assert lay["chunk_offset"] == c0 and lay["input_offset"] == b0 and sum(lay["input_bytes"]) == total
        merged = parallel.concat_to_rank0(local, chunk)
        if rank == 0:
            raw = bytes(merged.numpy())
            want = cpu_container(data, chunk, with_index)
            ret["equal_to_single_process_container"] = raw == want
            ret["round_trip"] = cpu_decode_container(raw) == data.tobytes()
        else:
            assert merged is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_index", [True, False])
def test_two_rank_gather_and_concat_matches_single_process(with_index):
    total, chunk, world = 9 * 4096 + 1234, 4096, 2
    with tempfile.TemporaryDirectory() as d:
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, os.path.join(d, "init"), total, chunk, with_index, ret), nprocs=world, join=True)
        assert ret["equal_to_single_process_container"], "stitched container differs from the one-process container"
        assert ret["round_trip"]


def _multi_worker(rank, world, initfile, total, chunk, ret):
    """Config 5's wire form: every rank's PAGED container (built on the CPU by the header's layout, tests/paged_cpu.py) travels as it stands behind the DHCM
    front matter; rank 0 gathers the blobs, a CPU reader decodes them rank by rank."""
    import paged_cpu
    from density_amd import container
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        data = datagen.mixed(total, seed=5)
        c0, c1, b0, b1 = parallel.shard_chunks(total, chunk, rank, world)
        blob = paged_cpu.build(data[b0:b1], chunk) if b1 > b0 else np.zeros(0, dtype=np.uint8)
        local = torch.frombuffer(bytearray(blob.tobytes()), dtype=torch.uint8) if blob.size else torch.zeros(0, dtype=torch.uint8)
        front, rows, length = parallel.exchange_multi_layout(local.numel(), b1 - b0, torch.device("cpu"), 0, chunk)
        assert rows[rank][1] == local.numel() and sum(r[2] for r in rows) == total and rows[rank][0] % 256 == 0
        merged = parallel.concat_multi_to_rank0(local, b1 - b0, 0, chunk)
        if rank == 0:
            hdr, got_rows = parallel.parse_multi(merged)
            assert got_rows == rows and hdr["total_len"] == total and hdr["container_len"] == merged.numel() == length and hdr["n_ranks"] == world
            out = bytearray()
            for off, ln, nb in got_rows:                                        # a CPU reader: blob by blob, chunk by chunk
                if not nb:
                    continue
                sub = merged[off:off + ln].numpy()
                h, streams = container.chunk_payloads(sub)
                assert h.flags & container.FLAG_PAGED
                for i, st in enumerate(streams):
                    out += pyoracle.decode(ALGO, st, min(h.chunk_size, h.total_len - i * h.chunk_size))
            ret["multi_round_trip"] = bytes(out) == data.tobytes()
            ret["own_blob_in_place"] = bytes(merged[rows[0][0]:rows[0][0] + rows[0][1]].numpy()) == blob.tobytes()
        else:
            assert merged is None
    finally:
        dist.destroy_process_group()


def test_two_rank_multi_container_of_paged_blobs():
    total, chunk, world = 5 * 65536 + 4321, 65536, 2
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_multi_worker, args=(world, os.path.join(d, "init"), total, chunk, ret), nprocs=world, join=True)
        assert ret["multi_round_trip"] and ret["own_blob_in_place"]


def test_bench_launcher_starts_the_ranks_itself():
    """bench.py --gpus N without a torchrun environment re-executes under torch.distributed.run with N ranks and reports the world size the
    process group saw (dry mode: CPU tensors, gloo)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-gpu", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] and rec["scaling"] == "weak"
