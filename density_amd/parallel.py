"""Multi-GPU sharding of the container path: one process per GPU, torch.distributed (RCCL on GPUs, gloo on
CPU for tests).

Chunks are independent reference streams, so a contiguous range of chunks is a shard: rank g of G encodes chunks
[g*n/G, (g+1)*n/G) of the global input into its own local container with NO data-path collective.  The only exchange is
metadata: an all-gather of each rank's (chunk count, payload bytes) — 16 bytes per rank — from which every rank derives
where its payload region sits in the global container.  Concatenating the payloads (gather-to-one or all-gather) is optional
and bounded by xGMI, not by the codec (7 links x ~153 GB/s per GPU), so it is a separate, separately timed
step.  Everything here is layout arithmetic on tensors + collectives; it runs unchanged on CPU tensors under gloo.
"""
import struct

import torch
import torch.distributed as dist

HEADER_BYTES = 32
MAGIC = 0x31434844
FLAG_BLOCK_INDEX = 1


def _align16(v):
    return (v + 15) // 16 * 16


def shard_chunks(total_len, chunk_size, rank, world):
    """Chunk range [c0, c1) and byte range [b0, b1) of `rank`: contiguous, chunk-aligned, balanced to within one chunk."""
    n_chunks = (total_len + chunk_size - 1) // chunk_size
    c0 = n_chunks * rank // world
    c1 = n_chunks * (rank + 1) // world
    b0 = min(c0 * chunk_size, total_len)
    b1 = min(c1 * chunk_size, total_len)
    return c0, c1, b0, b1


def parse_local(container):
    """Splits a local container (1-D uint8 tensor, any device) into (header dict, size table, block index or None, payload region).

    The payload region is the byte range from the first payload to container_len; payload offsets inside it are relative
    and every payload starts 16-byte aligned, so regions of consecutive shards concatenate after padding to 16."""
    head = bytes(container[:HEADER_BYTES].cpu().numpy())
    magic, algo, version, flags, chunk_size, n_chunks, total_len, container_len = struct.unpack("<IBBHIIQQ", head)
    if magic != MAGIC:
        raise ValueError("not a DHC1 container")
    table = container[HEADER_BYTES:HEADER_BYTES + 4 * n_chunks]
    idx_at = _align16(HEADER_BYTES + 4 * n_chunks)
    n_idx = (total_len + 255) // 256 if flags & FLAG_BLOCK_INDEX else 0
    index = container[idx_at:idx_at + n_idx] if n_idx else None
    pay_at = _align16(idx_at + n_idx)
    hdr = dict(algo=algo, version=version, flags=flags, chunk_size=chunk_size, n_chunks=n_chunks, total_len=total_len,
               container_len=container_len)
    return hdr, table, index, container[pay_at:container_len]


def exchange_layout(n_chunks_local, payload_bytes_local, total_len_local, device, group=None):
    """The path's only collective on the encode side: all-gather of (chunks, payload bytes, input bytes) per rank.

    Returns per-rank lists and this rank's chunk / payload-byte offsets in the global container's tables / payload area."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([n_chunks_local, payload_bytes_local, total_len_local], dtype=torch.int64, device=device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    rows = torch.stack(gathered).cpu().tolist()
    chunks = [r[0] for r in rows]
    # every region but the last non-empty one is padded so that the next payload starts 16-byte aligned
    last = max([i for i, r in enumerate(rows) if r[1] > 0], default=-1)
    pay = [(r[1] if i == last else _align16(r[1])) for i, r in enumerate(rows)]
    lens = [r[2] for r in rows]
    return dict(chunks=chunks, payload_bytes=pay, input_bytes=lens,
                chunk_offset=sum(chunks[:rank]), payload_offset=sum(pay[:rank]), input_offset=sum(lens[:rank]))


def global_layout(layout, chunk_size, flags):
    """Offsets of the tables and of the payload area in the concatenated global container."""
    n_chunks = sum(layout["chunks"])
    total_len = sum(layout["input_bytes"])
    idx_at = _align16(HEADER_BYTES + 4 * n_chunks)
    n_idx = (total_len + 255) // 256 if flags & FLAG_BLOCK_INDEX else 0
    pay_at = _align16(idx_at + n_idx)
    # the last rank's payload region is not padded at the end
    container_len = pay_at + sum(layout["payload_bytes"])
    return dict(n_chunks=n_chunks, total_len=total_len, index_at=idx_at, index_bytes=n_idx, payload_at=pay_at, container_len=container_len)


def concat_to_rank0(local_container, chunk_size, group=None):
    """Optional stitch across GPUs: rank 0 receives every rank's size table, block-index slice and payload region and writes
    the global container (valid input for density_hip_decode on one GPU).  Other ranks return None.

    Every shard but the last covers whole chunks (shard_chunks), so block-index slices concatenate without re-basing."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_container.device
    hdr, table, index, payload = parse_local(local_container)
    lay = exchange_layout(hdr["n_chunks"], payload.numel(), hdr["total_len"], dev, group)
    glob = global_layout(lay, chunk_size, hdr["flags"])
    pad = lay["payload_bytes"][rank] - payload.numel()
    if pad:
        payload = torch.cat([payload, torch.zeros(pad, dtype=torch.uint8, device=dev)])
    parts = [table.contiguous(), index.contiguous() if index is not None else torch.empty(0, dtype=torch.uint8, device=dev), payload.contiguous()]
    if rank != 0:
        ops = [dist.P2POp(dist.isend, p, 0, group=group) for p in parts if p.numel()]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return None
    out = torch.zeros(glob["container_len"], dtype=torch.uint8, device=dev)
    # every piece is received straight into its place in the global container, all ranks at once (one batched P2P group:
    # with RCCL the seven senders use their own xGMI links concurrently instead of one link at a time)
    t_at, i_at, p_at = HEADER_BYTES, glob["index_at"], glob["payload_at"]
    ops = []
    for r in range(world):
        sizes = [4 * lay["chunks"][r], (lay["input_bytes"][r] + 255) // 256 if hdr["flags"] & FLAG_BLOCK_INDEX else 0, lay["payload_bytes"][r]]
        dests = [out[t_at:t_at + sizes[0]], out[i_at:i_at + sizes[1]], out[p_at:p_at + sizes[2]]]
        for src, d in zip(parts, dests):
            if d.numel() == 0:
                continue
            if r == 0:
                d.copy_(src)
            else:
                ops.append(dist.P2POp(dist.irecv, d, r, group=group))
        t_at += sizes[0]; i_at += sizes[1]; p_at += sizes[2]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    head = struct.pack("<IBBHIIQQ", MAGIC, hdr["algo"], 1, hdr["flags"], chunk_size, glob["n_chunks"], glob["total_len"], glob["container_len"])
    out[:HEADER_BYTES] = torch.frombuffer(bytearray(head), dtype=torch.uint8).to(dev)
    return out


# ---- config 5's wire form: the multi-rank container "DHCM" (include/density_hip.h) ----
# Every rank's container travels AS IT STANDS — the paged blob the rank times, or a packed / slotted one: each is a self-describing DHC1 container of that
# rank's shard — behind a 32-byte super-header and one {offset, length, input bytes} row per rank.  Still ONE collective: an all-gather of two u64 per rank.
MULTI_MAGIC = 0x4D434844
MULTI_HEADER = "<IBBHIIQQ"
MULTI_ROW = "<QQQ"


def _align256(v):
    return (v + 255) // 256 * 256


def multi_layout(lengths, input_bytes, algo=0, chunk_size=0):
    """(header bytes, [(offset, length, input_bytes)], container_len) of the super-container over the ranks' blobs (the arithmetic of
    density_hip_multi_layout: tests/test_placement_abi.py holds the two against each other)."""
    world = len(lengths)
    at = _align256(32 + 24 * world)
    rows = []
    for r in range(world):
        rows.append((at, int(lengths[r]), int(input_bytes[r])))
        at += int(lengths[r])
        if r + 1 < world:
            at = _align256(at)
    head = struct.pack(MULTI_HEADER, MULTI_MAGIC, 1, algo, 0, world, chunk_size, sum(int(v) for v in input_bytes), at)
    return head + b"".join(struct.pack(MULTI_ROW, *row) for row in rows), rows, at


def exchange_multi_layout(container_len_local, input_bytes_local, device, algo=0, chunk_size=0, group=None):
    """The collective of config 5: all-gather of (container length, input bytes) per rank -> the super-container's front matter, every rank's row, its length."""
    world = dist.get_world_size(group)
    mine = torch.tensor([container_len_local, input_bytes_local], dtype=torch.int64, device=device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    rows = torch.stack(gathered).cpu().tolist()
    return multi_layout([r[0] for r in rows], [r[1] for r in rows], algo, chunk_size)


def parse_multi(blob):
    """(header dict, [(offset, length, input_bytes)]) of a super-container (1-D uint8 tensor, any device); raises ValueError on a malformed front."""
    head = bytes(blob[:32].cpu().numpy())
    if len(head) < 32:
        raise ValueError("not a DHCM container")
    magic, version, algo, flags, world, chunk_size, total_len, container_len = struct.unpack(MULTI_HEADER, head)
    if magic != MULTI_MAGIC or version != 1 or flags != 0 or world == 0 or 32 + 24 * world > blob.numel() or container_len > blob.numel():
        raise ValueError("not a DHCM container")
    raw = bytes(blob[32:32 + 24 * world].cpu().numpy())
    rows = [struct.unpack_from(MULTI_ROW, raw, 24 * r) for r in range(world)]
    at, total = 32 + 24 * world, 0
    for off, ln, nb in rows:
        if off < at or off % 256 or off + ln > container_len:
            raise ValueError("DHCM rows out of order or outside the container")
        at, total = off + ln, total + nb
    if total != total_len:
        raise ValueError("DHCM input bytes do not add up")
    return dict(algo=algo, n_ranks=world, chunk_size=chunk_size, total_len=total_len, container_len=container_len), rows


def concat_multi_to_rank0(local_blob, input_bytes_local, algo=0, chunk_size=0, group=None):
    """Optional gather of config 5's output onto one GPU: rank 0 receives every rank's blob straight into its place of the super-container (one batched P2P
    group: the senders use their own xGMI links at once) and returns it; the others return None.  xGMI-bound, never part of `value`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_blob.device
    front, rows, total = exchange_multi_layout(local_blob.numel(), input_bytes_local, dev, algo, chunk_size, group)
    if rank != 0:
        if local_blob.numel():
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_blob.contiguous(), 0, group=group)]):
                w.wait()
        return None
    out = torch.zeros(total, dtype=torch.uint8, device=dev)
    out[:len(front)] = torch.frombuffer(bytearray(front), dtype=torch.uint8).to(dev)
    ops = []
    for r, (off, ln, _) in enumerate(rows):
        if ln == 0:
            continue
        if r == 0:
            out[off:off + ln].copy_(local_blob)
        else:
            ops.append(dist.P2POp(dist.irecv, out[off:off + ln], r, group=group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return out


def decode_multi_device(blob, out):
    """Decodes a device-resident super-container rank by rank on THIS device (density_hip_decode_device per row, each blob in place); returns the bytes written."""
    from . import container
    hdr, rows = parse_multi(blob)
    at = 0
    for off, ln, nb in rows:
        if nb:
            got = container.decode_device(blob.data_ptr() + off, ln, out.data_ptr() + at, nb)
            assert got == nb
        at += nb
    return at
