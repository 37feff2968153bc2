"""A PAGED container (include/density_hip.h: DENSITY_HIP_FLAG_PAGED) assembled on the CPU from the oracle's chunk streams — what a CPU producer that follows
the header's layout would write: header, size table, block index, page directory, pages.  Test infrastructure (it calls the oracle): the GPU decoder must take such
a container like one of its own, and the CPU reader (density_amd.container.chunk_payloads) must give the streams back."""
import numpy as np

from density_amd import _lib
from oracle import pymodel, pyoracle

PAGE = 65536
ALGO = "chameleon"


def walk_records(stream, n_bytes):
    """Per 256-byte block of a chunk of n_bytes: (index byte, bytes of stream), by walking the records with the reference FSM (codec.rs:88-123)."""
    g, pos, out = pymodel.Guard(), 0, []
    for b0 in range(0, n_bytes, 256):
        blen = min(256, n_bytes - b0)
        ragged = blen < 256
        if g.next_is_copy():
            out.append((0x80 | (0x7F if ragged else 0), blen))
            pos += blen
            g.decay()
        else:
            sig = int.from_bytes(stream[pos:pos + 8], "little")
            hits = bin(sig).count("1")
            reclen = 8 + 4 * (blen // 4) - 2 * hits + blen % 4
            out.append((0x7F if ragged else hits, reclen))
            g.update(reclen >= 256)
            pos += reclen
    assert pos == len(stream)
    return out


def build(data, chunk, page_order=None):
    """The container as bytes.  Pages are taken in chunk order unless `page_order` (a permutation of the pages in use) says otherwise — the GPU takes them
    from an atomic counter, in whatever order its work-groups get there."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.size
    n_chunks = (n + chunk - 1) // chunk
    ppc = int(_lib.lib().density_hip_paged_pages_per_chunk(chunk))
    streams, index, chunk_pages = [], bytearray(), []
    for i in range(n_chunks):
        part = data[i * chunk:(i + 1) * chunk]
        s = pyoracle.encode(ALGO, part)
        blocks = walk_records(s, part.size)
        index += bytes(b for b, _ in blocks)
        # units that may not be split over pages: whole rounds of 16 blocks; the last whole round carries the partial round and the ragged block behind it
        nfull = part.size // 256
        nrounds = nfull // 16
        units = [[16 * r, sum(ln for _, ln in blocks[16 * r:16 * r + 16])] for r in range(nrounds)]
        tail = sum(ln for _, ln in blocks[16 * nrounds:])
        if units: units[-1][1] += tail
        else: units = [[0, tail]]
        pages, used, first = [], 0, 0                                   # (first block, bytes) per page of this chunk
        for blk, ln in units:
            if used and used + ln >= PAGE:                              # would not END INSIDE the page (strictly): the next page
                pages.append((first, used)); used, first = 0, blk
            used += ln
        pages.append((first, used))
        assert sum(u for _, u in pages) == len(s) and len(pages) <= ppc
        streams.append(s); chunk_pages.append(pages)
    total_pages = sum(len(p) for p in chunk_pages)
    order = list(range(total_pages)) if page_order is None else [int(v) for v in page_order]
    assert sorted(order) == list(range(total_pages))
    table = 32 + 4 * n_chunks
    ix0 = (table + 15) // 16 * 16
    dir0 = (ix0 + (n + 255) // 256 + 15) // 16 * 16
    pages0 = (dir0 + 16 * (ppc + 1) * n_chunks + 255) // 256 * 256
    out = np.zeros(pages0 + PAGE * total_pages, dtype=np.uint8)
    hdr = _lib.Header()
    hdr.magic, hdr.algo, hdr.version, hdr.flags = 0x31434844, 0, 1, 1 | 4
    hdr.chunk_size, hdr.n_chunks, hdr.total_len, hdr.container_len = chunk, n_chunks, n, out.size
    out[:32] = np.frombuffer(bytes(hdr), dtype=np.uint8)
    out[ix0:ix0 + len(index)] = np.frombuffer(bytes(index), dtype=np.uint8)
    k = 0
    for i, (s, pages) in enumerate(zip(streams, chunk_pages)):
        out[32 + 4 * i:36 + 4 * i] = np.frombuffer(len(s).to_bytes(4, "little"), dtype=np.uint8)
        d = dir0 + 16 * (ppc + 1) * i
        out[d:d + 4] = np.frombuffer(len(pages).to_bytes(4, "little"), dtype=np.uint8)
        at = 0
        for j, (first, used) in enumerate(pages):
            page = order[k]; k += 1
            e = d + 16 * (j + 1)
            out[e:e + 12] = np.frombuffer(page.to_bytes(4, "little") + first.to_bytes(4, "little") + used.to_bytes(4, "little"), dtype=np.uint8)
            out[pages0 + page * PAGE:pages0 + page * PAGE + used] = np.frombuffer(s[at:at + used], dtype=np.uint8)
            at += used
    return out
