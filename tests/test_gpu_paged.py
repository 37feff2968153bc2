"""The paged container (include/density_hip.h DENSITY_HIP_FLAG_PAGED): wire-ready without a stitch pass.  A chunk's stream — the used bytes of its
pages in directory order, reassembled on the CPU exactly as a CPU reader would — must be the oracle's stream of that chunk (codec.rs:72-80 per chunk),
page changes fall on multiples of 16 blocks, and the GPU decodes the pages in place."""
import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu


def _encode_paged(host, chunk):
    import torch
    from density_amd import container
    n = host.size
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound_paged("chameleon", n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    hdr = container.encode_device_paged("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    return x, cont, hdr


@pytest.mark.parametrize("kind,n,chunk", [("rep-text", 64 << 20, 4 << 20), ("mixed", 24 << 20, 1 << 20), ("random", 16 << 20, 2 << 20),
                                           ("zeros", 16 << 20, 1 << 20), ("rep-text", (32 << 20) + 12345, 4 << 20)])
def test_paged_streams_are_the_oracles(kind, n, chunk):
    from density_amd import container
    from oracle import pyoracle
    host = datagen.rep_text(n) if kind == "rep-text" else datagen.by_kind(kind, n, seed=11)
    x, cont, hdr = _encode_paged(host, chunk)
    assert hdr.flags & container.FLAG_PAGED, "this shape is what the paged form is for"
    blob = cont[:hdr.container_len].cpu().numpy()
    h, streams = container.chunk_payloads(blob)
    assert len(streams) == (n + chunk - 1) // chunk
    for i, s in enumerate(streams):
        assert s == pyoracle.encode("chameleon", host[i * chunk:(i + 1) * chunk]), f"chunk {i}"
    # density: the unused tails of the pages are all that separates the blob from the packed form
    packed = sum(len(s) for s in streams)
    assert hdr.container_len < 1.09 * packed + (2 << 20), (hdr.container_len, packed)
    # the directory: page changes on multiples of 16 blocks, pages used once
    b = bytes(blob)
    off = (32 + 4 * h.n_chunks + 15) // 16 * 16
    off = (off + (h.total_len + 255) // 256 + 15) // 16 * 16
    ppc = int(__import__("density_amd")._lib.lib().density_hip_paged_pages_per_chunk(h.chunk_size))
    seen = set()
    for i in range(h.n_chunks):
        d = off + 16 * (ppc + 1) * i
        for k in range(int.from_bytes(b[d:d + 4], "little")):
            e = d + 16 * (k + 1)
            page, first = int.from_bytes(b[e:e + 4], "little"), int.from_bytes(b[e + 4:e + 8], "little")
            assert first % 16 == 0 and page not in seen
            seen.add(page)


@pytest.mark.parametrize("kind,n,chunk", [("rep-text", 64 << 20, 4 << 20), ("mixed", 24 << 20, 1 << 20), ("random", 16 << 20, 2 << 20),
                                           ("zeros", 16 << 20, 1 << 20), ("rep-text", (32 << 20) + 12345, 4 << 20), ("prose", 40 << 20, 3 << 20)])
def test_paged_container_decodes_in_place(kind, n, chunk):
    import torch
    from density_amd import container
    host = datagen.rep_text(n) if kind == "rep-text" else datagen.by_kind(kind, n, seed=12)
    x, cont, hdr = _encode_paged(host, chunk)
    assert hdr.flags & container.FLAG_PAGED
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr) == n
    assert torch.equal(back, x)
    # ... and from a copy at another address, with the header read from the blob (what arrives over a wire)
    blob = cont[:hdr.container_len].clone()
    back.zero_()
    assert container.decode_device(blob.data_ptr(), hdr.container_len, back.data_ptr(), n) == n
    assert torch.equal(back, x)


def test_a_lying_page_directory_is_a_format_error():
    import torch
    from density_amd import container, _lib
    from density_amd.codec import DecodeError
    n, chunk = 16 << 20, 2 << 20
    host = datagen.rep_text(n)
    x, cont, hdr = _encode_paged(host, chunk)
    blob = cont[:hdr.container_len].cpu().numpy().copy()
    off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
    off = (off + (n + 255) // 256 + 15) // 16 * 16
    ppc = int(_lib.lib().density_hip_paged_pages_per_chunk(chunk))
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    def entry(c, k): return off + 16 * (ppc + 1) * c + 16 * (k + 1)
    for what, at, value in [("page beyond the container", entry(1, 1), 0x7fff), ("first block off the grid", entry(2, 1) + 4, 24), ("bytes used too many", entry(3, 0) + 8, 65536 + 2),
                            ("bytes used shifted", entry(0, 0) + 8, None), ("no pages", off + 16 * (ppc + 1) * 4, 0), ("too many pages", off + 16 * (ppc + 1) * 5, 200)]:
        bad = blob.copy()
        v = int.from_bytes(bad[at:at + 4].tobytes(), "little") - 2 if value is None else value
        bad[at:at + 4] = np.frombuffer(int(v).to_bytes(4, "little"), dtype=np.uint8)
        d = torch.from_numpy(bad).cuda()
        with pytest.raises(DecodeError):
            container.decode_device(d.data_ptr(), bad.size, back.data_ptr(), n)


@pytest.mark.parametrize("kind,n,chunk,shuffle", [("mixed", (6 << 20) + 4321, 1 << 20, False), ("rep-text", 16 << 20, 4 << 20, True), ("random", (4 << 20) + 256, 2 << 20, True)])
def test_gpu_decodes_a_cpu_built_paged_container(kind, n, chunk, shuffle):
    """A paged container assembled on the CPU from the oracle's streams by the header's layout alone (tests/paged_cpu.py), its pages in chunk order or shuffled:
    the GPU decoder takes it like one of its own."""
    import torch
    import paged_cpu
    from density_amd import container
    data = datagen.rep_text(n) if kind == "rep-text" else datagen.by_kind(kind, n, seed=7)
    blob = paged_cpu.build(data, chunk)
    if shuffle:
        from test_paged_cpu_reader import _dir_heads
        hdr, _ = container.chunk_payloads(blob)
        total = sum(int.from_bytes(bytes(blob[d:d + 4]), "little") for d in _dir_heads(blob, hdr, chunk))   # pages in use
        blob = paged_cpu.build(data, chunk, page_order=list(np.random.default_rng(3).permutation(total)))
    d = torch.from_numpy(blob).cuda()
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert container.decode_device(d.data_ptr(), blob.size, back.data_ptr(), n) == n
    assert np.array_equal(back.cpu().numpy(), data)



def _directory(blob, hdr):
    """[(page, first block, bytes used), ...] per chunk, and the offset of the directory."""
    from density_amd import _lib
    off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
    off = (off + (hdr.total_len + 255) // 256 + 15) // 16 * 16
    ppc = int(_lib.lib().density_hip_paged_pages_per_chunk(hdr.chunk_size))
    b = bytes(blob[:off + 16 * (ppc + 1) * hdr.n_chunks])                          # (the front matter only)
    out = []
    for i in range(hdr.n_chunks):
        d = off + 16 * (ppc + 1) * i
        k = int.from_bytes(b[d:d + 4], "little")
        out.append([tuple(int.from_bytes(b[d + 16 * (j + 1) + 4 * f:d + 16 * (j + 1) + 4 * f + 4], "little") for f in range(3)) for j in range(k)])
        tail = b[d + 16 * (k + 1):d + 16 * (ppc + 1)]
        assert tail == bytes(len(tail)), f"chunk {i}: directory entries behind the last page are part of the wire bytes: zeros"
    return off, ppc, out


def test_config2_full_size_paged_container_is_the_oracles_streams():
    """The HEADLINE form at the headline size (what bench.py times): 1 GiB of rep-text through density_hip_encode_device_paged — 256 work-groups racing
    for one page counter, ~10,000 pages.  Every chunk, reassembled on the CPU the way a CPU reader does (directory -> used bytes of the pages), equals the
    oracle's stream of that chunk (codec.rs:72-80 per chunk); the directory is sound (every page used once, first blocks on multiples of 16, the bytes
    used add up to the size table); the pages decode in place; and a SECOND encode gives the same chunk streams although its page order is its own."""
    import os
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from density_amd import container
    from oracle import pyoracle
    n, chunk = 1 << 30, 4 << 20
    host = datagen.rep_text(n)
    x, cont, hdr = _encode_paged(host, chunk)
    assert hdr.flags & container.FLAG_PAGED and (hdr.n_chunks, hdr.total_len, hdr.chunk_size) == (256, n, chunk)
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                                # (a null stream argument = the library's own stream: not ordered behind torch's memset)
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr) == n
    assert torch.equal(back, x)
    blob = cont[:hdr.container_len].cpu().numpy()
    h, streams = container.chunk_payloads(blob)

    def check(i):
        return streams[i] == pyoracle.encode("chameleon", host[i * chunk:(i + 1) * chunk])
    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        ok = list(ex.map(check, range(hdr.n_chunks)))
    assert all(ok), [i for i, v in enumerate(ok) if not v][:8]
    off, ppc, pages = _directory(blob, hdr)
    sizes = [int.from_bytes(bytes(blob[32 + 4 * i:36 + 4 * i]), "little") for i in range(hdr.n_chunks)]
    seen = set()
    for i, pp in enumerate(pages):
        assert 1 <= len(pp) <= ppc and pp[0][1] == 0
        assert sum(u for _, _, u in pp) == sizes[i] == len(streams[i])
        for k, (page, first, used) in enumerate(pp):
            assert first % 16 == 0 and used <= 65536 and page not in seen and (k == 0 or first > pp[k - 1][1])
            seen.add(page)
    pages_base = (off + 16 * (ppc + 1) * hdr.n_chunks + 255) // 256 * 256
    n_pages = (hdr.container_len - pages_base) // 65536
    assert (hdr.container_len - pages_base) % 65536 == 0 and max(seen) < n_pages
    # pages the counter handed out and nobody wrote to (spares): never more than one per chunk — and none for a chunk whose last page was, at the bytes
    # per block it had had, likely to hold the rest (rep-text's chunks are all alike and end with a nearly full page: most keep their spare)
    assert n_pages - len(seen) <= hdr.n_chunks, (n_pages, len(seen))
    print("pages", n_pages, "used", len(seen))
    order1 = [p for pp in pages for p, _, _ in pp]
    del blob, streams
    # again: the same streams, whatever the pages' order
    x2, cont2, hdr2 = _encode_paged(host, chunk)
    back.zero_(); torch.cuda.synchronize()
    assert container.decode_device(cont2.data_ptr(), hdr2.container_len, back.data_ptr(), n, header=hdr2) == n
    assert torch.equal(back, x)
    blob2 = cont2[:hdr2.container_len].cpu().numpy()
    _, streams2 = container.chunk_payloads(blob2)
    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        ok = list(ex.map(lambda i: streams2[i] == pyoracle.encode("chameleon", host[i * chunk:(i + 1) * chunk]), range(hdr2.n_chunks)))
    assert all(ok)
    _, _, pages2 = _directory(blob2, hdr2)
    assert [[(f, u) for _, f, u in pp] for pp in pages2] == [[(f, u) for _, f, u in pp] for pp in pages]     # page boundaries are the stream's, not the run's
    print("page order identical between the two runs:", order1 == [p for pp in pages2 for p, _, _ in pp])


def test_a_directory_shorter_than_its_stream_is_a_format_error():
    """the last page's `used` must end where the size table says the stream ends — a crafted container with ONE page, a full chunk's
    index and an inflated size would otherwise map stream positions past the page (and past the container)."""
    import torch
    from density_amd import container, _lib
    from density_amd.codec import DecodeError
    n, chunk = 8 << 20, 2 << 20
    host = datagen.rep_text(n)
    x, cont, hdr = _encode_paged(host, chunk)
    blob = cont[:hdr.container_len].cpu().numpy().copy()
    off, ppc, pages = _directory(blob, hdr)
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                                # (a null stream argument = the library's own stream: not ordered behind torch's memset)
    pages_base = (off + 16 * (ppc + 1) * hdr.n_chunks + 255) // 256 * 256
    last_page = (hdr.container_len - pages_base) // 65536 - 1

    def put(b, at, v): b[at:at + 4] = np.frombuffer(int(v).to_bytes(4, "little"), dtype=np.uint8)
    cases = {}
    # (a) the directory truncated to its first page, which is moved to the container's last page; the size table untouched
    bad = blob.copy(); d = off + 16 * (ppc + 1) * 1
    put(bad, d, 1); put(bad, d + 16, last_page); cases["one page, full size"] = bad
    # (b) the size table inflated over a true directory
    bad = blob.copy(); put(bad, 32 + 4 * 2, int(chunk * 1.03)); cases["inflated size"] = bad
    # (c) the size table deflated
    bad = blob.copy(); put(bad, 32 + 4 * 3, int.from_bytes(bytes(blob[32 + 12:32 + 16]), "little") - 264); cases["deflated size"] = bad
    # (d) the last page's used bytes grown
    bad = blob.copy(); e = off + 16 * (ppc + 1) * 0 + 16 * len(pages[0]); put(bad, e + 8, pages[0][-1][2] + 136); cases["last page grown"] = bad
    for what, bad in cases.items():
        d = torch.from_numpy(bad).cuda()
        with pytest.raises(DecodeError):
            container.decode_device(d.data_ptr(), bad.size, back.data_ptr(), n)
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n) == n and torch.equal(back, x)   # (the library is fine afterwards)


def test_host_pointer_decode_of_a_large_paged_blob():
    """density_hip_decode() on host pointers takes a wire-ready paged blob of 32 MiB and more (it used to enter the pipelined path,
    whose slices follow the PACKED layout)."""
    from density_amd import container
    n, chunk = 48 << 20, 4 << 20
    host = datagen.rep_text(n)
    x, cont, hdr = _encode_paged(host, chunk)
    assert hdr.flags & container.FLAG_PAGED and hdr.n_chunks >= 4
    blob = cont[:hdr.container_len].cpu().numpy()
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(blob, back) == n
    assert np.array_equal(back, host)


def test_pack_device_refuses_a_paged_container_and_large_chunks_come_out_slotted():
    """ADVICE r5 (medium x2): density_hip_pack_device is for slotted containers — a paged one is wire-ready and not laid out the way the pack reads —;
    and the paged encoder offers only what the paged decoder takes (chunks of at most 4 MiB): an 8 MiB chunk comes out slotted and decodes."""
    import torch
    from density_amd import container
    from density_amd.codec import EncodeError
    n, chunk = 16 << 20, 2 << 20
    host = datagen.rep_text(n)
    x, cont, hdr = _encode_paged(host, chunk)
    out = torch.empty(container.container_bound("chameleon", n, chunk), dtype=torch.uint8, device="cuda")
    with pytest.raises(EncodeError):
        container.pack_device(cont.data_ptr(), hdr.container_len, out.data_ptr(), out.numel(), header=hdr)
    n, chunk = 32 << 20, 8 << 20
    host = datagen.rep_text(n)
    x, cont, hdr = _encode_paged(host, chunk)
    assert not (hdr.flags & container.FLAG_PAGED) and (hdr.flags & container.FLAG_SLOTTED)
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                                # (a null stream argument = the library's own stream: not ordered behind torch's memset)
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr) == n
    assert torch.equal(back, x)


def test_multi_rank_container_of_paged_blobs_decodes_on_one_device():
    """Config 5's wire form (include/density_hip.h "DHCM"): two ranks' PAGED blobs behind the super-header, decoded rank by rank on one device — each blob in
    place at its offset of the super-container."""
    import torch
    from density_amd import container, parallel
    chunk, parts = 2 << 20, [datagen.rep_text(16 << 20), datagen.by_kind("mixed", (6 << 20) + 999, seed=5)]
    blobs = []
    for host in parts:
        x, cont, hdr = _encode_paged(host, chunk)
        assert hdr.flags & container.FLAG_PAGED
        blobs.append(cont[:hdr.container_len].clone())
    front, rows, total = parallel.multi_layout([b.numel() for b in blobs], [p.size for p in parts], 0, chunk)
    sup = torch.zeros(total, dtype=torch.uint8, device="cuda")
    sup[:len(front)] = torch.frombuffer(bytearray(front), dtype=torch.uint8).cuda()
    for (off, ln, _), b in zip(rows, blobs):
        sup[off:off + ln] = b
    out = torch.zeros(sum(p.size for p in parts), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert parallel.decode_multi_device(sup, out) == out.numel()
    assert np.array_equal(out.cpu().numpy(), np.concatenate(parts))
    bad = sup.clone(); bad[32 + 24] ^= 0x10                                         # rank 1's offset moved: the front matter no longer adds up
    with pytest.raises(ValueError):
        parallel.parse_multi(bad)
