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

