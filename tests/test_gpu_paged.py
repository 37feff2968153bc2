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
