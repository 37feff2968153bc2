"""The paged container's layout as the header states it, without a GPU: a container assembled on the CPU (tests/paged_cpu.py) read back by the CPU reader
(density_amd.container.chunk_payloads) gives the oracle's chunk streams, whatever the order of the pages."""
import numpy as np
import pytest

import datagen
import paged_cpu
from density_amd import container
from oracle import pyoracle


@pytest.mark.parametrize("kind,n,chunk", [("mixed", (3 << 20) + 777, 1 << 20), ("prose", 2 << 20, 1 << 20), ("random", (2 << 20) + 256, 1 << 20), ("zeros", 2 << 20, 1 << 20)])
def test_cpu_built_paged_container_reads_back(kind, n, chunk):
    data = datagen.by_kind(kind, n, seed=5)
    blob = paged_cpu.build(data, chunk)
    hdr, streams = container.chunk_payloads(blob)
    assert hdr.flags & container.FLAG_PAGED and hdr.total_len == n and hdr.container_len == blob.size
    for i, s in enumerate(streams):
        assert s == pyoracle.encode("chameleon", data[i * chunk:(i + 1) * chunk]), i
    # ... and with the pages in another order (the GPU's depends on the run)
    total = sum(int.from_bytes(bytes(blob[d:d + 4]), "little") for d in _dir_heads(blob, hdr, chunk))
    perm = list(np.random.default_rng(1).permutation(total))
    blob2 = paged_cpu.build(data, chunk, page_order=perm)
    assert container.chunk_payloads(blob2)[1] == streams


def _dir_heads(blob, hdr, chunk):
    from density_amd import _lib
    ppc = int(_lib.lib().density_hip_paged_pages_per_chunk(chunk))
    off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
    off = (off + (hdr.total_len + 255) // 256 + 15) // 16 * 16
    return [off + 16 * (ppc + 1) * i for i in range(hdr.n_chunks)]
