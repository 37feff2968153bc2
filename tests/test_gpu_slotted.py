"""The slotted container (DENSITY_HIP_FLAG_SLOTTED: every chunk stream left in its worst-case slot, no stitch pass) against the packed
one: the chunk streams are the oracle's, decode gives the input back, and density_hip_pack_device turns it into byte for byte what
density_hip_encode_device writes."""
import numpy as np
import pytest

import datagen
from density_amd import container
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo", ["chameleon", "cheetah", "lion"])
@pytest.mark.parametrize("kind,n,chunk", [("prose", 3 * (1 << 20) + 12345, 1 << 18), ("mixed", 40 * 4096 + 77, 4096), ("random", 900_001, 65536),
                                          ("rep", 8 << 20, 1 << 20), ("prose", 70_000, 1 << 20)])
def test_slotted_equals_packed(algo, kind, n, chunk):
    import torch
    host = datagen.by_kind(kind, n, seed=chunk % 97)
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound_slotted(algo, n, chunk)
    assert cap >= container.container_bound(algo, n, chunk)
    slotted = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    packed = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    repacked = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    hs = container.encode_device_slotted(algo, x.data_ptr(), n, slotted.data_ptr(), cap, chunk)
    hp = container.encode_device(algo, x.data_ptr(), n, packed.data_ptr(), cap, chunk)
    nch = -(-n // chunk)
    assert hs.n_chunks == hp.n_chunks == nch and hs.total_len == n
    assert bool(hs.flags & container.FLAG_SLOTTED) == (nch > 1) and not (hp.flags & container.FLAG_SLOTTED)
    assert hs.container_len <= cap and hp.container_len <= hs.container_len
    # the chunk streams in their slots are the oracle's streams
    _, pay_s = container.chunk_payloads(slotted[:hs.container_len].cpu().numpy())
    _, pay_p = container.chunk_payloads(packed[:hp.container_len].cpu().numpy())
    assert pay_s == pay_p
    for i in sorted(set([0, nch // 2, nch - 1])):
        assert pay_s[i] == pyoracle.encode(algo, host[i * chunk:(i + 1) * chunk]), (algo, i)
    # decode straight from the slots
    assert container.decode_device(slotted.data_ptr(), hs.container_len, back.data_ptr(), n, header=hs) == n
    assert torch.equal(back, x)
    # pack: byte for byte the packed container
    hr = container.pack_device(slotted.data_ptr(), hs.container_len, repacked.data_ptr(), cap, header=hs)
    assert (hr.container_len, hr.flags, hr.n_chunks) == (hp.container_len, hp.flags, hp.n_chunks)
    assert torch.equal(repacked[:hr.container_len], packed[:hp.container_len])
    # packing a packed container is a copy
    again = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    ha = container.pack_device(packed.data_ptr(), hp.container_len, again.data_ptr(), cap, header=hp)
    assert ha.container_len == hp.container_len and torch.equal(again[:ha.container_len], packed[:hp.container_len])


def test_slotted_container_with_a_lying_size_table_is_a_format_error():
    import torch
    from density_amd import DecodeError
    n, chunk = 600_000, 65536
    host = datagen.prose(n, seed=4)
    x = torch.from_numpy(host).cuda()
    cap = container.container_bound_slotted("chameleon", n, chunk)
    cont = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    h = container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    bad = cont.clone()
    bad[32 + 4 * 3:36 + 4 * 3] = torch.tensor(list(int(0x7FFFFFF0).to_bytes(4, "little")), dtype=torch.uint8, device="cuda")
    with pytest.raises(DecodeError):
        container.decode_device(bad.data_ptr(), h.container_len, back.data_ptr(), n, header=h)
