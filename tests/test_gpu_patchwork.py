"""Seeded patchworks of data kinds, sizes and chunk sizes through every encoder and decoder (the shape of tools/gpu_fuzz_encode.py, a dozen
cases of it): every chunk stream == the oracle's stream of that chunk, the reference symbols' streams == the oracle's, decode == input."""
import numpy as np
import pytest

import datagen
from density_amd import BY_NAME, container
from oracle import pyoracle

pytestmark = pytest.mark.gpu
KINDS = ["prose", "zeros", "random", "rep", "samehash", "lowzero", "saltzero", "binaryish", "mixed"]


def patchwork(rng, n):
    parts, left = [], n
    while left > 0:
        k = KINDS[int(rng.integers(0, len(KINDS)))]
        m = min(left, int(rng.choice([64, 300, 4096, 20_000, 70_000, 300_000, 1_000_000])) + int(rng.integers(0, 257)))
        parts.append(datagen.by_kind(k, max(m, 4), seed=int(rng.integers(1, 1 << 30)))[:m])
        left -= m
    return np.concatenate(parts)


@pytest.mark.parametrize("algo", ["chameleon", "cheetah", "lion"])
@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_patchwork(algo, seed):
    rng = np.random.default_rng(seed * 101 + len(algo))
    n = int(rng.choice([257, 4095, 70_001, 300_000, 1_048_576, 2_500_003, 5_000_000 if algo == "chameleon" else 1_500_000]))
    chunk = int(rng.choice([256, 4096, 65536, 262144, 1 << 20]))
    if algo != "chameleon" and n // chunk > 400:
        chunk = 65536
    data = patchwork(rng, n)
    cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
    cn = container.encode(algo, data, cont, chunk)
    _, payloads = container.chunk_payloads(cont[:cn])
    wrong = [i for i, p in enumerate(payloads) if p != pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk])]
    assert not wrong, (n, chunk, wrong[:8])
    back = np.zeros(n, dtype=np.uint8)
    assert container.decode(cont[:cn], back) == n and np.array_equal(back, data), (n, chunk)
    C = BY_NAME[algo]
    so = np.zeros(C.safe_encode_buffer_size(n), dtype=np.uint8)
    sn = C.encode(data, so)
    assert so[:sn].tobytes() == pyoracle.encode(algo, data), n
    sb = np.zeros(n, dtype=np.uint8)
    assert C.decode(so[:sn], sb) == n and np.array_equal(sb, data), n
