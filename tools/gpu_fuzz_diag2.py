"""Chameleon: a corrupted container accepted by the decoder — which kernel family agrees with the oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
from density_amd import container, DecodeError
from oracle import pyoracle
kind = sys.argv[1] if len(sys.argv) > 1 else "lowzero"
trials = 100
n, chunk = 3 * 262144 + 999, 262144
data = datagen.by_kind(kind, n, seed=21)
cont = np.zeros(container.container_bound("chameleon", n, chunk), dtype=np.uint8)
cn = container.encode("chameleon", data, cont, chunk)
raw = cont[:cn].copy()
hdr, payloads = container.chunk_payloads(raw)
off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
off = (off + (hdr.total_len + 255) // 256 + 15) // 16 * 16
offs = []
for p in payloads:
    offs.append(off); off = (off + len(p) + 15) // 16 * 16
rng = np.random.default_rng(hash(kind) & 0xfff)
shown = 0
for t in range(trials):
    bad = raw.copy()
    k = int(rng.integers(0, len(payloads)))
    at = offs[k] + int(rng.integers(0, len(payloads[k])))
    if t % 2: bad[at] ^= int(rng.integers(1, 256))
    else: bad[at:at + 4] = rng.integers(0, 256, size=min(4, len(bad) - at), dtype=np.uint8)
    res = {}
    for variant in (0, 4, 1):
        container.set_kernel_variant(variant)
        out = np.zeros(n, dtype=np.uint8)
        try:
            m = container.decode(bad, out); res[variant] = out[:m].tobytes()
        except DecodeError:
            res[variant] = None
    container.set_kernel_variant(0)
    if res[0] is None: continue
    want = b"".join(pyoracle.decode("chameleon", bytes(bad[offs[i]:offs[i] + len(payloads[i])]), min(chunk, n - i * chunk)) for i in range(len(payloads)))
    if res[0] != want:
        first = next(i for i in range(min(len(res[0]), len(want))) if res[0][i] != want[i])
        ndiff = sum(1 for i in range(0, min(len(res[0]), len(want)), 4) if res[0][i:i + 4] != want[i:i + 4])
        print(f"trial {t}: corrupt at chunk {k} +{at - offs[k]} ({'4 bytes' if t % 2 == 0 else '1 byte'}); first differing byte {first} (chunk {first // chunk} +{first % chunk}), {ndiff} quads differ; "
              f"pipelines == oracle: {res[4] == want}, one-wavefront == oracle: {res[1] == want}; lens {len(res[0])} {len(want)}")
        i = first & ~3
        print("   rotor ", res[0][i:i + 12].hex(), "\n   oracle", want[i:i + 12].hex(), "\n   input ", data[i:i + 12].tobytes().hex())
        shown += 1
        if shown >= 6: break
