"""Fuzz: corrupted Cheetah containers must end the same way in the decode passes and in the one-wave decoder (kernel variant 128), and corrupted
Chameleon containers (with block index) the same way on the rotation kernels and on the role pipelines (variant 4).
python tools/gpu_fuzz_passes.py [trials per kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_decode_passes as T
from density_amd import container, DecodeError
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad_total = 0
for kind in ("mixed", "prose", "patchy", "pairs"):
    n, chunk = 4 * 131072 + 555, 131072
    data = T.make(kind, n, seed=11)
    raw, streams = T.cpu_container(data, chunk)
    base = (32 + 4 * len(streams) + 15) // 16 * 16
    rng = np.random.default_rng(hash(kind) & 0xffff)
    div = 0
    for t in range(trials):
        bad = raw.copy()
        mode = t % 4
        if mode == 0:
            at = base + int(rng.integers(0, len(raw) - base)); bad[at] ^= int(rng.integers(1, 256))
        elif mode == 1:                                               # a burst of 8 random bytes
            at = base + int(rng.integers(0, len(raw) - base - 8)); bad[at:at + 8] = rng.integers(0, 256, size=8, dtype=np.uint8)
        elif mode == 2:                                               # a size-table entry changed
            k = int(rng.integers(0, len(streams))); sz = len(streams[k]); new = max(0, sz + int(rng.integers(-300, 300)))
            bad[32 + 4 * k:36 + 4 * k] = np.frombuffer(int(new).to_bytes(4, "little"), dtype=np.uint8)
        else:                                                         # a signature bit flipped near a record start
            at = base + int(rng.integers(0, (len(raw) - base) // 2)) & ~1; bad[at] ^= 1 << int(rng.integers(0, 8))
        a, b = T.decode_both(bad, n)
        if a[0] != b[0] or (a[0] == "ok" and a[1] != b[1]):
            div += 1
            print(f"DIVERGENCE {kind} trial {t} mode {mode}: passes {a[0]} / one wave {b[0]}" + (f", first differing byte {next(i for i in range(min(len(a[1]), len(b[1]))) if a[1][i] != b[1][i]) if a[0] == 'ok' and len(a[1]) == len(b[1]) else ''}" if a[0] == b[0] else ""), flush=True)
    print(f"cheetah {kind}: {trials} corruptions, {div} divergences", flush=True)
    bad_total += div

# Chameleon: containers made by the GPU encoder (with their block index), payload bytes corrupted; where the rotation decoder accepts the container
# (the index still describes the streams) its output must be the oracle's decode of the corrupted chunk streams
from oracle import pyoracle
for kind in ("mixed", "prose", "lowzero", "saltzero", "zeros"):
    n, chunk = 3 * 262144 + 999, 262144
    data = T.datagen.by_kind(kind, n, seed=21)
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
    div = acc = 0
    for t in range(trials):
        bad = raw.copy()
        k = int(rng.integers(0, len(payloads)))
        at = offs[k] + int(rng.integers(0, len(payloads[k])))
        if t % 2: bad[at] ^= int(rng.integers(1, 256))
        else: bad[at:at + 4] = rng.integers(0, 256, size=min(4, len(bad) - at), dtype=np.uint8)
        out = np.zeros(n, dtype=np.uint8)
        try:
            m = container.decode(bad, out)
        except DecodeError:
            continue
        acc += 1
        want = b"".join(pyoracle.decode("chameleon", bytes(bad[offs[i]:offs[i] + len(payloads[i])]), min(chunk, n - i * chunk)) for i in range(len(payloads)))
        if out[:m].tobytes() != want:
            div += 1
            print(f"DIVERGENCE chameleon {kind} trial {t}: corrupt at chunk {k} +{at - offs[k]}", flush=True)
    print(f"chameleon {kind}: {trials} corruptions, {acc} accepted, {div} divergences from the oracle", flush=True)
    bad_total += div
sys.exit(1 if bad_total else 0)
