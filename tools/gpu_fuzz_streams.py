"""Fuzz: corrupted reference streams through the nine symbols' decoders (one-wave, passes, sequential and segmented Chameleon) against the
oracle's decode of the same bytes: the same output, or an error where the oracle returns nothing.
python tools/gpu_fuzz_streams.py [trials]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
from density_amd import BY_NAME, DecodeError, container
from oracle import pyoracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
total_div = 0
cases = [("chameleon", "mixed", 70_000), ("chameleon", "lowzero", 300_000), ("chameleon", "rep", 6 << 20), ("chameleon", "mixed", 5 << 20),
         ("cheetah", "mixed", 50_000), ("cheetah", "prose", 200_000), ("cheetah", "samehash", 150_000),
         ("lion", "mixed", 50_000), ("lion", "prose", 120_000), ("lion", "binaryish", 90_000)]
for algo, kind, n in cases:
    data = datagen.by_kind(kind, n, seed=5)
    enc = np.frombuffer(pyoracle.encode(algo, data), dtype=np.uint8)
    C = BY_NAME[algo]
    rng = np.random.default_rng(len(kind) * 1000 + n % 977)
    div = errs = 0
    out = np.zeros(n, dtype=np.uint8)
    for t in range(trials if n < (1 << 20) else max(8, trials // 4)):
        bad = enc.copy()
        mode = t % 4
        at = int(rng.integers(0, len(bad) - 8))
        if mode == 0: bad[at] ^= int(rng.integers(1, 256))
        elif mode == 1: bad[at:at + 8] = rng.integers(0, 256, size=8, dtype=np.uint8)
        elif mode == 2: bad = bad[:len(bad) - int(rng.integers(1, 400))].copy()
        else: bad[at & ~1] ^= 1 << int(rng.integers(0, 8))
        want = pyoracle.decode(algo, bytes(bad), n)
        try:
            m = C.decode(bad, out); got = out[:m].tobytes()
        except DecodeError:
            got = b""; errs += 1
        if got != want:
            div += 1
            if div <= 3:
                first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
                print(f"DIVERGENCE {algo} {kind} {n}: trial {t} mode {mode} at {at}: gpu {len(got)} bytes, oracle {len(want)} bytes, first differing byte {first}", flush=True)
    print(f"{algo} {kind} {n}: {div} divergences ({errs} errors)", flush=True)
    total_div += div
# containers without a block index: the record-walking decoders (rotor-noindex = variant 2, pipelines = 6, one wavefront = 1)
import test_gpu_decode_passes as T
for variant in (2, 6, 1):
    container.set_kernel_variant(variant)
    n, chunk = 4 * 65536 + 321, 65536
    data = datagen.by_kind("mixed", n, seed=8)
    raw, streams = T.cpu_container(data, chunk, algo="chameleon", algo_id=0)
    base = (32 + 4 * len(streams) + 15) // 16 * 16
    offs, o = [], base
    for s in streams:
        offs.append(o); o = (o + len(s) + 15) // 16 * 16
    rng = np.random.default_rng(variant)
    div = 0
    for t in range(trials):
        bad = raw.copy()
        at = base + int(rng.integers(0, len(raw) - base - 8))
        if t % 2: bad[at] ^= int(rng.integers(1, 256))
        else: bad[at:at + 8] = rng.integers(0, 256, size=8, dtype=np.uint8)
        want_parts = [pyoracle.decode("chameleon", bytes(bad[offs[k]:offs[k] + len(s)]), min(chunk, n - k * chunk)) for k, s in enumerate(streams)]
        whole = all(len(w) == min(chunk, n - k * chunk) for k, w in enumerate(want_parts))
        out = np.zeros(n, dtype=np.uint8)
        try:
            m = container.decode(bad, out); got = out[:m].tobytes()
        except DecodeError:
            got = None
        if whole and got != b"".join(want_parts):
            div += 1; print(f"DIVERGENCE container without index, variant {variant}, trial {t}: gpu {'error' if got is None else len(got)}", flush=True)
        if not whole and got is not None:
            div += 1; print(f"DIVERGENCE container without index, variant {variant}, trial {t}: a chunk the oracle cannot decode was accepted", flush=True)
    print(f"chameleon container without index, variant {variant}: {div} divergences", flush=True)
    total_div += div
container.set_kernel_variant(0)
sys.exit(1 if total_div else 0)
