"""Fuzz: random patchworks of data kinds, sizes and chunk sizes through the container encoders (all three algorithms) and the Chameleon stream
encoder: every chunk stream == the oracle's, decode == input.   python tools/gpu_fuzz_encode.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
from density_amd import BY_NAME, container
from oracle import pyoracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ["prose", "zeros", "random", "rep", "samehash", "lowzero", "saltzero", "binaryish", "mixed"]
def patchwork(n):
    parts, left = [], n
    while left > 0:
        k = KINDS[int(rng.integers(0, len(KINDS)))]
        m = min(left, int(rng.choice([64, 300, 4096, 20_000, 70_000, 300_000, 1_000_000])) + int(rng.integers(0, 257)))
        parts.append(datagen.by_kind(k, max(m, 4), seed=int(rng.integers(1, 1 << 30)))[:m])
        left -= m
    return np.concatenate(parts)
bad = 0
for t in range(trials):
    algo = ["chameleon", "chameleon", "cheetah", "lion"][t % 4]
    n = int(rng.choice([1, 5, 255, 256, 257, 4095, 70_001, 300_000, 1_048_576, 2_500_003, 6_000_000 if algo == "chameleon" else 1_500_000]))
    chunk = int(rng.choice([256, 4096, 65536, 262144, 1 << 20, 4 << 20]))
    if algo != "chameleon" and n // chunk > 400: chunk = 65536
    data = patchwork(n)
    cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
    cn = container.encode(algo, data, cont, chunk)
    hdr, payloads = container.chunk_payloads(cont[:cn])
    wrong = [i for i, p in enumerate(payloads) if p != pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk])]
    back = np.zeros(n, dtype=np.uint8)
    ok = container.decode(cont[:cn], back) == n and np.array_equal(back, data)
    # the stream symbols too
    C = BY_NAME[algo]
    so = np.zeros(C.safe_encode_buffer_size(n), dtype=np.uint8)
    sn = C.encode(data, so)
    stream_ok = so[:sn].tobytes() == pyoracle.encode(algo, data)
    sb = np.zeros(n, dtype=np.uint8)
    stream_back = C.decode(so[:sn], sb) == n and np.array_equal(sb, data)
    if wrong or not ok or not stream_ok or not stream_back:
        bad += 1
        print(f"FAIL trial {t}: {algo} n {n} chunk {chunk}: wrong chunks {wrong[:6]}, container round trip {ok}, stream == oracle {stream_ok}, stream round trip {stream_back}", flush=True)
print(f"{trials} trials, {bad} failures")
sys.exit(1 if bad else 0)
