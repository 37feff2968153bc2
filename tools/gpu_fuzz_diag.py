"""Which decoder is right on a corrupted Cheetah chunk stream?  Compares the decode passes and the one-wave decoder with the oracle's decode of the same bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_decode_passes as T
from oracle import pyoracle
kind = "pairs"
n, chunk = 4 * 131072 + 555, 131072
data = T.make(kind, n, seed=11)
raw, streams = T.cpu_container(data, chunk)
base = (32 + 4 * len(streams) + 15) // 16 * 16
offs = []
o = base
for s in streams:
    offs.append(o); o = (o + len(s) + 15) // 16 * 16
rng = np.random.default_rng(hash(kind) & 0xffff)
shown = 0
for t in range(80):
    bad = raw.copy()
    mode = t % 4
    if mode == 0:
        at = base + int(rng.integers(0, len(raw) - base)); bad[at] ^= int(rng.integers(1, 256))
    elif mode == 1:
        at = base + int(rng.integers(0, len(raw) - base - 8)); bad[at:at + 8] = rng.integers(0, 256, size=8, dtype=np.uint8)
    elif mode == 2:
        k = int(rng.integers(0, len(streams))); sz = len(streams[k]); new = max(0, sz + int(rng.integers(-300, 300)))
        bad[32 + 4 * k:36 + 4 * k] = np.frombuffer(int(new).to_bytes(4, "little"), dtype=np.uint8)
    else:
        at = base + int(rng.integers(0, (len(raw) - base) // 2)) & ~1; bad[at] ^= 1 << int(rng.integers(0, 8))
    a, b = T.decode_both(bad, n)
    if a[0] == "ok" and b[0] == "ok" and a[1] != b[1] and mode != 2:
        # the oracle on every chunk stream of the corrupted container
        want = b""
        for k, s in enumerate(streams):
            want_len = min(chunk, n - k * chunk)
            want += pyoracle.decode(T.ALGO, bytes(bad[offs[k]:offs[k] + len(s)]), want_len)
        first = next(i for i in range(min(len(a[1]), len(b[1]))) if a[1][i] != b[1][i])
        ck = first // chunk
        print(f"trial {t} mode {mode} corrupt at {at} (chunk {max(i for i in range(len(offs)) if offs[i] <= at)} +{at - max(x for x in offs if x <= at)}): first differing byte {first} (chunk {ck} +{first % chunk}); passes == oracle: {a[1] == want}, one wave == oracle: {b[1] == want}; oracle len {len(want)}")
        i = first & ~3
        print("   passes  ", a[1][i:i + 16].hex(), "\n   one wave", b[1][i:i + 16].hex(), "\n   oracle  ", want[i:i + 16].hex())
        shown += 1
        if shown >= 5: break
