import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, datagen
from oracle import pyoracle
from density_amd import Cheetah
big = datagen.by_kind("prose", 5000, seed=31)
for n in [1, 5, 64, 127, 128, 129, 135, 136, 137, 255, 256, 257, 263, 1024, 4099]:
    data = big[:n].copy()
    want = pyoracle.encode("cheetah", data)
    out = np.zeros(Cheetah.safe_encode_buffer_size(n), np.uint8)
    m = Cheetah.encode(data, out); got = out[:m].tobytes()
    print(n, "enc", "ok" if got == want else "BAD", flush=True)
    back = np.zeros(n, np.uint8)
    k = Cheetah.decode(np.frombuffer(want, np.uint8), back)
    print(n, "dec", "ok" if back.tobytes() == data.tobytes() else "BAD", k, flush=True)
