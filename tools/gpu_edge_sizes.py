"""Stream round trips of Cheetah / Lion at the edge sizes, printed one per line (a hang shows where): python tools/gpu_edge_sizes.py [cheetah|lion]"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, datagen
from oracle import pyoracle
from density_amd import Cheetah, Lion
ALG = {"cheetah": Cheetah, "lion": Lion}[sys.argv[1] if len(sys.argv) > 1 else "cheetah"]; NAME = sys.argv[1] if len(sys.argv) > 1 else "cheetah"
big = datagen.by_kind("prose", 5000, seed=31)
for n in [1, 5, 64, 127, 128, 129, 135, 136, 137, 255, 256, 257, 263, 1024, 4099]:
    data = big[:n].copy()
    want = pyoracle.encode(NAME, data)
    out = np.zeros(ALG.safe_encode_buffer_size(n), np.uint8)
    m = ALG.encode(data, out); got = out[:m].tobytes()
    print(n, "enc", "ok" if got == want else "BAD", flush=True)
    back = np.zeros(n, np.uint8)
    k = ALG.decode(np.frombuffer(want, np.uint8), back)
    print(n, "dec", "ok" if back.tobytes() == data.tobytes() else "BAD", k, flush=True)
