"""where the pass decoder's output first differs from the input (tools; not part of the product)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
from density_amd import container
from test_gpu_decode_passes import cpu_container, make
kind, n, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
data = make(kind, n, seed=n % 1000 + 3)
raw, streams = cpu_container(data, chunk)
out = np.zeros(n, dtype=np.uint8)
m = container.decode(raw, out)
bad = np.flatnonzero(out != data)
print("decoded", m, "mismatching bytes", bad.size)
if bad.size:
    q = np.unique(bad // 4)
    print("first bad quads", q[:20], "chunk-relative", (q[:20] * 4) % chunk // 4, "bad quads", q.size)
    import collections
    print("bad quads per chunk", collections.Counter((q * 4 // chunk).tolist()))
    print("lane of bad quads (mod 64) histogram", np.bincount(q % 64, minlength=64))
