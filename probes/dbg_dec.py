import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, datagen
from density_amd import Chameleon, container
from oracle import pyoracle
ok = True
for kind in ("prose", "mixed", "zeros", "random", "lowzero", "samehash"):
    for n in (200, 256, 264, 1000, 2048, 4099, 20000, 65536, 300001):
        data = datagen.by_kind(kind, n, seed=3)
        enc = pyoracle.encode('chameleon', data)
        out = np.zeros(n, dtype=np.uint8)
        try:
            m = Chameleon.decode(enc, out)
            good = (m == n and out.tobytes() == data.tobytes())
        except Exception as ex:
            good = False; m = str(ex)
        if not good:
            ok = False
            bad = np.nonzero(out[:n] != data)[0]
            print("FAIL", kind, n, m, "first bad", bad[:5], "count", bad.size)
print("ALL OK" if ok else "SOME FAILED")
