"""Pipelined encoder vs oracle on a range of sizes and data kinds; prints the first mismatch (debug driver)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, datagen
from density_amd import Chameleon, container
from oracle import pyoracle
container.set_kernel_variant(0)
sizes = [256, 2048, 2048 + 256, 4096, 8192, 16384, 16384 + 512, 20480, 65536, 300000]
for kind in ("prose", "mixed", "random"):
    for n in sizes:
        data = datagen.by_kind(kind, n, 7) if hasattr(datagen, "by_kind") else datagen.prose(n, 7)
        data = np.ascontiguousarray(data[:n])
        want = pyoracle.encode('chameleon', data)
        out = np.zeros(Chameleon.safe_encode_buffer_size(n), dtype=np.uint8)
        m = Chameleon.encode(data, out)
        g = out[:m].tobytes()
        ok = g == want
        msg = ""
        if not ok:
            d = next((i for i in range(min(len(g), len(want))) if g[i] != want[i]), min(len(g), len(want)))
            # which record does the first difference fall in? walk the oracle stream (no copy mode assumed for prose)
            msg = " first diff at byte %d (len want %d got %d)" % (d, len(want), len(g))
        print(kind, n, "ok" if ok else "MISMATCH" + msg, flush=True)
