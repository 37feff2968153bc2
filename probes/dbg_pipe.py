import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, datagen
from density_amd import Chameleon, container
from oracle import pyoracle
for n in (256, 512, 1024, 2048):
    data = datagen.prose(5000, 21)[:n].copy()
    want = pyoracle.encode('chameleon', data)
    res = {}
    for v in (1, 0):
        container.set_kernel_variant(v)
        out = np.zeros(Chameleon.safe_encode_buffer_size(n), dtype=np.uint8)
        m = Chameleon.encode(data, out)
        res[v] = out[:m].tobytes()
    print(n, 'len want/simple/pipe', len(want), len(res[1]), len(res[0]), 'simple ok', res[1] == want, 'pipe ok', res[0] == want)
    g = res[0]
    diffs = [i for i in range(min(len(g), len(want))) if g[i] != want[i]]
    print(' first diffs at', diffs[:20], 'count', len(diffs))
    if diffs:
        i = max(diffs[0] - 8, 0)
        print(' want', want[i:i+48].hex())
        print(' got ', g[i:i+48].hex())
        print(' sig want', want[:8].hex(), 'got', g[:8].hex())
