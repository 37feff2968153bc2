"""How many zero-entry look-ups (MAP quads whose STORED dictionary entry is 0 outside slot 0: density_amd/csrc/chameleon_dev.hpp) a salt
multiplier costs on text samples: events per buffer and distinct quads behind them.  The multiplier is internal to the GPU table (streams do
not depend on it); this script is how kSaltMul was chosen.   python tools/salt_candidates.py [hex multipliers ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import datagen
M = 0x9D6EF916
def events(data, K):
    q = data[:data.size // 4 * 4].view("<u4").astype(np.uint64)
    P = (q * M) & 0xffffffff
    h = P >> 16
    z = ((((P & 0xfffe) | (q >> 31)) ^ ((h * K) & 0xffff)) == 0) & (h != 0)
    return int(z.sum()), len(np.unique(q[z]))
bufs = {"rep-text chunk 0": datagen.rep_text(4 << 20), "rep-text chunk 1": datagen.rep_text(8 << 20)[4 << 20:],
        "prose (configs 3/4 seed)": datagen.prose(4 << 20, seed=0xD1B54A32D192ED03), "prose (seed 12345)": datagen.prose(4 << 20, seed=12345)}
cands = [int(a, 16) for a in sys.argv[1:]] or [0x9e5b, 0x9e3b, 0x7f4b, 0xb5ad, 0x6a09, 0xbb67, 0x3c6f, 0xa54f, 0x510f, 0x9b05, 0x1f83, 0x5be1, 0xc2b3, 0x2545, 0x85eb, 0xca6b]
for K in cands:
    print(hex(K), {k: events(v, K) for k, v in bufs.items()})
