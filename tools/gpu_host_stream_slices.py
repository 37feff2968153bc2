"""How many slices the pipelined reference symbols should move a long stream in (DENSITY_HIP_{ENCODE,DECODE}_SLICES): 64 MiB and 256 MiB of text,
median of 5 warm calls each, torch initialised first (bench.py's process is one with torch in it)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import torch
torch.zeros(4).cuda()
import numpy as np
import datagen
from density_amd import Chameleon

for mib in (64, 256):
    data = datagen.rep_text(mib << 20, period=1_000_003)
    out = np.zeros(Chameleon.safe_encode_buffer_size(data.size), dtype=np.uint8)
    back = np.zeros(data.size, dtype=np.uint8)
    m = Chameleon.encode(data, out); Chameleon.decode(out[:m], back)
    for name, ks in (("DENSITY_HIP_ENCODE_SLICES", (6, 8, 12, 16, 24)), ("DENSITY_HIP_DECODE_SLICES", (3, 4, 5, 6, 8))):
        for k in ks:
            os.environ[name] = str(k)
            te, td = [], []
            for _ in range(5):
                t0 = time.perf_counter(); m = Chameleon.encode(data, out); t1 = time.perf_counter(); Chameleon.decode(out[:m], back); t2 = time.perf_counter()
                te.append(t1 - t0); td.append(t2 - t1)
            print(f"{mib} MiB {name}={k}: encode {data.size / sorted(te)[2] / 1e9:.1f} GB/s, decode {data.size / sorted(td)[2] / 1e9:.1f} GB/s", flush=True)
        os.environ.pop(name, None)
    assert np.array_equal(back, data)
