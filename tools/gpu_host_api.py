"""host-pointer API rates (PCIe inclusive): python tools/gpu_host_api.py — the bench's 64 MiB sample, and the container calls at several sizes with the
pipelined path on (default) and off (kernel variant 512)"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import datagen, bench
from density_amd import container
host = datagen.rep_text(64 << 20)
print(json.dumps(bench.host_api_rates("chameleon", host, 4 << 20, 64 << 20)["container"], indent=1))
for mib, chunk in ((16, 0), (64, 0), (64, 4 << 20), (256, 0), (256, 4 << 20), (1024, 4 << 20)):
    n = mib << 20
    data = datagen.rep_text(n)
    cont = np.zeros(container.container_bound("chameleon", n, chunk), dtype=np.uint8)
    back = np.zeros(n, dtype=np.uint8)
    for variant in (0, 512):
        container.set_kernel_variant(variant)
        cn = container.encode("chameleon", data, cont, chunk); container.decode(cont[:cn], back)
        te, td = [], []
        for _ in range(5):
            t0 = time.perf_counter(); cn = container.encode("chameleon", data, cont, chunk); t1 = time.perf_counter()
            m = container.decode(cont[:cn], back); t2 = time.perf_counter()
            te.append(t1 - t0); td.append(t2 - t1)
        assert m == n and np.array_equal(back, data)
        e, d = sorted(te)[2], sorted(td)[2]
        print(f"{mib:5d} MiB, chunk {(chunk >> 10) or 'auto'} KiB, {'pipelined' if variant == 0 else 'staged   '}: encode {n / e / 1e9:6.1f} GB/s ({e * 1e3:.2f} ms), decode {n / d / 1e9:6.1f} GB/s ({d * 1e3:.2f} ms)", flush=True)
container.set_kernel_variant(0)
