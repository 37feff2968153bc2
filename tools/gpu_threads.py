"""Thread safety of the host-pointer entry points (the header says: per-device mutex + internal stream): 6 threads, mixed calls, every result checked."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
from density_amd import BY_NAME, container
from oracle import pyoracle
errors = []
def worker(tid):
    rng = np.random.default_rng(tid)
    try:
        for it in range(25):
            algo = ["chameleon", "cheetah", "lion"][int(rng.integers(0, 3))]
            n = int(rng.choice([3000, 70_001, 400_000, 1_200_000]))
            data = datagen.by_kind(["prose", "mixed", "random", "zeros"][int(rng.integers(0, 4))], n, seed=int(rng.integers(1, 1 << 20)))
            if it % 2:
                C = BY_NAME[algo]
                so = np.zeros(C.safe_encode_buffer_size(n), dtype=np.uint8)
                sn = C.encode(data, so)
                assert so[:sn].tobytes() == pyoracle.encode(algo, data), ("stream", algo, n)
                back = np.zeros(n, dtype=np.uint8)
                assert C.decode(so[:sn], back) == n and np.array_equal(back, data)
            else:
                chunk = int(rng.choice([4096, 65536, 262144]))
                cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
                cn = container.encode(algo, data, cont, chunk)
                hdr, payloads = container.chunk_payloads(cont[:cn])
                assert all(p == pyoracle.encode(algo, data[i * chunk:(i + 1) * chunk]) for i, p in enumerate(payloads)), ("container", algo, n, chunk)
                back = np.zeros(n, dtype=np.uint8)
                assert container.decode(cont[:cn], back) == n and np.array_equal(back, data)
    except Exception as ex:
        errors.append((tid, repr(ex)))
ts = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
[t.start() for t in ts]; [t.join() for t in ts]
print("errors:", errors)
sys.exit(1 if errors else 0)
