"""One-shot diagnostics for a fresh GPU box: self-test bits, then stream / container parity of every kernel variant against the
oracle on a handful of inputs, reporting the first differing byte instead of stopping at the first failure."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from density_amd import Chameleon, container, _lib
from oracle import pyoracle

def first_diff(a, b):
    n = min(len(a), len(b))
    x = np.frombuffer(a[:n], np.uint8) != np.frombuffer(b[:n], np.uint8)
    return int(np.argmax(x)) if x.any() else n

print("selftest bits:", hex(_lib.lib().density_hip_selftest_bits() & 0xffffffff), "err:", _lib.last_error())
cases = [("prose", 20_000), ("prose", 300_000), ("prose", 2_100_000), ("mixed", 1_000_000), ("random", 200_000), ("zeros", 300_000),
         ("samehash", 300_000), ("saltzero", 300_000), ("rep", 3_000_000)]
for vname, v in (("rotor", 0), ("pipelined", 4), ("simple", 1)):
    container.set_kernel_variant(v)
    for kind, n in cases:
        data = datagen.by_kind(kind, n, seed=5)
        want = pyoracle.encode("chameleon", data)
        out = np.zeros(Chameleon.safe_encode_buffer_size(n), np.uint8)
        t0 = time.time()
        try:
            m = Chameleon.encode(data, out); got = out[:m].tobytes()
        except Exception as ex:
            print(vname, kind, n, "ENCODE EXC", ex); continue
        ok = got == want
        msg = "" if ok else f" len {len(got)} vs {len(want)} first diff at {first_diff(got, want)}"
        # container with index, decode
        chunk = 1 << 18
        cont = np.zeros(container.container_bound("chameleon", n, chunk), np.uint8)
        try:
            cn = container.encode("chameleon", data, cont, chunk)
            _, pl = container.chunk_payloads(cont[:cn])
            okc = all(p == pyoracle.encode("chameleon", data[i*chunk:(i+1)*chunk]) for i, p in enumerate(pl))
            back = np.zeros(n, np.uint8)
            okd = container.decode(cont[:cn], back) == n and np.array_equal(back, data)
            if not okd:
                d = np.argmax(back != data); msg += f" decode first diff at {int(d)}"
        except Exception as ex:
            okc = okd = False; msg += f" container EXC {ex}"
        print(f"{vname:10s} {kind:9s} {n:8d} stream {'ok' if ok else 'BAD'} chunks {'ok' if okc else 'BAD'} decode {'ok' if okd else 'BAD'} {time.time()-t0:.2f}s{msg}", flush=True)
container.set_kernel_variant(0)
