"""Where a GPU stream departs from the oracle's: record number, offset inside the record, what kind of byte."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen
from density_amd import Chameleon, container, _lib
from oracle import pyoracle

def records(enc, n):
    """(start, length, raw?) of every record of a chameleon stream, by walking it with the reference FSM"""
    from oracle import pymodel
    g, pos, out = pymodel.Guard(), 0, []
    for b0 in range(0, n, 256):
        blen = min(256, n - b0)
        if g.next_is_copy():
            out.append((pos, blen, True)); pos += blen; g.decay()
        else:
            sig = int.from_bytes(enc[pos:pos + 8], "little"); hits = bin(sig).count("1")
            l = 8 + 4 * (blen // 4) - 2 * hits + blen % 4
            out.append((pos, l, False)); g.update(l >= 256); pos += l
    return out

print("selftest bits:", hex(_lib.lib().density_hip_selftest_bits() & 0xffffffff))
for kind, n in (("prose", 65536), ("prose", 262144), ("zeros", 65536), ("rep", 1 << 20)):
    data = datagen.by_kind(kind, n, seed=5)
    want = pyoracle.encode("chameleon", data)
    out = np.zeros(Chameleon.safe_encode_buffer_size(n), np.uint8)
    m = Chameleon.encode(data, out); got = out[:m].tobytes()
    if got == want:
        print(kind, n, "encode ok")
    else:
        x = np.frombuffer(got[:min(len(got), len(want))], np.uint8) != np.frombuffer(want[:min(len(got), len(want))], np.uint8)
        bad = np.flatnonzero(x)
        recs = records(want, n)
        print(kind, n, "encode BAD: len", len(got), "vs", len(want), "diff bytes", bad.size, "first", bad[:12].tolist())
        shown = 0
        for k, (p, l, raw) in enumerate(recs):
            inside = bad[(bad >= p) & (bad < p + l)]
            if inside.size and shown < 6:
                print("   record", k, "start", p, "len", l, "raw" if raw else "coded", "bad offsets in record:", (inside - p)[:16].tolist(),
                      "want", want[inside[0]:inside[0] + 8].hex(), "got", got[inside[0]:inside[0] + 8].hex())
                shown += 1
    # decode of the oracle's stream packed as a one-chunk... use container path: encode on GPU with the old kernels, decode with rotor
    chunk = 1 << 16
    container.set_kernel_variant(4)
    cont = np.zeros(container.container_bound("chameleon", n, chunk), np.uint8)
    cn = container.encode("chameleon", data, cont, chunk)
    container.set_kernel_variant(0)
    back = np.zeros(n, np.uint8)
    try:
        ok = container.decode(cont[:cn], back) == n and np.array_equal(back, data)
        d = np.flatnonzero(back != data)
        print(kind, n, "rotor decode of a pipelined-encoded container:", "ok" if ok else f"BAD first diffs {d[:8].tolist()} count {d.size}")
    except Exception as ex:
        print(kind, n, "rotor decode EXC", ex)
