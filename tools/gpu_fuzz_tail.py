"""Fuzz: Chameleon containers whose LAST chunk is short and ragged, payload bytes of that chunk corrupted; where the decoder accepts the container its output must be
the oracle's decode of the corrupted chunk stream.   python tools/gpu_fuzz_tail.py [trials] [seed] [kernel variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
from density_amd import container, DecodeError
from oracle import pyoracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad_total = 0
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for kind in ("mixed", "prose"):
    for tail in (999, 263, 4099, 70001):
        n, chunk = 3 * 262144 + tail, 262144
        data = datagen.by_kind(kind, n, seed=21)
        cont = np.zeros(container.container_bound("chameleon", n, chunk), dtype=np.uint8)
        container.set_kernel_variant(0)
        cn = container.encode("chameleon", data, cont, chunk)
        container.set_kernel_variant(variant)
        raw = cont[:cn].copy()
        hdr, payloads = container.chunk_payloads(raw)
        off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
        off = (off + (hdr.total_len + 255) // 256 + 15) // 16 * 16
        offs = []
        for p in payloads:
            offs.append(off); off = (off + len(p) + 15) // 16 * 16
        k = len(payloads) - 1
        div = acc = 0
        for t in range(trials):
            bad = raw.copy()
            at = offs[k] + int(rng.integers(0, len(payloads[k])))
            if t % 2: bad[at] ^= int(rng.integers(1, 256))
            else: bad[at:at + 4] = rng.integers(0, 256, size=min(4, len(bad) - at), dtype=np.uint8)
            out = np.zeros(n, dtype=np.uint8)
            try:
                m = container.decode(bad, out)
            except DecodeError:
                continue
            acc += 1
            want = b"".join(pyoracle.decode("chameleon", bytes(bad[offs[i]:offs[i] + len(payloads[i])]), min(chunk, n - i * chunk)) for i in range(len(payloads)))
            got = out[:m].tobytes()
            if got != want:
                div += 1
                first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
                print(f"DIVERGENCE {kind} tail {tail} trial {t}: corrupt at +{at - offs[k]} of {len(payloads[k])} ({'xor' if t % 2 else '4 bytes'}): got {len(got)} want {len(want)} bytes, first difference at chunk offset {None if first is None else first - 3 * chunk}; "
                      f"bytes before {raw[at:at+4].tolist()} after {bad[at:at+4].tolist()}", flush=True)
        print(f"variant {variant} chameleon {kind} tail {tail}: {trials} corruptions, {acc} accepted, {div} divergences", flush=True)
        bad_total += div
sys.exit(1 if bad_total else 0)
