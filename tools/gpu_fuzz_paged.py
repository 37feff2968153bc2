"""Fuzz: PAGED Chameleon containers (made by density_hip_encode_device_paged) with page bytes, directory entries or index bytes corrupted; where the decoder accepts
the blob its output must be the oracle's decode of the chunk streams a CPU reader reassembles from the corrupted blob (container.chunk_payloads); a blob the CPU
reader cannot reassemble (directory bytes that do not add up) must be refused.   python tools/gpu_fuzz_paged.py [trials] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from density_amd import container, DecodeError
from oracle import pyoracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad_total = 0
for kind, n, chunk in (("prose", 5 * (1 << 20) + 70001, 1 << 20), ("mixed", 4 * (1 << 20) + 263, 1 << 20), ("rep", 3 * (2 << 20) + 999, 2 << 20)):
    data = datagen.by_kind(kind, n, seed=9)
    x = torch.from_numpy(data).cuda()
    cap = container.container_bound_paged("chameleon", n, chunk)
    cont = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    hdr = container.encode_device_paged("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    assert hdr.flags & container.FLAG_PAGED
    raw = cont[:hdr.container_len].cpu().numpy().copy()
    h, payloads = container.chunk_payloads(raw)
    front = (32 + 4 * h.n_chunks + 15) // 16 * 16
    div = acc = refused_ok = 0
    for t in range(trials):
        bad = raw.copy()
        mode = t % 3
        if mode == 0:                                            # anywhere in the pages
            at = int(rng.integers(len(raw) // 8, len(raw))); bad[at] ^= int(rng.integers(1, 256))
        elif mode == 1:                                          # four random bytes in the pages
            at = int(rng.integers(len(raw) // 8, len(raw) - 4)); bad[at:at + 4] = rng.integers(0, 256, size=4, dtype=np.uint8)
        else:                                                    # front matter behind the header: size table, block index, directory
            at = int(rng.integers(32, max(front + 64, len(raw) // 8))); bad[at] ^= 1 << int(rng.integers(0, 8))
        out = np.zeros(n, dtype=np.uint8)
        try:
            readable = True
            _, streams = container.chunk_payloads(bad)
        except Exception:
            readable = False
        try:
            m = container.decode(bad, out)
        except DecodeError:
            continue
        acc += 1
        if not readable:
            div += 1
            print(f"DIVERGENCE {kind} trial {t} mode {mode}: accepted a blob whose directory does not add up (corrupt at {at})", flush=True)
            continue
        want = b"".join(pyoracle.decode("chameleon", streams[i], min(chunk, n - i * chunk)) for i in range(len(streams)))
        if out[:m].tobytes() != want:
            div += 1
            first = next((i for i in range(min(m, len(want))) if out[i] != want[i]), None)
            print(f"DIVERGENCE {kind} trial {t} mode {mode}: corrupt at {at}: got {m} want {len(want)} bytes, first difference at {first}", flush=True)
    print(f"paged {kind}: {trials} corruptions, {acc} accepted, {div} divergences", flush=True)
    bad_total += div
sys.exit(1 if bad_total else 0)
