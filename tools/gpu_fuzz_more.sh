#!/bin/bash
# hardening runs beside the suite: threads, edge sizes, more corruption kinds (Cheetah containers of the skewed-vocabulary kinds through the dense / fall-back dictionary pass)
export TMPDIR=/tmp
timeout 600 python tools/gpu_threads.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python tools/gpu_edge_sizes.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_decode_passes as T
from density_amd import container, DecodeError
rng = np.random.default_rng(17)
for kind in ("vocab30", "vocab60", "samehash", "binaryish", "rep", "zeros"):
    for n, chunk in ((3 * 393216 + 4321, 393216), (5 * 131072 + 77, 131072)):
        data = T.make(kind, n, seed=3)
        raw, streams = T.cpu_container(data, chunk)
        base = (32 + 4 * len(streams) + 15) // 16 * 16
        div = 0
        for t in range(250):
            bad = raw.copy()
            mode = t % 4
            if mode == 0:
                at = base + int(rng.integers(0, len(raw) - base)); bad[at] ^= int(rng.integers(1, 256))
            elif mode == 1:
                at = base + int(rng.integers(0, len(raw) - base - 8)); bad[at:at + 8] = rng.integers(0, 256, size=8, dtype=np.uint8)
            elif mode == 2:
                k = int(rng.integers(0, len(streams))); sz = len(streams[k]); new = max(0, sz + int(rng.integers(-300, 300)))
                bad[32 + 4 * k:36 + 4 * k] = np.frombuffer(int(new).to_bytes(4, "little"), dtype=np.uint8)
            else:
                at = base + int(rng.integers(0, (len(raw) - base) // 2)) & ~1; bad[at] ^= 1 << int(rng.integers(0, 8))
            a, b = T.decode_both(bad, n)
            if a[0] != b[0] or (a[0] == "ok" and a[1] != b[1]):
                div += 1; print("DIVERGENCE", kind, n, chunk, t, mode, a[0], b[0], flush=True)
        print(f"cheetah {kind} n {n} chunk {chunk}: 250 corruptions, {div} divergences", flush=True)
PY
