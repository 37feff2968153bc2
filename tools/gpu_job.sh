#!/bin/bash
# r5y: paged containers with their pages in any order (the tail's 32-bit arithmetic), then the whole paged / chameleon files
timeout 600 python -m pytest tests/test_gpu_paged.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, datagen, paged_cpu
from density_amd import container
from test_paged_cpu_reader import _dir_heads
bad = 0
for seed in range(12):
    kind, n, chunk = [("random", 3 << 20, 1 << 20), ("mixed", (5 << 20) + 999, 1 << 20), ("rep", 8 << 20, 2 << 20)][seed % 3]
    data = datagen.rep_text(n) if kind == "rep" else datagen.by_kind(kind, n, seed=seed)
    blob = paged_cpu.build(data, chunk)
    hdr, _ = container.chunk_payloads(blob)
    total = sum(int.from_bytes(bytes(blob[d:d + 4]), "little") for d in _dir_heads(blob, hdr, chunk))
    blob = paged_cpu.build(data, chunk, page_order=list(np.random.default_rng(100 + seed).permutation(total)))
    d = torch.from_numpy(blob).cuda(); back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    try: ok = container.decode_device(d.data_ptr(), blob.size, back.data_ptr(), n) == n and np.array_equal(back.cpu().numpy(), data)
    except Exception as ex: ok = False
    bad += not ok
print("shuffled CPU-built paged containers: 12 tried,", bad, "failed")
PY
