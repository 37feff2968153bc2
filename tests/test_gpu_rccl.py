"""The multi-GPU path's process-group branch on real hardware ("run 1-rank RCCL on the box"): `bench.py --gpus 1`
under a torchrun-shaped environment of ONE rank initialises RCCL (`nccl` backend), runs the path's only collective (the size
all-gather, density_amd/parallel.py::exchange_layout) and the optional batched-P2P stitch (concat_to_rank0) on the GPU, and checks
that the stitched global container equals the local one and decodes to the input.  2/4/8 GPUs are the driver's to measure."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_runs_the_process_group_branch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", str(256 << 20),
                        "--no-cpu", "--no-sweep", "--no-extra", "--concat"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 1 and mg["process_group"] == "nccl (RCCL)"
    assert mg["concat_to_rank0_ms"] is not None and mg["concat_decodes_to_input"] is True
    assert mg["global_container_bytes"] == d["encoded_bytes_packed"]          # (the layout exchange is of the packed form; `encoded_bytes` describes the timed, paged one)


def test_exchange_layout_and_concat_on_device_tensors():
    """The same collectives called directly, on device tensors, in a one-rank RCCL group started inside a child process."""
    code = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch.distributed as dist
import datagen
from density_amd import container, parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, chunk = 8 << 20, 1 << 18
host = datagen.mixed(n, seed=3)
x = torch.from_numpy(host).cuda()
cap = container.container_bound("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
local = cont[:hdr.container_len]
h, table, index, payload = parallel.parse_local(local)
lay = parallel.exchange_layout(h["n_chunks"], payload.numel(), n, x.device)
assert lay["chunks"] == [hdr.n_chunks] and lay["chunk_offset"] == 0 and lay["payload_offset"] == 0
merged = parallel.concat_to_rank0(local, chunk)
assert torch.equal(merged, local)
dist.barrier(); dist.destroy_process_group()
print("RCCL-1 OK")
''' % (ROOT, ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-1 OK" in r.stdout, r.stderr[-3000:]
