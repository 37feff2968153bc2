"""More chunks than table slots (Cheetah / Lion keep their tables in a workspace of at most 8 GiB: 10922 / 4681 streams; beyond that a work-group takes its chunks one
after the other and clears its tables in between).  The debug build's DENSITY_HIP_SERIAL_SLOTS makes the slots few, so that a small container takes that road: the
chunk streams are the oracle's, the decode is the input.  (Lion's decoder has a parser wave that is through with a chunk long before its table wave: the clearing
waits for both.)  In a process of its own: the switch is read from the environment of a debug build."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, os.environ["ROOT"]); sys.path.insert(0, os.path.join(os.environ["ROOT"], "tests"))
import numpy as np, torch, datagen
from density_amd import container, _lib
from oracle import pyoracle
_lib.use_debug_build()
algo, kind, n, chunk = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
host = datagen.by_kind(kind, n, seed=9)
x = torch.from_numpy(host).cuda()
cap = container.container_bound(algo, n, chunk)
cont = torch.zeros(cap, dtype=torch.uint8, device="cuda"); back = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for rep in range(3):
    hdr = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    _, pays = container.chunk_payloads(cont[:hdr.container_len].cpu().numpy())
    nch = -(-n // chunk)
    assert len(pays) == nch
    for i in range(nch):
        assert pays[i] == pyoracle.encode(algo, host[i * chunk:(i + 1) * chunk]), (algo, i)
    back.zero_(); torch.cuda.synchronize()
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr) == n
    assert torch.equal(back, x)
print("ok")
'''


@pytest.mark.parametrize("algo", ["cheetah", "lion"])
@pytest.mark.parametrize("kind,n,chunk,slots", [("prose", 1_500_000, 65536, 3), ("mixed", 700_001, 32768, 2), ("prose", 2_000_000, 131072, 1)])
def test_more_chunks_than_table_slots(algo, kind, n, chunk, slots):
    env = dict(os.environ, DENSITY_HIP_SERIAL_SLOTS=str(slots), ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", SCRIPT, algo, kind, str(n), str(chunk)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-3000:])
