"""Kernel times of the headline workload WITHOUT any correctness check — for throw-away experiment builds whose results may be wrong
(tools; never a source of reported numbers): python tools/gpu_kernel_time.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, chunk = 1 << 30, 4 << 20
x = torch.from_numpy(datagen.rep_text(n)).cuda()
cap = container.container_bound_slotted("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
hdr = container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
def step():
    container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
    container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
for _ in range(3): step()
torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
for _ in range(steps): step()
torch.cuda.synchronize()
t = {}
for name, ms in container.last_timings(): t[name] = t.get(name, 0.0) + ms / steps
print({k: round(v, 4) for k, v in t.items()}, "round trip equal:", bool(torch.equal(back, x)))
