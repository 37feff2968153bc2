"""How the round-trip time of the headline workload changes over the first seconds of continuous running (clock ramp after an idle phase):
python tools/gpu_ramp.py [idle seconds before] [total steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
total = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n, chunk = 1 << 30, 4 << 20
x = torch.from_numpy(datagen.rep_text(n)).cuda()
cap = container.container_bound_slotted("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
hdr = container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
def step():
    container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
    container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
for trial in range(2):
    torch.cuda.synchronize(); time.sleep(idle)
    t_start = time.perf_counter(); out = []
    done = 0
    for block in (5, 20, 25, 50, 100, 100, 200, 500, 1000, 1000):
        if done >= total: break
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(block): step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        done += block
        out.append(f"steps {done - block}..{done} ({(t1 - t_start) * 1e3:.0f} ms in): {(t1 - t0) / block * 1e3:.4f} ms/step")
    print(f"after {idle} s idle:\n  " + "\n  ".join(out), flush=True)
