"""Kernel times of the headline workload WITHOUT any correctness claim — for throw-away experiment builds (tools; never a source of reported
numbers): python tools/gpu_kernel_time.py [steps] [other.so ...]
Every library named (the tree's own first) runs on the SAME device buffers in the same process, in alternation, so that memory placement
and the box are the same for all of them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
libs = [("tree", _lib.LIB_PATH)] + [(os.path.basename(p), os.path.abspath(p)) for p in sys.argv[2:]]
n, chunk = 1 << 30, 4 << 20
x = torch.from_numpy(datagen.rep_text(n)).cuda()
cap = container.container_bound_slotted("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
handles = {}
def use(name, path):
    if name not in handles:
        _lib._lib = None; _lib.LIB_PATH = path; handles[name] = _lib.lib()
    _lib._lib = handles[name]
def measure():
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
    container.set_profiling(False)
    return t, bool(torch.equal(back, x))
for rep in range(3 if len(libs) > 1 else 1):
    for name, path in libs:
        use(name, path)
        t, ok = measure()
        e, d = t.get("chameleon_encode_chunks", 0), t.get("chameleon_decode_chunks", 0)
        print(f"{name:>14}: encode {e:.4f} decode {d:.4f} sum {e + d:.4f} ms  round trip equal: {ok}", flush=True)
