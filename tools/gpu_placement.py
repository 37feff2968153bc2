"""Does the placement of the three buffers (input, container, output) move the kernel times?  One arena, the buffers carved at different offsets:
python tools/gpu_placement.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container
n, chunk = 1 << 30, 4 << 20
host = torch.from_numpy(datagen.rep_text(n))
cap = container.container_bound_slotted("chameleon", n, chunk)
capa = (cap + (1 << 21) - 1) >> 21 << 21
arena = torch.empty(n + capa + n + (64 << 20), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(ox, oc, ob, steps=60):
    x = arena[ox:ox + n]; cont = arena[oc:oc + cap]; back = arena[ob:ob + n]
    x.copy_(host)
    hdr = container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    def step():
        container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
        container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
    for _ in range(60): step()
    torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    t = {}
    for name, ms in container.last_timings(): t[name] = t.get(name, 0.0) + ms / steps
    container.set_profiling(False)
    return t["chameleon_encode_chunks"], t["chameleon_decode_chunks"], bool(torch.equal(back, x))
print("arena at", hex(arena.data_ptr()))
for pad in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, (5 << 20) + 65536 + 256):
    for pad2 in (0, pad):
        ox, oc, ob = 0, n + pad, n + pad + capa + pad2
        e, d, ok = run(ox, oc, ob)
        print(f"container +{pad:>8}, output +{pad2:>8}: encode {e:.4f} decode {d:.4f} sum {e + d:.4f} ok {ok}", flush=True)
